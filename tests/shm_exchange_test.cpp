// Host-only test of the shared-memory block exchange (daliti_amd/csrc/s2m_comm.cpp): N forked processes run many
// exchanges with jittered timing; every rank must see every rank's block of the SAME sequence, in rank order.
// Built and run by tests/test_sharding.py (no GPU involved: the exchange is host code).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "s2m_comm.h"

int main(int argc, char **argv)
{
    const int nranks = argc > 1 ? std::atoi(argv[1]) : 4;
    const int rounds = argc > 2 ? std::atoi(argv[2]) : 3000;
    const int count = 160;
    char name[64];
    std::snprintf(name, sizeof(name), "/s2m_test_%d", (int)getpid());
    {   // a leftover of a "crashed job" under the same name: every word 1 (sequence 1 of every slot looks published)
        const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
        if (fd >= 0) {
            const size_t bytes = 65536;
            if (ftruncate(fd, (off_t)bytes) == 0) {
                void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
                if (p != MAP_FAILED) { for (size_t i = 0; i < bytes / 8; ++i) static_cast<unsigned long long *>(p)[i] = 1ull; munmap(p, bytes); }
            }
            close(fd);
        }
    }
    std::vector<pid_t> kids;
    int rank = 0;
    for (int r = 1; r < nranks; ++r) {
        const pid_t p = fork();
        if (p < 0) return 3;
        if (p == 0) { rank = r; kids.clear(); break; }
        kids.push_back(p);
    }
    s2m::ShmExchange x;
    std::string err;
    std::vector<double> mine(count), all((size_t)nranks * count);
    unsigned seed = 1234u + 77u * (unsigned)rank;
    int bad = 0;
    // three generations under the SAME name, detach and re-attach without any barrier in between: the first
    // attach also has to get rid of the stale segment the parent left under the name (junk sequence words)
    for (int gen = 0; gen < 3 && !bad; ++gen) {
        if (!s2m::shm_exchange_init(x, name, nranks, rank, err)) { std::fprintf(stderr, "rank %d gen %d: %s\n", rank, gen, err.c_str()); bad = 2; break; }
        for (int it = 0; it < rounds && !bad; ++it) {
            for (int k = 0; k < count; ++k) mine[k] = 1000.0 * rank + it + 0.001 * k + 1e6 * gen;
            if ((rand_r(&seed) & 15) == 0) usleep(rand_r(&seed) % 200);  // shake the arrival order
            if (!s2m::shm_exchange(x, mine.data(), count, all.data(), err)) { std::fprintf(stderr, "rank %d: %s\n", rank, err.c_str()); bad = 1; break; }
            for (int r = 0; r < nranks && !bad; ++r)
                for (int k = 0; k < count; ++k)
                    if (all[(size_t)r * count + k] != 1000.0 * r + it + 0.001 * k + 1e6 * gen) {
                        std::fprintf(stderr, "rank %d gen %d round %d: slot %d word %d holds %.3f\n", rank, gen, it, r, k, all[(size_t)r * count + k]);
                        bad = 1;
                        break;
                    }
        }
        if ((rand_r(&seed) & 1) == 0) usleep(rand_r(&seed) % 3000);  // some ranks leave (and come back) late
        s2m::shm_exchange_destroy(x);
    }
    if (rank != 0) _exit(bad);
    for (pid_t p : kids) {
        int st = 0;
        waitpid(p, &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) bad = 1;
    }
    std::printf(bad ? "FAILED\n" : "ok: %d ranks x 3 attachments x %d exchanges\n", nranks, rounds);
    return bad;
}
