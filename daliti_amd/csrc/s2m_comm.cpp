// s2m_comm.cpp -- RCCL all-reduce of the normal block, called from the C++ host loop.
//
// Multi-GPU form of the path: every rank reduces its shard of the scan to the
// 160-double block, one ncclAllReduce(sum, ncclDouble) per ESKF iteration sums the blocks over xGMI,
// and every rank runs the identical fp64 update.  The message is 1.3 KB, so the collective is pure
// latency; it is issued on the engine's stream straight from the C++ loop (no Python in between).
// RCCL is resolved at run time (dlopen) so that single-GPU users carry no dependency on it; inside a
// PyTorch process the already loaded librccl is reused.
#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <rccl/rccl.h>

#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>

#include "s2m_comm.h"

namespace s2m {
namespace {
struct Api {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Api g_api;
std::mutex g_api_mutex;  // handles may attach communicators from different host threads

bool load_api(std::string &err)
{
    std::lock_guard<std::mutex> lock(g_api_mutex);
    if (g_api.lib) return true;
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)  // prefer an instance that is already in the process (PyTorch ships one)
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    for (int i = 0; !h && i < 3; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!h) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
    Api a;
    a.lib = h;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
    a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.AllReduce || !a.CommDestroy) { err = "librccl lacks the nccl* entry points"; return false; }
    g_api = a;
    return true;
}
}  // namespace

static_assert(sizeof(ncclUniqueId) == kCommIdBytes, "ncclUniqueId size");

bool comm_unique_id(unsigned char id[kCommIdBytes], std::string &err)
{
    if (!load_api(err)) return false;
    ncclUniqueId u;
    const ncclResult_t r = g_api.GetUniqueId(&u);
    if (r != ncclSuccess) { err = std::string("ncclGetUniqueId: ") + (g_api.GetErrorString ? g_api.GetErrorString(r) : "?"); return false; }
    std::memcpy(id, &u, kCommIdBytes);
    return true;
}

bool comm_init(Comm &c, const unsigned char id[kCommIdBytes], int nranks, int rank, std::string &err)
{
    if (!load_api(err)) return false;
    ncclUniqueId u;
    std::memcpy(&u, id, kCommIdBytes);
    ncclComm_t comm = nullptr;
    const ncclResult_t r = g_api.CommInitRank(&comm, nranks, u, rank);
    if (r != ncclSuccess) { err = std::string("ncclCommInitRank: ") + (g_api.GetErrorString ? g_api.GetErrorString(r) : "?"); return false; }
    c.handle = comm;
    c.nranks = nranks;
    c.rank = rank;
    return true;
}

bool comm_allreduce_sum_f64(Comm &c, double *d_buf, size_t count, hipStream_t st, std::string &err)
{
    const ncclResult_t r = g_api.AllReduce(d_buf, d_buf, count, ncclDouble, ncclSum, (ncclComm_t)c.handle, st);
    if (r != ncclSuccess) { err = std::string("ncclAllReduce: ") + (g_api.GetErrorString ? g_api.GetErrorString(r) : "?"); return false; }
    return true;
}

void comm_destroy(Comm &c)
{
    if (c.handle && g_api.CommDestroy) (void)g_api.CommDestroy((ncclComm_t)c.handle);
    c = Comm();
}

// ---- host shared-memory exchange (s2m_comm.h) --------------------------------------------------------------------
namespace {
// Header of the segment (the slots follow kShmHeaderDoubles doubles in).  Rank 0 OWNS the name: it removes whatever sits
// under it (the leftover of a crashed job carries old sequence words), creates the segment exclusively -- fresh, zero-
// filled -- and writes the magic word; the other ranks open it, count themselves in, and everybody starts when rank 0
// has seen all of them (`go`).  A rank that finds a segment already fully attached, or whose name has moved to another
// inode while it waits, has caught the previous generation of the name (re-attach under the same name without a
// barrier in between) and starts over on the new one.
struct ShmHeader {
    unsigned long long magic, nranks, attached, go;
};
constexpr unsigned long long kShmMagic = 0x5332'4d45'5843'4831ull;  // "S2MEXCH1"
constexpr double kShmAttachTimeout = 60.0;

double since(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
bool same_inode(const char *name, const struct stat &mine)
{
    const int fd = shm_open(name, O_RDWR, 0600);
    if (fd < 0) return false;
    struct stat st;
    const bool same = fstat(fd, &st) == 0 && st.st_ino == mine.st_ino && st.st_dev == mine.st_dev;
    close(fd);
    return same;
}
}  // namespace

bool shm_exchange_init(ShmExchange &x, const char *name, int nranks, int rank, std::string &err)
{
    shm_exchange_destroy(x);
    if (!name || name[0] != '/' || std::strlen(name) >= sizeof(x.name) || std::strchr(name + 1, '/')) {
        err = "shared-memory name must look like \"/something\"";
        return false;
    }
    const size_t bytes = ((size_t)kShmHeaderDoubles + (size_t)2 * nranks * kShmSlotDoubles) * sizeof(double);
    const auto t0 = std::chrono::steady_clock::now();
    void *base = MAP_FAILED;
    if (rank == 0) {
        (void)shm_unlink(name);
        const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) { err = std::string("shm_open: ") + std::strerror(errno); return false; }
        if (ftruncate(fd, (off_t)bytes) != 0) { err = std::string("ftruncate: ") + std::strerror(errno); close(fd); shm_unlink(name); return false; }
        base = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (base == MAP_FAILED) { err = std::string("mmap: ") + std::strerror(errno); shm_unlink(name); return false; }
        ShmHeader *h = static_cast<ShmHeader *>(base);
        h->nranks = (unsigned long long)nranks;
        __atomic_store_n(&h->attached, 1ull, __ATOMIC_RELAXED);
        __atomic_store_n(&h->magic, kShmMagic, __ATOMIC_RELEASE);
        while (__atomic_load_n(&h->attached, __ATOMIC_ACQUIRE) < (unsigned long long)nranks) {
            if (since(t0) > kShmAttachTimeout) {
                err = "shared-memory exchange: not every rank attached within 60 s";
                munmap(base, bytes); shm_unlink(name);
                return false;
            }
            usleep(50);
        }
        __atomic_store_n(&h->go, 1ull, __ATOMIC_RELEASE);
    } else {
        for (;;) {
            if (since(t0) > kShmAttachTimeout) { err = "shared-memory exchange: rank 0 did not create the segment within 60 s"; return false; }
            const int fd = shm_open(name, O_RDWR, 0600);
            struct stat st;
            if (fd < 0 || fstat(fd, &st) != 0 || (size_t)st.st_size < bytes) {  // not there yet, or not sized yet
                if (fd >= 0) close(fd);
                usleep(100);
                continue;
            }
            base = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            close(fd);
            if (base == MAP_FAILED) { err = std::string("mmap: ") + std::strerror(errno); return false; }
            ShmHeader *h = static_cast<ShmHeader *>(base);
            bool stale = false;
            while (__atomic_load_n(&h->magic, __ATOMIC_ACQUIRE) != kShmMagic && !stale) {
                if (since(t0) > kShmAttachTimeout || !same_inode(name, st)) stale = true;
                else usleep(50);
            }
            if (!stale && h->nranks != (unsigned long long)nranks) { err = "shared-memory exchange: the segment was created for another rank count"; munmap(base, bytes); return false; }
            // count in; a segment that is already full is the previous generation of this name
            if (!stale && __atomic_add_fetch(&h->attached, 1ull, __ATOMIC_ACQ_REL) > (unsigned long long)nranks) stale = true;
            while (!stale && __atomic_load_n(&h->go, __ATOMIC_ACQUIRE) == 0ull) {
                if (since(t0) > kShmAttachTimeout) { err = "shared-memory exchange: not every rank attached within 60 s"; munmap(base, bytes); return false; }
                if (!same_inode(name, st)) stale = true;   // rank 0 has replaced the segment under this name
                else usleep(50);
            }
            if (!stale) break;
            munmap(base, bytes);
            base = MAP_FAILED;
            usleep(200);
        }
    }
    x.base = base;
    x.bytes = bytes;
    x.nranks = nranks;
    x.rank = rank;
    x.seq = 0;
    std::snprintf(x.name, sizeof(x.name), "%s", name);
    return true;
}

bool shm_exchange(ShmExchange &x, const double *block, int count, double *out, std::string &err)
{
    if (!x.base || count < 1 || count >= kShmSlotDoubles) { err = "shared-memory exchange not initialised"; return false; }
    const unsigned long long seq = ++x.seq;
    double *slots = static_cast<double *>(x.base) + kShmHeaderDoubles + (size_t)(seq & 1ull) * x.nranks * kShmSlotDoubles;
    double *mine = slots + (size_t)x.rank * kShmSlotDoubles;
    std::memcpy(mine, block, (size_t)count * sizeof(double));
    __atomic_store_n(reinterpret_cast<unsigned long long *>(mine + kShmSlotDoubles - 1), seq, __ATOMIC_RELEASE);
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < x.nranks; ++r) {
        const double *slot = slots + (size_t)r * kShmSlotDoubles;
        const unsigned long long *flag = reinterpret_cast<const unsigned long long *>(slot + kShmSlotDoubles - 1);
        for (long spin = 0;; ++spin) {
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) break;
            __builtin_ia32_pause();
            if ((spin & 0xfffff) == 0xfffff &&
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 60.0) {
                err = "shared-memory exchange: rank " + std::to_string(r) + " did not publish sequence " + std::to_string(seq) +
                      " within 60 s";
                return false;
            }
        }
        std::memcpy(out + (size_t)r * count, slot, (size_t)count * sizeof(double));
    }
    return true;
}

void shm_exchange_destroy(ShmExchange &x)
{
    if (x.base) {
        munmap(x.base, x.bytes);
        if (x.name[0] && x.rank == 0) shm_unlink(x.name);  // the name belongs to rank 0; mappings of the others stay valid
    }
    x = ShmExchange();
}

}  // namespace s2m
