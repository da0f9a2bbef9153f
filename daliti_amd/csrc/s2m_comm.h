// s2m_comm.h -- thin wrapper over the RCCL calls the sharded update needs (see s2m_comm.cpp).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <string>

namespace s2m {

constexpr int kCommIdBytes = 128;  // NCCL_UNIQUE_ID_BYTES

struct Comm {
    void *handle = nullptr;  // ncclComm_t
    int nranks = 0, rank = 0;
};

// Exchange of the normal block between the PROCESSES of one node through POSIX shared memory, without a collective
// library and without the GPU: every rank's reduce kernel publishes its block into that rank's own pinned page (as in
// the single-GPU loop), the rank's host thread copies the 1.3 KB into its slot of the shared segment and raises the slot's
// sequence word, then reads everybody's slot of the same sequence.  Two slots per rank (sequence parity): a rank can
// only reach sequence k + 2 after every rank has published k + 1, i.e. after every rank has finished reading k.
struct ShmExchange {
    void *base = nullptr;   // mapped segment
    size_t bytes = 0;
    int nranks = 0, rank = 0;
    unsigned long long seq = 0;  // last sequence this rank published (the same on every rank: the loop runs in lock step)
    char name[96] = {0};
};
constexpr int kShmSlotDoubles = 256;  // 160-double block + sequence word, padded to 2 KB (no line shared between slots)
constexpr int kShmHeaderDoubles = 16;  // attach header in front of the slots: magic, rank count, ranks attached, go

// Collective over the ranks: rank 0 creates the segment under `name` (removing a leftover), the others wait for it, and
// every rank returns once all have attached (or fails after 60 s).  Re-attaching under the same name is safe.
bool shm_exchange_init(ShmExchange &x, const char *name, int nranks, int rank, std::string &err);
// publish `block` (count doubles) as this rank's contribution, wait for all ranks, return the per-rank blocks in rank order
// in `out` (nranks x count).  false on time-out (a rank died or left the loop).
bool shm_exchange(ShmExchange &x, const double *block, int count, double *out, std::string &err);
void shm_exchange_destroy(ShmExchange &x);

bool comm_unique_id(unsigned char id[kCommIdBytes], std::string &err);
bool comm_init(Comm &c, const unsigned char id[kCommIdBytes], int nranks, int rank, std::string &err);
bool comm_allreduce_sum_f64(Comm &c, double *d_buf, size_t count, hipStream_t st, std::string &err);
void comm_destroy(Comm &c);

}  // namespace s2m
