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

bool comm_unique_id(unsigned char id[kCommIdBytes], std::string &err);
bool comm_init(Comm &c, const unsigned char id[kCommIdBytes], int nranks, int rank, std::string &err);
bool comm_allreduce_sum_f64(Comm &c, double *d_buf, size_t count, hipStream_t st, std::string &err);
void comm_destroy(Comm &c);

}  // namespace s2m
