// s2m_comm.cpp -- RCCL all-reduce of the normal block, called from the C++ host loop.
//
// Multi-GPU form of the path: every rank reduces its shard of the scan to the
// 160-double block, one ncclAllReduce(sum, ncclDouble) per ESKF iteration sums the blocks over xGMI,
// and every rank runs the identical fp64 update.  The message is 1.3 KB, so the collective is pure
// latency; it is issued on the engine's stream straight from the C++ loop (no Python in between).
// RCCL is resolved at run time (dlopen) so that single-GPU users carry no dependency on it; inside a
// PyTorch process the already loaded librccl is reused.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

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

}  // namespace s2m
