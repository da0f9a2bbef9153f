// s2m_relay.hip -- kernels of the layout that is produced BESIDE the frames (s2m_engine_relay.cpp).
//
// ikd-Tree moves every large rebuild to a second thread (ikd_Tree.cpp:192-203, 229-367): the tree is flattened, rebuilt beside
// the node's loop, the operations that arrived meanwhile are applied again from a log (Rebuild_Logger) and the new subtree is
// swapped in.  The same here for the whole map: a snapshot of the live points with their ids (one pass, unordered compaction),
// a complete build from it on the handle's layout stream, the ids put back, the update calls that arrived meanwhile run again
// on the new map (the voxel rule is a function of the point SET: the same calls give the same set and the same ids), swap
// between two frames.  Also: the count of occupied cells of the live map, to see the density drift away from what the cell
// size was chosen for without waiting for a merge to count them.
#include "s2m_map_internal.h"

namespace s2m {

// every position that holds a point -> {x, y, z, bitcast(id)}, in no particular order (the build sorts anyway)
__global__ __launch_bounds__(256) void snapshot_kernel(const float4 *__restrict__ pts, const uint32_t *__restrict__ pidx, int64_t m,
                                                       float4 *__restrict__ out, uint32_t *__restrict__ count, uint32_t cap)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t id = 0xffffffffu;
    const bool live = j < m && (id = pidx[j]) != 0xffffffffu;
    const unsigned long long bal = __ballot(live);
    if (bal == 0ull) return;
    uint32_t at = 0;
    if (lane == __ffsll((long long)bal) - 1) at = atomicAdd(count, (uint32_t)__popcll(bal));
    at = __shfl(at, __ffsll((long long)bal) - 1, 64);
    if (live) {
        const uint32_t mine = at + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        if (mine < cap) {
            const float4 p = pts[j];
            out[mine] = make_float4(p.x, p.y, map_point_z(p), __uint_as_float(id));
        }
    }
}
// a map built from a snapshot numbers its points by their place in the snapshot: back to the ids they had
__global__ __launch_bounds__(256) void remap_ids_kernel(uint32_t *__restrict__ pidx, int64_t m, const float4 *__restrict__ snap)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const uint32_t i = pidx[j];
    if (i != 0xffffffffu) pidx[j] = __float_as_uint(snap[i].w);
}
// occupied cells of the bricks in use: one wave per brick over its 513 prefix words
__global__ __launch_bounds__(256) void count_cells_kernel(const uint32_t *__restrict__ bricks_dev, const uint32_t *__restrict__ tab,
                                                          uint32_t *__restrict__ out)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t id = (int64_t)blockIdx.x * 4 + wave;
    if (id >= (int64_t)*bricks_dev) return;
    const uint32_t *t = tab + id * kBrickStride;
    int cells = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) cells += t[lane * 8 + k + 1] > t[lane * 8 + k] ? 1 : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cells += __shfl_xor(cells, off, 64);
    if (lane == 0 && cells > 0) atomicAdd(out, (uint32_t)cells);
}
// {occupied cells, sequence word} into pinned host memory, and the device word back to zero for the next count
__global__ void cells_home_kernel(uint32_t *__restrict__ cells, uint32_t *__restrict__ host, uint32_t seq)
{
    __hip_atomic_store(host + 1, *cells, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __builtin_amdgcn_s_waitcnt(0);
    __hip_atomic_store(host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    *cells = 0u;
}

void launch_snapshot(const float4 *pts, const uint32_t *pidx, int64_t m, float4 *out, uint32_t *count, int64_t cap, hipStream_t st)
{
    (void)hipMemsetAsync(count, 0, sizeof(uint32_t), st);
    if (m > 0) hipLaunchKernelGGL(snapshot_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, pts, pidx, m, out, count, (uint32_t)cap);
}
void launch_remap_ids(uint32_t *pidx, int64_t m, const float4 *snap, hipStream_t st)
{
    if (m > 0) hipLaunchKernelGGL(remap_ids_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, pidx, m, snap);
}
void launch_count_cells(const uint32_t *bricks_dev, int64_t bricks_bound, const uint32_t *tab, uint32_t *cells_dev, uint32_t *host_dev, uint32_t seq,
                        hipStream_t st)
{
    if (bricks_bound > 0)
        hipLaunchKernelGGL(count_cells_kernel, dim3((unsigned)((bricks_bound + 3) / 4)), dim3(256), 0, st, bricks_dev, tab, cells_dev);
    hipLaunchKernelGGL(cells_home_kernel, dim3(1), dim3(1), 0, st, cells_dev, host_dev, seq);
}

}  // namespace s2m
