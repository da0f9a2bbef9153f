// s2m_relay.hip -- kernels of the layout that is produced BESIDE the frames (s2m_engine_relay.cpp).
//
// ikd-Tree moves every large rebuild to a second thread (ikd_Tree.cpp:192-203, 229-367): the tree is flattened, rebuilt beside
// the node's loop, the operations that arrived meanwhile are applied again from a log (Rebuild_Logger) and the new subtree is
// swapped in.  The same here for the whole map: a snapshot of the live points with their ids (in position order),
// a complete build from it on the handle's layout stream, the ids put back, the update calls that arrived meanwhile run again
// on the new map (the voxel rule is a function of the point SET: the same calls give the same set and the same ids), swap
// between two frames.  Also: the count of occupied cells of the live map, to see the density drift away from what the cell
// size was chosen for without waiting for a merge to count them.
#include <rocprim/device/device_radix_sort.hpp>

#include "s2m_map_internal.h"

namespace s2m {

// every position that holds a point -> {x, y, z, bitcast(id)}, IN POSITION ORDER: inside a cell the live map's positions ascend
// with the point id, the build's sort is stable, so the new layout has the documented order (brick, cell, id) again -- the order
// ties at the same float distance are ranked by.  Three launches: live positions per block of 1 024, their exclusive sums (one
// workgroup), the write.
constexpr int kSnapPer = 4, kSnapBlock = 256 * kSnapPer;
__global__ __launch_bounds__(256) void snapshot_count_kernel(const uint32_t *__restrict__ pidx, int64_t m, uint32_t *__restrict__ blk)
{
    __shared__ uint32_t s[4];
    const int64_t j0 = (int64_t)blockIdx.x * kSnapBlock + (int64_t)threadIdx.x * kSnapPer;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < kSnapPer; ++k) c += (j0 + k < m && pidx[j0 + k] != 0xffffffffu) ? 1u : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blk[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}
__global__ __launch_bounds__(1024) void snapshot_scan_kernel(uint32_t *__restrict__ blk, int nb, uint32_t *__restrict__ total)
{
    __shared__ uint32_t part[1024];
    const int t = threadIdx.x, per = (nb + 1023) / 1024, lo = t * per, hi = min(lo + per, nb);
    uint32_t sum = 0;
    for (int i = lo; i < hi; ++i) sum += blk[i];
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {   // inclusive scan of the 1 024 partial sums
        const uint32_t v = t >= off ? part[t - off] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;
    for (int i = lo; i < hi; ++i) { const uint32_t c = blk[i]; blk[i] = run; run += c; }
    if (t == 1023) *total = part[1023];
}
__global__ __launch_bounds__(256) void snapshot_write_kernel(const float4 *__restrict__ pts, const uint32_t *__restrict__ pidx, int64_t m,
                                                             const uint32_t *__restrict__ blk, float4 *__restrict__ out, uint32_t cap)
{
    __shared__ uint32_t s[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t j0 = (int64_t)blockIdx.x * kSnapBlock + (int64_t)threadIdx.x * kSnapPer;
    uint32_t id[kSnapPer], c = 0;
#pragma unroll
    for (int k = 0; k < kSnapPer; ++k) {
        id[k] = j0 + k < m ? pidx[j0 + k] : 0xffffffffu;
        c += id[k] != 0xffffffffu ? 1u : 0u;
    }
    uint32_t inc = c;   // inclusive prefix over the wave's lanes
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(inc, off, 64);
        if (lane >= off) inc += v;
    }
    if (lane == 63) s[wave] = inc;
    __syncthreads();
    uint32_t at = blk[blockIdx.x] + inc - c;
    for (int w = 0; w < wave; ++w) at += s[w];
#pragma unroll
    for (int k = 0; k < kSnapPer; ++k) {
        if (id[k] == 0xffffffffu) continue;
        if (at < cap) {
            const float4 p = pts[j0 + k];
            out[at] = make_float4(p.x, p.y, map_point_z(p), __uint_as_float(id[k]));
        }
        ++at;
    }
}
// The snapshot in ascending id order: the build's sort is stable, so inside a cell of the NEW grid -- whatever its size -- the
// points then stand in id order, the documented order of ties (s2m_map_get_order).  (Position order gives that only while the
// cells stay the same.)
__global__ __launch_bounds__(256) void snap_keys_kernel(const float4 *__restrict__ snap, int64_t n, uint32_t *__restrict__ key, uint32_t *__restrict__ val)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { key[i] = __float_as_uint(snap[i].w); val[i] = (uint32_t)i; }
}
__global__ __launch_bounds__(256) void snap_gather_kernel(const float4 *__restrict__ snap, const uint32_t *__restrict__ val, int64_t n, float4 *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = snap[val[i]];
}
// (rocprim sorts up to 2^20 items by a merge sort: a block sort and twenty merge passes -- twenty-odd launches of a few
// microseconds each.  Beside the frames such a chain of tiny dependent kernels on a third queue delays the dispatch of the
// frame's and the side stream's kernels (15 - 50 us between kernels instead of 0 - 5: NOTEBOOK round 6); the radix passes are
// six launches)
using SnapSortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 4096>;
size_t snapshot_sort_tmp_bytes(int64_t n)
{
    size_t tmp = 0;
    uint32_t *p = nullptr;
    (void)rocprim::radix_sort_pairs<SnapSortConfig>(nullptr, tmp, p, p, p, p, (size_t)std::max<int64_t>(n, 1), 0, 32, (hipStream_t) nullptr);
    size_t tmp2 = 0;   // (either road's storage: the limit above is compared with the count at run time)
    (void)rocprim::radix_sort_pairs<SnapSortConfig>(nullptr, tmp2, p, p, p, p, (size_t)4096, 0, 32, (hipStream_t) nullptr);
    tmp = std::max(tmp, tmp2);
    return tmp;
}
// work: 4 n words (keys, sorted keys, values, sorted values); out: n points
hipError_t snapshot_sort_by_id(const float4 *snap, int64_t n, float4 *out, uint32_t *work, void *tmp, size_t tmp_bytes, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    uint32_t *key = work, *key2 = work + n, *val = work + 2 * n, *val2 = work + 3 * n;
    hipLaunchKernelGGL(snap_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, snap, n, key, val);
    S2M_TRY(rocprim::radix_sort_pairs<SnapSortConfig>(tmp, tmp_bytes, key, key2, val, val2, (size_t)n, 0, 32, st));
    hipLaunchKernelGGL(snap_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, snap, val2, n, out);
    return hipGetLastError();
}

// a map built from a snapshot numbers its points by their place in the snapshot: back to the ids they had
__global__ __launch_bounds__(256) void remap_ids_kernel(uint32_t *__restrict__ pidx, int64_t m, const float4 *__restrict__ snap)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const uint32_t i = pidx[j];
    if (i != 0xffffffffu) pidx[j] = __float_as_uint(snap[i].w);
}
// occupied cells of the bricks in use: a wave per brick over its 513 prefix words (word i + 1 > word i: cell i holds points), the
// workgroups stride over the bricks and add ONE number each to the count (an atomic per brick on one word was 0.2 ms of a frame
// at 20 k bricks: they are served one after the other)
__global__ __launch_bounds__(256) void count_cells_kernel(const uint32_t *__restrict__ bricks_dev, const uint32_t *__restrict__ tab,
                                                          uint32_t *__restrict__ out)
{
    __shared__ int part[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t n = (int64_t)*bricks_dev;
    int cells = 0;
    for (int64_t id = (int64_t)blockIdx.x * 4 + wave; id < n; id += (int64_t)gridDim.x * 4) {
        const uint32_t *t = tab + id * kBrickStride;
#pragma unroll
        for (int k = 0; k < 8; ++k) cells += t[k * 64 + lane + 1] > t[k * 64 + lane] ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cells += __shfl_xor(cells, off, 64);
    if (lane == 0) part[wave] = cells;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int all = part[0] + part[1] + part[2] + part[3];
        if (all > 0) atomicAdd(out, (uint32_t)all);
    }
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

int64_t snapshot_blocks(int64_t m) { return (m + kSnapBlock - 1) / kSnapBlock; }
// blk: snapshot_blocks(m) words of scratch; *count = the number of points written (the live points of the map)
void launch_snapshot(const float4 *pts, const uint32_t *pidx, int64_t m, float4 *out, uint32_t *count, int64_t cap, uint32_t *blk, hipStream_t st)
{
    (void)hipMemsetAsync(count, 0, sizeof(uint32_t), st);
    if (m <= 0) return;
    const int nb = (int)snapshot_blocks(m);
    hipLaunchKernelGGL(snapshot_count_kernel, dim3((unsigned)nb), dim3(256), 0, st, pidx, m, blk);
    hipLaunchKernelGGL(snapshot_scan_kernel, dim3(1), dim3(1024), 0, st, blk, nb, count);
    hipLaunchKernelGGL(snapshot_write_kernel, dim3((unsigned)nb), dim3(256), 0, st, pts, pidx, m, blk, out, (uint32_t)cap);
}
void launch_remap_ids(uint32_t *pidx, int64_t m, const float4 *snap, hipStream_t st)
{
    if (m > 0) hipLaunchKernelGGL(remap_ids_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, pidx, m, snap);
}
void launch_count_cells(const uint32_t *bricks_dev, int64_t bricks_bound, const uint32_t *tab, uint32_t *cells_dev, uint32_t *host_dev, uint32_t seq,
                        hipStream_t st)
{
    if (bricks_bound > 0)
        hipLaunchKernelGGL(count_cells_kernel, dim3((unsigned)std::min<int64_t>((bricks_bound + 3) / 4, 1024)), dim3(256), 0, st, bricks_dev, tab, cells_dev);
    hipLaunchKernelGGL(cells_home_kernel, dim3(1), dim3(1), 0, st, cells_dev, host_dev, seq);
}

}  // namespace s2m
