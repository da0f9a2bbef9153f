// s2m_map.hip -- builds the brick-grid map (layout: s2m_device.h) from an unordered point cloud.
//
// Replaces the map seed of the reference: ikdtree.Build(feats_down_world->points)
// (eskf_lio/src/laserMapping.cpp:784-790; KD_TREE::Build / BuildTree,
// eskf_lio/include/ikd-Tree/ikd_Tree.cpp:408-423, 678-733).  Instead of a pointer tree of 176-byte
// nodes the points are radix-sorted by (brick, cell-in-brick; stable, i.e. then by caller index) into one float4
// array whose third word is the sorted position itself, with the caller indices beside it (pidx); a toroidally addressed
// top-level array over the bricks' signed integer coordinates and a 513-entry prefix table per occupied brick locate
// any run of cells along x with two 4-byte loads -- and none of it depends on the box the cloud happens to occupy.
//
// This is the per-map part of the path, not the per-iteration hot loop: the full build (radix sort of all points) and
// the tables that both the build and a merged update derive from the sorted keys; how the map changes after a scan is
// s2m_mapedit.hip.  The device-wide radix sort and scans come from rocPRIM (ROCm's native primitives library),
// everything else is hand-written below.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>
#include <rocprim/iterator/zip_iterator.hpp>

#include "s2m_map_internal.h"

namespace s2m {

// ---- bounding box of an AoS cloud ---------------------------------------------------------------------
// Per-workgroup partial boxes, then one workgroup folds them: no atomics (same-line atomics cost ~11 ns
// each on MI355X; a version with six atomics per wave spent 0.5 ms of 0.57 ms on them at 5 M points).
__device__ __forceinline__ void block_minmax(float (&mn)[3], float (&mx)[3], float *__restrict__ out6)
{
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[k] = fminf(mn[k], __shfl_xor(mn[k], off, 64));
            mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], off, 64));
        }
    }
    __shared__ float smn[4][3], smx[4][3];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { smn[wave][k] = mn[k]; smx[wave][k] = mx[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        out6[k] = fminf(fminf(smn[0][k], smn[1][k]), fminf(smn[2][k], smn[3][k]));
        out6[3 + k] = fmaxf(fmaxf(smx[0][k], smx[1][k]), fmaxf(smx[2][k], smx[3][k]));
    }
}

__global__ __launch_bounds__(256) void bbox_partial_kernel(const float *__restrict__ xyz, int64_t stride, int64_t m,
                                                           float *__restrict__ partial /* blocks x 6 */)
{
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float v = xyz[i * stride + k];
            mn[k] = fminf(mn[k], v);
            mx[k] = fmaxf(mx[k], v);
        }
    }
    block_minmax(mn, mx, partial + (int64_t)blockIdx.x * 6);
}

// (dims, optional: the voxel grid's numbers derived from the box right here -- s2m_voxel.hip, voxel_downsample -- so that the
// host need not see the box before it launches the grid's kernels)
__global__ __launch_bounds__(256) void bbox_final_kernel(const float *__restrict__ partial, int blocks,
                                                         float *__restrict__ out6, float inv_leaf, VoxelDimsDev *__restrict__ dims)
{
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int b = threadIdx.x; b < blocks; b += blockDim.x) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            mn[k] = fminf(mn[k], partial[b * 6 + k]);
            mx[k] = fmaxf(mx[k], partial[b * 6 + 3 + k]);
        }
    }
    block_minmax(mn, mx, out6);
    if (dims) {
        __syncthreads();
        if (threadIdx.x == 0) {
            // pcl::VoxelGrid::applyFilter: min_b = floor(min * inv_leaf), div_b = max_b - min_b + 1, divb_mul = {1, div0, div0 * div1}
            int64_t div[3];
            for (int k = 0; k < 3; ++k) {
                dims->min_b[k] = (int)floorf(out6[k] * inv_leaf);
                div[k] = (int64_t)(int)floorf(out6[3 + k] * inv_leaf) - dims->min_b[k] + 1;
            }
            dims->mul1 = div[0];
            dims->mul2 = div[0] * div[1];
            const int64_t total = div[0] * div[1] * div[2];
            dims->too_fine = total > (int64_t)2147483647 ? 1u : 0u;
            uint32_t bits = 1u;
            while (bits < 32u && ((int64_t)1 << bits) < total) ++bits;
            dims->bits = bits;
        }
    }
}

hipError_t cloud_bbox_launch(const float *xyz, int64_t stride, int64_t n, float *scratch, float inv_leaf, VoxelDimsDev *dims, hipStream_t st)
{
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, kBboxBlocks);
    float *out = scratch + (int64_t)kBboxBlocks * 6;
    hipLaunchKernelGGL(bbox_partial_kernel, dim3(blocks), dim3(256), 0, st, xyz, stride, n, scratch);
    hipLaunchKernelGGL(bbox_final_kernel, dim3(1), dim3(256), 0, st, scratch, blocks, out, inv_leaf, dims);
    return hipGetLastError();
}

hipError_t cloud_bbox(const float *xyz, int64_t stride, int64_t n, float *scratch, Mailbox &mail, float lo[3], float hi[3],
                      hipStream_t st)
{
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, kBboxBlocks);
    float *out = scratch + (int64_t)kBboxBlocks * 6;
    hipLaunchKernelGGL(bbox_partial_kernel, dim3(blocks), dim3(256), 0, st, xyz, stride, n, scratch);
    hipLaunchKernelGGL(bbox_final_kernel, dim3(1), dim3(256), 0, st, scratch, blocks, out, 0.0f, (VoxelDimsDev *)nullptr);
    const uint32_t *w = reinterpret_cast<const uint32_t *>(out);
    const uint32_t *src[6] = {w, w + 1, w + 2, w + 3, w + 4, w + 5};
    uint32_t bits[6];
    hipError_t e = mail_fetch(mail, src, 6, bits, st);
    if (e != hipSuccess) return e;
    float box[6];
    std::memcpy(box, bits, sizeof(box));
    for (int k = 0; k < 3; ++k) { lo[k] = box[k]; hi[k] = box[3 + k]; }
    return hipGetLastError();
}

// Sort key of the BUILD: the box-independent key (point_key, s2m_device.h) orders bricks lexicographically by their signed
// (z, y, x); inside the build's bounding box the linear index ((bz - lo) * ny + (by - lo)) * nx + (bx - lo) orders them the
// same way with far fewer bits for the radix sort (22 instead of 63 at 5 M points: three passes instead of eight).  The
// gather kernel expands the sorted keys to the box-independent form, which is what the map keeps.
struct BuildBox {
    int lo[3], n[3];
};
__global__ __launch_bounds__(256) void key_kernel(const float *__restrict__ xyz, int64_t stride, int64_t m, Grid g, BuildBox bb,
                                                  uint64_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int cx = cell_coord(xyz[i * stride + 0], g.ox, g.inv_c);
    const int cy = cell_coord(xyz[i * stride + 1], g.oy, g.inv_c);
    const int cz = cell_coord(xyz[i * stride + 2], g.oz, g.inv_c);
    const uint64_t brick = ((uint64_t)((cz >> 3) - bb.lo[2]) * (uint64_t)bb.n[1] + (uint64_t)((cy >> 3) - bb.lo[1])) * (uint64_t)bb.n[0] +
                           (uint64_t)((cx >> 3) - bb.lo[0]);
    const uint32_t local = (uint32_t)((((cz & 7) << 3) | (cy & 7)) << 3 | (cx & 7));
    keys[i] = (brick << 9) | local;
    vals[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void gather_kernel(const float *__restrict__ xyz, int64_t stride, int64_t m,
                                                     const uint32_t *__restrict__ vals, BuildBox bb, float4 *__restrict__ pts,
                                                     uint32_t *__restrict__ pidx, uint64_t *__restrict__ keys)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const uint32_t src = vals[j];
    pts[j] = make_map_point(xyz[(int64_t)src * stride], xyz[(int64_t)src * stride + 1], xyz[(int64_t)src * stride + 2],
                            (uint32_t)j);
    pidx[j] = src;
    const uint64_t k = keys[j], lin = k >> 9;
    const int bx = (int)(lin % (uint64_t)bb.n[0]) + bb.lo[0], by = (int)((lin / (uint64_t)bb.n[0]) % (uint64_t)bb.n[1]) + bb.lo[1],
              bz = (int)(lin / ((uint64_t)bb.n[0] * (uint64_t)bb.n[1])) + bb.lo[2];
    keys[j] = (brick_key(bx, by, bz) << 9) | (k & 511ull);
}

// ---- top entries and per-brick prefix tables from the SORTED keys -------------------------------------------
// Used by the full build and by the merge update alike.  (1) a run-length encoding of the keys' brick parts gives the
// occupied bricks in key order -- their keys, their point counts, their number; (2) an exclusive scan of the counts gives
// every brick's first position; (3) every brick takes its slot of the (zeroed) top array; (4) one wave per occupied brick
// reads the brick's keys once, coalesced, drops the first position of every cell into a 512-entry LDS table, turns it into
// the exclusive prefix of the cell counts (an empty cell takes the start of the next non-empty one), and derives the row
// mask and the occupied-cell count from it.  No per-point brick id, no global atomics per cell.
struct BrickOfKey {
    __host__ __device__ uint64_t operator()(const uint64_t &k) const { return k >> 9; }
};

constexpr int kOccShards = 64;  // occupied-cell counters, 128 B apart (same-address atomics cost ~11 ns each)

__global__ __launch_bounds__(256) void brick_assign_kernel(const uint32_t *__restrict__ bricks_dev, Grid g, uint4 *__restrict__ top,
                                                           const uint64_t *__restrict__ bkey, uint32_t *__restrict__ occ)
{
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id < kOccShards) occ[id * 32] = 0u;  // the occupied-cell counters brick_table_kernel adds to
    if (id >= (int64_t)*bricks_dev) return;
    top[top_slot_of_key(g, bkey[id])] = make_uint4((uint32_t)id + 1u, 0u, 0u, 0u);
}

// box of the bricks in `bkey` (one workgroup: a few thousand to a few hundred thousand entries) and, when given, of the
// brick parts of the n keys in `nk`; out6 = {lo xyz, hi xyz} as int32 (lo > hi: nothing)
__global__ __launch_bounds__(1024) void brick_box_kernel(const uint32_t *__restrict__ bricks_dev, const uint64_t *__restrict__ bkey,
                                                         const uint32_t *__restrict__ tab, const uint64_t *__restrict__ nk, int n,
                                                         int32_t *__restrict__ out6)
{
    int lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
    auto take = [&](uint64_t bk) {
        int b[3];
        brick_coords(bk, b[0], b[1], b[2]);
#pragma unroll
        for (int k = 0; k < 3; ++k) { lo[k] = min(lo[k], b[k]); hi[k] = max(hi[k], b[k]); }
    };
    const int64_t bricks = bricks_dev ? (int64_t)*bricks_dev : 0;
    for (int64_t id = threadIdx.x; id < bricks; id += blockDim.x)
        if (!tab || tab[id * kBrickStride + kBrickCells] > tab[id * kBrickStride]) take(bkey[id]);  // (bricks that hold points)
    for (int i = threadIdx.x; i < n; i += blockDim.x)
        if (nk[i] != ~0ull) take(nk[i] >> 9);  // (~0: a place behind the device's count of the staged points, slab_key_kernel)
    __shared__ int slo[16][3], shi[16][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[k] = min(lo[k], __shfl_xor(lo[k], off, 64));
            hi[k] = max(hi[k], __shfl_xor(hi[k], off, 64));
        }
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < 3; ++k) { slo[wave][k] = lo[k]; shi[wave][k] = hi[k]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        int l = INT32_MAX, h = INT32_MIN;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { l = min(l, slo[w][threadIdx.x]); h = max(h, shi[w][threadIdx.x]); }
        out6[threadIdx.x] = l;
        out6[3 + threadIdx.x] = h;
    }
}

__global__ void set_word_kernel(uint32_t *dst, uint32_t value) { *dst = value; }
void launch_set_word(uint32_t *dst, uint32_t value, hipStream_t st) { hipLaunchKernelGGL(set_word_kernel, dim3(1), dim3(1), 0, st, dst, value); }

void launch_brick_box(const uint32_t *bricks_dev, const uint64_t *bkey, const uint32_t *tab, const uint64_t *nk, int n, int32_t *out6,
                      hipStream_t st)
{
    hipLaunchKernelGGL(brick_box_kernel, dim3(1), dim3(1024), 0, st, bricks_dev, bkey, tab, nk, n, out6);
}

// bricks_dev: the number of occupied bricks (the host may only know an upper bound for it)
__global__ __launch_bounds__(256) void brick_table_kernel(const uint32_t *__restrict__ bricks_dev, int64_t m, Grid g,
                                                          const uint64_t *__restrict__ keys,
                                                          const uint32_t *__restrict__ bstart, uint4 *__restrict__ top,
                                                          uint32_t *__restrict__ tab, uint32_t *__restrict__ occ,
                                                          uint8_t *__restrict__ bmark, uint32_t *__restrict__ bend)
{
    __shared__ uint32_t lds[4][kBrickCells];
    const int64_t bricks = (int64_t)*bricks_dev;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t id = (int64_t)blockIdx.x * 4 + wave;
    int cells = 0;
    if (id < bricks) {
        uint32_t *t = lds[wave];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[lane * 8 + k] = 0xffffffffu;
        const uint32_t s = bstart[id];
        const uint32_t e = (id + 1 < bricks) ? bstart[id + 1] : (uint32_t)m;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        for (uint32_t j = s + (uint32_t)lane; j < e; j += 64u) {
            const uint64_t k = keys[j];
            if (j == s || keys[j - 1] != k) t[(uint32_t)k & 511u] = j;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        unsigned long long mask;
        cells = table_from_firsts(t, e, tab + id * kBrickStride, lane, mask);
        if (lane == 0) {
            uint32_t *te = reinterpret_cast<uint32_t *>(&top[top_slot_of_key(g, keys[s] >> 9)]);
            te[2] = (uint32_t)mask;
            te[3] = (uint32_t)(mask >> 32);
            bmark[id] = 0;  // no in-place update pending on a fresh layout (slab_update)
            bend[id] = e;   // the brick's stretch ends where the next one starts (dense)
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cells += __shfl_xor(cells, off, 64);
    __shared__ int wsum[4];
    if (lane == 0) wsum[wave] = cells;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int c = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (c) atomicAdd(&occ[(blockIdx.x % kOccShards) * 32], (uint32_t)c);
    }
}

// `need` is the bare requirement; a (re)allocation adds `headroom` elements on top, so that a map that grows a
// little with every scan does not reallocate -- two device-wide syncs and ~100 MB of hipMalloc at 5 M points --
// on every update (round 2 compared the stored capacity against need + headroom: the headroom was never usable)
static int64_t g_map_allocations = 0;  // diagnostic only (s2m_map_update_stats); racy increments are harmless
static int64_t g_map_alloc_bytes = 0;
// S2M_TRACE_ALLOC=1: every (re)allocation of the map / update code is reported on stderr (a device allocation stalls the
// stream for ~0.1-1 ms: in a frame loop each one is a slow frame, and this is how they are found)
static void trace_alloc(const char *what, size_t bytes)
{
    static const bool on = std::getenv("S2M_TRACE_ALLOC") != nullptr;
    g_map_alloc_bytes += (int64_t)bytes;
    if (on) std::fprintf(stderr, "[s2m alloc] %s: %.1f MB (allocation %lld)\n", what, (double)bytes / 1048576.0, (long long)g_map_allocations);
}
hipError_t map_ensure(void **p, int64_t *cap, int64_t need, size_t elem, int64_t headroom)
{
    if (*cap >= need && *p) return hipSuccess;
    if (*p) S2M_TRY(hipFree(*p));
    *p = nullptr;
    *cap = 0;
    const int64_t c = std::max<int64_t>(need + headroom, 1);
    S2M_TRY(hipMalloc(p, (size_t)c * elem));
    *cap = c;
    ++g_map_allocations;
    trace_alloc("map_ensure", (size_t)c * elem);
    return hipSuccess;
}
int64_t map_allocations() { return g_map_allocations; }
int64_t map_allocated_bytes() { return g_map_alloc_bytes; }
void note_allocation(const char *what, size_t bytes) { ++g_map_allocations; trace_alloc(what, bytes); }
// (with room for the slack and the tail a maintained map is laid out with: s2m_mapedit.hip)
int64_t map_headroom_for(int64_t m) { return m + ((int64_t)1 << 20); }

void free_map(MapBuffers &b)
{
    void *ptrs[] = {b.pts, b.pidx, b.pts2, b.pidx2, b.top, b.tab, b.keys, b.keys_alt, b.vals, b.vals_alt, b.work_a, b.work_b,
                    b.work_c, b.top2, b.bstart, b.bkey, b.bmark, b.bend, b.bmove, b.bplan, b.blist, b.grow, b.run, b.mk, b.mv, b.dword, b.sort_tmp, b.bbox, b.counters};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (b.h_stats) { (void)hipHostFree(b.h_stats); (void)hipEventDestroy(b.stats_event); }
    free_mailbox(b.mail);
    b = MapBuffers();
}

hipError_t map_ensure_sort_tmp(MapBuffers &buf, size_t bytes)
{
    if (bytes <= buf.sort_tmp_bytes && buf.sort_tmp) return hipSuccess;
    if (buf.sort_tmp) S2M_TRY(hipFree(buf.sort_tmp));
    buf.sort_tmp = nullptr;
    buf.sort_tmp_bytes = 0;
    const size_t want = std::max<size_t>(2 * bytes, (size_t)1 << 20);  // with room to spare (s2m_mapupd.hip, ensure_tmp)
    S2M_TRY(hipMalloc(&buf.sort_tmp, want));
    ++g_map_allocations;
    trace_alloc("map sort_tmp", want);
    buf.sort_tmp_bytes = want;
    return hipSuccess;
}

// per-point scratch (keys, sorted keys, source indices, three work arrays of cap + 1 words); with headroom, so
// that a map that grows a little with every scan does not reallocate (and fall back to a full rebuild) each time
static hipError_t ensure_scratch(MapBuffers &buf, int64_t m)
{
    if (buf.scratch_cap >= m) return hipSuccess;
    const int64_t cap = m + map_headroom_for(m);
    void **ps[] = {(void **)&buf.keys, (void **)&buf.keys_alt, (void **)&buf.vals, (void **)&buf.vals_alt,
                   (void **)&buf.work_a, (void **)&buf.work_b, (void **)&buf.work_c};
    const size_t es[] = {8, 8, 4, 4, 4, 4, 4};
    buf.scratch_cap = 0;
    for (int k = 0; k < 7; ++k) {
        int64_t c = 0;
        if (*ps[k]) { S2M_TRY(hipFree(*ps[k])); *ps[k] = nullptr; }
        S2M_TRY(map_ensure(ps[k], &c, cap + 1, es[k]));
    }
    buf.scratch_cap = cap;
    return hipSuccess;
}

// The build's sort: rocprim would take a merge sort (a block sort and ~20 merge passes, one or two launches each) for up to
// 2^20 points; a build beside the frames (s2m_engine_relay.cpp) is a third queue whose chain of tiny kernels delays the
// dispatch of the frame's own, so few launches matter more than the last microsecond: the radix passes above 4096 points.  Both are stable.
using BuildSortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 4096>;

// The arrays of `dst` get (at least) the capacities of `src`: the map a layout beside the frames will be built into follows the
// live map's growth at the moment the live map allocates -- a frame that has stalled for its allocations anyway -- instead of
// allocating beside the frames of the next layout (s2m_engine_relay.cpp).  What `dst` held is lost where an array grows: only for
// a map that is not in use and will be built from scratch.
hipError_t map_reserve_like(MapBuffers &dst, const MapBuffers &src, int64_t build_points)
{
    struct Arr { void **p; int64_t *cap; int64_t want; size_t elem; };
    const Arr arrs[] = {
        {(void **)&dst.pts, &dst.pts_cap, src.pts_cap, sizeof(float4)},       {(void **)&dst.pidx, &dst.pidx_cap, src.pidx_cap, sizeof(uint32_t)},
        {(void **)&dst.pts2, &dst.pts2_cap, src.pts2_cap, sizeof(float4)},    {(void **)&dst.pidx2, &dst.pidx2_cap, src.pidx2_cap, sizeof(uint32_t)},
        {(void **)&dst.top, &dst.top_cap, src.top_cap, sizeof(uint4)},        {(void **)&dst.top2, &dst.top2_cap, src.top2_cap, sizeof(uint4)},
        {(void **)&dst.tab, &dst.tab_cap, src.tab_cap, sizeof(uint32_t)},     {(void **)&dst.bstart, &dst.bstart_cap, src.bstart_cap, sizeof(uint32_t)},
        {(void **)&dst.bkey, &dst.bkey_cap, src.bkey_cap, sizeof(uint64_t)},  {(void **)&dst.bmark, &dst.bmark_cap, src.bmark_cap, sizeof(uint8_t)},
        {(void **)&dst.bend, &dst.bend_cap, src.bend_cap, sizeof(uint32_t)},  {(void **)&dst.bmove, &dst.bmove_cap, src.bmove_cap, sizeof(uint32_t)},
        {(void **)&dst.bplan, &dst.bplan_cap, src.bplan_cap, 32},             {(void **)&dst.blist, &dst.blist_cap, src.blist_cap, sizeof(uint32_t)},
        {(void **)&dst.run, &dst.run_cap, src.run_cap, sizeof(uint2)},        {(void **)&dst.grow, &dst.grow_cap, src.grow_cap, sizeof(uint32_t)},
        {(void **)&dst.mk, &dst.mk_cap, src.mk_cap, sizeof(uint64_t)},        {(void **)&dst.mv, &dst.mv_cap, src.mv_cap, sizeof(uint32_t)},
        {(void **)&dst.dword, &dst.dword_cap, src.dword_cap, sizeof(unsigned long long)},
    };
    for (const Arr &a : arrs)
        if (a.want > 0 && *a.cap < a.want) S2M_TRY(map_ensure(a.p, a.cap, a.want, a.elem, 0));
    if (dst.scratch_cap < src.scratch_cap) {
        void **ps[] = {(void **)&dst.keys, (void **)&dst.keys_alt, (void **)&dst.vals, (void **)&dst.vals_alt,
                       (void **)&dst.work_a, (void **)&dst.work_b, (void **)&dst.work_c};
        const size_t es[] = {8, 8, 4, 4, 4, 4, 4};
        dst.scratch_cap = 0;
        for (int k = 0; k < 7; ++k) {
            int64_t c = 0;
            if (*ps[k]) { S2M_TRY(hipFree(*ps[k])); *ps[k] = nullptr; }
            S2M_TRY(map_ensure(ps[k], &c, src.scratch_cap + 1, es[k], 0));
        }
        dst.scratch_cap = src.scratch_cap;
    }
    // the temporary storage a BUILD of build_points points sorts with (the live map may never have been built at its present size:
    // its own storage is what its merges needed)
    size_t want_tmp = src.sort_tmp_bytes;
    if (build_points > 0) {
        size_t t = 0;
        uint64_t *k = nullptr;
        uint32_t *v = nullptr;
        if (rocprim::radix_sort_pairs<BuildSortConfig>(nullptr, t, k, k, v, v, (size_t)build_points, 0, 64u, (hipStream_t) nullptr) == hipSuccess) want_tmp = std::max(want_tmp, t);
    }
    if (dst.sort_tmp_bytes < want_tmp) {   // (exactly as much: the two maps change places, neither may outbid the other)
        if (dst.sort_tmp) S2M_TRY(hipFree(dst.sort_tmp));
        dst.sort_tmp = nullptr;
        dst.sort_tmp_bytes = 0;
        S2M_TRY(hipMalloc(&dst.sort_tmp, want_tmp));
        note_allocation("map sort_tmp (the other map)", want_tmp);
        dst.sort_tmp_bytes = want_tmp;
    }
    return hipSuccess;
}

// The same arrays for a map that has outgrown them in the middle of an update: everything is re-allocated (room for twice the
// need: a map that is driven through keeps growing), the sorted keys of the current map -- the one array an update reads --
// carried over.  A slow frame (allocations stall the stream) instead of a rebuild; the reference's tree allocates per node.
hipError_t map_grow_scratch(MapBuffers &buf, int64_t need, int64_t keys_in_use, hipStream_t st)
{
    if (buf.scratch_cap >= need) return hipSuccess;
    const int64_t cap = 2 * need + ((int64_t)1 << 20);
    uint64_t *old_sorted = buf.keys_alt;
    buf.keys_alt = nullptr;
    void **ps[] = {(void **)&buf.keys, (void **)&buf.keys_alt, (void **)&buf.vals, (void **)&buf.vals_alt,
                   (void **)&buf.work_a, (void **)&buf.work_b, (void **)&buf.work_c};
    const size_t es[] = {8, 8, 4, 4, 4, 4, 4};
    S2M_TRY(wait_stream(nullptr, st, "the stream, before the map's scratch arrays are reallocated"));  // (kernels in flight may still read them)
    for (int k = 0; k < 7; ++k) {
        int64_t c = 0;
        if (*ps[k]) { S2M_TRY(hipFree(*ps[k])); *ps[k] = nullptr; }
        S2M_TRY(map_ensure(ps[k], &c, cap + 1, es[k]));
    }
    if (old_sorted) {
        if (keys_in_use > 0) S2M_TRY(hipMemcpyAsync(buf.keys_alt, old_sorted, (size_t)keys_in_use * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
        S2M_TRY(wait_stream(nullptr, st, "the copy of the sorted keys into the grown scratch"));
        S2M_TRY(hipFree(old_sorted));
    }
    buf.scratch_cap = cap;
    return hipSuccess;
}

hipError_t map_put_sentinels(float4 *pts, int64_t m, hipStream_t st)
{
    // kSentinelPoints extra elements: the sentinel points the search kernels load for the padding slots of a batch
    // (far enough for the squared distance to overflow to +inf; index word 0xffffffff)
    struct Block {
        uint32_t w[kSentinelPoints][4];
        Block() { for (auto &q : w) { q[0] = q[1] = q[3] = 0x7f61b1e6u; q[2] = 0xffffffffu; } }  // make_map_point(3e38f x3, ~0)
    };
    static const Block sentinel;  // built once (thread-safe static initialisation), read-only afterwards
    return hipMemcpyAsync(pts + m, sentinel.w, sizeof(sentinel.w), hipMemcpyHostToDevice, st);
}

__global__ void stats_mail_kernel(const uint32_t *__restrict__ bricks, const uint32_t *__restrict__ occ, uint32_t *__restrict__ out)
{
    const int i = threadIdx.x;
    if (i == 0) __hip_atomic_store(out, *bricks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if (i <= kOccShards) __hip_atomic_store(out + i, occ[(i - 1) * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// occupied-brick and occupied-cell counts of a merge update travel to pinned host memory behind the kernels and
// are read when somebody asks (resolve_stats): the update itself does not wait for them
hipError_t resolve_stats(MapBuffers &buf, MapStats &stats)
{
    if (!buf.stats_pending) return hipSuccess;
    S2M_TRY(wait_event(nullptr, buf.stats_event, "the counts of the last merge"));
    buf.stats_pending = false;
    int64_t cells = 0;
    for (int k = 0; k < kOccShards; ++k) cells += buf.h_stats[1 + k];
    stats.bricks = buf.h_stats[0];
    stats.occupied_cells = cells;
    return hipSuccess;
}

// ---- the window of the top array ---------------------------------------------------------------------------
static int log2_ceil(int64_t v)
{
    int l = 0;
    while (((int64_t)1 << l) < v) ++l;
    return l;
}
// The box of bricks [lo, hi] becomes the grid's bounds.  The toroidal top array keeps its sizes when the box fits them
// (nothing moves: a slot depends on the brick's coordinates and the sizes only); otherwise every axis gets the next power of
// two above its extent plus half of it (at least 8 bricks more), so that the box can wander that far before the next
// re-lay.  *resized says which.  An empty box (hi < lo) gets a one-entry array.  Sets the fields of g only.
hipError_t map_window_for(Grid &g, const int lo[3], const int hi[3], bool &too_large, bool *resized)
{
    too_large = false;
    if (resized) *resized = false;
    int64_t ext[3];
    for (int k = 0; k < 3; ++k) {
        ext[k] = std::max<int64_t>((int64_t)hi[k] - lo[k] + 1, 0);
        if (ext[k] > ((int64_t)1 << kBrickBits)) { too_large = true; return hipSuccess; }
    }
    const bool fits = g.top != nullptr && ext[0] <= (int64_t)g.tmx + 1 && ext[1] <= (int64_t)g.tmy + 1 && ext[2] <= (int64_t)g.tmz + 1;
    if (!fits) {
        int lg[3];
        for (int k = 0; k < 3; ++k) lg[k] = ext[k] > 0 ? log2_ceil(ext[k] + std::max<int64_t>(ext[k] / 2, 8)) : 0;
        // a box that is thin along some axes and long along others must not pay for slack it cannot afford
        while (lg[0] + lg[1] + lg[2] > 28) {
            int k = 0;
            for (int j = 1; j < 3; ++j)
                if (((int64_t)1 << lg[j]) - ext[j] > ((int64_t)1 << lg[k]) - ext[k]) k = j;
            if (lg[k] == 0 || ((int64_t)1 << (lg[k] - 1)) < ext[k]) { too_large = true; return hipSuccess; }
            --lg[k];
        }
        g.tmx = (1u << lg[0]) - 1u; g.tmy = (1u << lg[1]) - 1u; g.tmz = (1u << lg[2]) - 1u;
        g.tsy = (uint32_t)lg[0]; g.tsz = (uint32_t)(lg[0] + lg[1]);
        if (resized) *resized = true;
    }
    for (int k = 0; k < 3; ++k) { g.blo[k] = lo[k]; g.bhi[k] = hi[k]; }
    return hipSuccess;
}
// the same for the map's own array, which the caller is about to fill from scratch: (re)allocated when the sizes change
hipError_t map_set_window(MapBuffers &buf, Grid &g, const int lo[3], const int hi[3], bool &too_large, bool *resized, hipStream_t st)
{
    bool r = false;
    if (g.top != buf.top || !buf.top) g.top = nullptr;  // no array yet (or not ours): sizes are chosen
    S2M_TRY(map_window_for(g, lo, hi, too_large, &r));
    if (resized) *resized = r;
    if (too_large) return hipSuccess;
    const int64_t slots = top_slots(g);
    if (r || buf.top_cap < slots + 1 || buf.top2_cap < slots + 1 || buf.grow_cap < slots + 1) {
        // (with room for the window to double twice -- a map that is driven through grows along the drive until the
        // field-of-view trim bounds it -- and the spare a re-lay writes: no allocation in a frame)
        S2M_TRY(map_ensure((void **)&buf.top, &buf.top_cap, slots + 1, sizeof(uint4), 3 * slots));
        S2M_TRY(map_ensure((void **)&buf.top2, &buf.top2_cap, slots + 1, sizeof(uint4), 3 * slots));
        S2M_TRY(map_ensure((void **)&buf.grow, &buf.grow_cap, slots + 1, sizeof(uint32_t), 3 * slots));
        // (the growth history is kept per slot: it does not survive a change of the slots)
        S2M_TRY(hipMemsetAsync(buf.grow, 0, (size_t)(slots + 1) * sizeof(uint32_t), st));
    }
    g.top = buf.top;
    return hipSuccess;
}

// top entries + brick tables of the m points whose sorted (box-independent) keys are `keys`; buf.keys is free scratch.
// brick_bound >= 0: an upper bound of the number of occupied bricks known to the host -- nothing is read back
// before the tables are built, and the counts arrive later (resolve_stats).  find_bounds: the grid's bounds (and, if they
// do not fit it, the top array's sizes) are taken from the bricks found -- one hand-back; otherwise g's bounds hold them.
hipError_t map_build_tables(MapBuffers &buf, Grid &g, const uint64_t *keys, int64_t m, MapStats &stats, hipStream_t st,
                            int64_t brick_bound, bool find_bounds, bool *too_large)
{
    if (too_large) *too_large = false;
    uint32_t *bricks_dev = buf.counters + kBricksWord;
    uint64_t *ukeys = buf.keys;        // the bricks in key order (scratch: as many as there are points at most)
    uint32_t *ucount = buf.work_a;     // their point counts
    auto brick_of = rocprim::make_transform_iterator(keys, BrickOfKey());
    size_t tmp = 0;
    S2M_TRY(rocprim::run_length_encode(nullptr, tmp, brick_of, (unsigned int)m, ukeys, ucount, bricks_dev, st));
    S2M_TRY(map_ensure_sort_tmp(buf, tmp));
    size_t t1 = buf.sort_tmp_bytes;
    S2M_TRY(rocprim::run_length_encode(buf.sort_tmp, t1, brick_of, (unsigned int)m, ukeys, ucount, bricks_dev, st));
    const bool lazy = brick_bound >= 0 && !find_bounds;
    int64_t bricks = brick_bound;
    if (find_bounds) {
        int32_t *box = reinterpret_cast<int32_t *>(buf.counters + kBoxWords);
        hipLaunchKernelGGL(brick_box_kernel, dim3(1), dim3(1024), 0, st, bricks_dev, ukeys, (const uint32_t *)nullptr,
                           (const uint64_t *)nullptr, 0, box);
        const uint32_t *src[7] = {bricks_dev, buf.counters + kBoxWords, buf.counters + kBoxWords + 1, buf.counters + kBoxWords + 2,
                                  buf.counters + kBoxWords + 3, buf.counters + kBoxWords + 4, buf.counters + kBoxWords + 5};
        uint32_t v[7];
        S2M_TRY(mail_fetch(buf.mail, src, 7, v, st));
        bricks = v[0];
        int lo[3], hi[3];
        for (int k = 0; k < 3; ++k) { lo[k] = (int32_t)v[1 + k]; hi[k] = (int32_t)v[4 + k]; }
        if (bricks == 0) { lo[0] = lo[1] = lo[2] = 0; hi[0] = hi[1] = hi[2] = -1; }
        bool big = false;
        S2M_TRY(map_set_window(buf, g, lo, hi, big, nullptr, st));
        if (big) {
            if (too_large) *too_large = true;
            return hipSuccess;
        }
    } else if (!lazy) {
        uint32_t b32 = 0;
        const uint32_t *src[1] = {bricks_dev};
        S2M_TRY(mail_fetch(buf.mail, src, 1, &b32, st));
        bricks = b32;
    }
    const int64_t slots = top_slots(g);
    S2M_TRY(hipMemsetAsync(buf.top, 0, (size_t)(slots + 1) * sizeof(uint4), st));
    // (rows to spare for the bricks that in-place updates open until the next merge: a moving sensor opens dozens per frame)
    const int64_t spare = bricks / 2 + 16384;
    S2M_TRY(map_ensure((void **)&buf.tab, &buf.tab_cap, bricks * kBrickStride, sizeof(uint32_t), spare * kBrickStride));
    S2M_TRY(map_ensure((void **)&buf.bstart, &buf.bstart_cap, bricks + 1, sizeof(uint32_t), spare));
    S2M_TRY(map_ensure((void **)&buf.bkey, &buf.bkey_cap, bricks, sizeof(uint64_t), spare));
    S2M_TRY(map_ensure((void **)&buf.bmark, &buf.bmark_cap, bricks, sizeof(uint8_t), spare));
    S2M_TRY(map_ensure((void **)&buf.bend, &buf.bend_cap, bricks, sizeof(uint32_t), spare));
    S2M_TRY(map_ensure((void **)&buf.bmove, &buf.bmove_cap, bricks, sizeof(uint32_t), spare));
    if (bricks > 0) {
        // first position of every brick (entries beyond the actual number, in the lazy case, are never read)
        size_t ts = 0;
        S2M_TRY(rocprim::exclusive_scan(nullptr, ts, ucount, buf.bstart, 0u, (size_t)bricks, rocprim::plus<uint32_t>(), st));
        S2M_TRY(map_ensure_sort_tmp(buf, ts));
        size_t t2 = buf.sort_tmp_bytes;
        S2M_TRY(rocprim::exclusive_scan(buf.sort_tmp, t2, ucount, buf.bstart, 0u, (size_t)bricks, rocprim::plus<uint32_t>(), st));
        S2M_TRY(hipMemcpyAsync(buf.bkey, ukeys, (size_t)bricks * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
    }
    hipLaunchKernelGGL(brick_assign_kernel, dim3((unsigned)((std::max<int64_t>(bricks, kOccShards) + 255) / 256)), dim3(256), 0, st,
                       bricks_dev, g, buf.top, ukeys, buf.counters + 64);
    if (bricks > 0)
        hipLaunchKernelGGL(brick_table_kernel, dim3((unsigned)((bricks + 3) / 4)), dim3(256), 0, st, bricks_dev, m, g, keys,
                           buf.bstart, buf.top, buf.tab, buf.counters + 64, buf.bmark, buf.bend);
    ++buf.layout_gen;  // a fresh dense layout: every position below m holds a point, and there is no tail
    buf.main_ext = m;
    buf.tail_used = 0;
    launch_set_word(buf.counters + kTailWord, (uint32_t)m, st);
    if (!buf.h_stats) {
        S2M_TRY(hipHostMalloc((void **)&buf.h_stats, (1 + kOccShards) * sizeof(uint32_t), hipHostMallocMapped));
        S2M_TRY(hipHostGetDevicePointer((void **)&buf.h_stats_dev, buf.h_stats, 0));
        S2M_TRY(hipEventCreateWithFlags(&buf.stats_event, hipEventDisableTiming));
    }
    hipLaunchKernelGGL(stats_mail_kernel, dim3(1), dim3(128), 0, st, bricks_dev, buf.counters + 64, buf.h_stats_dev);
    S2M_TRY(hipEventRecord(buf.stats_event, st));
    buf.stats_pending = true;
    stats.layout_points = m;
    stats.top_entries = slots;
    if (!lazy) {
        S2M_TRY(resolve_stats(buf, stats));
        S2M_TRY(hipGetLastError());
    } else {
        stats.bricks = bricks;  // the bound, until the counts have arrived
    }
    g.top = buf.top; g.tab = buf.tab;
    return hipSuccess;
}

// Would a build of a cloud with the box [lo, hi] at this cell size (around keep_origin, or around an origin of its own) find the
// box representable?  The checks of build_once / map_window_for without touching anything: an update that has to fall back
// to a rebuild asks BEFORE the map it holds is given up (one absurd coordinate in a scan must not cost a node its map).
bool map_build_would_fit(const float lo[3], const float hi[3], float cell, const float *keep_origin)
{
    if (!(cell > 0.0f)) return true;   // (chosen from the density by the build itself)
    const float shift[3] = {0.37f, 0.41f, 0.29f};
    const float inv_c = 1.0f / cell;
    int blo[3], bhi[3];
    double box_bricks = 1.0;
    for (int k = 0; k < 3; ++k) {
        const float o = keep_origin ? keep_origin[k] : std::floor(lo[k] / cell) * cell - cell - shift[k] * cell;
        const int c0 = cell_coord(lo[k], o, inv_c), c1 = cell_coord(hi[k], o, inv_c);
        if (!(lo[k] <= hi[k]) || c0 <= -kCellLimit || c1 >= kCellLimit) return false;
        blo[k] = c0 >> 3; bhi[k] = c1 >> 3;
        box_bricks *= (double)std::max(bhi[k] - blo[k] + 1, 1);
    }
    if (box_bricks >= 9.0e15) return false;
    Grid g;
    bool too_large = false;
    if (map_window_for(g, blo, bhi, too_large, nullptr) != hipSuccess) return false;
    return !too_large;
}

// The grid's origin is chosen by the FIRST build of a map (keep_origin == nullptr) and kept by every later one: a cell's
// box-independent coordinates -- and with them the order of the points -- never change while the cell size stands.
static hipError_t build_once(const float *xyz, int64_t stride, int64_t m, float cell, const float lo[3],
                             const float hi[3], const float *keep_origin, MapBuffers &buf, Grid &g, MapStats &stats,
                             bool &too_large, hipStream_t st)
{
    too_large = false;
    const uint4 *old_top = g.top;
    const uint32_t otm[3] = {g.tmx, g.tmy, g.tmz}, ots[2] = {g.tsy, g.tsz};
    g = Grid();
    // (the window of a previous map on the same buffers is kept when the new box fits it: no reallocation on a rebuild)
    g.top = old_top; g.tmx = otm[0]; g.tmy = otm[1]; g.tmz = otm[2]; g.tsy = ots[0]; g.tsz = ots[1];
    g.c = cell;
    g.inv_c = 1.0f / cell;
    g.m = m;
    g.live = m;
    g.sent_off = (m + kSentinelPoints) < ((int64_t)1 << 28) ? (uint32_t)(m << 4) : 0u;
    // The origin sits one cell (and an odd fraction of a cell) below the first cloud's box: man-made scenes have planes at
    // round coordinates, and a plane that coincides with a cell face splits its points over two cell layers and leaves
    // every query on it with zero margin to the face (measured 2x slower searches).
    const float shift[3] = {0.37f, 0.41f, 0.29f};
    float o[3];
    for (int k = 0; k < 3; ++k) o[k] = keep_origin ? keep_origin[k] : std::floor(lo[k] / cell) * cell - cell - shift[k] * cell;
    g.ox = o[0]; g.oy = o[1]; g.oz = o[2];
    // the termination bound of the search works with in-cell positions rounded to float and with the float reciprocal
    // of the cell size: both errors are relative (cell_pos, s2m_device.h), far below this margin wherever the map lies
    g.slop = 1.0e-4f;
    BuildBox bb;
    int blo[3] = {0, 0, 0}, bhi[3] = {-1, -1, -1};
    for (int k = 0; k < 3 && m > 0; ++k) {
        const int c0 = cell_coord(lo[k], o[k], g.inv_c), c1 = cell_coord(hi[k], o[k], g.inv_c);  // (monotone: the extreme cells)
        if (c0 <= -kCellLimit || c1 >= kCellLimit) { too_large = true; return hipSuccess; }
        blo[k] = c0 >> 3; bhi[k] = c1 >> 3;
    }
    for (int k = 0; k < 3; ++k) { bb.lo[k] = blo[k]; bb.n[k] = std::max(bhi[k] - blo[k] + 1, 1); }
    const double box_bricks = (double)bb.n[0] * (double)bb.n[1] * (double)bb.n[2];
    if (box_bricks >= 9.0e15) { too_large = true; return hipSuccess; }  // (brick << 9 | cell) must fit 64 bits

    S2M_TRY(map_ensure((void **)&buf.pts, &buf.pts_cap, m + kSentinelPoints, sizeof(float4), map_headroom_for(m)));
    S2M_TRY(map_put_sentinels(buf.pts, m, st));
    S2M_TRY(map_ensure((void **)&buf.pidx, &buf.pidx_cap, m + 1, sizeof(uint32_t), map_headroom_for(m)));
    bool resized = false;
    S2M_TRY(map_set_window(buf, g, blo, bhi, too_large, &resized, st));
    if (too_large) return hipSuccess;
    S2M_TRY(ensure_scratch(buf, m));
    if (!buf.counters) S2M_TRY(hipMalloc((void **)&buf.counters, (64 + kOccShards * 32) * sizeof(uint32_t)));
    // net growth of every brick of this grid since its room was last laid out (s2m_mapedit.hip): a new grid starts from zero
    S2M_TRY(hipMemsetAsync(buf.grow, 0, (size_t)(top_slots(g) + 1) * sizeof(uint32_t), st));
    buf.added_since_layout = 0;
    S2M_TRY(hipMemsetAsync(buf.counters, 0, 64 * sizeof(uint32_t), st));
    stats = MapStats();
    stats.top_entries = top_slots(g);
    g.top = buf.top; g.tab = buf.tab; g.pts = buf.pts; g.pidx = buf.pidx;
    if (m == 0) {
        S2M_TRY(hipMemsetAsync(buf.top, 0, (size_t)(top_slots(g) + 1) * sizeof(uint4), st));
        return wait_stream(nullptr, st, "the build of an empty map");
    }

    const int blocks = (int)((m + 255) / 256);
    hipLaunchKernelGGL(key_kernel, dim3(blocks), dim3(256), 0, st, xyz, stride, m, g, bb, buf.keys, buf.vals);
    int bits = 9 + log2_ceil((int64_t)box_bricks);
    bits = std::min(bits + 1, 64);
    size_t tmp = 0;
    S2M_TRY(rocprim::radix_sort_pairs<BuildSortConfig>(nullptr, tmp, buf.keys, buf.keys_alt, buf.vals, buf.vals_alt, (size_t)m, 0,
                                      (unsigned)bits, st));
    S2M_TRY(map_ensure_sort_tmp(buf, tmp));
    size_t t1 = buf.sort_tmp_bytes;
    S2M_TRY(rocprim::radix_sort_pairs<BuildSortConfig>(buf.sort_tmp, t1, buf.keys, buf.keys_alt, buf.vals, buf.vals_alt, (size_t)m,
                                      0, (unsigned)bits, st));
    hipLaunchKernelGGL(gather_kernel, dim3(blocks), dim3(256), 0, st, xyz, stride, m, buf.vals_alt, bb, buf.pts, buf.pidx, buf.keys_alt);
    buf.next_id = m;        // the ids of a fresh build are the caller's indices
    buf.ids_dense = true;
    S2M_TRY(map_build_tables(buf, g, buf.keys_alt, m, stats, st, -1, false, nullptr));
    g.top = buf.top; g.tab = buf.tab; g.pts = buf.pts; g.pidx = buf.pidx;
    return hipSuccess;
}

hipError_t build_map(const float *xyz, int64_t stride, int64_t m, float cell, MapBuffers &buf, Grid &grid,
                     MapStats &stats, bool &too_large, hipStream_t st, const float *keep_origin)
{
    float lo[3] = {0.f, 0.f, 0.f}, hi[3] = {0.f, 0.f, 0.f};
    if (m > 0) {
        if (!buf.bbox) S2M_TRY(hipMalloc((void **)&buf.bbox, kBboxScratchFloats * sizeof(float)));
        S2M_TRY(cloud_bbox(xyz, stride, m, buf.bbox, buf.mail, lo, hi, st));
    }
    if (cell > 0.0f) return build_once(xyz, stride, m, cell, lo, hi, keep_origin, buf, grid, stats, too_large, st);
    // density-driven cell size: LiDAR maps are surfaces, so points per occupied cell ~ c^2; aim for
    // ~11 points per occupied cell (cell edge about twice the 5-NN radius): measured fastest on
    // MI355X for the first-shell + hard-list search, including the large-displacement first pass
    float c = 0.5f;
    for (int attempt = 0; attempt < 4; ++attempt) {
        S2M_TRY(build_once(xyz, stride, m, c, lo, hi, nullptr, buf, grid, stats, too_large, st));
        if (too_large) { c *= 2.0f; continue; }
        if (m == 0 || stats.occupied_cells == 0) return hipSuccess;
        const double mean = (double)m / (double)stats.occupied_cells;
        if (mean >= 8.5 && mean <= 14.0) return hipSuccess;
        float cn = c * (float)std::sqrt(11.0 / mean);
        cn = std::min(std::max(cn, 0.02f), 64.0f);
        if (std::fabs(cn - c) < 0.05f * c) return hipSuccess;
        c = cn;
    }
    return build_once(xyz, stride, m, c, lo, hi, nullptr, buf, grid, stats, too_large, st);
}

// AoS (caller stride) -> SoA scan arrays; feats_down keeps only x, y, z on this path
__global__ __launch_bounds__(256) void deinterleave_kernel(const float *__restrict__ src, int64_t stride, int64_t n,
                                                           float *__restrict__ sx, float *__restrict__ sy,
                                                           float *__restrict__ sz)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    sx[i] = src[i * stride];
    sy[i] = src[i * stride + 1];
    sz[i] = src[i * stride + 2];
}

// per-scan state of a new scan in one launch: point_selected_surf = true (laserMapping.cpp:812), nothing effective,
// no flags (three memsets are six launches on this stack)
__global__ __launch_bounds__(256) void scan_reset_kernel(int64_t n, uint8_t *__restrict__ sel, uint8_t *__restrict__ eff,
                                                         uint8_t *__restrict__ flags)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { sel[i] = 1; eff[i] = 0; flags[i] = 0; }
}

void launch_scan_reset(int64_t n, uint8_t *sel, uint8_t *eff, uint8_t *flags, hipStream_t st)
{
    const int64_t m = std::max<int64_t>(n, 1);
    hipLaunchKernelGGL(scan_reset_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, m, sel, eff, flags);
}

void launch_deinterleave(const float *src, int64_t stride, int64_t n, float *sx, float *sy, float *sz, hipStream_t st)
{
    const int blocks = (int)((n + 255) / 256);
    if (blocks > 0) hipLaunchKernelGGL(deinterleave_kernel, dim3(blocks), dim3(256), 0, st, src, stride, n, sx, sy, sz);
}

}  // namespace s2m
