// s2m_map.hip -- builds the brick-grid map (layout: s2m_device.h) from an unordered point cloud.
//
// Replaces the map seed of the reference: ikdtree.Build(feats_down_world->points)
// (eskf_lio/src/laserMapping.cpp:784-790; KD_TREE::Build / BuildTree,
// eskf_lio/include/ikd-Tree/ikd_Tree.cpp:408-423, 678-733).  Instead of a pointer tree of 176-byte
// nodes the points are radix-sorted by (brick, cell-in-brick; stable, i.e. then by caller index) into one float4
// array whose third word is the sorted position itself, with the caller indices beside it (pidx); a dense
// top-level array over the bounding box and a 513-entry prefix table per occupied brick locate
// any run of cells along x with two 4-byte loads.
//
// This is the per-map part of the path, not the per-iteration hot loop: the full build (radix sort of all points) and
// the tables that both the build and a merged update derive from the sorted keys; how the map changes after a scan is
// s2m_mapedit.hip.  The device-wide radix sort and scans come from rocPRIM (ROCm's native primitives library),
// everything else is hand-written below.
#include <algorithm>
#include <cmath>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
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

__global__ __launch_bounds__(256) void bbox_final_kernel(const float *__restrict__ partial, int blocks,
                                                         float *__restrict__ out6)
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
}

hipError_t cloud_bbox(const float *xyz, int64_t stride, int64_t n, float *scratch, Mailbox &mail, float lo[3], float hi[3],
                      hipStream_t st)
{
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, kBboxBlocks);
    float *out = scratch + (int64_t)kBboxBlocks * 6;
    hipLaunchKernelGGL(bbox_partial_kernel, dim3(blocks), dim3(256), 0, st, xyz, stride, n, scratch);
    hipLaunchKernelGGL(bbox_final_kernel, dim3(1), dim3(256), 0, st, scratch, blocks, out);
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

__global__ __launch_bounds__(256) void key_kernel(const float *__restrict__ xyz, int64_t stride, int64_t m, Grid g,
                                                  uint64_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int cx = cell_of(xyz[i * stride + 0], g.ox, g.inv_c, g.ncx);
    const int cy = cell_of(xyz[i * stride + 1], g.oy, g.inv_c, g.ncy);
    const int cz = cell_of(xyz[i * stride + 2], g.oz, g.inv_c, g.ncz);
    const uint64_t brick = ((uint64_t)(cz >> 3) * g.nby + (cy >> 3)) * g.nbx + (cx >> 3);
    const uint32_t local = (uint32_t)((((cz & 7) << 3) | (cy & 7)) << 3 | (cx & 7));
    keys[i] = (brick << 9) | local;
    vals[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void gather_kernel(const float *__restrict__ xyz, int64_t stride, int64_t m,
                                                     const uint32_t *__restrict__ vals, float4 *__restrict__ pts,
                                                     uint32_t *__restrict__ pidx)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const uint32_t src = vals[j];
    pts[j] = make_map_point(xyz[(int64_t)src * stride], xyz[(int64_t)src * stride + 1], xyz[(int64_t)src * stride + 2],
                            (uint32_t)j);
    pidx[j] = src;
}

// ---- top entries and per-brick prefix tables from the SORTED keys -------------------------------------------
// Used by the full build and by the merge update alike.  (1) every first point of a brick leaves its position in
// the second word of the brick's top entry; (2) an exclusive scan over the dense top array numbers the occupied
// bricks in key order; (3) ids and brick starts are written; (4) one wave per occupied brick reads the brick's
// keys once, coalesced, drops the first position of every cell into a 512-entry LDS table, turns it into the
// exclusive prefix of the cell counts (an empty cell takes the start of the next non-empty one), and derives the
// row mask and the occupied-cell count from it.  No per-point brick id, no global atomics per cell.
__global__ __launch_bounds__(256) void brick_head_kernel(int64_t m, const uint64_t *__restrict__ keys, uint4 *__restrict__ top)
{
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t k = keys[j];
        if (j == 0 || (keys[j - 1] >> 9) != (k >> 9)) reinterpret_cast<uint32_t *>(&top[k >> 9])[1] = (uint32_t)(j + 1);
    }
}

constexpr int kOccShards = 64;  // occupied-cell counters, 128 B apart (same-address atomics cost ~11 ns each)

struct TopOccupied {
    __host__ __device__ uint32_t operator()(const uint4 &t) const { return t.y != 0u ? 1u : 0u; }
};

__global__ __launch_bounds__(256) void brick_assign_kernel(int64_t top_entries, uint4 *__restrict__ top,
                                                           const uint32_t *__restrict__ rank,
                                                           uint32_t *__restrict__ bstart, uint32_t *__restrict__ bkey,
                                                           uint32_t *__restrict__ occ)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b < kOccShards) occ[b * 32] = 0u;  // the occupied-cell counters brick_table_kernel adds to
    if (b >= top_entries) return;
    const uint32_t y = top[b].y;
    if (y == 0u) return;
    const uint32_t id = rank[b];
    reinterpret_cast<uint32_t *>(&top[b])[0] = id + 1u;
    bstart[id] = y - 1u;
    bkey[id] = (uint32_t)b;  // the brick's place in the top array = the high part of its points' keys
}


// bricks_dev: the number of occupied bricks when the host only knows an upper bound for it (merge update)
__global__ __launch_bounds__(256) void brick_table_kernel(int64_t bricks, const uint32_t *__restrict__ bricks_dev, int64_t m,
                                                          const uint64_t *__restrict__ keys,
                                                          const uint32_t *__restrict__ bstart, uint4 *__restrict__ top,
                                                          uint32_t *__restrict__ tab, uint32_t *__restrict__ occ,
                                                          uint8_t *__restrict__ bmark, uint32_t *__restrict__ bend)
{
    __shared__ uint32_t lds[4][kBrickCells];
    if (bricks_dev) bricks = (int64_t)*bricks_dev;
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
            uint32_t *te = reinterpret_cast<uint32_t *>(&top[keys[s] >> 9]);
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
    return hipSuccess;
}
int64_t map_allocations() { return g_map_allocations; }
void note_allocation() { ++g_map_allocations; }
int64_t map_headroom_for(int64_t m) { return m / 4 + 65536; }

void free_map(MapBuffers &b)
{
    void *ptrs[] = {b.pts, b.pidx, b.pts2, b.pidx2, b.top, b.tab, b.keys, b.keys_alt, b.vals, b.vals_alt, b.work_a, b.work_b,
                    b.work_c, b.rank, b.bstart, b.bkey, b.bmark, b.bend, b.grow, b.mk, b.mv, b.dword, b.sort_tmp, b.bbox, b.counters};
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
    S2M_TRY(hipEventSynchronize(buf.stats_event));
    buf.stats_pending = false;
    int64_t cells = 0;
    for (int k = 0; k < kOccShards; ++k) cells += buf.h_stats[1 + k];
    stats.bricks = buf.h_stats[0];
    stats.occupied_cells = cells;
    return hipSuccess;
}

// top entries + brick tables of the m points whose sorted keys are `keys` (buf.top zeroed by the caller).
// brick_bound >= 0: an upper bound of the number of occupied bricks known to the host -- nothing is read back
// before the tables are built, and the counts arrive later (resolve_stats).
hipError_t map_build_tables(MapBuffers &buf, const uint64_t *keys, int64_t m, int64_t top_entries, MapStats &stats,
                            hipStream_t st, int64_t brick_bound)
{
    const int blocks = (int)std::min<int64_t>((m + 255) / 256, 4096);
    hipLaunchKernelGGL(brick_head_kernel, dim3(blocks), dim3(256), 0, st, m, keys, buf.top);
    S2M_TRY(map_ensure((void **)&buf.rank, &buf.rank_cap, top_entries + 1, sizeof(uint32_t)));
    auto occupied = rocprim::make_transform_iterator(static_cast<const uint4 *>(buf.top), TopOccupied());
    size_t tmp = 0;
    S2M_TRY(rocprim::exclusive_scan(nullptr, tmp, occupied, buf.rank, 0u, (size_t)top_entries + 1, rocprim::plus<uint32_t>(), st));
    S2M_TRY(map_ensure_sort_tmp(buf, tmp));
    size_t t1 = buf.sort_tmp_bytes;
    // one element past the end (top has a zero spare entry): rank[top_entries] = number of occupied bricks
    S2M_TRY(rocprim::exclusive_scan(buf.sort_tmp, t1, occupied, buf.rank, 0u, (size_t)top_entries + 1, rocprim::plus<uint32_t>(), st));
    const bool lazy = brick_bound >= 0;
    int64_t bricks = brick_bound;
    if (!lazy) {
        uint32_t b32 = 0;
        const uint32_t *src[1] = {buf.rank + top_entries};
        S2M_TRY(mail_fetch(buf.mail, src, 1, &b32, st));
        bricks = b32;
    }
    S2M_TRY(map_ensure((void **)&buf.tab, &buf.tab_cap, bricks * kBrickStride, sizeof(uint32_t), (bricks / 4 + 64) * kBrickStride));
    S2M_TRY(map_ensure((void **)&buf.bstart, &buf.bstart_cap, bricks, sizeof(uint32_t), bricks / 4 + 64));
    S2M_TRY(map_ensure((void **)&buf.bkey, &buf.bkey_cap, bricks, sizeof(uint32_t), bricks / 4 + 64));
    S2M_TRY(map_ensure((void **)&buf.bmark, &buf.bmark_cap, bricks, sizeof(uint8_t), bricks / 4 + 64));
    S2M_TRY(map_ensure((void **)&buf.bend, &buf.bend_cap, bricks, sizeof(uint32_t), bricks / 4 + 64));
    hipLaunchKernelGGL(brick_assign_kernel, dim3((unsigned)((top_entries + 255) / 256)), dim3(256), 0, st, top_entries,
                       buf.top, buf.rank, buf.bstart, buf.bkey, buf.counters + 64);
    if (bricks > 0)
        hipLaunchKernelGGL(brick_table_kernel, dim3((unsigned)((bricks + 3) / 4)), dim3(256), 0, st, bricks,
                           lazy ? buf.rank + top_entries : (const uint32_t *)nullptr, m, keys, buf.bstart, buf.top, buf.tab,
                           buf.counters + 64, buf.bmark, buf.bend);
    ++buf.layout_gen;  // a fresh dense layout: every position below m holds a point
    if (!buf.h_stats) {
        S2M_TRY(hipHostMalloc((void **)&buf.h_stats, (1 + kOccShards) * sizeof(uint32_t), hipHostMallocMapped));
        S2M_TRY(hipHostGetDevicePointer((void **)&buf.h_stats_dev, buf.h_stats, 0));
        S2M_TRY(hipEventCreateWithFlags(&buf.stats_event, hipEventDisableTiming));
    }
    hipLaunchKernelGGL(stats_mail_kernel, dim3(1), dim3(128), 0, st, buf.rank + top_entries, buf.counters + 64, buf.h_stats_dev);
    S2M_TRY(hipEventRecord(buf.stats_event, st));
    buf.stats_pending = true;
    if (!lazy) {
        S2M_TRY(resolve_stats(buf, stats));
        S2M_TRY(hipGetLastError());
    } else {
        stats.bricks = bricks;  // the bound, until the counts have arrived
    }
    return hipSuccess;
}

// margin_cells: free cells kept around the bounding box on every side (rounded up to whole bricks by the
// caller's arithmetic below); a map that is maintained incrementally gets one so that points just outside the
// current box can be merged in without a new grid
static hipError_t build_once(const float *xyz, int64_t stride, int64_t m, float cell, const float lo[3],
                             const float hi[3], int margin_cells, MapBuffers &buf, Grid &g, MapStats &stats,
                             bool &too_large, hipStream_t st)
{
    too_large = false;
    g = Grid();
    g.c = cell;
    g.inv_c = 1.0f / cell;
    g.m = m;
    g.live = m;
    g.sent_off = (m + kSentinelPoints) < ((int64_t)1 << 28) ? (uint32_t)(m << 4) : 0u;
    // one cell of padding below the box; extents rounded up to whole bricks.  The origin is shifted
    // by an odd fraction of a cell per axis: man-made scenes have planes at round coordinates, and a
    // plane that coincides with a cell face splits its points over two cell layers and leaves every
    // query on it with zero margin to the face (measured 2x slower searches).
    const float shift[3] = {0.37f, 0.41f, 0.29f};
    int nc[3];
    float o[3];
    for (int k = 0; k < 3; ++k) {
        o[k] = std::floor(lo[k] / cell) * cell - cell - shift[k] * cell - (float)margin_cells * cell;
        const double span = ((double)hi[k] - (double)o[k]) / (double)cell;
        const int64_t cells = (int64_t)std::floor(span) + 2 + margin_cells;
        const int64_t rounded = ((cells + kBrick - 1) / kBrick) * kBrick;
        if (rounded > (int64_t)1 << 24) { too_large = true; return hipSuccess; }
        nc[k] = (int)rounded;
    }
    g.ox = o[0]; g.oy = o[1]; g.oz = o[2];
    g.ncx = nc[0]; g.ncy = nc[1]; g.ncz = nc[2];
    g.nbx = nc[0] / kBrick; g.nby = nc[1] / kBrick; g.nbz = nc[2] / kBrick;
    const int64_t top_entries = (int64_t)g.nbx * g.nby * g.nbz;
    if (top_entries > ((int64_t)1 << 31)) { too_large = true; return hipSuccess; }
    // the bound of the search uses coordinates in cell units; their float rounding error is a few
    // ulp of the largest cell coordinate
    const float max_cells = (float)std::max(std::max(nc[0], nc[1]), nc[2]);
    g.slop = std::max(1.0e-4f, 16.0f * max_cells * 1.1920929e-7f);

    S2M_TRY(map_ensure((void **)&buf.pts, &buf.pts_cap, m + kSentinelPoints, sizeof(float4), map_headroom_for(m)));
    S2M_TRY(map_put_sentinels(buf.pts, m, st));
    S2M_TRY(map_ensure((void **)&buf.pidx, &buf.pidx_cap, m + 1, sizeof(uint32_t), map_headroom_for(m)));
    S2M_TRY(map_ensure((void **)&buf.top, &buf.top_cap, top_entries + 1, sizeof(uint4)));
    S2M_TRY(ensure_scratch(buf, m));
    if (!buf.counters) S2M_TRY(hipMalloc((void **)&buf.counters, (64 + kOccShards * 32) * sizeof(uint32_t)));
    S2M_TRY(hipMemsetAsync(buf.top, 0, (size_t)(top_entries + 1) * sizeof(uint4), st));
    // net growth of every brick of this grid since its room was last laid out (s2m_mapedit.hip): a new grid starts from zero
    S2M_TRY(map_ensure((void **)&buf.grow, &buf.grow_cap, top_entries + 1, sizeof(uint32_t)));
    S2M_TRY(hipMemsetAsync(buf.grow, 0, (size_t)(top_entries + 1) * sizeof(uint32_t), st));
    buf.added_since_layout = 0;
    S2M_TRY(hipMemsetAsync(buf.counters, 0, 64 * sizeof(uint32_t), st));
    stats = MapStats();
    stats.top_entries = top_entries;
    if (m == 0) {
        g.top = buf.top; g.tab = buf.tab; g.pts = buf.pts; g.pidx = buf.pidx;
        return hipStreamSynchronize(st);
    }

    const int blocks = (int)((m + 255) / 256);
    hipLaunchKernelGGL(key_kernel, dim3(blocks), dim3(256), 0, st, xyz, stride, m, g, buf.keys, buf.vals);
    int bits = 9;
    while (((int64_t)1 << (bits - 9)) < top_entries) ++bits;
    bits = std::min(bits + 1, 64);
    size_t tmp = 0;
    S2M_TRY(rocprim::radix_sort_pairs(nullptr, tmp, buf.keys, buf.keys_alt, buf.vals, buf.vals_alt, (size_t)m, 0,
                                      (unsigned)bits, st));
    S2M_TRY(map_ensure_sort_tmp(buf, tmp));
    size_t t1 = buf.sort_tmp_bytes;
    S2M_TRY(rocprim::radix_sort_pairs(buf.sort_tmp, t1, buf.keys, buf.keys_alt, buf.vals, buf.vals_alt, (size_t)m,
                                      0, (unsigned)bits, st));
    hipLaunchKernelGGL(gather_kernel, dim3(blocks), dim3(256), 0, st, xyz, stride, m, buf.vals_alt, buf.pts, buf.pidx);
    buf.next_id = m;        // the ids of a fresh build are the caller's indices
    buf.ids_dense = true;
    S2M_TRY(map_build_tables(buf, buf.keys_alt, m, top_entries, stats, st));
    stats.top_entries = top_entries;
    g.top = buf.top; g.tab = buf.tab; g.pts = buf.pts; g.pidx = buf.pidx;
    return hipSuccess;
}

hipError_t build_map(const float *xyz, int64_t stride, int64_t m, float cell, MapBuffers &buf, Grid &grid,
                     MapStats &stats, bool &too_large, hipStream_t st, bool with_margin)
{
    float lo[3] = {0.f, 0.f, 0.f}, hi[3] = {0.f, 0.f, 0.f};
    if (m > 0) {
        if (!buf.bbox) S2M_TRY(hipMalloc((void **)&buf.bbox, kBboxScratchFloats * sizeof(float)));
        S2M_TRY(cloud_bbox(xyz, stride, m, buf.bbox, buf.mail, lo, hi, st));
    }
    // margin of an incrementally maintained map: an eighth of the longest extent, 2 to 64 bricks
    auto margin_for = [&](float c) {
        if (!with_margin) return 0;
        const double ext = std::max(std::max((double)hi[0] - lo[0], (double)hi[1] - lo[1]), (double)hi[2] - lo[2]);
        const int64_t cells = (int64_t)(ext / (8.0 * (double)c));
        return (int)(std::min<int64_t>(std::max<int64_t>(cells / kBrick, 2), 64) * kBrick);
    };
    if (cell > 0.0f) return build_once(xyz, stride, m, cell, lo, hi, margin_for(cell), buf, grid, stats, too_large, st);
    // density-driven cell size: LiDAR maps are surfaces, so points per occupied cell ~ c^2; aim for
    // ~11 points per occupied cell (cell edge about twice the 5-NN radius): measured fastest on
    // MI355X for the first-shell + hard-list search, including the large-displacement first pass
    float c = 0.5f;
    for (int attempt = 0; attempt < 4; ++attempt) {
        S2M_TRY(build_once(xyz, stride, m, c, lo, hi, margin_for(c), buf, grid, stats, too_large, st));
        if (too_large) { c *= 2.0f; continue; }
        if (m == 0 || stats.occupied_cells == 0) return hipSuccess;
        const double mean = (double)m / (double)stats.occupied_cells;
        if (mean >= 8.5 && mean <= 14.0) return hipSuccess;
        float cn = c * (float)std::sqrt(11.0 / mean);
        cn = std::min(std::max(cn, 0.02f), 64.0f);
        if (std::fabs(cn - c) < 0.05f * c) return hipSuccess;
        c = cn;
    }
    return build_once(xyz, stride, m, c, lo, hi, margin_for(c), buf, grid, stats, too_large, st);
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
