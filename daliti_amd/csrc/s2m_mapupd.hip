// s2m_mapupd.hip -- incremental map maintenance on the GPU.
//
// Replaces what the node does to the ikd-Tree after every scan:
//   map_incremental()                     eskf_lio/src/laserMapping.cpp:582-630
//   KD_TREE::Add_Points(pts, downsample)  eskf_lio/include/ikd-Tree/ikd_Tree.cpp:477-573
//   KD_TREE::Delete_Point_Boxes           ikd_Tree.cpp:631-658 (called by lasermap_fov_segment,
//                                         laserMapping.cpp:313-369)
// The reference inserts point by point: with downsampling on, the new point's voxel [min, max) of
// edge downsample_size is searched, and the voxel is rewritten to hold only the point closest to its
// centre when it held several points or when the new point wins (strict "<" against the new point,
// so a tie goes to the new point).  Applied to a whole batch the end state of a voxel is therefore
// {argmin of centre distance over old and new points; new beats old on ties; the last of tied new
// points wins} -- with one exception that keeps an untouched voxel untouched: a single old point
// that is strictly closer than every new point.  That closed form is what runs here, in parallel:
//   probe    eight lanes per new point: count / best old point inside its voxel via the brick grid; the batch's winner
//            per voxel by one 64-bit atomic minimum of (centre distance, ~batch index) in a direct-address table over the
//            voxel box of the grid (map_incremental), or
//   sort     radix sort of the new points by voxel key (stable: batch order inside a voxel) when no such box is known
//   resolve  eight lanes per voxel: winner among the new points, verdict against the best old point, and the old
//            points of a rewritten voxel (except the keeper) marked removed by sorted position, their bricks flagged
//   stage    the winning new points in batch order
// The new map is "survivors in point-id order, then the staged points": written into the current grid by
// s2m_mapedit.hip (the touched bricks rewritten in place, or the whole map merged) or, when a new point lies outside the
// grid, compacted here (update_finish) and rebuilt.  Counts travel to the host through the pinned mailbox (mail_post /
// mail_collect), not through 4-byte copies.  Among several OLD points tied for the smallest centre distance the lowest
// id wins (the reference takes the first in its tree traversal, which has no GPU counterpart); such ties need two points
// at exactly the same float distance inside one voxel.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "s2m_map_internal.h"

namespace s2m {

struct Voxel {
    float mn[3], mx[3], mid[3];
};

// Box_of_Point / mid_point of Add_Points (ikd_Tree.cpp:491-499), float arithmetic as written there
__device__ __forceinline__ Voxel voxel_of(float x, float y, float z, float ds)
{
    Voxel v;
    const float p[3] = {x, y, z};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        v.mn[k] = floorf(p[k] / ds) * ds;
        v.mx[k] = v.mn[k] + ds;
        v.mid[k] = (float)((double)v.mn[k] + (double)(v.mx[k] - v.mn[k]) / 2.0);
    }
    return v;
}
__device__ __forceinline__ bool in_box(const float4 &p, const float (&mn)[3], const float (&mx)[3])
{
    // Search_by_range / Delete_by_range membership: min <= p < max (ikd_Tree.cpp:1259, 794)
    return mn[0] <= p.x && mx[0] > p.x && mn[1] <= p.y && mx[1] > p.y && mn[2] <= p.z && mx[2] > p.z;
}
__device__ __forceinline__ float dist2(float ax, float ay, float az, const float (&b)[3])
{
    float d = (ax - b[0]) * (ax - b[0]) + (ay - b[1]) * (ay - b[1]);
    d = d + (az - b[2]) * (az - b[2]);
    return d;
}
// the same voxel as a LINEAR index over a box of voxels known to hold every point of the batch (VoxBox, s2m_kernels.h):
// the sort that groups a batch by voxel then runs over `bits` instead of 63 bits (five launches less per scan)
__device__ __forceinline__ bool voxel_in_box(float x, float y, float z, float ds, const VoxBox &b, uint64_t &lin)
{
    const int64_t kx = (int64_t)floorf(x / ds) - b.lo[0], ky = (int64_t)floorf(y / ds) - b.lo[1], kz = (int64_t)floorf(z / ds) - b.lo[2];
    lin = ((uint64_t)kz * (uint64_t)b.d[1] + (uint64_t)ky) * (uint64_t)b.d[0] + (uint64_t)kx;
    return kx >= 0 && kx < b.d[0] && ky >= 0 && ky < b.d[1] && kz >= 0 && kz < b.d[2];
}
__device__ __forceinline__ uint64_t voxel_key(float x, float y, float z, float ds)
{
    // 21 bits per axis of floor(p / ds), biased; identical floor() to the one that makes the box
    const int64_t kx = (int64_t)floorf(x / ds) + (1 << 20), ky = (int64_t)floorf(y / ds) + (1 << 20),
                  kz = (int64_t)floorf(z / ds) + (1 << 20);
    return ((uint64_t)(kx & 0x1fffff) << 42) | ((uint64_t)(ky & 0x1fffff) << 21) | (uint64_t)(kz & 0x1fffff);
}

// visit every old point inside box [mn, mx) through the brick grid.  The cell of a coordinate (cell_coord,
// s2m_device.h) is a monotone function of v (every step in it is), and the map points were binned with the same
// expression: a point with mn <= p < mx lies in a cell between cell(mn) and cell(mx), so no slack cells are needed.
// Per x-row the cells of one brick are one contiguous run of the sorted array.
// `sub` of `stride`: the lanes of a group share one box and take every stride-th point of each run -- a lane that
// walks a run alone waits a full memory latency per point (measured: 75 us for 11 k boxes of ~100 points).
template <class F>
__device__ __forceinline__ void for_points_in_box(const Grid &g, const float (&mn)[3], const float (&mx)[3], F &&f,
                                                  uint32_t sub = 0, uint32_t stride = 1)
{
    if (g.m == 0) return;
    int c0[3], c1[3];
    const float o[3] = {g.ox, g.oy, g.oz};
#pragma unroll
    for (int k = 0; k < 3; ++k) {  // clipped to the bricks in use (an empty range when the box lies outside them)
        c0[k] = max(cell_coord(mn[k], o[k], g.inv_c), g.blo[k] * 8);
        c1[k] = min(cell_coord(mx[k], o[k], g.inv_c), g.bhi[k] * 8 + 7);
    }
    for (int zz = c0[2]; zz <= c1[2]; ++zz)
        for (int yy = c0[1]; yy <= c1[1]; ++yy) {
            const int rowbit = ((zz & 7) << 3) | (yy & 7);
            const uint32_t toprow = top_row(g, yy >> 3, zz >> 3);
            for (int bx = c0[0] >> 3; bx <= (c1[0] >> 3); ++bx) {
                const uint4 te = g.top[toprow | ((uint32_t)bx & g.tmx)];
                const uint32_t mword = (rowbit & 32) ? te.w : te.z;
                if (te.x == 0 || ((mword >> (rowbit & 31)) & 1u) == 0) continue;
                const int l0 = max(c0[0], bx * 8) & 7, l1 = min(c1[0], bx * 8 + 7) & 7;
                const uint32_t *tb = g.tab + (int64_t)(te.x - 1) * kBrickStride + (rowbit << 3);
                const uint32_t e = tb[l1 + 1];
                for (uint32_t i = tb[l0] + sub; i < e; i += stride) {
                    const float4 q = g.pts[i];  // {x, y, position, z}
                    const float4 p = make_float4(q.x, q.y, map_point_z(q), 0.0f);
                    if (in_box(p, mn, mx)) f(p, g.pidx[i], i, te.x - 1u);  // point id (4-byte read beside the point), position, brick
                }
            }
        }
}

constexpr int kBoxLanes = 8;  // lanes that share one voxel box

__global__ __launch_bounds__(256) void add_probe_kernel(Grid g, const float4 *__restrict__ np, int n, float ds,
                                                        uint64_t *__restrict__ key, uint32_t *__restrict__ val,
                                                        float *__restrict__ dnew, uint32_t *__restrict__ cnt,
                                                        uint32_t *__restrict__ best_idx, uint32_t *__restrict__ best_pos,
                                                        float *__restrict__ best_d, uint32_t *__restrict__ add_flag, VoxBox vb,
                                                        unsigned long long *__restrict__ vtab)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = tid / kBoxLanes;
    const uint32_t sub = (uint32_t)(tid % kBoxLanes);
    const bool live = i < n;
    const float4 p = np[live ? i : 0];
    const Voxel v = voxel_of(p.x, p.y, p.z, ds);
    uint32_t c = 0, bi = 0xffffffffu, bp = 0u;
    float bd = INFINITY;
    if (live)
        for_points_in_box(g, v.mn, v.mx, [&](const float4 &q, uint32_t idx, uint32_t pos, uint32_t) {
            ++c;
            const float d = dist2(q.x, q.y, q.z, v.mid);
            if (d < bd || (d == bd && idx < bi)) { bd = d; bi = idx; bp = pos; }
        }, sub, kBoxLanes);
    // group result: counts add up; the best old point is the smallest (distance, caller index) pair
#pragma unroll
    for (int off = kBoxLanes / 2; off > 0; off >>= 1) {
        c += __shfl_xor(c, off, kBoxLanes);
        const float od = __shfl_xor(bd, off, kBoxLanes);
        const uint32_t oi = __shfl_xor(bi, off, kBoxLanes);
        const uint32_t op = __shfl_xor(bp, off, kBoxLanes);
        if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; bp = op; }
    }
    if (!live || sub != 0) return;
    uint64_t lin = 0;
    if (vb.bits > 0) (void)voxel_in_box(p.x, p.y, p.z, ds, vb, lin);  // (the caller has made sure every point is inside)
    key[i] = vb.bits > 0 ? lin : voxel_key(p.x, p.y, p.z, ds);
    val[i] = (uint32_t)i;
    add_flag[i] = 0u;  // set by add_resolve_kernel for the winners (was a memset of its own)
    const float dn = dist2(p.x, p.y, p.z, v.mid);
    dnew[i] = dn;
    // winner of the voxel among the batch -- smallest centre distance, the LAST one on ties -- without a sort: one 64-bit
    // atomic minimum of (distance bits, ~batch index) into the voxel's slot of a direct-address table over the voxel box
    // (all slots ~0 between batches: add_stage_kernel puts them back)
    if (vtab) atomicMin(vtab + lin, ((unsigned long long)__float_as_uint(dn) << 32) | (unsigned long long)(~(uint32_t)i));
    cnt[i] = c;
    best_idx[i] = bi;
    best_pos[i] = bp;
    best_d[i] = bd;
}

// one lane per sorted position; the first position of a voxel segment decides for the voxel
__global__ __launch_bounds__(256) void add_resolve_kernel(Grid g, const float4 *__restrict__ np, int n, float ds,
                                                          const uint64_t *__restrict__ skey,
                                                          const uint32_t *__restrict__ sval,
                                                          const float *__restrict__ dnew,
                                                          const uint32_t *__restrict__ cnt,
                                                          const uint32_t *__restrict__ best_idx,
                                                          const uint32_t *__restrict__ best_pos,
                                                          const float *__restrict__ best_d,
                                                          uint8_t *__restrict__ alive_s,
                                                          uint32_t *__restrict__ add_flag, uint32_t *__restrict__ counters,
                                                          const unsigned long long *__restrict__ vtab, uint8_t *__restrict__ bmark)
{
    // kBoxLanes lanes per sorted position: all of them take the (cheap) decision, the walk that marks the voxel's old
    // points is shared between them
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int s = tid / kBoxLanes;
    const uint32_t sub = (uint32_t)(tid % kBoxLanes);
    // table form (vtab): s is a batch index, and the point decides for its voxel when the table names it the winner;
    // sorted form: s is a sorted position, and the first position of a voxel segment decides
    bool head = false;
    bool add_new = false, rewrite = false;
    uint32_t keep = 0xffffffffu, w = 0, c = 0;
    float wd = 0.0f;
    float4 pw = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vtab) {
        if (s < n) {
            const unsigned long long e = vtab[skey[s]];  // (skey = the unsorted linear voxel indices)
            head = (uint32_t)s == ~(uint32_t)e;
            w = (uint32_t)s;
            wd = __uint_as_float((uint32_t)(e >> 32));
        }
    } else {
        head = s < n && !(s > 0 && skey[s - 1] == skey[s]);
        if (head) {
            // winner among the new points of this voxel: smallest centre distance, the last one on ties
            w = sval[s];
            wd = dnew[w];
            for (int t = s + 1; t < n && skey[t] == skey[s]; ++t) {
                const uint32_t i = sval[t];
                if (dnew[i] <= wd) { wd = dnew[i]; w = i; }
            }
        }
    }
    if (head) {
        c = cnt[w];
        const uint32_t bi = best_idx[w];
        const float bd = best_d[w];
        pw = np[w];
        if (c == 0) {
            add_new = true;
            rewrite = true;
        } else if (!(bd < wd)) {  // the new point is not strictly farther: it wins (:506 is a strict "<")
            add_new = true;
            rewrite = true;
        } else {
            // an old point stays the closest.  The reference rewrites the voxel when it held several points
            // or when the result "is" the new point within EPSS (ikd_Tree.cpp:514, 1676-1680)
            const float4 pe = g.pts[best_pos[w]];
            const bool same = fabs((double)(pw.x - pe.x)) < 1e-6 && fabs((double)(pw.y - pe.y)) < 1e-6 &&
                              fabs((double)(pw.z - map_point_z(pe))) < 1e-6;
            keep = bi;
            rewrite = (c > 1) || same;
        }
    }
    if (add_new && sub == 0) add_flag[w] = 1u;
    // tmp_counter of Add_Points: one atomic per wave (same-address atomics serialise at ~11 ns each)
    const unsigned long long rw = __ballot(rewrite && sub == 0);
    if (rw != 0ull && (threadIdx.x & 63) == (unsigned)(__ffsll((long long)rw) - 1)) atomicAdd(&counters[1], (uint32_t)__popcll(rw));
    if (rewrite && c > (keep != 0xffffffffu ? 1u : 0u)) {
        const Voxel v = voxel_of(pw.x, pw.y, pw.z, ds);
        for_points_in_box(g, v.mn, v.mx, [&](const float4 &, uint32_t idx, uint32_t pos, uint32_t brick) {
            if (idx != keep) { alive_s[pos] = 0; bmark[brick] |= 1u; }  // removed (by sorted position); its brick is touched
        }, sub, kBoxLanes);
    }
}

// over the SORTED array
__global__ __launch_bounds__(256) void delete_boxes_kernel(Grid g, const float4 *__restrict__ pts, int64_t m,
                                                           const float *__restrict__ boxes, int nb,
                                                           uint8_t *__restrict__ alive_s, uint8_t *__restrict__ bmark,
                                                           uint32_t *__restrict__ counters)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool hit = false;
    if (j < m && alive_s[j]) {
        const float4 q = pts[j];
        const float4 p = make_float4(q.x, q.y, map_point_z(q), 0.0f);
        for (int b = 0; b < nb && !hit; ++b) {
            const float *bx = boxes + 6 * b;
            const float mn[3] = {bx[0], bx[1], bx[2]}, mx[3] = {bx[3], bx[4], bx[5]};
            hit = in_box(p, mn, mx);
        }
        if (hit) {
            alive_s[j] = 0;
            // the brick that holds the point is touched (same cell arithmetic as the build; a map point lies inside the bounds)
            const int cx = cell_coord(p.x, g.ox, g.inv_c), cy = cell_coord(p.y, g.oy, g.inv_c), cz = cell_coord(p.z, g.oz, g.inv_c);
            const uint32_t idp1 = brick_in_bounds(g, cx >> 3, cy >> 3, cz >> 3) ? g.top[top_slot(g, cx >> 3, cy >> 3, cz >> 3)].x : 0u;
            if (idp1) bmark[idp1 - 1u] |= 1u;
        }
    }
    // one atomic per workgroup: a field-of-view trim deletes 1e5..1e6 points, per-point atomics on one
    // address would take milliseconds
    const int cnt = __syncthreads_count(hit ? 1 : 0);
    if (threadIdx.x == 0 && cnt) atomicAdd(&counters[2], (uint32_t)cnt);
}

// the same deletion by BRICK, for a few large boxes (the slabs of a field-of-view move: laserMapping.cpp:346-368): one
// wave per brick of the box's brick range (clipped to the bricks in use).  A brick whose cells all lie strictly between the
// box's end cells is inside it whole (cell_coord is monotone: cell(p) > cell(mn) means p > mn, cell(p) < cell(mx) means
// p < mx) and only its `alive` bytes are touched; the shell of bricks on the box's faces tests its points.  The kernel over
// every position above read 15 M positions and chained three memory latencies per deleted point (183 us for 2.6 M points).
struct DeleteBox {
    float mn[3], mx[3];
    int c0[3], c1[3];   // the cells of the box's corners
    int b0[3], nbk[3];  // its brick range, clipped to the bricks in use
};
__global__ __launch_bounds__(256) void delete_box_bricks_kernel(Grid g, DeleteBox d, uint8_t *__restrict__ alive_s,
                                                                uint8_t *__restrict__ bmark, uint32_t *__restrict__ counters)
{
    __shared__ uint32_t s_cnt;
    if (threadIdx.x == 0) s_cnt = 0u;
    __syncthreads();
    const int64_t w = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    const int64_t total = (int64_t)d.nbk[0] * d.nbk[1] * d.nbk[2];
    uint32_t cnt = 0;
    if (w < total) {
        const int bx = d.b0[0] + (int)(w % d.nbk[0]), by = d.b0[1] + (int)((w / d.nbk[0]) % d.nbk[1]);
        const int bz = d.b0[2] + (int)(w / ((int64_t)d.nbk[0] * d.nbk[1]));
        const uint4 te = g.top[top_slot(g, bx, by, bz)];
        if (te.x != 0u) {
            const uint32_t *tb = g.tab + (int64_t)(te.x - 1u) * kBrickStride;
            const uint32_t i0 = tb[0], i1 = tb[512];
            const bool whole = bx * 8 > d.c0[0] && bx * 8 + 7 < d.c1[0] && by * 8 > d.c0[1] && by * 8 + 7 < d.c1[1] &&
                               bz * 8 > d.c0[2] && bz * 8 + 7 < d.c1[2];
            for (uint32_t i = i0 + lane; i < i1; i += 64u) {
                if (!alive_s[i]) continue;
                bool hit = whole;
                if (!whole) {
                    const float4 q = g.pts[i];
                    hit = in_box(make_float4(q.x, q.y, map_point_z(q), 0.0f), d.mn, d.mx);
                }
                if (hit) { alive_s[i] = 0; ++cnt; }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
            if (lane == 0u && cnt) {
                bmark[te.x - 1u] |= 1u;
                atomicAdd(&s_cnt, cnt);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt) atomicAdd(&counters[2], s_cnt);
}

// survivors of the map in CALLER order = ascending point id: (id or ~0 for a removed point, position) pairs are sorted by
// id, the first `survivors` positions are gathered
__global__ __launch_bounds__(256) void id_key_kernel(const uint32_t *__restrict__ pidx, const uint8_t *__restrict__ alive_s, int64_t m,
                                                     uint32_t *__restrict__ key, uint32_t *__restrict__ val)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    key[j] = (!alive_s || alive_s[j]) ? pidx[j] : 0xffffffffu;
    val[j] = (uint32_t)j;
}
// number of keys below ~0 in the sorted keys (one thread: a binary search)
__global__ void live_count_kernel(const uint32_t *__restrict__ skey, int64_t m, uint32_t *__restrict__ out)
{
    int64_t lo = 0, hi = m;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (skey[mid] != 0xffffffffu) lo = mid + 1; else hi = mid;
    }
    *out = (uint32_t)lo;
}
__global__ __launch_bounds__(256) void gather_by_order_kernel(const float4 *__restrict__ pts, const uint32_t *__restrict__ order,
                                                              const uint32_t *__restrict__ count, float4 *__restrict__ out)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= (int64_t)*count) return;
    const float4 q = pts[order[r]];
    out[r] = make_float4(q.x, q.y, map_point_z(q), 0.0f);
}
// rank[position] = r for the r-th smallest live id: the dense caller index of every live position
__global__ __launch_bounds__(256) void rank_scatter_kernel(const uint32_t *__restrict__ order, const uint32_t *__restrict__ skey, int64_t m,
                                                           uint32_t *__restrict__ rank)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m) return;
    rank[order[r]] = skey[r] != 0xffffffffu ? (uint32_t)r : 0xffffffffu;
}

// out[pos[i]] = src[i] for flagged i (pos = exclusive scan of the flags), shifted by base; with a winner table, every
// point also puts its voxel's slot back to "empty" (all points of a voxel write the same value)
// base_in / count_out (deferred count): the base is read from the device and the count behind this batch left in ANOTHER
// word (every thread of the launch reads the first)
__global__ __launch_bounds__(256) void scatter_kernel(const float4 *__restrict__ src, const uint32_t *__restrict__ flag,
                                                      const uint32_t *__restrict__ pos, int64_t m, int64_t base,
                                                      float4 *__restrict__ out, const uint64_t *__restrict__ vkey,
                                                      unsigned long long *__restrict__ vtab,
                                                      const uint32_t *__restrict__ base_in, uint32_t *__restrict__ count_out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (base_in) base = (int64_t)*base_in;
    if (i < m && vtab) vtab[vkey[i]] = ~0ull;
    if (i < m && flag[i]) {
        float4 p = src[i];
        p.w = 0.0f;
        out[base + pos[i]] = p;
    }
    if (count_out && i == m - 1) *count_out = (uint32_t)base + pos[i] + flag[i];
}
// Add_Points(points, false) behind a count that lives on the device
__global__ __launch_bounds__(256) void append_kernel(const float4 *__restrict__ src, int64_t n, float4 *__restrict__ out,
                                                     const uint32_t *__restrict__ base_in, uint32_t *__restrict__ count_out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t base = (int64_t)*base_in;
    if (i < n) out[base + i] = src[i];
    if (i == n - 1) *count_out = (uint32_t)(base + n);
}

// map_incremental() (laserMapping.cpp:582-630): class 1 = PointToAdd, 2 = PointNoNeedDownsample, 0 = skip
struct VoxReach {
    int r[6];
};
__global__ __launch_bounds__(256) void incr_classify_kernel(Pose pose, const float *__restrict__ sx,
                                                            const float *__restrict__ sy,
                                                            const float *__restrict__ sz, int n,
                                                            const int32_t *__restrict__ nn_idx,
                                                            const float4 *__restrict__ pts, int have_nn, int map_has5, double fs,
                                                            float4 *__restrict__ pw_out,
                                                            unsigned long long *__restrict__ cls_out, int ax, int ay, int az,
                                                            VoxReach reach, uint32_t *__restrict__ vox_ext,
                                                            unsigned long long *__restrict__ blk_cnt)
{
    __shared__ unsigned long long s_cnt[4];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float wx = 0.0f, wy = 0.0f, wz = 0.0f;
    int cls = 0;
    if (i < n) {
    body_to_world(pose, sx[i], sy[i], sz[i], wx, wy, wz);  // pointBodyToWorld, :591
    pw_out[i] = make_float4(wx, wy, wz, 0.0f);
    cls = 1;
    int cnt = 0;
    if (have_nn) {
#pragma unroll
        for (int k = 0; k < kK; ++k) cnt += nn_idx[(int64_t)i * kK + k] >= 0 ? 1 : 0;
    }
    if (cnt > 0) {  // !Nearest_Points[i].empty() && flg_EKF_inited (:593)
        const float mid[3] = {(float)(floor((double)wx / fs) * fs + 0.5 * fs), (float)(floor((double)wy / fs) * fs + 0.5 * fs),
                              (float)(floor((double)wz / fs) * fs + 0.5 * fs)};  // :599-601
        const float dist = dist2(wx, wy, wz, mid);                              // :602
        const float4 n0 = pts[nn_idx[(int64_t)i * kK]];  // Nearest_Points as sorted positions
        if (fabs((double)(n0.x - mid[0])) > 0.5 * fs && fabs((double)(n0.y - mid[1])) > 0.5 * fs &&
            fabs((double)(map_point_z(n0) - mid[2])) > 0.5 * fs) {              // :603
            cls = 2;
        } else {
            bool need_add = true;
            // :610-616 run over the five neighbours of the reference's UNBOUNDED search whenever the map holds five points.
            // The lists here are exact up to the gate (and in their first entry); what is missing or unproven beyond it lies
            // more than sqrt(5) m from the point, i.e. more than 1.8 m from the voxel centre, and cannot be closer to it
            // than the point itself (at most 0.44 m): only the entries present can say "no need to add".
            if (map_has5) {
#pragma unroll
                for (int r = 0; r < kK; ++r) {
                    const int32_t j = nn_idx[(int64_t)i * kK + r];
                    if (j < 0) continue;
                    const float4 q = pts[j];
                    if (need_add && dist2(q.x, q.y, map_point_z(q), mid) < dist) need_add = false;  // :612-616
                }
            }
            cls = need_add ? 1 : 0;
        }
    }
    // both list flags in one word (low half: PointToAdd, high half: PointNoNeedDownsample): one scan gives both positions
    cls_out[i] = cls == 1 ? 1ull : (cls == 2 ? (1ull << 32) : 0ull);
    }
    // PointToAdd goes through the voxel rule: the box of its voxels sizes the direct-address table of the batch's winners
    // (VoxBox) -- a box that follows the SCAN, whatever the map has grown to.  Reported as the excess over the reach of the
    // last scans around the anchor voxel (the sensor's): in a steady stream no wave has anything to report
    uint32_t e[6] = {0u, 0u, 0u, 0u, 0u, 0u};
    if (cls == 1) {
        const float ds = (float)fs;
        const int64_t k[3] = {(int64_t)floorf(wx / ds), (int64_t)floorf(wy / ds), (int64_t)floorf(wz / ds)};  // (voxel_in_box's floor)
        const int64_t a[3] = {ax, ay, az};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            e[q] = (uint32_t)min(max(a[q] - k[q] - (int64_t)reach.r[q], (int64_t)0), (int64_t)0x7fffffff);
            e[3 + q] = (uint32_t)min(max(k[q] - a[q] - (int64_t)reach.r[3 + q], (int64_t)0), (int64_t)0x7fffffff);
        }
    }
    wave_max6_to(e, vox_ext);
    // the workgroup's two counts, packed like the flags: scatter2_blocks_kernel adds up the workgroups in front of its own
    unsigned long long c = cls == 1 ? 1ull : (cls == 2 ? (1ull << 32) : 0ull);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// Lists A and B without a device-wide scan: the classification left every workgroup's counts (blk_cnt), a workgroup here
// adds up the ones in front of it, scans its own 256 flags and writes its points -- one launch where the scan (two kernels
// of rocprim) and the scatter below were three.  counts2 = the lengths of the two lists.
__global__ __launch_bounds__(256) void scatter2_blocks_kernel(const float4 *__restrict__ src, const unsigned long long *__restrict__ flag,
                                                              const unsigned long long *__restrict__ blk_cnt, int n,
                                                              float4 *__restrict__ out_a, float4 *__restrict__ out_b,
                                                              uint32_t *__restrict__ counts2)
{
    __shared__ unsigned long long s_part[4], s_wave[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    unsigned long long before = 0ull;
    for (int b = tid; b < (int)blockIdx.x; b += 256) before += blk_cnt[b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off, 64);
    const int i = blockIdx.x * 256 + tid;
    const unsigned long long f = i < n ? flag[i] : 0ull;
    unsigned long long in = f;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long a = __shfl_up(in, off, 64);
        if (lane >= off) in += a;
    }
    if (lane == 0) s_part[wave] = before;
    if (lane == 63) s_wave[wave] = in;
    __syncthreads();
    unsigned long long at = s_part[0] + s_part[1] + s_part[2] + s_part[3] + in - f;
    for (int w = 0; w < wave; ++w) at += s_wave[w];
    if (f != 0ull) {
        float4 p = src[i];
        p.w = 0.0f;
        if (f & 1ull) out_a[(uint32_t)at] = p; else out_b[(uint32_t)(at >> 32)] = p;
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 255) {
        const unsigned long long total = at + f;
        counts2[0] = (uint32_t)total;
        counts2[1] = (uint32_t)(total >> 32);
    }
}

// The winners of update_add's batch behind the staged points by ONE workgroup, for the batch of a scan (up to kCompactMax
// points): the flags come in with coalesced reads and wait in LDS as bytes, every thread counts a contiguous run of them,
// one prefix sum over the workgroup, every thread writes its run's winners out.  The device-wide scan (two kernels of
// rocprim) and the scatter behind it were three launches for a few thousand points, and the host's launches are what bounds
// a frame.  (scatter_kernel's arguments; batch_count = the winners.)
constexpr int kCompactThreads = 1024, kCompactMax = 16384;
__global__ __launch_bounds__(kCompactThreads) void compact1_kernel(const float4 *__restrict__ src, const uint32_t *__restrict__ flag, int n,
                                                                   int64_t base, float4 *__restrict__ out, const uint64_t *__restrict__ vkey,
                                                                   unsigned long long *__restrict__ vtab, const uint32_t *__restrict__ base_in,
                                                                   uint32_t *__restrict__ count_out, uint32_t *__restrict__ batch_count)
{
    __shared__ uint8_t l_f[kCompactMax];
    __shared__ uint32_t w_part[kCompactThreads / 64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (base_in) base = (int64_t)*base_in;
    for (int i = tid; i < n; i += kCompactThreads) {
        l_f[i] = (uint8_t)(flag[i] != 0u);
        if (vtab) vtab[vkey[i]] = ~0ull;  // the voxel's slot of the winner table back to "empty"
    }
    __syncthreads();
    const int per = (n + kCompactThreads - 1) / kCompactThreads;
    const int i0 = min(tid * per, n), i1 = min(i0 + per, n);
    uint32_t c = 0u;
    for (int i = i0; i < i1; ++i) c += l_f[i];
    uint32_t in = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t a = __shfl_up(in, off, 64);
        if (lane >= off) in += a;
    }
    if (lane == 63) w_part[wave] = in;
    __syncthreads();
    uint32_t at = in - c, total = 0u;
    for (int w = 0; w < kCompactThreads / 64; ++w) {
        if (w < wave) at += w_part[w];
        total += w_part[w];
    }
    if (c != 0u)
        for (int i = i0; i < i1; ++i) {
            if (!l_f[i]) continue;
            float4 p = src[i];
            p.w = 0.0f;
            out[base + at++] = p;
        }
    if (tid == 0) {
        if (count_out) *count_out = (uint32_t)base + total;
        *batch_count = total;
    }
}

__global__ __launch_bounds__(256) void xyz_to_float4_kernel(const float *__restrict__ xyz, int64_t stride, int64_t n,
                                                            float4 *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = make_float4(xyz[i * stride], xyz[i * stride + 1], xyz[i * stride + 2], 0.0f);
}

// ikdtree.flatten's counterpart: the map in the caller's index order, packed xyz
__global__ __launch_bounds__(256) void map_to_xyz_kernel(const float4 *__restrict__ pts, const uint32_t *__restrict__ rank,
                                                         int64_t m, float *__restrict__ xyz)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m && rank[j] != 0xffffffffu) {
        const float4 p = pts[j];
        const int64_t i = rank[j];
        xyz[3 * i] = p.x; xyz[3 * i + 1] = p.y; xyz[3 * i + 2] = map_point_z(p);
    }
}

__global__ __launch_bounds__(256) void positions_to_indices_kernel(const int32_t *__restrict__ nn, const uint32_t *__restrict__ pidx,
                                                                   int64_t count, int32_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) {
        const int32_t p = nn[i];
        out[i] = p >= 0 ? (int32_t)pidx[p] : -1;
    }
}

struct MailArgs {
    const uint32_t *src[kMailSlots];
    int k;
    uint32_t seq;
};
// one wave: the values, then (system-scope release) the sequence number the host is polling for
__global__ void mail_kernel(MailArgs a, uint32_t *__restrict__ out)
{
    const int i = threadIdx.x;
    if (i < a.k) __hip_atomic_store(out + 1 + i, *a.src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __builtin_amdgcn_s_waitcnt(0);
    if (i == 0) __hip_atomic_store(out, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- host side ---------------------------------------------------------------------------------------
void free_mailbox(Mailbox &mb)
{
    if (mb.h) (void)hipHostFree(mb.h);
    mb = Mailbox();
}

static hipError_t mail_ensure(Mailbox &mb)
{
    if (mb.h) return hipSuccess;
    S2M_TRY(hipHostMalloc((void **)&mb.h, 64 * sizeof(uint32_t), hipHostMallocMapped));
    S2M_TRY(hipHostGetDevicePointer((void **)&mb.dev, mb.h, 0));
    mb.h[0] = 0u;
    mb.seq = 0u;
    return hipSuccess;
}

// a hand-back in two halves: mail_post enqueues the kernel that copies the words and raises the sequence number,
// mail_collect waits for it -- whatever the caller enqueues in between runs while the words travel
hipError_t mail_post(Mailbox &mb, const uint32_t *const *src, int k, hipStream_t st)
{
    if (k <= 0 || k > kMailSlots) return hipErrorInvalidValue;
    S2M_TRY(mail_ensure(mb));
    MailArgs a;
    for (int i = 0; i < kMailSlots; ++i) a.src[i] = src[i < k ? i : 0];
    a.k = k;
    if (++mb.seq == 0u) mb.seq = 1u;
    a.seq = mb.seq;
    if (cur_wait()->withhold(kStallMail)) return hipSuccess;  // (fault injection: this hand-back never leaves)
    hipLaunchKernelGGL(mail_kernel, dim3(1), dim3(64), 0, st, a, mb.dev);
    return hipSuccess;
}

hipError_t mail_collect(Mailbox &mb, int k, uint32_t *out, hipStream_t st)
{
    // When the kernel's sequence number shows up in pinned memory everything enqueued before it has finished.  Polling it
    // costs 2-3 us after the kernel ends; hipStreamSynchronize costs 10-20 us.  The wait follows the handle's policy and
    // ends at its deadline (s2m_wait.h).
    S2M_TRY(wait_word(nullptr, (const volatile uint32_t *)mb.h, mb.seq, st, "a hand-back of device words (mailbox)"));
    for (int i = 0; i < k; ++i) out[i] = mb.h[1 + i];
    return hipSuccess;
}

hipError_t mail_fetch(Mailbox &mb, const uint32_t *const *src, int k, uint32_t *out, hipStream_t st)
{
    S2M_TRY(mail_post(mb, src, k, st));
    return mail_collect(mb, k, out, st);
}

hipError_t mail_wait(Mailbox &mb, hipStream_t st)
{
    S2M_TRY(mail_ensure(mb));  // the fetched word is the mailbox's own sequence word
    const uint32_t *src[1] = {mb.dev};
    uint32_t v = 0;
    return mail_fetch(mb, src, 1, &v, st);
}

static inline int nblk(int64_t n) { return (int)((n + 255) / 256); }

template <class T>
static hipError_t grow(T **p, int64_t *cap, int64_t need, bool keep = false, hipStream_t st = nullptr)
{
    if (*cap >= need && *p) return hipSuccess;
    const int64_t c = std::max<int64_t>(need + need / 4, 1024);
    T *q = nullptr;
    S2M_TRY(hipMalloc((void **)&q, (size_t)c * sizeof(T)));
    note_allocation("update grow", (size_t)c * sizeof(T));
    if (*p) {
        if (keep && *cap > 0) {
            S2M_TRY(hipMemcpyAsync(q, *p, (size_t)*cap * sizeof(T), hipMemcpyDeviceToDevice, st));
            S2M_TRY(wait_stream(nullptr, st, "the copy into a grown update buffer"));
        }
        S2M_TRY(hipFree(*p));
    }
    *p = q;
    *cap = c;
    return hipSuccess;
}

void free_update(UpdateBuffers &u)
{
    void *ptrs[] = {u.alive_s, u.counters, u.stage, u.key, u.key2, u.val, u.val2, u.dnew, u.cnt, u.best_idx, u.best_pos, u.best_d,
                    u.add_flag, u.pos, u.flag32, u.pos_old, u.ord_key, u.ord_val, u.list, u.tmp, u.boxes, u.cvt, u.vtab};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    free_mailbox(u.mail);
    u = UpdateBuffers();
}

static hipError_t ensure_tmp(UpdateBuffers &u, size_t bytes)
{
    if (bytes <= u.tmp_bytes && u.tmp) return hipSuccess;
    if (u.tmp) S2M_TRY(hipFree(u.tmp));
    u.tmp = nullptr;
    // with room to spare: the scratch of a sort grows with the batch, and a hipFree + hipMalloc in the middle of a frame
    // stalls the device for ~0.1 ms (seen as isolated slow frames while the map's growth per frame crept up)
    const size_t want = std::max<size_t>(2 * bytes, (size_t)1 << 20);
    S2M_TRY(hipMalloc(&u.tmp, want));
    note_allocation("update tmp", want);
    u.tmp_bytes = want;
    return hipSuccess;
}

// dst gets the per-batch and staging capacities src has grown to (the other map of a handle, s2m_engine_relay.cpp: what its first
// real update would otherwise allocate beside a frame)
hipError_t update_reserve_like(UpdateBuffers &dst, const UpdateBuffers &src, hipStream_t st)
{
    if (src.batch_cap > dst.batch_cap) {
        const int64_t want = src.batch_cap;
        int64_t c;
        c = dst.batch_cap; S2M_TRY(grow(&dst.key, &c, want));
        c = dst.batch_cap; S2M_TRY(grow(&dst.key2, &c, want));
        c = dst.batch_cap; S2M_TRY(grow(&dst.val, &c, want));
        c = dst.batch_cap; S2M_TRY(grow(&dst.val2, &c, want));
        c = dst.batch_cap; S2M_TRY(grow(&dst.dnew, &c, want));
        c = dst.batch_cap; S2M_TRY(grow(&dst.cnt, &c, want));
        c = dst.batch_cap; S2M_TRY(grow(&dst.best_idx, &c, want));
        c = dst.batch_cap; S2M_TRY(grow(&dst.best_pos, &c, want));
        c = dst.batch_cap; S2M_TRY(grow(&dst.best_d, &c, want));
        c = dst.batch_cap; S2M_TRY(grow(&dst.add_flag, &c, want));
        c = dst.batch_cap; S2M_TRY(grow(&dst.pos, &c, want));
        dst.batch_cap = c;
    }
    if (src.stage_cap > dst.stage_cap) S2M_TRY(grow(&dst.stage, &dst.stage_cap, src.stage_cap, true, st));
    if (src.tmp_bytes > dst.tmp_bytes) S2M_TRY(ensure_tmp(dst, src.tmp_bytes / 2 + 1));
    if (src.vtab_cap > dst.vtab_cap) {
        if (dst.vtab) S2M_TRY(hipFree(dst.vtab));
        dst.vtab = nullptr;
        dst.vtab_cap = 0;
        S2M_TRY(hipMalloc((void **)&dst.vtab, (size_t)src.vtab_cap * sizeof(unsigned long long)));
        note_allocation("voxel table (the other map)", (size_t)src.vtab_cap * sizeof(unsigned long long));
        S2M_TRY(hipMemsetAsync(dst.vtab, 0xff, (size_t)src.vtab_cap * sizeof(unsigned long long), st));
        dst.vtab_cap = src.vtab_cap;
    }
    dst.reserve_hint = std::max(dst.reserve_hint, src.reserve_hint);
    return hipSuccess;
}

static hipError_t scan_u32(UpdateBuffers &u, const uint32_t *in, uint32_t *out, int64_t n, hipStream_t st)
{
    size_t bytes = 0;
    S2M_TRY(rocprim::exclusive_scan(nullptr, bytes, in, out, 0u, (size_t)n, rocprim::plus<uint32_t>(), st));
    S2M_TRY(ensure_tmp(u, bytes));
    size_t b2 = u.tmp_bytes;
    return rocprim::exclusive_scan(u.tmp, b2, in, out, 0u, (size_t)n, rocprim::plus<uint32_t>(), st);
}

// every old point alive, the update's counters zero: one launch (memsets are two launches each on this stack)
__global__ __launch_bounds__(256) void update_reset_kernel(int64_t bytes, uint8_t *__restrict__ alive_s, const uint32_t *__restrict__ pidx,
                                                           uint32_t *__restrict__ counters)
{
    // a position holds a point unless its id says "hole" (the slack a merge leaves behind every brick)
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (blockIdx.x == 0 && threadIdx.x < kUpdWords) counters[threadIdx.x] = 0u;
    if (i + 16 <= bytes - 1) {
        uint32_t w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 id = *reinterpret_cast<const uint4 *>(pidx + i + 4 * q);
            w[q] = (id.x != 0xffffffffu ? 1u : 0u) | (id.y != 0xffffffffu ? 0x100u : 0u) | (id.z != 0xffffffffu ? 0x10000u : 0u) |
                   (id.w != 0xffffffffu ? 0x1000000u : 0u);
        }
        *reinterpret_cast<uint4 *>(alive_s + i) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
        for (int64_t k = i; k < bytes; ++k) alive_s[k] = (k < bytes - 1 && pidx[k] != 0xffffffffu) ? 1 : 0;
    }
}

hipError_t update_begin(UpdateBuffers &u, const Grid &g, hipStream_t st)
{
    const uint8_t *before = u.alive_s;
    S2M_TRY(grow(&u.alive_s, &u.alive_s_cap, g.m + 1));
    if (!u.counters) S2M_TRY(hipMalloc((void **)&u.counters, kUpdWords * sizeof(uint32_t)));
    if (!u.boxes) S2M_TRY(grow(&u.boxes, &u.boxes_cap, 4096 * 6));  // (the ABI's maximum: the first field-of-view trim allocates nothing)
    {
        // alive_s says which positions hold a point.  It is all ones on a fresh layout (build / merge: layout_gen moved) and
        // stays as the in-place updates left it otherwise (holes at the ends of rewritten bricks): then only the counters
        // are zeroed
        const bool fresh = u.alive_gen != u.layout_gen || before != u.alive_s || before == nullptr;
        u.alive_gen = u.layout_gen;
        const int64_t bytes = fresh ? g.m + 1 : 0;
        hipLaunchKernelGGL(update_reset_kernel, dim3((unsigned)std::max<int64_t>((bytes + 4095) / 4096, 1)), dim3(256), 0, st, bytes, u.alive_s,
                           g.pidx, u.counters);
    }
    u.stage_n = 0;
    u.stage_deferred = false;
    u.stage_ops = 0;
    u.pend = StagePending();
    u.deleted_reported = 0;
    return hipSuccess;
}

const uint32_t *update_stage_word(const UpdateBuffers &u)
{
    return u.stage_deferred ? u.counters + kUpdStageWord + (u.stage_ops & 1) : nullptr;
}
hipError_t update_materialize(UpdateBuffers &u, hipStream_t st)
{
    if (!u.pend.on) return hipSuccess;
    const StagePending p = u.pend;
    u.pend = StagePending();
    if (p.na > 0) {
        hipLaunchKernelGGL(compact1_kernel, dim3(1), dim3(kCompactThreads), 0, st, p.la, p.flag, p.na, (int64_t)0, u.stage, p.vkey, p.vtab,
                           u.counters + kUpdStageWord + (u.stage_ops & 1), u.counters + kUpdStageWord + ((u.stage_ops + 1) & 1),
                           u.counters + kUpdBatchWord);
        ++u.stage_ops;
    }
    if (p.nb > 0) {
        hipLaunchKernelGGL(append_kernel, dim3(nblk(p.nb)), dim3(256), 0, st, p.lb, (int64_t)p.nb, u.stage, u.counters + kUpdStageWord + (u.stage_ops & 1),
                           u.counters + kUpdStageWord + ((u.stage_ops + 1) & 1));
        ++u.stage_ops;
    }
    return hipGetLastError();
}
hipError_t update_stage_count(UpdateBuffers &u, hipStream_t st)
{
    S2M_TRY(update_materialize(u, st));
    if (!u.stage_deferred) return hipSuccess;
    const uint32_t *src[1] = {update_stage_word(u)};
    uint32_t v = 0;
    S2M_TRY(mail_fetch(u.mail, src, 1, &v, st));
    u.stage_n = (int64_t)v;
    u.stage_deferred = false;
    return hipSuccess;
}

hipError_t update_add(UpdateBuffers &u, const Grid &g, const float4 *np, int64_t n, bool downsample, float ds,
                      int64_t *n_added, hipStream_t st, const VoxBox *vox, bool defer)
{
    if (n_added) *n_added = 0;
    if (n <= 0) return hipSuccess;
    // defer: nobody asks how many points this batch adds before the update is committed -- the count stays on the device and
    // the host goes on launching with a bound (one round trip less per scan, and the launches behind it no longer wait for
    // the host: s2m_map_incremental)
    defer = defer && !n_added;
    // a scan's two batches stay where they are (StagePending): the in-place update's preparation stages them itself
    const bool can_pend = defer && u.fuse_stage && n <= kCompactMax / 2 && (downsample ? !u.pend.on && u.stage_n == 0
                                                                                         : (u.pend.on ? u.pend.lb == nullptr && u.pend.na + n <= kCompactMax / 2
                                                                                                      : u.stage_n == 0));
    if (!can_pend) S2M_TRY(update_materialize(u, st));
    if (!defer && u.stage_deferred) S2M_TRY(update_stage_count(u, st));
    if (defer && !u.stage_deferred) {  // the count so far moves to the device
        if (u.stage_n != 0) defer = false;  // (an exact batch came first: stay exact)
        else u.stage_deferred = true;
    }
    S2M_TRY(grow(&u.stage, &u.stage_cap, std::max(u.stage_n + n, u.reserve_hint), true, st));
    const uint32_t *w_in = defer ? u.counters + kUpdStageWord + (u.stage_ops & 1) : nullptr;
    uint32_t *w_out = defer ? u.counters + kUpdStageWord + ((u.stage_ops + 1) & 1) : nullptr;
    if (!downsample) {  // Add_Points(points, false): every point is inserted (ikd_Tree.cpp:549-570)
        if (can_pend) {
            u.pend.lb = np;
            u.pend.nb = (int)n;
            u.pend.on = true;
        } else if (defer) {
            hipLaunchKernelGGL(append_kernel, dim3(nblk(n)), dim3(256), 0, st, np, n, u.stage, w_in, w_out);
            ++u.stage_ops;
        } else {
            S2M_TRY(hipMemcpyAsync(u.stage + u.stage_n, np, (size_t)n * sizeof(float4), hipMemcpyDeviceToDevice, st));
        }
        u.stage_n += n;
        if (n_added) *n_added = n;
        return hipGetLastError();
    }
    if (u.batch_cap < n) {
        const int64_t want = std::max<int64_t>(n, u.reserve_hint);
        int64_t c;
        c = u.batch_cap; S2M_TRY(grow(&u.key, &c, want));
        c = u.batch_cap; S2M_TRY(grow(&u.key2, &c, want));
        c = u.batch_cap; S2M_TRY(grow(&u.val, &c, want));
        c = u.batch_cap; S2M_TRY(grow(&u.val2, &c, want));
        c = u.batch_cap; S2M_TRY(grow(&u.dnew, &c, want));
        c = u.batch_cap; S2M_TRY(grow(&u.cnt, &c, want));
        c = u.batch_cap; S2M_TRY(grow(&u.best_idx, &c, want));
        c = u.batch_cap; S2M_TRY(grow(&u.best_pos, &c, want));
        c = u.batch_cap; S2M_TRY(grow(&u.best_d, &c, want));
        c = u.batch_cap; S2M_TRY(grow(&u.add_flag, &c, want));
        c = u.batch_cap; S2M_TRY(grow(&u.pos, &c, want));
        u.batch_cap = c;
    }
    const int in = (int)n;
    // with the batch inside a known box of voxels (map_incremental: VoxBox) whose direct-address table fits, the batch's
    // voxel winners come from 64-bit atomic minima in that table; otherwise the batch is sorted by voxel key
    unsigned long long *vtab = nullptr;
    if (vox && vox->bits > 0 && vox->bits <= kVoxTableBits) {
        const int64_t slots = (int64_t)vox->d[0] * vox->d[1] * vox->d[2];
        if (u.vtab_cap < slots) {
            // (the box follows the scan and breathes from frame to frame: with room to spare, up to the largest box a
            // table is used for)
            const int64_t want = std::min<int64_t>(slots + slots / 2, (int64_t)1 << kVoxTableBits);
            if (u.vtab) S2M_TRY(hipFree(u.vtab));
            u.vtab = nullptr;
            u.vtab_cap = 0;
            S2M_TRY(hipMalloc((void **)&u.vtab, (size_t)want * sizeof(unsigned long long)));
            note_allocation("voxel table", (size_t)want * sizeof(unsigned long long));
            S2M_TRY(hipMemsetAsync(u.vtab, 0xff, (size_t)want * sizeof(unsigned long long), st));
            u.vtab_cap = want;
        }
        vtab = u.vtab;
    }
    hipLaunchKernelGGL(add_probe_kernel, dim3(nblk(n * kBoxLanes)), dim3(256), 0, st, g, np, in, ds, u.key, u.val, u.dnew, u.cnt,
                       u.best_idx, u.best_pos, u.best_d, u.add_flag, vox ? *vox : VoxBox{}, vtab);
    const uint64_t *rkey = u.key;   // what add_resolve_kernel reads as keys / values: unsorted with the table
    const uint32_t *rval = u.val;
    if (!vtab) {
        const unsigned kbits = (vox && vox->bits > 0) ? (unsigned)vox->bits : 63u;
        size_t bytes = 0;
        S2M_TRY(rocprim::radix_sort_pairs(nullptr, bytes, u.key, u.key2, u.val, u.val2, (size_t)n, 0, kbits, st));
        S2M_TRY(ensure_tmp(u, bytes));
        size_t b2 = u.tmp_bytes;
        S2M_TRY(rocprim::radix_sort_pairs(u.tmp, b2, u.key, u.key2, u.val, u.val2, (size_t)n, 0, kbits, st));
        rkey = u.key2;
        rval = u.val2;
    }
    uint32_t before = 0;
    if (n_added) {  // tmp_counter of Add_Points before this batch (zero unless several batches share one update)
        const uint32_t *src[1] = {u.counters + 1};
        S2M_TRY(mail_fetch(u.mail, src, 1, &before, st));
    }
    hipLaunchKernelGGL(add_resolve_kernel, dim3(nblk(n * kBoxLanes)), dim3(256), 0, st, g, np, in, ds, rkey, rval, u.dnew,
                       u.cnt, u.best_idx, u.best_pos, u.best_d, u.alive_s, u.add_flag, u.counters, vtab, u.bmark);
    // winners, in batch order, go to the staging list
    const bool one_group = n <= kCompactMax;
    if (can_pend) {
        u.pend.la = np;
        u.pend.flag = u.add_flag;
        u.pend.na = in;
        u.pend.vkey = u.key;
        u.pend.vtab = vtab;
        u.pend.on = true;
        u.stage_n += n;  // (a bound)
        return hipGetLastError();
    }
    if (one_group) {
        hipLaunchKernelGGL(compact1_kernel, dim3(1), dim3(kCompactThreads), 0, st, np, u.add_flag, in, u.stage_n, u.stage, u.key, vtab, w_in, w_out,
                           u.counters + kUpdBatchWord);
    } else {
        S2M_TRY(scan_u32(u, u.add_flag, u.pos, n, st));
        hipLaunchKernelGGL(scatter_kernel, dim3(nblk(n)), dim3(256), 0, st, np, u.add_flag, u.pos, n, u.stage_n, u.stage, u.key, vtab, w_in, w_out);
    }
    if (defer) {
        ++u.stage_ops;
        u.stage_n += n;  // (a bound: every point of the batch may have won its voxel)
        return hipGetLastError();
    }
    const uint32_t *src[3] = {one_group ? u.counters + kUpdBatchWord : u.pos + (n - 1), one_group ? u.counters + kUpdBatchWord + 1 : u.add_flag + (n - 1),
                              u.counters + 1};   // (word kUpdBatchWord + 1 is never written: zero)
    uint32_t v[3] = {0, 0, 0};
    S2M_TRY(mail_fetch(u.mail, src, 3, v, st));
    u.stage_n += (int64_t)v[0] + v[1];
    if (n_added) *n_added = (int64_t)v[2] - before;  // tmp_counter of Add_Points
    return hipGetLastError();
}

constexpr int kBrickBoxes = 8;  // up to this many boxes per call go brick by brick
// the brick range of one box, clipped to the bricks in use; false when the box misses them
static bool delete_box_range(const Grid &g, const float *box, DeleteBox &d)
{
    const float o[3] = {g.ox, g.oy, g.oz};
    for (int k = 0; k < 3; ++k) {
        d.mn[k] = box[k];
        d.mx[k] = box[3 + k];
        if (!(box[k] < box[3 + k])) return false;  // (an empty or NaN interval holds no point: min <= p < max)
        d.c0[k] = cell_coord(box[k], o[k], g.inv_c);
        d.c1[k] = cell_coord(box[3 + k], o[k], g.inv_c);
        const int lo = std::max(d.c0[k], g.blo[k] * 8), hi = std::min(d.c1[k], g.bhi[k] * 8 + 7);
        if (lo > hi) return false;
        d.b0[k] = lo >> 3;
        d.nbk[k] = (hi >> 3) - d.b0[k] + 1;
    }
    return true;
}
bool delete_touches_map(const Grid &g, const float *boxes_host, int nb)
{
    if (g.m == 0) return false;
    if (nb > kBrickBoxes) return true;
    DeleteBox d;
    for (int b = 0; b < nb; ++b)
        if (delete_box_range(g, boxes_host + 6 * b, d)) return true;
    return false;
}

hipError_t update_delete(UpdateBuffers &u, const Grid &g, const float *boxes_host, int nb, int64_t *n_deleted,
                         hipStream_t st)
{
    if (n_deleted) *n_deleted = 0;
    if (nb <= 0 || g.m == 0) return hipSuccess;
    // counters[2] is zero at update_begin and only this entry point adds to it: the count after the launch, minus
    // what earlier calls of the same update reported
    if (nb <= kBrickBoxes) {
        // a few boxes: one launch per box over its bricks, one after the other (overlapping boxes count a point once)
        int launched = 0;
        for (int b = 0; b < nb; ++b) {
            DeleteBox d;
            if (!delete_box_range(g, boxes_host + 6 * b, d)) continue;  // outside the bricks in use: nothing to look at
            const int64_t total = (int64_t)d.nbk[0] * d.nbk[1] * d.nbk[2];
            hipLaunchKernelGGL(delete_box_bricks_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, st, g, d, u.alive_s, u.bmark, u.counters);
            ++launched;
        }
        if (!launched) return hipSuccess;
    } else {
        S2M_TRY(grow(&u.boxes, &u.boxes_cap, std::max<int64_t>((int64_t)nb * 6, 4096 * 6)));  // (the ABI's maximum: no growth in a frame)
        S2M_TRY(hipMemcpyAsync(u.boxes, boxes_host, (size_t)nb * 6 * sizeof(float), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(delete_boxes_kernel, dim3(nblk(g.m)), dim3(256), 0, st, g, g.pts, g.m, u.boxes, nb, u.alive_s, u.bmark, u.counters);
    }
    const uint32_t *src[1] = {u.counters + 2};
    uint32_t after = 0;
    S2M_TRY(mail_fetch(u.mail, src, 1, &after, st));
    if (n_deleted) *n_deleted = (int64_t)after - u.deleted_reported;
    u.deleted_reported = after;
    return hipGetLastError();
}

// (id, position) pairs of the map sorted by id into u.ord_key / u.ord_val (removed points last); *live = their number
static hipError_t order_by_id(UpdateBuffers &u, const Grid &g, const uint8_t *alive_s, int64_t *live, hipStream_t st)
{
    *live = 0;
    if (g.m <= 0) return hipSuccess;
    if (u.ord_cap < g.m) {
        int64_t c;
        c = u.ord_cap; S2M_TRY(grow(&u.flag32, &c, g.m));
        c = u.ord_cap; S2M_TRY(grow(&u.pos_old, &c, g.m));
        c = u.ord_cap; S2M_TRY(grow(&u.ord_key, &c, g.m));
        c = u.ord_cap; S2M_TRY(grow(&u.ord_val, &c, g.m));
        u.ord_cap = c;
    }
    hipLaunchKernelGGL(id_key_kernel, dim3(nblk(g.m)), dim3(256), 0, st, g.pidx, alive_s, g.m, u.flag32, u.pos_old);
    size_t bytes = 0;
    S2M_TRY(rocprim::radix_sort_pairs(nullptr, bytes, u.flag32, u.ord_key, u.pos_old, u.ord_val, (size_t)g.m, 0, 32, st));
    S2M_TRY(ensure_tmp(u, bytes));
    size_t b2 = u.tmp_bytes;
    S2M_TRY(rocprim::radix_sort_pairs(u.tmp, b2, u.flag32, u.ord_key, u.pos_old, u.ord_val, (size_t)g.m, 0, 32, st));
    if (!u.counters) {
        S2M_TRY(hipMalloc((void **)&u.counters, kUpdWords * sizeof(uint32_t)));
        S2M_TRY(hipMemsetAsync(u.counters, 0, kUpdWords * sizeof(uint32_t), st));
    }
    hipLaunchKernelGGL(live_count_kernel, dim3(1), dim3(1), 0, st, u.ord_key, g.m, u.counters + 14);
    const uint32_t *src[1] = {u.counters + 14};
    uint32_t v = 0;
    S2M_TRY(mail_fetch(u.mail, src, 1, &v, st));
    *live = v;
    return hipSuccess;
}

// survivors (ascending id = caller order) followed by the staged appends -> u.list; *m_out = new size
hipError_t update_finish(UpdateBuffers &u, const Grid &g, int64_t *m_out, hipStream_t st)
{
    S2M_TRY(update_stage_count(u, st));  // (pending batches staged, a count left on the device read)
    int64_t survivors = 0;
    S2M_TRY(order_by_id(u, g, u.alive_s, &survivors, st));
    S2M_TRY(grow(&u.list, &u.list_cap, survivors + u.stage_n));
    if (survivors > 0)
        hipLaunchKernelGGL(gather_by_order_kernel, dim3(nblk(survivors)), dim3(256), 0, st, g.pts, u.ord_val, u.counters + 14, u.list);
    if (u.stage_n > 0)
        S2M_TRY(hipMemcpyAsync(u.list + survivors, u.stage, (size_t)u.stage_n * sizeof(float4), hipMemcpyDeviceToDevice, st));
    *m_out = survivors + u.stage_n;
    return hipGetLastError();
}

// rank[position] = dense caller index (the rank of the point's id among the live points); alive_s == nullptr: every
// position below g.m is live.  Cold path: the getters that answer in caller indices / caller order.
hipError_t caller_ranks(UpdateBuffers &u, const Grid &g, const uint8_t *alive_s, const uint32_t **rank, int64_t *live, hipStream_t st)
{
    *rank = nullptr;
    S2M_TRY(order_by_id(u, g, alive_s, live, st));
    if (g.m <= 0) return hipSuccess;
    hipLaunchKernelGGL(rank_scatter_kernel, dim3(nblk(g.m)), dim3(256), 0, st, u.ord_val, u.ord_key, g.m, u.flag32);
    *rank = u.flag32;
    return hipGetLastError();
}

hipError_t incr_classify(UpdateBuffers &u, const Pose &pose, const float *sx, const float *sy, const float *sz, int n,
                         const int32_t *nn_idx, const Grid &g, bool have_nn, double fs, float4 **to_add, int64_t *n_add,
                         float4 **no_down, int64_t *n_no_down, hipStream_t st, VoxBox *vox, bool begin_update, const uint32_t *extra,
                         uint32_t *extra_out)
{
    if (extra_out) *extra_out = 0u;
    if (vox) *vox = VoxBox{};
    *n_add = 0;
    *n_no_down = 0;
    *to_add = nullptr;
    *no_down = nullptr;
    if (n <= 0) return begin_update ? update_begin(u, g, st) : hipSuccess;
    // scratch: cvt holds [pw (n) | list A (n) | list B (n)], flags in add_flag / cnt, positions in pos / best_idx
    S2M_TRY(grow(&u.cvt, &u.cvt_cap, (int64_t)3 * std::max<int64_t>(n, u.reserve_hint)));
    if (u.batch_cap < n) {
        const int64_t want = std::max<int64_t>(n, u.reserve_hint);
        int64_t c;
        c = u.batch_cap; S2M_TRY(grow(&u.key, &c, want));
        c = u.batch_cap; S2M_TRY(grow(&u.key2, &c, want));
        c = u.batch_cap; S2M_TRY(grow(&u.val, &c, want));
        c = u.batch_cap; S2M_TRY(grow(&u.val2, &c, want));
        c = u.batch_cap; S2M_TRY(grow(&u.dnew, &c, want));
        c = u.batch_cap; S2M_TRY(grow(&u.cnt, &c, want));
        c = u.batch_cap; S2M_TRY(grow(&u.best_idx, &c, want));
        c = u.batch_cap; S2M_TRY(grow(&u.best_pos, &c, want));
        c = u.batch_cap; S2M_TRY(grow(&u.best_d, &c, want));
        c = u.batch_cap; S2M_TRY(grow(&u.add_flag, &c, want));
        c = u.batch_cap; S2M_TRY(grow(&u.pos, &c, want));
        u.batch_cap = c;
    }
    float4 *pw = u.cvt, *la = u.cvt + n, *lb = u.cvt + 2 * (int64_t)n;
    unsigned long long *fl = reinterpret_cast<unsigned long long *>(u.key), *ps = reinterpret_cast<unsigned long long *>(u.key2);
    // the anchor of the voxel box: the sensor's voxel (any voxel would do: the kernel reports distances from it)
    int anchor[3];
    for (int k = 0; k < 3; ++k) anchor[k] = (int)std::fmin(std::fmax(std::floor(pose.t[k] / fs), -1.0e9), 1.0e9);
    VoxReach reach;
    for (int k = 0; k < 6; ++k) reach.r[k] = u.vox_reach[k];
    if (!u.counters) {  // (the box words are zero here, and zeroed again by every update_begin behind the read-back)
        S2M_TRY(hipMalloc((void **)&u.counters, kUpdWords * sizeof(uint32_t)));
        S2M_TRY(hipMemsetAsync(u.counters, 0, kUpdWords * sizeof(uint32_t), st));
    }
    hipLaunchKernelGGL(incr_classify_kernel, dim3(nblk(n)), dim3(256), 0, st, pose, sx, sy, sz, n, nn_idx, g.pts,
                       have_nn ? 1 : 0, g.live >= kK ? 1 : 0, fs, pw, fl, anchor[0], anchor[1], anchor[2], reach, u.counters + kUpdVoxWord, ps);
    hipLaunchKernelGGL(scatter2_blocks_kernel, dim3(nblk(n)), dim3(256), 0, st, pw, fl, ps, n, la, lb, u.counters + kUpdListWord);
    {   // both list lengths with one hand-back (and two words that stay zero)
        const uint32_t *c2 = u.counters + kUpdListWord;
        const uint32_t *p32 = c2, *f32 = c2 + 2;
        const uint32_t *v = u.counters + kUpdVoxWord;
        const uint32_t *src[11] = {p32, f32, p32 + 1, f32 + 1, v, v + 1, v + 2, v + 3, v + 4, v + 5, extra ? extra : p32};
        uint32_t h[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        S2M_TRY(mail_post(u.mail, src, 11, st));
        // the update this classification feeds begins while the counts travel (its reset kernel needs none of them and
        // zeroes the box words behind the read above, in stream order)
        if (begin_update) S2M_TRY(update_begin(u, g, st));
        else S2M_TRY(hipMemsetAsync(u.counters + kUpdVoxWord, 0, 6 * sizeof(uint32_t), st));
        S2M_TRY(mail_collect(u.mail, 11, h, st));
        if (extra && extra_out) *extra_out = h[10];
        *n_add = (int64_t)h[0] + h[1];
        *n_no_down = (int64_t)h[2] + h[3];
        if (vox && *n_add > 0) {  // the box of the PointToAdd voxels: every one of them lies inside by construction
            VoxBox b{};
            double prod = 1.0;
            bool ok = true;
            for (int k = 0; k < 3; ++k) {
                const int64_t below = (int64_t)reach.r[k] + (int64_t)h[4 + k], above = (int64_t)reach.r[3 + k] + (int64_t)h[7 + k];
                const int64_t lo = (int64_t)anchor[k] - below, d = below + above + 1;
                // the reach the next scan is measured against: what this one needed and a little more where it grew, a
                // little less (never below 16 voxels) where it did not
                u.vox_reach[k] = (int)std::min<int64_t>(h[4 + k] ? below + 8 : std::max<int64_t>(below - 2, 16), 1 << 20);
                u.vox_reach[3 + k] = (int)std::min<int64_t>(h[7 + k] ? above + 8 : std::max<int64_t>(above - 2, 16), 1 << 20);
                ok = ok && d < 2000000 && lo > -1000000000ll && lo < 1000000000ll;
                b.lo[k] = (int)lo;
                b.d[k] = (int)d;
                prod *= (double)d;
            }
            if (ok && prod < 1.0e12) {
                b.bits = 1;
                while (((uint64_t)1 << b.bits) < (uint64_t)prod) ++b.bits;
                *vox = b;
            }
        }
    }
    *to_add = la;
    *no_down = lb;
    return hipGetLastError();
}

hipError_t xyz_to_float4(UpdateBuffers &u, const float *xyz_dev, int64_t stride, int64_t n, float4 **out, hipStream_t st)
{
    S2M_TRY(grow(&u.cvt, &u.cvt_cap, std::max<int64_t>(n, 1)));
    if (n > 0) hipLaunchKernelGGL(xyz_to_float4_kernel, dim3(nblk(n)), dim3(256), 0, st, xyz_dev, stride, n, u.cvt);
    *out = u.cvt;
    return hipGetLastError();
}

void launch_map_to_xyz(const float4 *pts, const uint32_t *pidx, int64_t m, float *xyz, hipStream_t st)
{
    if (m > 0) hipLaunchKernelGGL(map_to_xyz_kernel, dim3(nblk(m)), dim3(256), 0, st, pts, pidx, m, xyz);
}

void launch_positions_to_indices(const int32_t *nn, const uint32_t *pidx, int64_t count, int32_t *out, hipStream_t st)
{
    if (count > 0) hipLaunchKernelGGL(positions_to_indices_kernel, dim3(nblk(count)), dim3(256), 0, st, nn, pidx, count, out);
}


// ---- the change log ---------------------------------------------------------------------------------------------------
// The reference flattens and publishes the whole map every frame (laserMapping.cpp:1170-1175, 1229-1235); a node that follows
// this engine's map keeps a mirror and asks for what changed (s2m_map_get_changes).  Removed points come from
// the bricks the update's verdict kernels marked (bit 0 of bmark: a point of the brick was removed): one wave per brick id,
// the marked ones walk their stretch -- the plan kernel of the in-place update does the same walk.  A removed point is logged
// with its coordinates (the follower finds it by place, then by id).  Box deletes (the field-of-view trim: millions of points)
// are not logged point by point at all: the boxes themselves are the entry, with the log's two counts at that moment
// (log_mark_kernel) so that the follower applies them in sequence with the other entries.
__global__ __launch_bounds__(256) void log_removed_kernel(const uint32_t *__restrict__ bricks_dev, const uint8_t *__restrict__ bmark,
                                                          const uint32_t *__restrict__ tab, const uint8_t *__restrict__ alive_s,
                                                          const uint32_t *__restrict__ pidx, const float4 *__restrict__ pts,
                                                          float4 *__restrict__ removed, uint32_t *__restrict__ counts, uint32_t cap)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t id = (int64_t)blockIdx.x * 4 + wave;
    if (id >= (int64_t)*bricks_dev || (bmark[id] & 1u) == 0) return;
    const uint32_t base = tab[id * kBrickStride], end = tab[id * kBrickStride + kBrickCells];
    for (uint32_t j0 = base; j0 < end; j0 += 64u) {
        const uint32_t j = j0 + (uint32_t)lane;
        uint32_t pid = 0xffffffffu;
        const bool gone = j < end && alive_s[j] == 0 && (pid = pidx[j]) != 0xffffffffu;
        const unsigned long long bal = __ballot(gone);
        if (bal == 0ull) continue;
        uint32_t at = 0;
        if (lane == 0) at = atomicAdd(counts + 1, (uint32_t)__popcll(bal));
        at = __shfl(at, 0, 64);
        const uint32_t mine = at + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        if (gone) {
            if (mine < cap) {
                const float4 p = pts[j];
                removed[mine] = make_float4(p.x, p.y, map_point_z(p), __uint_as_float(pid));
            } else {
                counts[2] = 1u;
            }
        }
    }
}
__global__ __launch_bounds__(256) void log_added_kernel(const float4 *__restrict__ stage, int64_t n, uint32_t first_id, float4 *__restrict__ added,
                                                        uint32_t *__restrict__ counts, uint32_t cap)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t at = counts[0];  // (the count moves on in a separate launch: every thread reads the same value)
    if (i >= n) return;
    if ((uint64_t)at + (uint64_t)i < cap) {
        float4 p = stage[i];
        p.w = __uint_as_float(first_id + (uint32_t)i);
        added[at + i] = p;
    } else {
        counts[2] = 1u;
    }
}
__global__ void log_count_kernel(uint32_t *counts, uint32_t add_n, int reset)
{
    if (reset) { counts[0] = 0u; counts[1] = 0u; counts[2] = 0u; counts[3] = 0u; }
    else counts[0] += add_n;
}
// a box delete at this point of the log: the counts as they stand are its place in the sequence
__global__ void log_mark_kernel(uint32_t *counts)
{
    const uint32_t k = counts[3];
    if (k < (uint32_t)kLogMarks) {
        counts[4 + 2 * k] = counts[0];
        counts[5 + 2 * k] = counts[1];
        counts[3] = k + 1u;
    } else {
        counts[2] = 1u;
    }
}
// The log on its way to the follower: the entries into pinned host memory (a fixed grid strides over them), then -- a second
// launch, one workgroup -- the header (counts, marks), the sequence word the host polls for, and the log is empty again.
__global__ __launch_bounds__(256) void log_flush_kernel(const float4 *__restrict__ added, const float4 *__restrict__ removed,
                                                        const uint32_t *__restrict__ counts, uint32_t cap, float4 *__restrict__ h_added,
                                                        float4 *__restrict__ h_removed)
{
    if (counts[2] != 0u) return;  // overflow: the follower will fetch the whole map
    const uint32_t na = counts[0] < cap ? counts[0] : cap, nr = counts[1] < cap ? counts[1] : cap;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < na; i += stride) h_added[i] = added[i];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nr; i += stride) h_removed[i] = removed[i];
}
__global__ void log_seal_kernel(uint32_t *counts, uint32_t *__restrict__ h_head, uint32_t seq)
{
    const int i = threadIdx.x;
    if (i < 4 + 2 * kLogMarks) __hip_atomic_store(h_head + 1 + i, counts[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (i == 0) {
        __hip_atomic_store(h_head, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        counts[0] = 0u; counts[1] = 0u; counts[2] = 0u; counts[3] = 0u;
    }
}
__global__ __launch_bounds__(256) void ids_by_rank_kernel(const uint32_t *__restrict__ pidx, const uint32_t *__restrict__ rank, int64_t m,
                                                          uint32_t *__restrict__ ids)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m && rank[j] != 0xffffffffu) ids[rank[j]] = pidx[j];
}

void free_changelog(ChangeLog &c)
{
    if (c.added) (void)hipFree(c.added);
    if (c.removed) (void)hipFree(c.removed);
    if (c.counts) (void)hipFree(c.counts);
    if (c.h_head) (void)hipHostFree(c.h_head);
    if (c.h_added) (void)hipHostFree(c.h_added);
    if (c.h_removed) (void)hipHostFree(c.h_removed);
    c = ChangeLog();
}
hipError_t changelog_ensure(ChangeLog &c, int64_t cap, hipStream_t st)
{
    if (c.cap >= cap && c.added) return hipSuccess;
    free_changelog(c);
    S2M_TRY(hipMalloc((void **)&c.added, (size_t)cap * sizeof(float4)));
    S2M_TRY(hipMalloc((void **)&c.removed, (size_t)cap * sizeof(float4)));
    S2M_TRY(hipMalloc((void **)&c.counts, kLogWords * sizeof(uint32_t)));
    S2M_TRY(hipMemsetAsync(c.counts, 0, kLogWords * sizeof(uint32_t), st));
    S2M_TRY(hipHostMalloc((void **)&c.h_head, 64 * sizeof(uint32_t), hipHostMallocMapped));
    S2M_TRY(hipHostGetDevicePointer((void **)&c.h_head_dev, c.h_head, 0));
    S2M_TRY(hipHostMalloc((void **)&c.h_added, (size_t)cap * sizeof(float4), hipHostMallocMapped));
    S2M_TRY(hipHostGetDevicePointer((void **)&c.h_added_dev, c.h_added, 0));
    S2M_TRY(hipHostMalloc((void **)&c.h_removed, (size_t)cap * sizeof(float4), hipHostMallocMapped));
    S2M_TRY(hipHostGetDevicePointer((void **)&c.h_removed_dev, c.h_removed, 0));
    std::memset(c.h_head, 0, 64 * sizeof(uint32_t));
    c.cap = cap;
    c.flush_seq = 0;
    c.posted = false;
    return hipSuccess;
}
void launch_log_removed(ChangeLog &c, const uint32_t *bricks_dev, int64_t bricks_bound, const uint8_t *bmark, const uint32_t *tab,
                        const uint8_t *alive_s, const uint32_t *pidx, const float4 *pts, hipStream_t st)
{
    if (bricks_bound > 0)
        hipLaunchKernelGGL(log_removed_kernel, dim3((unsigned)((bricks_bound + 3) / 4)), dim3(256), 0, st, bricks_dev, bmark, tab, alive_s, pidx,
                           pts, c.removed, c.counts, (uint32_t)c.cap);
}
void launch_log_added(ChangeLog &c, const float4 *stage, int64_t n, uint32_t first_id, hipStream_t st)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(log_added_kernel, dim3(nblk(n)), dim3(256), 0, st, stage, n, first_id, c.added, c.counts, (uint32_t)c.cap);
    hipLaunchKernelGGL(log_count_kernel, dim3(1), dim3(1), 0, st, c.counts, (uint32_t)n, 0);
}
void launch_log_reset(ChangeLog &c, hipStream_t st) { hipLaunchKernelGGL(log_count_kernel, dim3(1), dim3(1), 0, st, c.counts, 0u, 1); }
void launch_log_mark(ChangeLog &c, hipStream_t st) { hipLaunchKernelGGL(log_mark_kernel, dim3(1), dim3(1), 0, st, c.counts); }
// post: the log leaves for pinned host memory and starts again empty; collect: wait for it (under the handle's wait policy and
// deadline) -- whatever the caller enqueues or does in between runs while the entries travel
void changelog_post(ChangeLog &c, hipStream_t st)
{
    if (++c.flush_seq == 0u) c.flush_seq = 1u;
    hipLaunchKernelGGL(log_flush_kernel, dim3(32), dim3(256), 0, st, c.added, c.removed, c.counts, (uint32_t)c.cap, c.h_added_dev, c.h_removed_dev);
    if (!cur_wait()->withhold(kStallMail))  // (fault injection: this hand-back never leaves)
        hipLaunchKernelGGL(log_seal_kernel, dim3(1), dim3(64), 0, st, c.counts, c.h_head_dev, c.flush_seq);
    c.posted = true;
}
hipError_t changelog_collect(ChangeLog &c, hipStream_t st)
{
    S2M_TRY(wait_word(nullptr, (const volatile uint32_t *)c.h_head, c.flush_seq, st, "the map's change log on its way to the host (a hand-back)"));
    c.posted = false;
    return hipSuccess;
}
void launch_ids_by_rank(const uint32_t *pidx, const uint32_t *rank, int64_t m, uint32_t *ids, hipStream_t st)
{
    if (m > 0) hipLaunchKernelGGL(ids_by_rank_kernel, dim3(nblk(m)), dim3(256), 0, st, pidx, rank, m, ids);
}

}  // namespace s2m
