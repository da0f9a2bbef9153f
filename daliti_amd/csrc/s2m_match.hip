// s2m_match.hip -- rematch pass, part 1: exact 5 nearest map points of every scan point.
//
// Replaces, per scan point (eskf_lio/src/laserMapping.cpp:835-850): the body->world transform
// (:835-841) and ikdtree.Nearest_Search(point_world, 5, points_near, pointSearchSqDis) (:850;
// eskf_lio/include/ikd-Tree/ikd_Tree.cpp:425-461, 1061-1244, 1682-1709).  Output is
// Nearest_Points as indices into the caller's map array plus the ascending squared distances; the
// neighbour gate (:852-854) and esti_plane (:863) run in the thread-per-point kernel of
// s2m_reduce.hip, where one wave instruction serves 64 scan points instead of 4.
//
// Candidates are ranked by the 64-bit key (float bits of d2) << 32 | original index: d2 >= 0 so the
// bit pattern orders like the value, the index makes keys unique, and a top-5 insertion is a
// handful of 64-bit compare/selects with no tie branches (ikd-Tree ranks by d2, then x,
// ikd_Tree.h:102-108; exact ties are ~1e-7 of queries and either choice is a valid exact 5-NN).
//
// Two kernels, because measured cost is VALU issue + a long tail of far queries:
//   match_easy<G> : G lanes (1..8) per scan point scan the 3x3x3 cells around it -- 9 x-rows, each
//                   one or two (top entry, table pair, point run) lookups, issued phase by phase so
//                   every lane has all its loads of a phase in flight together.  If the 5th-best
//                   distance is provably inside the cube the result is final; otherwise the point
//                   is appended to the hard list.
//   match_hard    : one wave per hard point; the 64 lanes split the x-rows of a growing cube (only
//                   new shells are scanned), jump straight to the radius implied by the current
//                   5th-best distance, and stop once the bound passes the d2 <= 5 gate (:853).
// Brick row masks (s2m_device.h) skip the table lookups of rows that hold no points.
//
// Arithmetic contract: compiled with -ffp-contract=off; d2 = ((dx*dx + dy*dy) + dz*dz) in float
// exactly like calc_dist (ikd_Tree.cpp:1682-1688) and oracle/s2m_oracle.c.
#include <algorithm>
#include <cfloat>
#include <cmath>

#include "s2m_device.h"
#include "s2m_kernels.h"

namespace s2m {

typedef unsigned long long u64;
#ifndef S2M_INSERT_F64
#define S2M_INSERT_F64 1
#endif
// empty slot: +infinity as a double when the f64 exchange chain is used (orders after every key)
constexpr u64 kEmptyKey = S2M_INSERT_F64 ? 0x7ff0000000000000ull : ~0ull;
#ifndef S2M_EASY_BATCH
#define S2M_EASY_BATCH 8
#endif
constexpr int kEasyBatch = S2M_EASY_BATCH;  // point loads in flight per lane in the first-shell kernel
#ifndef S2M_HARD_BATCH
#define S2M_HARD_BATCH 8
#endif
constexpr int kHardBatch = S2M_HARD_BATCH;  // same for the one-cell-per-lane kernel

__device__ __forceinline__ u64 make_key(float d2, uint32_t orig)
{
    return ((u64)__float_as_uint(d2) << 32) | (u64)orig;
}

// Sorted ascending top-5 of unique keys.  A key with a non-negative float in its high word is a
// finite positive double whose IEEE order equals the unsigned order of the bits (float exponent
// 0xFF maps to double exponent <= 0x7FC, still finite), so one compare-exchange is v_min_f64 +
// v_max_f64: the insertion is ten branch-free instructions with no SGPR/exec traffic.  The empty
// key ~0 is a NaN as a double, so it is tested on the integer pattern before the exchange chain.
__device__ __forceinline__ void insert5(u64 (&t)[kK], u64 k)
{
#if S2M_INSERT_F64
    double kd = __longlong_as_double((long long)k);
#pragma unroll
    for (int i = 0; i < kK; ++i) {
        const double ti = __longlong_as_double((long long)t[i]);
        const double lo = fmin(ti, kd), hi = fmax(ti, kd);
        t[i] = (u64)__double_as_longlong(lo);
        kd = hi;
    }
#else
    if (k < t[kK - 1]) {
        t[kK - 1] = k;
#pragma unroll
        for (int i = kK - 2; i >= 0; --i) {
            const u64 a = t[i], b = t[i + 1];
            const bool sw = b < a;
            t[i] = sw ? b : a;
            t[i + 1] = sw ? a : b;
        }
    }
#endif
}

// points pts[s, e) -> top-5, B independent 16-byte loads per batch
template <int B>
__device__ __forceinline__ void scan_points(const float4 *__restrict__ pts, uint32_t s, uint32_t e, float wx,
                                            float wy, float wz, u64 (&t)[kK])
{
    for (uint32_t i = s; i < e; i += B) {
        float4 p[B];
#pragma unroll
        for (int u = 0; u < B; ++u) p[u] = pts[min(i + u, e - 1)];
#pragma unroll
        for (int u = 0; u < B; ++u) {
            const float dx = wx - p[u].x, dy = wy - p[u].y, dz = wz - p[u].z;
            float d = dx * dx + dy * dy;
            d = d + dz * dz;
            const u64 k = (i + u < e) ? make_key(d, __float_as_uint(p[u].w)) : kEmptyKey;
            insert5(t, k);
        }
    }
}

// cells [xa, xb] of x-row (yy, zz): per brick one 16-byte top entry (id + row mask), two table
// words, then the point run
__device__ __forceinline__ void scan_row(const Grid &g, int yy, int zz, int xa, int xb, float wx, float wy,
                                         float wz, u64 (&t)[kK])
{
    const int by = yy >> 3, bz = zz >> 3;
    const int rowbit = ((zz & 7) << 3) | (yy & 7);
    const int64_t toprow = ((int64_t)bz * g.nby + by) * g.nbx;
    for (int bx = xa >> 3; bx <= (xb >> 3); ++bx) {
        const uint4 te = g.top[toprow + bx];
        const uint32_t mword = (rowbit & 32) ? te.w : te.z;
        if (te.x == 0 || ((mword >> (rowbit & 31)) & 1u) == 0) continue;
        const int l0 = max(xa, bx << 3) & 7, l1 = min(xb, (bx << 3) + 7) & 7;
        const uint32_t *tb = g.tab + (int64_t)(te.x - 1) * kBrickStride + (rowbit << 3);
        scan_points<4>(g.pts, tb[l0], tb[l1 + 1], wx, wy, wz, t);
    }
}

template <int G>
__device__ __forceinline__ u64 group_min_u64(u64 v)
{
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) {
        const u64 o = __shfl_xor(v, off, G);
        v = o < v ? o : v;
    }
    return v;
}

// group-wide sorted top-5 of the G private lists (keys are unique, so the owner of the minimum is
// the one lane whose head equals it); non-destructive
template <int G>
__device__ __forceinline__ void merge_lists(const u64 (&priv)[kK], u64 (&best)[kK])
{
    if (G == 1) {
#pragma unroll
        for (int k = 0; k < kK; ++k) best[k] = priv[k];
        return;
    }
    u64 t[kK];
#pragma unroll
    for (int k = 0; k < kK; ++k) t[k] = priv[k];
#pragma unroll
    for (int k = 0; k < kK; ++k) {
        const u64 m = group_min_u64<G>(t[0]);
        best[k] = m;
        if (t[0] == m && m != kEmptyKey) {
#pragma unroll
            for (int s = 0; s < kK - 1; ++s) t[s] = t[s + 1];
            t[kK - 1] = kEmptyKey;
        }
    }
}

struct Query {
    float wx, wy, wz;
    int cx, cy, cz;
    float frx, fry, frz;  // position inside the home cell, in cells, [0, 1)
    float fmin;           // distance from the query to the nearest face of its home cell, in cells
};

__device__ __forceinline__ Query make_query(const Grid &g, const Pose &pose, float bx, float by, float bz)
{
    Query q;
    body_to_world(pose, bx, by, bz, q.wx, q.wy, q.wz);
    const float fx = (q.wx - g.ox) * g.inv_c, fy = (q.wy - g.oy) * g.inv_c, fz = (q.wz - g.oz) * g.inv_c;
    const float flx = floorf(fx), fly = floorf(fy), flz = floorf(fz);
    // clamp far-away queries so the int conversion is defined; the bound stays valid because the
    // clamped cells lie outside the grid and hold no points
    const float lim = 1.0e9f;
    q.cx = (int)fminf(fmaxf(flx, -lim), lim);
    q.cy = (int)fminf(fmaxf(fly, -lim), lim);
    q.cz = (int)fminf(fmaxf(flz, -lim), lim);
    q.frx = fx - flx; q.fry = fy - fly; q.frz = fz - flz;
    float f = fminf(fminf(q.frx, 1.0f - q.frx), fminf(q.fry, 1.0f - q.fry));
    q.fmin = fminf(f, fminf(q.frz, 1.0f - q.frz));
    return q;
}

// every point outside the cube of radius r (cells) around the home cell is at least this far
// (squared) from the query; slop covers the float rounding of cell coordinates
__device__ __forceinline__ float cube_bound2(const Grid &g, const Query &q, int r)
{
    float lb = ((float)r + q.fmin - g.slop) * g.c;
    lb = fmaxf(lb, 0.0f) * 0.999999f;
    return lb * lb;
}

// lower bound (squared) of the distance from the query to any point of the cell at offset
// (dx, dy, dz) cells from the home cell; slop as in cube_bound2
__device__ __forceinline__ float cell_bound2(const Grid &g, const Query &q, int dx, int dy, int dz)
{
    const float gx = dx > 0 ? (float)dx - q.frx : (dx < 0 ? q.frx - (float)(dx + 1) : 0.0f);
    const float gy = dy > 0 ? (float)dy - q.fry : (dy < 0 ? q.fry - (float)(dy + 1) : 0.0f);
    const float gz = dz > 0 ? (float)dz - q.frz : (dz < 0 ? q.frz - (float)(dz + 1) : 0.0f);
    const float ax = fmaxf(gx - g.slop, 0.0f), ay = fmaxf(gy - g.slop, 0.0f), az = fmaxf(gz - g.slop, 0.0f);
    return ((ax * ax + ay * ay) + az * az) * (g.c * g.c) * 0.99999f;
}

__device__ __forceinline__ void store_result(const u64 (&best)[kK], int64_t q, int32_t *__restrict__ nn_idx,
                                             float *__restrict__ nn_d2)
{
#pragma unroll
    for (int k = 0; k < kK; ++k) {
        const bool has = best[k] != kEmptyKey;
        nn_idx[q * kK + k] = has ? (int32_t)(uint32_t)(best[k] & 0xffffffffull) : -1;
        nn_d2[q * kK + k] = has ? __uint_as_float((uint32_t)(best[k] >> 32)) : INFINITY;
    }
}

// ---- first shell: 3x3x3 cells, G lanes per scan point -----------------------------------------------
template <int G>
__global__ __launch_bounds__(256) void match_easy(MatchArgs a)
{
    constexpr int R = (9 + G - 1) / G;  // x-rows per lane
    const long long t0 = a.dbg ? wall_clock64() : 0;
    const Grid &g = a.grid;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int qi = tid / G;
    const int j = tid % G;
    if (qi >= a.n) return;  // group-uniform
    const Query q = make_query(g, a.pose, a.sx[qi], a.sy[qi], a.sz[qi]);

    const int xlo = max(q.cx - 1, 0), xhi = min(q.cx + 1, g.ncx - 1);
    const int b0 = xlo >> 3, b1 = xhi >> 3;
    // phase 1: the top entries of every (row, brick) part of this lane
    uint32_t id[R][2];
    int rowoff[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int row = j + i * G;
        const int yy = q.cy + (row % 3) - 1, zz = q.cz + (row / 3) - 1;
        const bool ok = row < 9 && xlo <= xhi && yy >= 0 && yy < g.ncy && zz >= 0 && zz < g.ncz;
        const int rowbit = ((zz & 7) << 3) | (yy & 7);
        rowoff[i] = rowbit << 3;
        const int64_t toprow = ((int64_t)(zz >> 3) * g.nby + (yy >> 3)) * g.nbx;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            id[i][p] = 0;
            if (ok && (p == 0 || b1 != b0)) {
                const uint4 te = g.top[toprow + (p == 0 ? b0 : b1)];
                const uint32_t mword = (rowbit & 32) ? te.w : te.z;
                if ((mword >> (rowbit & 31)) & 1u) id[i][p] = te.x;
            }
        }
    }
    // phase 2: the table words of every non-empty part
    uint32_t s[R][2], e[R][2];
#pragma unroll
    for (int i = 0; i < R; ++i) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            s[i][p] = 0;
            e[i][p] = 0;
            if (id[i][p] != 0) {
                const int bx = (p == 0) ? b0 : b1;
                const int l0 = max(xlo, bx << 3) & 7, l1 = min(xhi, (bx << 3) + 7) & 7;
                const uint32_t *tb = g.tab + (int64_t)(id[i][p] - 1) * kBrickStride + rowoff[i];
                s[i][p] = tb[l0];
                e[i][p] = tb[l1 + 1];
            }
        }
    }
    // phase 3: the points
    u64 t[kK], best[kK];
#pragma unroll
    for (int k = 0; k < kK; ++k) t[k] = kEmptyKey;
#pragma unroll
    for (int i = 0; i < R; ++i) {
#pragma unroll
        for (int p = 0; p < 2; ++p) scan_points<kEasyBatch>(g.pts, s[i][p], e[i][p], q.wx, q.wy, q.wz, t);
    }
    merge_lists<G>(t, best);
    const bool found5 = best[kK - 1] != kEmptyKey;
    const float d5 = __uint_as_float((uint32_t)(best[kK - 1] >> 32));
    const bool done = found5 && d5 <= cube_bound2(g, q, 1);
    // Unresolved points go to one of three lists by expected cost (cube radius implied by the current
    // 5th-best distance; unknown when fewer than five were found): match_hard starts the expensive ones
    // first so they do not form the tail of the launch.  One atomic per wave and list: same-address
    // atomics serialise in L2 (~90 per microsecond), ten thousand per-lane atomics would cost > 100 us.
    const int rn_est = found5 ? (int)ceilf(sqrtf(d5) * g.inv_c * 1.000002f - q.fmin + g.slop) : 4;
    const int cls = (j == 0 && !done) ? (rn_est <= 2 ? 2 : (rn_est == 3 ? 1 : 0)) : -1;
    const int lane64 = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const unsigned long long mask = __ballot(cls == c);
        if (mask == 0ull) continue;  // wave-uniform
        const int leader = __ffsll((long long)mask) - 1;
        uint32_t base = 0;
        if (lane64 == leader) base = atomicAdd(a.hard_count + c, (uint32_t)__popcll(mask));
        base = __shfl(base, leader, 64);
        if (cls == c) {
            const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane64) - 1ull));
            a.hard_list[(int64_t)c * a.n + base + rank] = (uint32_t)qi;
        }
    }
    if (j == 0) {
        // unresolved points keep their first-shell list too: match_hard continues from it
        store_result(best, qi, a.nn_idx, a.nn_d2);
        if (a.dbg) {
            a.dbg[4 * (int64_t)qi + 0] = (uint32_t)(wall_clock64() - t0);
            a.dbg[4 * (int64_t)qi + 1] = done ? 1u : 0u;
            a.dbg[4 * (int64_t)qi + 2] = 0;
            a.dbg[4 * (int64_t)qi + 3] = 1;
        }
    }
}

// ---- the rest: G lanes (a whole or a fraction of a wave) per hard scan point ------------------------
template <int G>
__global__ __launch_bounds__(256) void match_hard(MatchArgs a)
{
    const Grid &g = a.grid;
    const int lane = threadIdx.x & (G - 1);                           // lane within the group
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / G;     // group index
    const int nwaves = (gridDim.x * blockDim.x) / G;
    const uint32_t c0 = a.hard_count[0], c1 = a.hard_count[1], c2 = a.hard_count[2];
    const uint32_t count = c0 + c1 + c2;
    const int rcap = max(max(g.ncx, g.ncy), g.ncz);
    for (uint32_t h = wave; h < count; h += nwaves) {
        const long long t0 = a.dbg ? wall_clock64() : 0;
        // concatenation [far | mid | near]: the most expensive points are handed out first
        const int qi = (int)(h < c0 ? a.hard_list[h]
                                    : (h < c0 + c1 ? a.hard_list[(int64_t)a.n + (h - c0)]
                                                   : a.hard_list[2 * (int64_t)a.n + (h - c0 - c1)]));
        const Query q = make_query(g, a.pose, a.sx[qi], a.sy[qi], a.sz[qi]);
        // smallest radius whose bound passes the d2 gate: beyond it the 5th neighbour cannot matter
        const int rgate = (int)ceilf(sqrtf(a.gates.knn_d2_gate) * g.inv_c * 1.000002f - q.fmin + g.slop) + 1;
        // continue from the first shell: lane 0 carries its five keys, the cube of radius 1 is done
        u64 t[kK], best[kK];
#pragma unroll
        for (int k = 0; k < kK; ++k) {
            const int32_t ci = a.nn_idx[(int64_t)qi * kK + k];
            const float cd = a.nn_d2[(int64_t)qi * kK + k];
            t[k] = (lane == 0 && ci >= 0) ? make_key(cd, (uint32_t)ci) : kEmptyKey;
            best[k] = (ci >= 0) ? make_key(cd, (uint32_t)ci) : kEmptyKey;
        }
        int rdone = 1;
        int r;
        {
            const bool f5 = best[kK - 1] != kEmptyKey;
            const float d5 = __uint_as_float((uint32_t)(best[kK - 1] >> 32));
            const int rn = f5 ? (int)ceilf(sqrtf(d5) * g.inv_c * 1.000002f - q.fmin + g.slop) : 2;
            r = min(min(max(rn, 2), max(rgate, 2)), max(rcap, 2));
        }
        uint32_t rounds = 0;
        for (;;) {
            // scan the shell (rdone, r]: one cell per lane, so every lane's top entry, table pair and
            // point loads are in flight together; cells of the already scanned inner cube and cells
            // that cannot hold anything closer than the current 5th-best are skipped
            const float tau = __uint_as_float((uint32_t)(best[kK - 1] >> 32));  // +inf/NaN pattern when < 5 found
            const bool have_tau = best[kK - 1] != kEmptyKey;
            const int side = 2 * r + 1;
            const int ncell = side * side * side;
            for (int ci = lane; ci < ncell; ci += G) {
                const int dx = (ci % side) - r, dy = ((ci / side) % side) - r, dz = (ci / (side * side)) - r;
                if (max(max(abs(dx), abs(dy)), abs(dz)) <= rdone) continue;
                const int xx = q.cx + dx, yy = q.cy + dy, zz = q.cz + dz;
                if (xx < 0 || xx >= g.ncx || yy < 0 || yy >= g.ncy || zz < 0 || zz >= g.ncz) continue;
                if (have_tau && cell_bound2(g, q, dx, dy, dz) > tau) continue;
                const int rowbit = ((zz & 7) << 3) | (yy & 7);
                const uint4 te = g.top[((int64_t)(zz >> 3) * g.nby + (yy >> 3)) * g.nbx + (xx >> 3)];
                const uint32_t mword = (rowbit & 32) ? te.w : te.z;
                if (te.x == 0 || ((mword >> (rowbit & 31)) & 1u) == 0) continue;
                const uint32_t *tb = g.tab + (int64_t)(te.x - 1) * kBrickStride + (rowbit << 3) + (xx & 7);
                scan_points<kHardBatch>(g.pts, tb[0], tb[1], q.wx, q.wy, q.wz, t);
            }
            rdone = r;
            ++rounds;
            merge_lists<G>(t, best);
            const bool found5 = best[kK - 1] != kEmptyKey;
            const float d5 = __uint_as_float((uint32_t)(best[kK - 1] >> 32));
            const float lb2 = cube_bound2(g, q, r);
            if (found5 && d5 <= lb2) break;          // exact 5-NN found
            if (lb2 > a.gates.knn_d2_gate) break;    // the 5th neighbour is beyond the gate (:853)
            if (r >= rcap) break;                    // whole grid scanned
            // all better candidates lie within sqrt(d5): jump straight to the radius covering it
            int rn = found5 ? (int)ceilf(sqrtf(d5) * g.inv_c * 1.000002f - q.fmin + g.slop) : 2 * r;
            r = min(max(rn, r + 1), max(rgate, r + 1));
            r = min(r, rcap);
        }
        if (lane == 0) {
            store_result(best, qi, a.nn_idx, a.nn_d2);
            if (a.dbg) {
                a.dbg[4 * (int64_t)qi + 0] += (uint32_t)(wall_clock64() - t0);
                a.dbg[4 * (int64_t)qi + 1] = (uint32_t)rdone;
                a.dbg[4 * (int64_t)qi + 3] = rounds + 1;
            }
        }
    }
}

template <int G>
static void launch_easy(const MatchArgs &a, hipStream_t st)
{
    const int64_t threads = (int64_t)a.n * G;
    const int blocks = (int)((threads + 255) / 256);
    hipLaunchKernelGGL(match_easy<G>, dim3(blocks), dim3(256), 0, st, a);
}

void launch_match(const MatchArgs &a, int group, hipStream_t st)
{
    if (a.n <= 0) return;
    switch (group & 0xff) {
        case 1: launch_easy<1>(a, st); break;
        case 2: launch_easy<2>(a, st); break;
        case 8: launch_easy<8>(a, st); break;
        default: launch_easy<4>(a, st); break;
    }
    // fixed grid, groups stride over the hard list whose length is only known on the device
    const int hg = (group >> 8) ? (group >> 8) : 64;  // hard-kernel group width rides in bits 8.. (tuning hook;
                                                      // measured: 64 lanes per hard point is fastest, the far tail is latency-bound)
    const int64_t groups = std::min<int64_t>(a.n, 8192 * (64 / hg));
    const int blocks = (int)((groups * hg + 255) / 256);
    if (hg == 16) hipLaunchKernelGGL(match_hard<16>, dim3(blocks), dim3(256), 0, st, a);
    else if (hg == 64) hipLaunchKernelGGL(match_hard<64>, dim3(blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(match_hard<32>, dim3(blocks), dim3(256), 0, st, a);
}

}  // namespace s2m
