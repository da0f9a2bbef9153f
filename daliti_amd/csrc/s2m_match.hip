// s2m_match.hip -- rematch pass, part 1: exact 5 nearest map points of every scan point.
//
// Replaces, per scan point (eskf_lio/src/laserMapping.cpp:835-850): the body->world transform
// (:835-841) and ikdtree.Nearest_Search(point_world, 5, points_near, pointSearchSqDis) (:850;
// eskf_lio/include/ikd-Tree/ikd_Tree.cpp:425-461, 1061-1244, 1682-1709).  Output is
// Nearest_Points as indices into the caller's map array plus the ascending squared distances; the
// neighbour gate (:852-854) and esti_plane (:863) run in the thread-per-point kernel of
// s2m_reduce.hip, where one wave instruction serves 64 scan points instead of 4.
//
// Candidates are ranked by the 64-bit key (float bits of d2) << 32 | sorted position: d2 >= 0 so the
// bit pattern orders like the value, the position makes keys unique, and a top-5 insertion is a
// handful of 64-bit compare/selects with no tie branches.  Sorted position = (brick, cell, caller index), a
// total order the oracle computes from s2m_map_info (ikd-Tree ranks by d2, then x, ikd_Tree.h:102-108, with
// a traversal-dependent choice at the 5th place; exact ties are ~1e-7 of queries and every choice is a valid
// exact 5-NN).  The position doubles as the gather address of the plane fit (s2m_reduce.hip).
//
// Two kernels, because measured cost is a long tail of far queries on top of the first-shell work:
//   match_rows<G> : G lanes (default 2) per scan point scan the 3x3x3 cells around it as nine x-ROWS (the G lanes read
//                   adjacent points of every run): the three
//                   cells of a row are one contiguous run of points (two when the row straddles a brick), the
//                   home row first, the other rows trimmed by its 5th-best distance, all of a lane's runs walked
//                   as one flat sequence of 8-point batches, three batches (24 loads) in flight per trip.  If the
//                   5th-best distance is provably inside the cube the result is final; otherwise the point is
//                   appended to the hard list.  (match_easy<G> is the earlier per-CELL form of the same search,
//                   kept behind S2M_EASY_CELLS=1 for A/B measurements: 2.6x the VALU and 2.8x the vector-memory
//                   instructions for the same answer.)
//   match_hard    : one wave per hard point.  The x-rows that can hold a point within the current radius
//                   (the first shell's 5th distance, else a growing band) are found either directly -- the
//                   7x7 rows around the home row while the radius is within three cells -- or from the
//                   row masks of the surrounding bricks; their cells are expanded into an LDS cell list by
//                   a DPP prefix sum and dealt to the 64 lanes, so the point loads of all rows are in
//                   flight together.  Stops once the bound passes the d2 <= 5 gate (:853).
// Padding slots of a point batch load a sentinel point beyond the array (distance +inf), so the inner loop
// has no predicate; keys go into the top-5 eight at a time through a pruned sorting network.
// Brick row masks (s2m_device.h) skip the table lookups of rows that hold no points.
//
// Arithmetic contract: compiled with -ffp-contract=off; d2 = ((dx*dx + dy*dy) + dz*dz) in float
// exactly like calc_dist (ikd_Tree.cpp:1682-1688) and oracle/s2m_oracle.c.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>

#include "s2m_device.h"
#include "s2m_kernels.h"

namespace s2m {

typedef unsigned long long u64;
#ifndef S2M_INSERT_F64
#define S2M_INSERT_F64 1
#endif
// empty slot: +infinity as a double when the f64 exchange chain is used (orders after every key)
constexpr u64 kEmptyKey = S2M_INSERT_F64 ? 0x7ff0000000000000ull : ~0ull;
// A slot is empty when the float in its high word is not a finite distance: the initial +inf pattern
// above, or the key of the sentinel point pts[m] = (3e38, 3e38, 3e38) that padding lanes of a batch load
// (its squared distance overflows to +inf), so padding needs no predicate anywhere after the address.
__device__ __forceinline__ bool is_empty(u64 k) { return (uint32_t)(k >> 32) >= 0x7f800000u; }
#ifndef S2M_EASY_BATCH
#define S2M_EASY_BATCH 8
#endif
constexpr int kEasyBatch = S2M_EASY_BATCH;  // point loads in flight per lane in the first-shell kernel
#ifndef S2M_HARD_BATCH
#define S2M_HARD_BATCH 8
#endif
constexpr int kHardBatch = S2M_HARD_BATCH;  // same for the one-cell-per-lane kernel
#ifndef S2M_HARD_BAND
#define S2M_HARD_BAND 1.7f  // first band of match_hard in cells (measured sweeps in DESIGN.md)
#endif

struct Query;
__device__ __forceinline__ void append_rec(HardRec *__restrict__ list, uint32_t *__restrict__ counter, bool want,
                                           const HardRec &rec);
__device__ __forceinline__ u64 make_key(float d2, uint32_t orig)
{
    return ((u64)__float_as_uint(d2) << 32) | (u64)orig;
}

// v_min_f64 / v_max_f64 issued directly: fmin()/fmax() make the compiler canonicalise every operand first
// (one extra v_max_f64 x, x, x per key -- 13 of ~61 operations per batch); the keys are never NaN as
// doubles (a float's bits in the high word give an exponent below 0x7fd, see insert5), so the result is
// the same.  Not volatile: unused halves of a comparator are still removed.
__device__ __forceinline__ double min_raw(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double max_raw(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
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
        const double lo = min_raw(ti, kd), hi = max_raw(ti, kd);
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

#ifndef S2M_BATCH_SORT
#define S2M_BATCH_SORT 1
#endif
__device__ __forceinline__ void cex(double &a, double &b)
{
    const double lo = min_raw(a, b), hi = max_raw(a, b);
    a = lo; b = hi;
}
// Eight new keys into the sorted top-5 in 48 min/max instead of 80: the 19-comparator sorting network for
// eight inputs with everything that only feeds outputs 5..7 left to dead-code elimination (33 operations),
// then c[i] = min(t[i], s[4-i]) -- the five smallest of both lists, as an up-down sequence -- and the
// five-comparator network that sorts every up-down sequence of five (found by exhaustive search over the
// 0/1 inputs).  Keys are unique (or +inf), so the result is the same list insert5 produces one by one.
__device__ __forceinline__ void insert_batch8(u64 (&t)[kK], const u64 (&k)[8])
{
    double v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = __longlong_as_double((long long)k[i]);
    cex(v[0], v[2]); cex(v[1], v[3]); cex(v[4], v[6]); cex(v[5], v[7]);
    cex(v[0], v[4]); cex(v[1], v[5]); cex(v[2], v[6]); cex(v[3], v[7]);
    cex(v[0], v[1]); cex(v[2], v[3]); cex(v[4], v[5]); cex(v[6], v[7]);
    cex(v[2], v[4]); cex(v[3], v[5]);
    cex(v[1], v[4]); cex(v[3], v[6]);
    cex(v[1], v[2]); cex(v[3], v[4]); cex(v[5], v[6]);
    double c[kK];
#pragma unroll
    for (int i = 0; i < kK; ++i) c[i] = min_raw(__longlong_as_double((long long)t[i]), v[kK - 1 - i]);
    cex(c[0], c[4]); cex(c[1], c[3]); cex(c[1], c[4]); cex(c[2], c[4]); cex(c[3], c[4]);
#pragma unroll
    for (int i = 0; i < kK; ++i) t[i] = (u64)__double_as_longlong(c[i]);
}

// points pts[s, e) -> top-5, B independent 16-byte loads per batch.  Addresses are a uniform base plus a
// 32-bit byte offset (the saddr form of global_load: no 64-bit address arithmetic per slot); padding slots
// select the offset of the sentinel block pts[m .. m+B), which the instruction's immediate offset then
// indexes like any other batch.  Maps beyond 2^28 - 8 points take 64-bit addresses (WIDE).
template <int B, bool WIDE = false>
__device__ __forceinline__ void scan_points(const Grid &g, uint32_t s, uint32_t e, float wx, float wy, float wz,
                                            u64 (&t)[kK])
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    const float4 *__restrict__ pts = g.pts;
    const f2 wxy = {wx, wy};
    for (uint32_t i = s; i < e; i += B) {
        float4 p[B];
        if (WIDE) {
            const uint32_t sent = (uint32_t)g.m;
#pragma unroll
            for (int u = 0; u < B; ++u) p[u] = pts[(i + u < e) ? i + u : sent];
        } else {
            const uint32_t left = e - i, off = i << 4, soff = g.sent_off;
            const char *base = reinterpret_cast<const char *>(pts);
#pragma unroll
            for (int u = 0; u < B; ++u)
                p[u] = *reinterpret_cast<const float4 *>(base + (size_t)(((uint32_t)u < left) ? off : soff) + 16 * u);
        }
        u64 key[B];
#pragma unroll
        for (int u = 0; u < B; ++u) {
            // (dx, dy) as one packed pair straight from the loaded words; same operations and order as
            // the scalar form: (dx*dx + dy*dy) + dz*dz
            const f2 dxy = wxy - f2{p[u].x, p[u].y};
            const f2 sq = dxy * dxy;
            const float dz = wz - map_point_z(p[u]);
            float d = sq.x + sq.y;
            d = d + dz * dz;
            key[u] = make_key(d, map_point_pos(p[u]));
        }
        if (S2M_BATCH_SORT && S2M_INSERT_F64 && B == 8) {
            u64 k8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) k8[u] = key[u < B ? u : 0];
            insert_batch8(t, k8);
        } else {
#pragma unroll
            for (int u = 0; u < B; ++u) insert5(t, key[u]);
        }
    }
}

// ---- group-wide minimum of a 64-bit key ------------------------------------------------------------
// Generic form: xor-shuffle butterfly (ds_bpermute, ~100+ cycles per step).  For a whole wave and for
// quads the minimum is taken with DPP instead (v_min_u32_dpp: no LDS pipe, a few cycles per step), in
// two 32-bit phases: the smallest high word first, then the smallest low word among its holders.
__device__ __forceinline__ uint32_t dpp_min_step(uint32_t v, const int ctrl_tag)
{
    // the control word must be a compile-time constant, hence the switch
    uint32_t o;
    switch (ctrl_tag) {
        case 0: o = (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, 0x111, 0xf, 0xf, false); break;  // row_shr:1
        case 1: o = (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, 0x112, 0xf, 0xf, false); break;  // row_shr:2
        case 2: o = (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, 0x114, 0xf, 0xf, false); break;  // row_shr:4
        case 3: o = (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, 0x118, 0xf, 0xf, false); break;  // row_shr:8
        case 4: o = (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, 0x142, 0xa, 0xf, false); break;  // row_bcast:15
        case 5: o = (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, 0x143, 0xc, 0xf, false); break;  // row_bcast:31
        case 6: o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xf, 0xf, true); break;              // quad_perm [1,0,3,2]
        default: o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xf, 0xf, true); break;             // quad_perm [2,3,0,1]
    }
    return min(v, o);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
#pragma unroll
    for (int k = 0; k < 6; ++k) v = dpp_min_step(v, k);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);  // lane 63 holds the minimum of the whole wave
}
// inclusive prefix sum over the wave with the same six DPP steps (lanes without a source add 0): no LDS
// traffic, ~6 instructions instead of six ds_bpermute round trips
__device__ __forceinline__ int wave_incl_scan(int x)
{
    int v = x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ uint32_t quad_min_u32(uint32_t v)
{
    v = dpp_min_step(v, 6);
    return dpp_min_step(v, 7);
}

template <int G>
__device__ __forceinline__ u64 group_min_u64(u64 v)
{
    if (G == 64 || G == 4) {
        const uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
        const uint32_t mh = (G == 64) ? wave_min_u32(hi) : quad_min_u32(hi);
        const uint32_t lo2 = (hi == mh) ? lo : 0xffffffffu;
        const uint32_t ml = (G == 64) ? wave_min_u32(lo2) : quad_min_u32(lo2);
        return ((u64)mh << 32) | (u64)ml;
    }
    if (G == 2) {  // the partner is lane ^ 1: one quad_perm DPP move per word instead of a trip through ds_bpermute
        const uint32_t oh = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), 0xB1, 0xf, 0xf, false);
        const uint32_t ol = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, 0xB1, 0xf, 0xf, false);
        const u64 o = ((u64)oh << 32) | (u64)ol;
        return o < v ? o : v;
    }
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
        if (t[0] == m && !is_empty(m)) {
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

// home cell and in-cell position of a world-frame query point
__device__ __forceinline__ Query query_at(const Grid &g, float wx, float wy, float wz)
{
    Query q;
    q.wx = wx; q.wy = wy; q.wz = wz;
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
__device__ __forceinline__ Query make_query(const Grid &g, const Pose &pose, float bx, float by, float bz)
{
    float wx, wy, wz;
    body_to_world(pose, bx, by, bz, wx, wy, wz);
    return query_at(g, wx, wy, wz);
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
        const bool has = !is_empty(best[k]);
        nn_idx[q * kK + k] = has ? (int32_t)(uint32_t)(best[k] & 0xffffffffull) : -1;
        nn_d2[q * kK + k] = has ? __uint_as_float((uint32_t)(best[k] >> 32)) : INFINITY;
    }
}

// ---- first shell: 3x3x3 cells, G lanes per scan point -----------------------------------------------
// one cell of the first shell: brick id (0 = nothing to read) and position of its table word
struct CellRef {
    uint32_t id;
    int word;  // row offset + x within the brick
};

__device__ __forceinline__ CellRef cell_ref(const Grid &g, int xx, int yy, int zz, bool ok)
{
    CellRef r;
    r.id = 0;
    r.word = 0;
    if (ok && xx >= 0 && xx < g.ncx && yy >= 0 && yy < g.ncy && zz >= 0 && zz < g.ncz) {
        const int rowbit = ((zz & 7) << 3) | (yy & 7);
        const uint4 te = g.top[((int64_t)(zz >> 3) * g.nby + (yy >> 3)) * g.nbx + (xx >> 3)];
        const uint32_t mword = (rowbit & 32) ? te.w : te.z;
        if ((mword >> (rowbit & 31)) & 1u) r.id = te.x;
        r.word = (rowbit << 3) + (xx & 7);
    }
    return r;
}

template <int G, bool WIDE>
__global__ __launch_bounds__(256) void match_easy(MatchArgs a)
{
    constexpr int HC = (3 + G - 1) / G;  // home-row cells per lane
    constexpr int RR = (8 + G - 1) / G;  // other x-rows per lane
    const long long t0 = a.dbg ? wall_clock64() : 0;
    const Grid &g = a.grid;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int qi = tid / G;
    const int j = tid % G;
    if (qi >= a.n) return;  // group-uniform
    const Query q = make_query(g, a.pose, a.sx[qi], a.sy[qi], a.sz[qi]);

    // phase 1: top entries (brick id + row mask).  Home row first: its three cells one by one; then the
    // eight other rows, three cells each.  Everything a lane needs in a phase is requested together.
    CellRef hc[HC], rc[RR][3];
    int rdy[RR], rdz[RR];
#pragma unroll
    for (int i = 0; i < HC; ++i) {
        const int c = j + i * G;  // 0..2 -> dx = -1..1
        hc[i] = cell_ref(g, q.cx + c - 1, q.cy, q.cz, c < 3);
    }
#pragma unroll
    for (int i = 0; i < RR; ++i) {
        const int r8 = j + i * G;             // 0..7 -> the rows (dy,dz) != (0,0)
        const int row = r8 + (r8 >= 4 ? 1 : 0);  // skip the centre of the 3x3
        rdy[i] = (row % 3) - 1;
        rdz[i] = (row / 3) - 1;
#pragma unroll
        for (int c = 0; c < 3; ++c) rc[i][c] = cell_ref(g, q.cx + c - 1, q.cy + rdy[i], q.cz + rdz[i], r8 < 8);
    }
    // phase 2: table words [start, end) of every non-empty cell
    uint32_t hs[HC], he[HC], rs[RR][3], re[RR][3];
#pragma unroll
    for (int i = 0; i < HC; ++i) {
        hs[i] = 0; he[i] = 0;
        if (hc[i].id) {
            const uint32_t *tb = g.tab + (int64_t)(hc[i].id - 1) * kBrickStride + hc[i].word;
            hs[i] = tb[0]; he[i] = tb[1];
        }
    }
#pragma unroll
    for (int i = 0; i < RR; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            rs[i][c] = 0; re[i][c] = 0;
            if (rc[i][c].id) {
                const uint32_t *tb = g.tab + (int64_t)(rc[i][c].id - 1) * kBrickStride + rc[i][c].word;
                rs[i][c] = tb[0]; re[i][c] = tb[1];
            }
        }
    // phase 3a: the home row; its 5th-best distance (if it holds five points) prunes the rest
    u64 t[kK], best[kK];
#pragma unroll
    for (int k = 0; k < kK; ++k) t[k] = kEmptyKey;
#pragma unroll
    for (int i = 0; i < HC; ++i) scan_points<kEasyBatch, WIDE>(g, hs[i], he[i], q.wx, q.wy, q.wz, t);
    merge_lists<G>(t, best);
    const bool have_tau = !is_empty(best[kK - 1]);
    const float tau = __uint_as_float((uint32_t)(best[kK - 1] >> 32));
    // phase 3b: the other 24 cells, skipping those that cannot hold anything closer than tau
#pragma unroll
    for (int i = 0; i < RR; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (rs[i][c] < re[i][c] && !(have_tau && cell_bound2(g, q, c - 1, rdy[i], rdz[i]) > tau))
                scan_points<kEasyBatch, WIDE>(g, rs[i][c], re[i][c], q.wx, q.wy, q.wz, t);
        }
    merge_lists<G>(t, best);
    const bool found5 = !is_empty(best[kK - 1]);
    const float d5 = __uint_as_float((uint32_t)(best[kK - 1] >> 32));
    const bool done = found5 && d5 <= cube_bound2(g, q, 1);
    // Unresolved points go to one of three lists by expected cost (cube radius implied by the current
    // 5th-best distance; unknown when fewer than five were found): match_hard starts the expensive ones
    // first so they do not form the tail of the launch.  One atomic per wave and list: same-address
    // atomics serialise in L2 (~90 per microsecond), ten thousand per-lane atomics would cost > 100 us.
    // unresolved: match_hard's list, in two parts -- the points without a radius (fewer than five neighbours in the
    // shell: two rounds there, ~11 us against ~5.5 us) are handed out first so that they do not form the launch's tail
    {
        uint32_t found = 0;
#pragma unroll
        for (int k = 0; k < kK; ++k) found += is_empty(best[k]) ? 0u : 1u;
        const HardRec rec = {q.wx, q.wy, q.wz, (uint32_t)qi, d5, found, a.slot, 0u};
        append_rec(a.hard_rec, a.hard_count, j == 0 && !done && !found5, rec);
        append_rec(a.hard_rec + a.hard_off1, a.hard_count + 1, j == 0 && !done && found5, rec);
    }
    if (j == 0) {
        // unresolved points keep their first-shell list too: match_hard takes its radius from it, and its
        // query point from here (the pose alone is 48 SGPRs that kernel would spill around every point)
        store_result(best, qi, a.nn_idx, a.nn_d2);
        if (a.dbg) {
            a.dbg[4 * (int64_t)qi + 0] = (uint32_t)(wall_clock64() - t0);
            a.dbg[4 * (int64_t)qi + 1] = done ? 1u : 0u;
            a.dbg[4 * (int64_t)qi + 2] = 0;
            a.dbg[4 * (int64_t)qi + 3] = 1;
        }
    }
}

// ---- first shell, second form: whole x-rows as point runs, one flattened work list per lane ---------------
// The cells of one x-row of a brick are contiguous in pts, so the three cells (cx-1, cx, cx+1) of a row are ONE run
// of points -- two when the row straddles a brick boundary in x (cx & 7 is 0 or 7) -- found with one top entry and
// four consecutive prefix words of the brick table.  Padding a batch of eight then costs at most seven slots per
// RUN instead of per CELL (measured on the cell form: 488 point-load slots per query for ~108 real candidates;
// 8.48 M VALU and 0.50 M vector-memory wave-instructions per launch at C3 against 3.3-4.7 M and 0.17-0.21 M here).
// A lane's runs (its contiguous share of each run when G lanes serve one query) are chained through a small
// per-lane list in LDS into ONE flat sequence of batches, walked with the next batch's eight loads already in
// flight while the current one goes through the distance arithmetic and the top-5 network -- with half the
// instructions the kernel is latency- rather than issue-bound.  The home row goes first; its 5th-best distance
// (group-wide for G > 1) trims every other row to the cells that can still hold something closer.
struct __attribute__((packed, aligned(4))) TabQuad {
    uint32_t w[4];
};
__device__ __forceinline__ uint32_t sel4(const TabQuad &q, int k)
{
    return k == 0 ? q.w[0] : (k == 1 ? q.w[1] : (k == 2 ? q.w[2] : q.w[3]));
}
// run of the cells [lo, hi] (absolute x indices) of a segment that starts at cell sx0 and whose prefix words are q
__device__ __forceinline__ void seg_run(const TabQuad &q, int sx0, int sx1, int lo, int hi, uint32_t &s, uint32_t &e)
{
    lo = max(lo, sx0);
    hi = min(hi, sx1);
    s = 0; e = 0;
    if (lo <= hi) { s = sel4(q, lo - sx0); e = sel4(q, hi - sx0 + 1); }
}

// eight candidates of the run [i, e) for lane j of the S lanes that share it: pts[i + S u + j], u = 0..7 -- the S lanes
// of a query read ADJACENT points in every load instruction, so a pair's two 16-byte requests fall into one cache line
// (the texture addresser spends a cycle per distinct line and instruction: with a chunk per lane every request was its
// own line, and at saturation that unit was the busiest one, TA_BUSY 63 % of the batched first-shell kernel).  Slots
// beyond e (all of them when i >= e) read the sentinel block.  S = 1, j = 0: eight consecutive points.
#ifndef S2M_ROWS_PRED
#define S2M_ROWS_PRED 0  // padding slots of match_rows' batches: 1 = the load is skipped (exec-masked), 0 = it reads the sentinel block
#endif
template <bool WIDE, int S = 1, bool PRED = false>
__device__ __forceinline__ void load_batch(const Grid &g, uint32_t i, uint32_t e, float4 (&p)[8], uint32_t j = 0)
{
    static_assert(8 * S <= kSentinelPoints, "sentinel block too short");
    const float4 *__restrict__ pts = g.pts;
    const uint32_t left = e > i + j ? e - i - j : 0u;  // slot u is real iff S u < left
    if (PRED) {
        // padding slots issue no request at all (four slots in ten are padding, and the texture addresser moves 64 bytes a
        // clock whatever they hold); their registers keep whatever was there and consume_batch sets the distance to +inf
        const uint32_t off = (i + j) << 4;
        const char *base = reinterpret_cast<const char *>(pts);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            asm volatile("" : "=v"(p[u].x), "=v"(p[u].y), "=v"(p[u].z), "=v"(p[u].w));  // unspecified, not undefined
            if ((uint32_t)(S * u) < left)
                p[u] = WIDE ? pts[i + j + S * u] : *reinterpret_cast<const float4 *>(base + (size_t)off + 16 * S * u);
        }
        return;
    }
    if (WIDE) {
        const uint32_t sent = (uint32_t)g.m;
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] = pts[((uint32_t)(S * u) < left) ? i + j + S * u : sent + S * u];
    } else {
        const uint32_t off = (i + j) << 4, soff = g.sent_off;
        const char *base = reinterpret_cast<const char *>(pts);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            p[u] = *reinterpret_cast<const float4 *>(base + (size_t)(((uint32_t)(S * u) < left) ? off : soff) + 16 * S * u);
    }
}
// `left` (PRED): slots u with S u >= left hold no point
template <int S = 1, bool PRED = false>
__device__ __forceinline__ void consume_batch(const float4 (&p)[8], float wx, float wy, float wz, u64 (&t)[kK], uint32_t left = 0)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 wxy = {wx, wy};
    u64 key[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const f2 dxy = wxy - f2{p[u].x, p[u].y};
        const f2 sq = dxy * dxy;
        const float dz = wz - map_point_z(p[u]);
        float d = sq.x + sq.y;
        d = d + dz * dz;
        if (PRED) d = ((uint32_t)(S * u) < left) ? d : INFINITY;
        key[u] = make_key(d, map_point_pos(p[u]));
    }
#ifdef S2M_EXP_BATCH_REJECT
    // experiment (DESIGN, rejected table): skip the 48-operation network when no lane of the wave holds a candidate
    // closer than its current 5th best
    bool better = false;
#pragma unroll
    for (int u = 0; u < 8; ++u) better = better || key[u] < t[kK - 1];
    if (!__any(better)) return;
#endif
    insert_batch8(t, key);
}

constexpr int kRunSlots = 16;  // 8 rows x 2 segments besides the home row
#ifndef S2M_ROWS_EXT
#define S2M_ROWS_EXT 0  // experiment (round 3, DESIGN "tried and rejected"): 2 or 3 = the first-shell kernel finishes the points whose
                        // radius is known in place
#endif
#ifndef S2M_ROWS_QUADS
#define S2M_ROWS_QUADS 0  // experiment: the two prefix words of an outer-row piece as ONE 4-byte-aligned dwordx4 (half the requests)
#endif
#ifndef S2M_ROWS_SHARE
#define S2M_ROWS_SHARE 1  // the G lanes of a query read adjacent points of the same run (0: a chunk of the run each)
#endif

// position in a lane's chain of runs: run k of nr, next batch at i, run end e (i = e = 0 once exhausted)
struct RunCursor {
    uint32_t k, i, e;
};

#ifdef S2M_EXP_ROWS_TIMELINE
// experiment: a stamp that waits for `dep` (a value that depends on the stage's loads) before reading the clock
#define S2M_ROWS_STAMP(k, dep) do { auto d_ = (dep); asm volatile("" : "+v"(d_)); st_[k] = wall_clock64(); } while (0)
#else
#define S2M_ROWS_STAMP(k, dep) do { } while (0)
#endif

template <int G, bool WIDE, int NB>
__device__ __forceinline__ void match_rows_body(const MatchArgs &a, uint2 *__restrict__ runs)
{
    const long long t0 = a.dbg ? wall_clock64() : 0;
#ifdef S2M_EXP_ROWS_TIMELINE
    long long st_[4] = {0, 0, 0, 0};
#endif
    const Grid &g = a.grid;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int qi = tid / G;
    const int j = tid % G;
    if (qi >= a.n) return;  // group-uniform
    constexpr int kShare = (S2M_ROWS_SHARE && !S2M_ROWS_EXT && G <= 4) ? G : 1;  // lanes that share a run point by point
    constexpr bool kPred = S2M_ROWS_PRED != 0;
    const Query q = make_query(g, a.pose, a.sx[qi], a.sy[qi], a.sz[qi]);
    S2M_ROWS_STAMP(0, q.cx);
    // x extent of the neighbourhood inside the grid; segment A lies in brick bA, segment B (if any) in bA + 1
    const int x_lo = max(q.cx - 1, 0), x_hi = min(q.cx + 1, g.ncx - 1);
    const bool xok = x_lo <= x_hi;
    const int bA = x_lo >> 3, bB = x_hi >> 3;
    const bool split = xok && bB != bA;
    const int ax1 = split ? (bA << 3) + 7 : x_hi;  // last cell of segment A
    const int bx0 = bB << 3;                         // first cell of segment B
    // phase 1: brick ids of the nine rows (two per row where the row is split).  The row masks are not consulted:
    // an empty row's prefix words are equal, i.e. an empty run.
    uint32_t idA[9], idB[9];
    int rowbit[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const int yy = q.cy + (r % 3) - 1, zz = q.cz + (r / 3) - 1;
        const bool ok = xok && yy >= 0 && yy < g.ncy && zz >= 0 && zz < g.ncz;
        rowbit[r] = ((zz & 7) << 3) | (yy & 7);
        const int64_t toprow = ((int64_t)(zz >> 3) * g.nby + (yy >> 3)) * g.nbx;
        idA[r] = 0u;
        idB[r] = 0u;
        if (ok) idA[r] = g.top[toprow + bA].x;
        if (ok && split) idB[r] = g.top[toprow + bB].x;
    }
#ifdef S2M_EXP_ROWS_TIMELINE
    {
        uint32_t all = 0;
#pragma unroll
        for (int r = 0; r < 9; ++r) all |= idA[r] | idB[r];
        S2M_ROWS_STAMP(1, all);
    }
#endif
    // phase 2: the home row's prefix words (four consecutive words cover its three cells); the other rows' words are
    // fetched after the home row has produced a bound -- only for the rows and cells that survive it
    TabQuad qa4 = TabQuad{{0u, 0u, 0u, 0u}}, qb4 = TabQuad{{0u, 0u, 0u, 0u}};
    if (idA[4]) qa4 = *reinterpret_cast<const TabQuad *>(g.tab + (int64_t)(idA[4] - 1) * kBrickStride + (rowbit[4] << 3) + (x_lo & 7));
    if (idB[4]) qb4 = *reinterpret_cast<const TabQuad *>(g.tab + (int64_t)(idB[4] - 1) * kBrickStride + (rowbit[4] << 3));
    u64 t[kK], best[kK];
#pragma unroll
    for (int k = 0; k < kK; ++k) t[k] = kEmptyKey;
    uint32_t nr = 0;
    // the G lanes of a query share every run point by point (load_batch): all of them list the whole run and walk it
    // in steps of 8 G points.  (S2M_ROWS_SHARE=0, the round-2 form: a contiguous chunk of ceil(len / G) points, rounded up
    // to whole batches, per lane.)
    auto push_run = [&](uint32_t s, uint32_t e) {
        if (s >= e) return;
        if (kShare > 1) {
            runs[nr * 256 + threadIdx.x] = make_uint2(s, e); ++nr;
            return;
        }
        const uint32_t chunk = (((e - s) + 8u * G - 1u) / (8u * G)) * 8u;
        const uint32_t ms = s + (uint32_t)j * chunk, me = min(ms + chunk, e);
        if (ms < me) { runs[nr * 256 + threadIdx.x] = make_uint2(ms, me); ++nr; }
    };
    // the lane's batches, NB at a time: all 8 * NB point loads of a trip are in flight together (an exhausted cursor
    // loads the sentinel block -- distance +inf, nothing is inserted -- so the trip has no branches around its loads)
    auto walk_runs = [&]() {
        if (nr == 0) return;
        RunCursor c;
        c.k = 0;
        { const uint2 r0 = runs[threadIdx.x]; c.i = r0.x; c.e = r0.y; }
        auto advance = [&]() {
            c.i += 8u * kShare;
            if (c.i >= c.e) {
                ++c.k;
                c.i = 0u; c.e = 0u;
                if (c.k < nr) { const uint2 rk = runs[c.k * 256 + threadIdx.x]; c.i = rk.x; c.e = rk.y; }
            }
        };
        while (c.k < nr) {
            float4 p[NB][8];
            uint32_t left[NB];
            const uint32_t jj = kShare > 1 ? (uint32_t)j : 0u;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                left[b] = c.e > c.i + jj ? c.e - c.i - jj : 0u;
                load_batch<WIDE, kShare, kPred>(g, c.i, c.e, p[b], jj);
                advance();
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) consume_batch<kShare, kPred>(p[b], q.wx, q.wy, q.wz, t, left[b]);
        }
    };
    // phase 3a: the home row (r = 4), all G lanes of the group on it
    {
        uint32_t s, e;
        seg_run(qa4, x_lo, ax1, x_lo, x_hi, s, e);
        push_run(s, e);
        seg_run(qb4, bx0, x_hi, x_lo, x_hi, s, e);
        push_run(s, e);
    }
    walk_runs();
    merge_lists<G>(t, best);
    const bool have_tau = !is_empty(best[kK - 1]);
    const float tau = __uint_as_float((uint32_t)(best[kK - 1] >> 32));
    S2M_ROWS_STAMP(2, tau);
    // phase 3b: the other rows, trimmed to the cells that can hold something closer than tau, as one work list.
    // First the two prefix words that delimit each surviving piece (all requested together), then the list.
    nr = 0;
    uint32_t sA[9], eA[9], sB[9], eB[9];
    // squared per-axis gaps to the neighbouring cells, once: cell_bound2(dx, dy, dz) = ((gx2 + gy2) + gz2) * c^2 * 0.99999
    // with exactly the products and sums of that function (24 bounds from 9 squares instead of 24 x 3)
    const float gxm = fmaxf(q.frx - g.slop, 0.0f), gxp = fmaxf((1.0f - q.frx) - g.slop, 0.0f);
    const float gym = fmaxf(q.fry - g.slop, 0.0f), gyp = fmaxf((1.0f - q.fry) - g.slop, 0.0f);
    const float gzm = fmaxf(q.frz - g.slop, 0.0f), gzp = fmaxf((1.0f - q.frz) - g.slop, 0.0f);
    const float gx2[3] = {gxm * gxm, 0.0f, gxp * gxp}, gy2[3] = {gym * gym, 0.0f, gyp * gyp}, gz2[3] = {gzm * gzm, 0.0f, gzp * gzp};
    const float cc = g.c * g.c;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        sA[r] = eA[r] = sB[r] = eB[r] = 0u;
        if (r == 4) continue;
        const float gyz0 = gy2[r % 3];  // dy = r % 3 - 1, dz = r / 3 - 1
        // the cell bound grows with |dx|: the cells that survive are a contiguous range around dx = 0
        if (have_tau && ((gx2[1] + gyz0) + gz2[r / 3]) * cc * 0.99999f > tau) continue;
        int xa = q.cx, xb = q.cx;
        if (!(have_tau && ((gx2[0] + gyz0) + gz2[r / 3]) * cc * 0.99999f > tau)) xa = q.cx - 1;
        if (!(have_tau && ((gx2[2] + gyz0) + gz2[r / 3]) * cc * 0.99999f > tau)) xb = q.cx + 1;
        // (two dword loads per piece; one 4-byte-aligned dwordx4 covering both words measured slower: 3.0 vs 2.6 us
        // for this stage)
        const int la = max(xa, x_lo), ha = min(xb, ax1);  // piece inside segment A
        if (idA[r] && la <= ha) {
            const uint32_t *tb = g.tab + (int64_t)(idA[r] - 1) * kBrickStride + (rowbit[r] << 3);
#if S2M_ROWS_QUADS
            const TabQuad qd = *reinterpret_cast<const TabQuad *>(tb + (la & 7));
            sA[r] = qd.w[0]; eA[r] = sel4(qd, ha - la + 1);
#else
            sA[r] = tb[la & 7]; eA[r] = tb[(ha & 7) + 1];
#endif
        }
        const int lb = max(xa, bx0), hb = min(xb, x_hi);  // piece inside segment B
        if (idB[r] && lb <= hb) {
            const uint32_t *tb = g.tab + (int64_t)(idB[r] - 1) * kBrickStride + (rowbit[r] << 3);
#if S2M_ROWS_QUADS
            const TabQuad qd = *reinterpret_cast<const TabQuad *>(tb + (lb & 7));
            sB[r] = qd.w[0]; eB[r] = sel4(qd, hb - lb + 1);
#else
            sB[r] = tb[lb & 7]; eB[r] = tb[(hb & 7) + 1];
#endif
        }
    }
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        if (r == 4) continue;
        push_run(sA[r], eA[r]);
        push_run(sB[r], eB[r]);
    }
    S2M_ROWS_STAMP(3, nr);
    walk_runs();
    merge_lists<G>(t, best);
    bool found5 = !is_empty(best[kK - 1]);
    float d5 = __uint_as_float((uint32_t)(best[kK - 1] >> 32));
    bool done = found5 && d5 <= cube_bound2(g, q, 1);
#if S2M_ROWS_EXT
    // ---- phase 4 (round 3 experiment, off by default: measured slower, see DESIGN): the second / third shell, in place,
    // for the points whose radius is already known ------
    // A point that holds five neighbours but not provably the nearest five used to go to match_hard -- a second kernel,
    // its own dependent-load chain of ~6 us per point, and a launch as long as its busiest wave.  Measured on the first
    // pass (predicted pose off by 1 deg / 5 cm: far returns sit 0.5-3 m off their surface): these points come in WHOLE
    // WAVES -- 21 % of this kernel's waves hold any at C3 (23 % at C4), and those hold 23-25 of 32 -- and for 96 % of
    // them (75 % at C4) the 5th distance is within three cells.  So the waves that hold such points (wave-uniform
    // branch; the others leave as before) finish them here: the x-rows within R cells of the home row whose (y,z)
    // bound is inside the radius, minus what the first shell has already read, listed per lane (the G lanes of a point
    // split the rows), top entries of all listed pieces in one trip, their prefix words in a second, then the point
    // runs through the walk above.  Every cell within the radius has then been read: the list is exact and final.
    // A lane with more pieces than slots gives up and the point takes the old route.
    {
        constexpr int R = S2M_ROWS_EXT;  // rows |dy|, |dz| <= R
        constexpr int W = 2 * R + 1;
        const float lim = ((float)R - g.slop) * g.c * 0.9999f;  // a row R + 1 cells away is bounded below by (R - slop) cells
        const bool ext = !done && found5 && d5 <= lim * lim;
        if (__any(ext)) {
            nr = 0;
            bool over = false;
            if (ext) {
                const float fxq = (float)q.cx + q.frx;
                for (int rr = j; rr < W * W; rr += G) {
                    const int dy = rr % W - R, dz = rr / W - R;
                    const int yy = q.cy + dy, zz = q.cz + dz;
                    if (yy < 0 || yy >= g.ncy || zz < 0 || zz >= g.ncz) continue;
                    const float gy = dy > 0 ? (float)dy - q.fry : (dy < 0 ? q.fry - (float)(dy + 1) : 0.0f);
                    const float gz = dz > 0 ? (float)dz - q.frz : (dz < 0 ? q.frz - (float)(dz + 1) : 0.0f);
                    const float ay = fmaxf(gy - g.slop, 0.0f), az = fmaxf(gz - g.slop, 0.0f);
                    const float b2 = (ay * ay + az * az) * (g.c * g.c) * 0.99999f;
                    if (b2 > d5) continue;
                    const float reach = sqrtf(fmaxf(d5 - b2, 0.0f)) * g.inv_c * 1.00001f + g.slop;
                    const int xa = max((int)floorf(fxq - reach), 0), xb = min((int)floorf(fxq + reach), g.ncx - 1);
                    // the nine inner rows: cells x_lo .. x_hi were read by the first shell, or skipped there because their
                    // box bound exceeds a 5th distance that was already no smaller than today's radius
                    const bool inner = xok && dy >= -1 && dy <= 1 && dz >= -1 && dz <= 1;
                    const int plo[2] = {xa, inner ? x_hi + 1 : 1}, phi[2] = {inner ? x_lo - 1 : xb, inner ? xb : 0};
                    const int rowb = ((zz & 7) << 3) | (yy & 7);
                    const int64_t toprow = ((int64_t)(zz >> 3) * g.nby + (yy >> 3)) * g.nbx;
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) {
                        const int lo = plo[pc], hi = phi[pc];
                        if (lo > hi) continue;
                        for (int b = lo >> 3; b <= (hi >> 3); ++b) {  // a piece spans at most 2 R + 1 cells: two bricks
                            const int sl = max(lo, b << 3) & 7, sh = min(hi, (b << 3) + 7) & 7;
                            if (nr == (uint32_t)kRunSlots) { over = true; break; }
                            runs[nr * 256 + threadIdx.x] = make_uint2((uint32_t)(toprow + b), (uint32_t)((rowb << 8) | (sl << 4) | sh));
                            ++nr;
                        }
                    }
                    if (over) break;
                }
            }
            // the G lanes of a point stand or fall together
            bool over_g = over;
#pragma unroll
            for (int off = G / 2; off > 0; off >>= 1) over_g = over_g || (__shfl_xor((int)over_g, off, G) != 0);
            if (over_g) nr = 0;
            // trip 1: the top entries of all listed pieces; trip 2: their two prefix words
            uint32_t ids[kRunSlots];
#pragma unroll
            for (int k = 0; k < kRunSlots; ++k) {
                ids[k] = 0u;
                if ((uint32_t)k < nr) {
                    const uint2 dsc = runs[k * 256 + threadIdx.x];
                    const uint4 te = g.top[dsc.x];
                    const uint32_t rowb = dsc.y >> 8;
                    const uint32_t mword = (rowb & 32u) ? te.w : te.z;
                    if (te.x != 0u && ((mword >> (rowb & 31u)) & 1u)) ids[k] = te.x;
                }
            }
            uint32_t rs[kRunSlots], re[kRunSlots];
#pragma unroll
            for (int k = 0; k < kRunSlots; ++k) {
                rs[k] = 0u; re[k] = 0u;
                if ((uint32_t)k < nr && ids[k] != 0u) {
                    const uint32_t w = runs[k * 256 + threadIdx.x].y;
                    const uint32_t *tb = g.tab + (int64_t)(ids[k] - 1) * kBrickStride + ((w >> 8) << 3);
                    rs[k] = tb[(w >> 4) & 7u];
                    re[k] = tb[(w & 7u) + 1u];
                }
            }
            uint32_t nr2 = 0;
#pragma unroll
            for (int k = 0; k < kRunSlots; ++k)
                if ((uint32_t)k < nr && rs[k] < re[k]) { runs[nr2 * 256 + threadIdx.x] = make_uint2(rs[k], re[k]); ++nr2; }
            nr = nr2;
            walk_runs();
            merge_lists<G>(t, best);
            found5 = !is_empty(best[kK - 1]);
            d5 = __uint_as_float((uint32_t)(best[kK - 1] >> 32));
            done = done || (ext && !over_g);
        }
    }
#endif
    // unresolved: match_hard's list, in two parts -- the points without a radius (fewer than five neighbours in the
    // shell: two rounds there, ~11 us against ~5.5 us) are handed out first so that they do not form the launch's tail
    {
        uint32_t found = 0;
#pragma unroll
        for (int k = 0; k < kK; ++k) found += is_empty(best[k]) ? 0u : 1u;
        const HardRec rec = {q.wx, q.wy, q.wz, (uint32_t)qi, d5, found, a.slot, 0u};
        append_rec(a.hard_rec, a.hard_count, j == 0 && !done && !found5, rec);
        append_rec(a.hard_rec + a.hard_off1, a.hard_count + 1, j == 0 && !done && found5, rec);
    }
    if (j == 0) {
        store_result(best, qi, a.nn_idx, a.nn_d2);
        if (a.dbg) {
            a.dbg[4 * (int64_t)qi + 0] = (uint32_t)(wall_clock64() - t0);
            a.dbg[4 * (int64_t)qi + 1] = done ? 1u : 0u;
            a.dbg[4 * (int64_t)qi + 2] = 0;
            a.dbg[4 * (int64_t)qi + 3] = 1;
#ifdef S2M_EXP_ROWS_TIMELINE
            // stage stamps relative to the kernel's first instruction of this wave, 16 bits each (100 MHz ticks)
            a.dbg[4 * (int64_t)qi + 2] = (uint32_t)((st_[0] - t0) & 0xffff) | ((uint32_t)((st_[1] - t0) & 0xffff) << 16);
            a.dbg[4 * (int64_t)qi + 3] = (uint32_t)((st_[2] - t0) & 0xffff) | ((uint32_t)((st_[3] - t0) & 0xffff) << 16);
#endif
        }
    }
}

template <int G, bool WIDE, int NB>
__global__ __launch_bounds__(256) void match_rows(MatchArgs a)
{
    __shared__ uint2 runs[kRunSlots * 256];  // [slot][thread]: conflict-free whatever the per-lane fill
    match_rows_body<G, WIDE, NB>(a, runs);
}

// the search arguments of scan blockIdx.y of a batched launch (everything uniform: scalar loads from the table)
__device__ __forceinline__ MatchArgs batch_match_args(const BatchArgs &b, uint32_t slot)
{
    const ScanDesc &d = b.d[slot];
    MatchArgs m;
    m.grid = b.grid; m.pose = d.pose; m.gates = b.gates;
    m.sx = d.sx; m.sy = d.sy; m.sz = d.sz; m.n = d.n;
    m.nn_idx = d.nn_idx; m.nn_d2 = d.nn_d2;
    m.hard_rec = b.hard_rec; m.hard_off1 = b.hard_off1; m.slot = slot;
    m.hard_count = b.hard_count; m.qheads = b.qheads; m.dbg = nullptr;
    return m;
}
// K scans, one grid: blockIdx.y = scan.  The scans that do not search in this pass leave at once.
template <int G, bool WIDE, int NB>
__global__ __launch_bounds__(256) void match_rows_batch(BatchArgs b)
{
    __shared__ uint2 runs[kRunSlots * 256];
    const ScanDesc &d = b.d[blockIdx.y];
    if (!d.active || !d.rematch) return;
    const MatchArgs a = batch_match_args(b, blockIdx.y);
    match_rows_body<G, WIDE, NB>(a, runs);
}

// one atomic per wave for a list append: same-address atomics serialise in L2 (~90 per microsecond), ten thousand
// per-lane atomics would cost > 100 us
__device__ __forceinline__ void append_rec(HardRec *__restrict__ list, uint32_t *__restrict__ counter, bool want,
                                           const HardRec &rec)
{
    const unsigned long long mask = __ballot(want);
    if (mask == 0ull) return;  // wave-uniform
    const int lane64 = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if (lane64 == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, leader, 64);
    if (want) {
        uint4 *dst = reinterpret_cast<uint4 *>(list + (base + (uint32_t)__popcll(mask & ((1ull << lane64) - 1ull))));
        dst[0] = make_uint4(__float_as_uint(rec.wx), __float_as_uint(rec.wy), __float_as_uint(rec.wz), rec.qi);
        dst[1] = make_uint4(__float_as_uint(rec.d5), rec.found, rec.slot, 0u);
    }
}

// ---- the rest: one wave per hard scan point, occupied rows only ------------------------------------
// k-th (0-based) set bit of a 64-bit mask; k < popcount(m)
__device__ __forceinline__ int kth_set_bit(uint64_t m, int k)
{
    int pos = 0;
    uint32_t lo = (uint32_t)m;
    int c = __popc(lo);
    if (k >= c) { k -= c; pos = 32; lo = (uint32_t)(m >> 32); }
#pragma unroll
    for (int w = 16; w >= 1; w >>= 1) {
        const uint32_t part = lo & ((1u << w) - 1u);
        c = __popc(part);
        if (k >= c) { k -= c; pos += w; lo >>= w; } else { lo = part; }
    }
    return pos;
}

// A hard point is one whose 5th neighbour is not provably inside the 3x3x3 cells.  Growing a cube
// cell by cell costs O(r^3) lookups although LiDAR maps are surfaces; instead the wave reads the top
// entries of the surrounding bricks once (lane b = brick b) and enumerates only their OCCUPIED
// (y,z) rows from the 64-bit row masks.  Each (brick,row) pair gets a lower bound from its (y,z)
// offset; a pair is scanned -- restricted to the x-cells the current radius can reach -- only if that
// bound is within the radius.  With five neighbours already known from the first shell their 5th
// distance is the radius and a single round finishes the point; otherwise the radius grows band by
// band over the same pair list until five are found, and stops at the d2 <= 5 gate (:853).
// While the radius is within three cells (98 % of the hard points of the benchmark scan) the enumeration
// is skipped altogether: only the 7x7 x-rows around the home row can qualify, so lane l < 49 addresses
// row l directly (measured: first-pass launch 68 -> 60 us; occupancy 3 vs 4 waves/SIMD and point batches
// of 4 vs 8 make no difference -- the kernel is VALU-issue bound at ~58 %, TA ~27 % busy).
constexpr int kPairSlots = 6;  // (brick,row) pairs a lane can hold per chunk of 64 bricks

#ifndef S2M_HARD_CHUNKS
#define S2M_HARD_CHUNKS 4  // chunks of 64 listed cells whose table words are fetched per round trip
#endif
#ifndef S2M_HARD_BAND_EMPTY
#define S2M_HARD_BAND_EMPTY 2.8f  // first band (cells) of a far point whose first shell held nothing
#endif
#ifndef S2M_ROWS_EXT
#define S2M_ROWS_EXT 0  // experiment (round 3, DESIGN "tried and rejected"): 2 or 3 = the first-shell kernel finishes the points whose
                        // 5th distance is within that many cells itself; 0 = every unresolved point goes to match_hard
#endif
#ifndef S2M_HARD_PIECES
#define S2M_HARD_PIECES 1  // idle lanes of the far-point kernel take the later 8-point pieces of the listed cells (0: one run per lane)
#endif
#ifndef S2M_HARD_OCC
#define S2M_HARD_OCC 4  // waves per SIMD match_hard is compiled for (116 VGPRs at 4; 5 needs spills) = resident waves / 1024
#endif
// FAR: the instantiation behind s2m_complete_neighbors.  Its radius is not the gate but whatever it takes to find
// five points, so the brick neighbourhood is intersected with the grid per point (a 100 m radius would otherwise
// enumerate millions of bricks that do not exist); the per-iteration instantiation keeps the unclamped cube, whose
// lane -> brick mapping is computed once per wave.
constexpr int kMaxCells = 1024;  // an append adds at most 64 rows x 8 cells
// `out(slot, idx, d2)` names the neighbour arrays of the scan a record belongs to: the launch's own arrays for one scan,
// a look-up in the table for a batched launch
template <bool WIDE, bool FAR, class Out>
__device__ __forceinline__ void match_hard_body(const MatchArgs &a, uint2 *__restrict__ cells, Out &&out)
{
    constexpr int G = 64;
    const Grid &g = a.grid;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t c0 = a.hard_count[0], count = c0 + a.hard_count[1];  // [no radius yet | radius known]
    // brick rings needed so that the neighbourhood covers the gate radius from anywhere in the home brick
    const float gate_r = sqrtf(a.gates.knn_d2_gate);
    const int NB = max(1, (int)fminf(ceilf(gate_r * g.inv_c * 0.125f + 1e-3f), 1048576.0f));
    const int bside = 2 * NB + 1, nbricks = FAR ? 0 : bside * bside * bside;  // FAR clips the cube to the grid per point
    // the brick this lane inspects in the first chunk of 64 bricks, relative to the home brick (the common
    // case NB = 1 has 27 bricks, one chunk): computed once, not per point
    const int ob0 = lane < nbricks ? lane : 0;
    const int odx0 = (ob0 % bside) - NB, ody0 = ((ob0 / bside) % bside) - NB, odz0 = (ob0 / (bside * bside)) - NB;
    // the x-row of the 7x7 around the home row this lane takes when the radius is within three cells
    const int ndy = (lane % 7) - 3, ndz = (lane / 7) - 3;
    const float near_r = (3.0f - g.slop) * g.c * 0.9999f;
    const float near_r2 = near_r * near_r;
    // Dynamic hand-out: the point of the wave's own index first (no atomic: an empty or short list costs nothing),
    // then tickets from the wave's shard head; shard s, ticket t is point nwaves + t * kQueueShards + s.  A static
    // stride left the launch waiting for the waves that happened to draw two expensive points (measured at C3: 9,981
    // points of 8 us mean on 4,096 resident waves took 36 us).
    const uint32_t shard = (uint32_t)wave % kQueueShards;
    uint32_t h = (uint32_t)wave;
    while (h < count) {
        const long long t0 = a.dbg ? wall_clock64() : 0;
        // the point's record: query, index, and the radius when the first shell found five (then one round is exact)
        const uint4 *rp = reinterpret_cast<const uint4 *>(h < c0 ? a.hard_rec + h : a.hard_rec + (a.hard_off1 + (h - c0)));
        const uint4 r0 = rp[0], r1 = rp[1];
        const int qi = (int)r0.w;
        const Query q = query_at(g, __uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r0.z));
        const float fxq = (float)q.cx + q.frx;  // query x in cell units
        bool have_tau = r1.y == (uint32_t)kK;
        float tau = have_tau ? __uint_as_float(r1.x) : 0.0f;  // squared
        u64 t[kK], best[kK];
#pragma unroll
        for (int k = 0; k < kK; ++k) { t[k] = kEmptyKey; best[k] = kEmptyKey; }
        // band radius while no radius is known: the first shell covered (1 + fmin) c; 1.7 c measured best at C3
        // (1.3 / 1.5 / 1.7 / 2.0 / 2.5 c -> 68 / 67 / 61 / 64 / 68 us for the first-pass launch)
        float band = S2M_HARD_BAND * g.c;
        // A point whose first shell was EMPTY (the predicted pose put it more than a cell off every surface) starts wider:
        // 2.8 cells is the widest band that still takes the direct 7x7-row path below (radius within 3 cells), and its
        // successor is the gate.  Measured with the gate clamps in place, search kernels per rematch pass, 1.7 -> 2.8
        // cells for these points: C3 38.0 -> 37.7 us, C4 77.4 -> 65.6, R1 24.1 -> 23.8, C2 27.2 -> 27.3; 3.1 cells (the
        // general path) 41.3 / 74.2 / 24.0 / 29.9.  Re-checked at the end of round 3 (search kernels per rematch pass): 2.2 cells
        // C3 32.6 / C4 59.7, 2.5 cells 32.5 / 53.9, against 31.7 / 53.0 with 2.8.
        if (r1.y == 0u) band = S2M_HARD_BAND_EMPTY * g.c;
        band = fminf(band, sqrtf(a.gates.knn_d2_gate * 1.0001f));  // no first band beyond the gate either (coarse grids)
        uint32_t rounds = 0;
        const int hbx = q.cx >> 3, hby = q.cy >> 3, hbz = q.cz >> 3;
        int nc = 0;  // cells waiting in the wave's list (wave-uniform)
        // Every qualifying row piece (cells xa .. xa+ncell-1 of one x-row of one brick) is expanded into the
        // wave's cell list {brick id, table word}: positions come from a DPP prefix sum over the lanes, so
        // the list needs no search afterwards.  flush_cells hands cell j to lane j % 64 -- the point loads of
        // all qualifying rows are in flight together, however unevenly the rows are filled.
        auto append_cells = [&](uint32_t id, int rowbit, int xa, int ncell) {
            const int incl = wave_incl_scan(ncell);
            const int total = __builtin_amdgcn_readlane(incl, 63);
            if (total == 0) return;  // wave-uniform
            const int at = nc + incl - ncell;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c < ncell) cells[at + c] = make_uint2(id, (uint32_t)((rowbit << 3) + ((xa + c) & 7)));
            nc += total;
        };
        auto flush_cells = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // list stores before the loads below
            // Pass 1: the two prefix words of EVERY listed cell, four chunks of 64 cells per round trip; only the cells
            // that hold points stay, compacted in place as point runs {start, end} (a compacted entry never lands beyond
            // the entries already read).  Most listed cells of a wide band are empty -- the point is far from every
            // surface, that is why it is here -- and each chunk of 64 cells used to cost a dependent table-then-points
            // round trip whether or not it held anything.
            int no = 0;
            for (int jb = 0; jb < nc; jb += 64 * S2M_HARD_CHUNKS) {  // wave-uniform trip count
                uint32_t rs[S2M_HARD_CHUNKS], re[S2M_HARD_CHUNKS];
#pragma unroll
                for (int u = 0; u < S2M_HARD_CHUNKS; ++u) {
                    const int j = jb + u * 64 + lane;
                    rs[u] = 0u; re[u] = 0u;
                    if (j < nc) {
                        const uint2 ce = cells[j];
                        const uint32_t *tb = g.tab + (int64_t)(ce.x - 1) * kBrickStride + ce.y;
                        rs[u] = tb[0]; re[u] = tb[1];
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // the reads above before the in-place writes
#pragma unroll
                for (int u = 0; u < S2M_HARD_CHUNKS; ++u) {
                    const bool holds = rs[u] < re[u];
                    const unsigned long long m = __ballot(holds);
                    if (holds) cells[no + __popcll(m & ((1ull << lane) - 1ull))] = make_uint2(rs[u], re[u]);
                    no += __popcll(m);
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            }
            // Pass 2: the runs.  A run is one cell -- 12 points on average at the tuned density, i.e. usually TWO batches
            // of eight -- and a lane that walks its run alone pays one dependent load trip per batch while most lanes of
            // the wave hold no run at all (a far point lists 10-30 non-empty cells).  With few runs the idle lanes take
            // the later pieces of the same runs instead: lane l serves piece l / no of run l % no (the last piece takes
            // whatever is left), so a cell of up to 32 points (16 with more than 16 runs) is read in ONE trip.  Same
            // candidates, same top-5 (the merge of the private lists does not depend on who scanned what).
            if (S2M_HARD_PIECES && no > 0 && no <= 32) {  // wave-uniform
                const int P = no <= 16 ? 4 : 2;
                const int piece = lane / no;
                if (piece < P) {
                    const uint2 run = cells[lane - piece * no];
                    const uint32_t s0 = run.x + 8u * (uint32_t)piece;
                    const uint32_t e0 = (piece == P - 1) ? run.y : min(s0 + 8u, run.y);
                    if (s0 < e0) scan_points<kHardBatch, WIDE>(g, s0, e0, q.wx, q.wy, q.wz, t);
                }
            } else
            for (int jb = 0; jb < no; jb += 64) {  // wave-uniform trip count
                const int j = jb + lane;
                if (j < no) {
                    const uint2 run = cells[j];
                    scan_points<kHardBatch, WIDE>(g, run.x, run.y, q.wx, q.wy, q.wz, t);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // loads above before the next stores
            nc = 0;
        };
        for (;;) {
            const float r2 = have_tau ? tau : band * band;  // scan every pair whose bound is within r2
            // x reach (cells) as a function of the pair's bound is computed per pair below
            if (r2 <= near_r2) {
                // Radius inside three cells: only the 7x7 x-rows around the home row can qualify (a row four
                // cells away is bounded below by (3 - slop) cells) and each reaches at most seven cells, i.e.
                // two bricks.  One row per lane, addressed directly: no brick enumeration, no owner search.
                uint32_t nid[2] = {0u, 0u};
                int nxa[2] = {0, 0}, ncl[2] = {0, 0}, nrow = 0;
                const int yy = q.cy + ndy, zz = q.cz + ndz;
                if (lane < 49 && yy >= 0 && yy < g.ncy && zz >= 0 && zz < g.ncz) {
                    const float gy = ndy > 0 ? (float)ndy - q.fry : (ndy < 0 ? q.fry - (float)(ndy + 1) : 0.0f);
                    const float gz = ndz > 0 ? (float)ndz - q.frz : (ndz < 0 ? q.frz - (float)(ndz + 1) : 0.0f);
                    const float ay = fmaxf(gy - g.slop, 0.0f), az = fmaxf(gz - g.slop, 0.0f);
                    const float b2 = (ay * ay + az * az) * (g.c * g.c) * 0.99999f;
                    if (b2 <= r2) {
                        const float reach = sqrtf(fmaxf(r2 - b2, 0.0f)) * g.inv_c * 1.00001f + g.slop;
                        const int xa = max((int)floorf(fxq - reach), 0), xb = min((int)floorf(fxq + reach), g.ncx - 1);
                        nrow = ((zz & 7) << 3) | (yy & 7);
                        const int64_t toprow = ((int64_t)(zz >> 3) * g.nby + (yy >> 3)) * g.nbx;
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const int bx = (xa >> 3) + k;
                            if (xa > xb || bx > (xb >> 3)) continue;
                            const uint4 te = g.top[toprow + bx];
                            const uint32_t mword = (nrow & 32) ? te.w : te.z;
                            if (te.x == 0 || ((mword >> (nrow & 31)) & 1u) == 0) continue;
                            nid[k] = te.x;
                            nxa[k] = max(xa, bx << 3);
                            ncl[k] = min(xb, (bx << 3) + 7) - nxa[k] + 1;
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) append_cells(nid[k], nrow, nxa[k], ncl[k]);  // <= 2 x 49 x 7 cells
            } else {
            // FAR: the neighbourhood clipped to the grid (empty when the point lies further outside than the radius)
            const int flx = max(hbx - NB, 0), fly = max(hby - NB, 0), flz = max(hbz - NB, 0);
            const int fsx = FAR ? max(min(hbx + NB, g.nbx - 1) - flx + 1, 0) : 0;
            const int fsy = FAR ? max(min(hby + NB, g.nby - 1) - fly + 1, 0) : 0;
            const int fsz = FAR ? max(min(hbz + NB, g.nbz - 1) - flz + 1, 0) : 0;
            const int nbr = FAR ? (int)min((long long)fsx * fsy * fsz, 0x7fffffc0ll) : nbricks;
            for (int bbase = 0; bbase < nbr; bbase += 64) {
                // 1. top entries of up to 64 bricks, one per lane
                const int b = bbase + lane;
                uint32_t my_id = 0;
                uint64_t my_mask = 0;
                int bx = 0, by = 0, bz = 0;
                if (b < nbr) {
                    if (FAR) {
                        bx = flx + b % fsx; by = fly + (b / fsx) % fsy; bz = flz + b / (fsx * fsy);
                    } else if (bbase == 0) {
                        bx = hbx + odx0; by = hby + ody0; bz = hbz + odz0;
                    } else {
                        bx = hbx + (b % bside) - NB;
                        by = hby + ((b / bside) % bside) - NB;
                        bz = hbz + (b / (bside * bside)) - NB;
                    }
                    if (bx >= 0 && bx < g.nbx && by >= 0 && by < g.nby && bz >= 0 && bz < g.nbz) {
                        const uint4 te = g.top[((int64_t)bz * g.nby + by) * g.nbx + bx];
                        my_id = te.x;
                        my_mask = te.x ? ((uint64_t)te.w << 32 | te.z) : 0ull;
                    }
                }
                // 2. exclusive prefix of the occupied-row counts over the lanes
                const int cnt = __popcll(my_mask);
                const int incl = wave_incl_scan(cnt);
                const int excl = incl - cnt;
                const int total = __shfl(incl, 63, 64);
                // 3. (brick,row) pairs, round-robin over the lanes
                for (int pbase = 0; pbase < total; pbase += 64 * kPairSlots) {
#pragma unroll
                    for (int slot = 0; slot < kPairSlots; ++slot) {
                        if (pbase + slot * 64 >= total) break;  // wave-uniform
                        const int p = pbase + slot * 64 + lane;
                        // owner lane o: the last lane whose exclusive prefix is <= p (uniform loop of shuffles)
                        int o = 0;
#pragma unroll
                        for (int step = 32; step >= 1; step >>= 1) {
                            const int cand = o + step;
                            const int pc = __shfl(excl, min(cand, 63), 64);
                            if (cand < 64 && pc <= p) o = cand;
                        }
                        const uint32_t mlo = __shfl((uint32_t)my_mask, o, 64), mhi = __shfl((uint32_t)(my_mask >> 32), o, 64);
                        const uint32_t oid = __shfl(my_id, o, 64);
                        const int obx = __shfl(bx, o, 64), oby = __shfl(by, o, 64), obz = __shfl(bz, o, 64);
                        const int oex = __shfl(excl, o, 64);
                        // a lane's pair qualifies when its (y,z) bound is within the radius; its cells are then
                        // spread over the whole wave (a row can hold ~100 points: one lane walking it alone
                        // was measured to be the whole cost of this kernel)
                        int rowbit = 0, xa = 0, ncell = 0;
                        if (p < total) {
                            const uint64_t om = ((uint64_t)mhi << 32) | mlo;
                            rowbit = kth_set_bit(om, p - oex);
                            const int yy = (oby << 3) + (rowbit & 7), zz = (obz << 3) + (rowbit >> 3);
                            // lower bound of the (y,z) distance from the query to this row, in cells
                            const int dy = yy - q.cy, dz = zz - q.cz;
                            const float gy = dy > 0 ? (float)dy - q.fry : (dy < 0 ? q.fry - (float)(dy + 1) : 0.0f);
                            const float gz = dz > 0 ? (float)dz - q.frz : (dz < 0 ? q.frz - (float)(dz + 1) : 0.0f);
                            const float ay = fmaxf(gy - g.slop, 0.0f), az = fmaxf(gz - g.slop, 0.0f);
                            const float b2 = (ay * ay + az * az) * (g.c * g.c) * 0.99999f;  // metres^2
                            if (b2 <= r2) {
                                // x cells the radius can reach in this row: |x - qx| <= sqrt(r2 - b2)
                                const float reach = sqrtf(fmaxf(r2 - b2, 0.0f)) * g.inv_c * 1.00001f + g.slop;
                                xa = max((int)floorf(fxq - reach), obx << 3);
                                const int xb = min((int)floorf(fxq + reach), (obx << 3) + 7);
                                ncell = max(xb - xa + 1, 0);
                            }
                        }
                        // the qualifying pairs' cells go to the wave's list
                        if (nc + 512 > kMaxCells) flush_cells();
                        append_cells(oid, rowbit, xa, min(ncell, 8));
                    }
                }
            }
            }
            flush_cells();
            ++rounds;
            merge_lists<G>(t, best);
            const bool found5 = !is_empty(best[kK - 1]);
            const float d5 = __uint_as_float((uint32_t)(best[kK - 1] >> 32));
            if (have_tau) break;  // every point within tau was visited: exact
            // band mode: rows with bound <= band^2 were scanned over their whole reach of this band only,
            // so restart the private lists when the radius changes (rows are rescanned with the new reach)
            if (found5 && d5 <= band * band) break;          // five found inside the fully scanned band
            if (band * band > a.gates.knn_d2_gate) break;    // beyond the gate: result is "not five within it"
            // Nothing beyond the gate matters (a 5th neighbour past it is rejected, :853): neither the exact round nor a
            // grown band goes further than just past the gate radius.  (At C4, where the predicted pose displaces far
            // returns by metres, the band used to double from 3.4 to 6.8 cells -- 3.4 m against a 2.24 m gate -- for every
            // point whose five neighbours lie 1.7-2.2 m away: 20 us per point, 141 us for the first-pass launch.)
            const float gate2_up = a.gates.knn_d2_gate * 1.0001f;
            if (found5) { have_tau = true; tau = fminf(d5, gate2_up); }  // radius known now: one exact round
            else band = fminf(band * 2.0f, sqrtf(gate2_up));
#pragma unroll
            for (int k = 0; k < kK; ++k) t[k] = kEmptyKey;
        }
        if (lane == 0) {
            int32_t *o_idx;
            float *o_d2;
            out(r1.z, o_idx, o_d2);
            store_result(best, qi, o_idx, o_d2);
            if (a.dbg) {
                a.dbg[4 * (int64_t)qi + 0] += (uint32_t)(wall_clock64() - t0);
                a.dbg[4 * (int64_t)qi + 2] = (uint32_t)t0;  // absolute start tick (100 MHz) of the hard part
                a.dbg[4 * (int64_t)qi + 1] = 2u + rounds;
                a.dbg[4 * (int64_t)qi + 3] = (rounds + 1) | ((uint32_t)(wall_clock64() - t0) << 8);
            }
        }
        if (a.qheads) {
            uint32_t ticket = 0;
            if (lane == 0) ticket = atomicAdd(a.qheads + shard * kQueueStride, 1u);
            ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket);
            h = (uint32_t)nwaves + ticket * kQueueShards + shard;
        } else {
            h += (uint32_t)nwaves;
        }
    }
}

// ---- the same search with 32 lanes per point: two points per wave (round 3) ---------------------------------------
// A far point is ~6.5 us of DEPENDENT load round trips whatever the lane count.  With 32 lanes per point every resident
// wave carries two points that advance independently (each half of the wave runs its own sequence of points and
// rounds; all primitives below are half-local: DPP scans and minima that do not cross lane 31|32, ballots split in
// two, width-32 shuffles), so 8,192 points are in flight at the same register and LDS budget.  The halves share one
// instruction stream: a half idles while the other fetches its next record, and their rounds issue one after the
// other, so a single point takes longer -- this form is for the batched launches, where the list is long and only
// throughput counts (hard_half_waves below has the measurements).
__device__ __forceinline__ int half_incl_scan(int x)
{
    int v = x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3: stays inside a half
    return v;
}
// the value lane 31 (half 0) / lane 63 (half 1) holds
__device__ __forceinline__ int half_last(int v, int half)
{
    const int a = __builtin_amdgcn_readlane(v, 31), b = __builtin_amdgcn_readlane(v, 63);
    return half ? b : a;
}
__device__ __forceinline__ int half_first(int v, int half)
{
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 32);
    return half ? b : a;
}
__device__ __forceinline__ uint32_t half_ballot(bool p, int half)
{
    const unsigned long long m = __ballot(p);
    return half ? (uint32_t)(m >> 32) : (uint32_t)m;
}
__device__ __forceinline__ uint32_t half_min_u32(uint32_t v, int half)
{
#pragma unroll
    for (int k = 0; k < 5; ++k) v = dpp_min_step(v, k);  // row_shr 1, 2, 4, 8, row_bcast:15: lane 31 / 63 hold their half's minimum
    return (uint32_t)half_last((int)v, half);
}
__device__ __forceinline__ u64 half_min_u64(u64 v, int half)
{
    const uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
    const uint32_t mh = half_min_u32(hi, half);
    const uint32_t ml = half_min_u32((hi == mh) ? lo : 0xffffffffu, half);
    return ((u64)mh << 32) | (u64)ml;
}
// sorted top-5 of the 32 private lists of a half (as merge_lists)
__device__ __forceinline__ void merge_lists_half(const u64 (&priv)[kK], u64 (&best)[kK], int half)
{
    u64 t[kK];
#pragma unroll
    for (int k = 0; k < kK; ++k) t[k] = priv[k];
#pragma unroll
    for (int k = 0; k < kK; ++k) {
        const u64 m = half_min_u64(t[0], half);
        best[k] = m;
        if (t[0] == m && !is_empty(m)) {
#pragma unroll
            for (int s = 0; s < kK - 1; ++s) t[s] = t[s + 1];
            t[kK - 1] = kEmptyKey;
        }
    }
}

constexpr int kHalfCells = 512;  // cell list of one half; an append adds at most 32 rows x 8 cells
template <bool WIDE, class Out>
__device__ __forceinline__ void match_hard32_body(const MatchArgs &a, uint2 *__restrict__ cells, Out &&out)
{
    constexpr int GL = 32;
    const Grid &g = a.grid;
    const int lane = threadIdx.x & 31;         // lane inside the half
    const int half = (threadIdx.x >> 5) & 1;
    const int grp = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const int ngrp = (gridDim.x * blockDim.x) >> 5;
    const uint32_t c0 = a.hard_count[0], count = c0 + a.hard_count[1];  // [no radius yet | radius known]
    const float gate_r = sqrtf(a.gates.knn_d2_gate);
    const int NB = max(1, (int)fminf(ceilf(gate_r * g.inv_c * 0.125f + 1e-3f), 1048576.0f));
    const int bside = 2 * NB + 1, nbricks = bside * bside * bside;
    const int ob0 = lane < nbricks ? lane : 0;
    const int odx0 = (ob0 % bside) - NB, ody0 = ((ob0 / bside) % bside) - NB, odz0 = (ob0 / (bside * bside)) - NB;
    const float near_r = (3.0f - g.slop) * g.c * 0.9999f;
    const float near_r2 = near_r * near_r;
    const float gate2_up = a.gates.knn_d2_gate * 1.0001f;
    const uint32_t shard = (uint32_t)grp % kQueueShards;
    uint32_t h = (uint32_t)grp;
    bool fresh = true;
    // state of the half's current point
    uint32_t qi = 0, slot = 0, found = 0;
    Query q = {};
    float fxq = 0.0f, tau = 0.0f, band = 0.0f;
    bool have_tau = false;
    int hbx = 0, hby = 0, hbz = 0;
    u64 t[kK], best[kK];
#pragma unroll
    for (int k = 0; k < kK; ++k) { t[k] = kEmptyKey; best[k] = kEmptyKey; }
    while (h < count) {  // divergent between the halves: each runs its own sequence of points, one ROUND per trip
        if (fresh) {
            const uint4 *rp = reinterpret_cast<const uint4 *>(h < c0 ? a.hard_rec + h : a.hard_rec + (a.hard_off1 + (h - c0)));
            const uint4 r0 = rp[0], r1 = rp[1];
            qi = r0.w; found = r1.y; slot = r1.z;
            q = query_at(g, __uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r0.z));
            fxq = (float)q.cx + q.frx;
            have_tau = found == (uint32_t)kK;
            tau = have_tau ? __uint_as_float(r1.x) : 0.0f;
            band = (found == 0u ? S2M_HARD_BAND_EMPTY : S2M_HARD_BAND) * g.c;   // as in match_hard_body
            band = fminf(band, sqrtf(gate2_up));
            hbx = q.cx >> 3; hby = q.cy >> 3; hbz = q.cz >> 3;
#pragma unroll
            for (int k = 0; k < kK; ++k) { t[k] = kEmptyKey; best[k] = kEmptyKey; }
            fresh = false;
        }
        int nc = 0;  // cells waiting in the half's list (uniform inside the half)
        auto append_cells = [&](uint32_t id, int rowbit, int xa, int ncell) {
            const int incl = half_incl_scan(ncell);
            const int total = half_last(incl, half);
            if (total == 0) return;
            const int at = nc + incl - ncell;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c < ncell) cells[at + c] = make_uint2(id, (uint32_t)((rowbit << 3) + ((xa + c) & 7)));
            nc += total;
        };
        auto flush_cells = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            int no = 0;
            for (int jb = 0; jb < nc; jb += GL * S2M_HARD_CHUNKS) {
                uint32_t rs[S2M_HARD_CHUNKS], re[S2M_HARD_CHUNKS];
#pragma unroll
                for (int u = 0; u < S2M_HARD_CHUNKS; ++u) {
                    const int j = jb + u * GL + lane;
                    rs[u] = 0u; re[u] = 0u;
                    if (j < nc) {
                        const uint2 ce = cells[j];
                        const uint32_t *tb = g.tab + (int64_t)(ce.x - 1) * kBrickStride + ce.y;
                        rs[u] = tb[0]; re[u] = tb[1];
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#pragma unroll
                for (int u = 0; u < S2M_HARD_CHUNKS; ++u) {
                    const bool holds = rs[u] < re[u];
                    const uint32_t m = half_ballot(holds, half);
                    if (holds) cells[no + __popc(m & ((1u << lane) - 1u))] = make_uint2(rs[u], re[u]);
                    no += __popc(m);
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            }
            if (S2M_HARD_PIECES && no > 0 && no <= 16) {  // idle lanes take the later pieces of the runs (see match_hard_body)
                const int P = no <= 8 ? 4 : 2;
                const int piece = lane / no;
                if (piece < P) {
                    const uint2 run = cells[lane - piece * no];
                    const uint32_t s0 = run.x + 8u * (uint32_t)piece;
                    const uint32_t e0 = (piece == P - 1) ? run.y : min(s0 + 8u, run.y);
                    if (s0 < e0) scan_points<kHardBatch, WIDE>(g, s0, e0, q.wx, q.wy, q.wz, t);
                }
            } else
            for (int jb = 0; jb < no; jb += GL) {
                const int j = jb + lane;
                if (j < no) {
                    const uint2 run = cells[j];
                    scan_points<kHardBatch, WIDE>(g, run.x, run.y, q.wx, q.wy, q.wz, t);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            nc = 0;
        };
        const float r2 = have_tau ? tau : band * band;
        if (r2 <= near_r2) {
            // the 7x7 x-rows around the home row, 32 + 17 of them per half
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int rl = lane + 32 * sub;
                const int ndy = (rl % 7) - 3, ndz = (rl / 7) - 3;
                uint32_t nid[2] = {0u, 0u};
                int nxa[2] = {0, 0}, ncl[2] = {0, 0}, nrow = 0;
                const int yy = q.cy + ndy, zz = q.cz + ndz;
                if (rl < 49 && yy >= 0 && yy < g.ncy && zz >= 0 && zz < g.ncz) {
                    const float gy = ndy > 0 ? (float)ndy - q.fry : (ndy < 0 ? q.fry - (float)(ndy + 1) : 0.0f);
                    const float gz = ndz > 0 ? (float)ndz - q.frz : (ndz < 0 ? q.frz - (float)(ndz + 1) : 0.0f);
                    const float ay = fmaxf(gy - g.slop, 0.0f), az = fmaxf(gz - g.slop, 0.0f);
                    const float b2 = (ay * ay + az * az) * (g.c * g.c) * 0.99999f;
                    if (b2 <= r2) {
                        const float reach = sqrtf(fmaxf(r2 - b2, 0.0f)) * g.inv_c * 1.00001f + g.slop;
                        const int xa = max((int)floorf(fxq - reach), 0), xb = min((int)floorf(fxq + reach), g.ncx - 1);
                        nrow = ((zz & 7) << 3) | (yy & 7);
                        const int64_t toprow = ((int64_t)(zz >> 3) * g.nby + (yy >> 3)) * g.nbx;
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const int bx = (xa >> 3) + k;
                            if (xa > xb || bx > (xb >> 3)) continue;
                            const uint4 te = g.top[toprow + bx];
                            const uint32_t mword = (nrow & 32) ? te.w : te.z;
                            if (te.x == 0 || ((mword >> (nrow & 31)) & 1u) == 0) continue;
                            nid[k] = te.x;
                            nxa[k] = max(xa, bx << 3);
                            ncl[k] = min(xb, (bx << 3) + 7) - nxa[k] + 1;
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    if (nc + GL * 8 > kHalfCells) flush_cells();
                    append_cells(nid[k], nrow, nxa[k], ncl[k]);
                }
            }
        } else {
            for (int bbase = 0; bbase < nbricks; bbase += GL) {
                const int b = bbase + lane;
                uint32_t my_id = 0;
                uint64_t my_mask = 0;
                int bx = 0, by = 0, bz = 0;
                if (b < nbricks) {
                    if (bbase == 0) {
                        bx = hbx + odx0; by = hby + ody0; bz = hbz + odz0;
                    } else {
                        bx = hbx + (b % bside) - NB;
                        by = hby + ((b / bside) % bside) - NB;
                        bz = hbz + (b / (bside * bside)) - NB;
                    }
                    if (bx >= 0 && bx < g.nbx && by >= 0 && by < g.nby && bz >= 0 && bz < g.nbz) {
                        const uint4 te = g.top[((int64_t)bz * g.nby + by) * g.nbx + bx];
                        my_id = te.x;
                        my_mask = te.x ? ((uint64_t)te.w << 32 | te.z) : 0ull;
                    }
                }
                const int cnt = __popcll(my_mask);
                const int incl = half_incl_scan(cnt);
                const int excl = incl - cnt;
                const int total = half_last(incl, half);
                for (int pbase = 0; pbase < total; pbase += GL * kPairSlots) {
#pragma unroll
                    for (int sl = 0; sl < kPairSlots; ++sl) {
                        if (pbase + sl * GL >= total) break;  // uniform inside the half
                        const int p = pbase + sl * GL + lane;
                        int o = 0;
#pragma unroll
                        for (int step = 16; step >= 1; step >>= 1) {
                            const int cand = o + step;
                            const int pc = __shfl(excl, min(cand, GL - 1), GL);
                            if (cand < GL && pc <= p) o = cand;
                        }
                        const uint32_t mlo = __shfl((uint32_t)my_mask, o, GL), mhi = __shfl((uint32_t)(my_mask >> 32), o, GL);
                        const uint32_t oid = __shfl(my_id, o, GL);
                        const int obx = __shfl(bx, o, GL), oby = __shfl(by, o, GL), obz = __shfl(bz, o, GL);
                        const int oex = __shfl(excl, o, GL);
                        int rowbit = 0, xa = 0, ncell = 0;
                        if (p < total) {
                            const uint64_t om = ((uint64_t)mhi << 32) | mlo;
                            rowbit = kth_set_bit(om, p - oex);
                            const int yy = (oby << 3) + (rowbit & 7), zz = (obz << 3) + (rowbit >> 3);
                            const int dy = yy - q.cy, dz = zz - q.cz;
                            const float gy = dy > 0 ? (float)dy - q.fry : (dy < 0 ? q.fry - (float)(dy + 1) : 0.0f);
                            const float gz = dz > 0 ? (float)dz - q.frz : (dz < 0 ? q.frz - (float)(dz + 1) : 0.0f);
                            const float ay = fmaxf(gy - g.slop, 0.0f), az = fmaxf(gz - g.slop, 0.0f);
                            const float b2 = (ay * ay + az * az) * (g.c * g.c) * 0.99999f;
                            if (b2 <= r2) {
                                const float reach = sqrtf(fmaxf(r2 - b2, 0.0f)) * g.inv_c * 1.00001f + g.slop;
                                xa = max((int)floorf(fxq - reach), obx << 3);
                                const int xb = min((int)floorf(fxq + reach), (obx << 3) + 7);
                                ncell = max(xb - xa + 1, 0);
                            }
                        }
                        if (nc + GL * 8 > kHalfCells) flush_cells();
                        append_cells(oid, rowbit, xa, min(ncell, 8));
                    }
                }
            }
        }
        flush_cells();
        merge_lists_half(t, best, half);
        const bool found5 = !is_empty(best[kK - 1]);
        const float d5 = __uint_as_float((uint32_t)(best[kK - 1] >> 32));
        // the decisions of match_hard_body, one round at a time
        bool done = have_tau;                                             // every point within tau was visited: exact
        done = done || (found5 && d5 <= band * band);                     // five found inside the fully scanned band
        done = done || (band * band > a.gates.knn_d2_gate);               // beyond the gate: "not five within it"
        if (!done) {
            if (found5) { have_tau = true; tau = fminf(d5, gate2_up); }   // radius known now: one exact round
            else band = fminf(band * 2.0f, sqrtf(gate2_up));
#pragma unroll
            for (int k = 0; k < kK; ++k) t[k] = kEmptyKey;
        } else {
            if (lane == 0) {
                int32_t *o_idx;
                float *o_d2;
                out(slot, o_idx, o_d2);
                store_result(best, (int64_t)qi, o_idx, o_d2);
            }
            uint32_t ticket = 0;
            if (lane == 0) ticket = atomicAdd(a.qheads + shard * kQueueStride, 1u);
            ticket = (uint32_t)half_first((int)ticket, half);
            h = (uint32_t)ngrp + ticket * kQueueShards + shard;
            fresh = true;
        }
    }
}

template <bool WIDE>
__global__ __launch_bounds__(256, S2M_HARD_OCC) void match_hard32(MatchArgs a)
{
    __shared__ uint2 cells_all[8][kHalfCells];  // one cell list per half-wave of the workgroup
    match_hard32_body<WIDE>(a, cells_all[threadIdx.x >> 5], [&](uint32_t, int32_t *&idx, float *&d2) {
        idx = a.nn_idx;
        d2 = a.nn_d2;
    });
}

template <bool WIDE, bool FAR = false>
__global__ __launch_bounds__(256, S2M_HARD_OCC) void match_hard(MatchArgs a)
{
    __shared__ uint2 cells_all[4][kMaxCells];  // one cell list per wave of the workgroup
    match_hard_body<WIDE, FAR>(a, cells_all[threadIdx.x >> 6], [&](uint32_t, int32_t *&idx, float *&d2) {
        idx = a.nn_idx;
        d2 = a.nn_d2;
    });
}

// the far points of ALL scans of a batched launch, from one list through one set of queue heads
template <bool WIDE>
__global__ __launch_bounds__(256, S2M_HARD_OCC) void match_hard_batch(BatchArgs b)
{
    __shared__ uint2 cells_all[4][kMaxCells];
    MatchArgs a;
    a.grid = b.grid; a.gates = b.gates;
    a.sx = a.sy = a.sz = nullptr; a.n = b.n_max;
    a.nn_idx = nullptr; a.nn_d2 = nullptr;
    a.hard_rec = b.hard_rec; a.hard_off1 = b.hard_off1; a.slot = 0;
    a.hard_count = b.hard_count; a.qheads = b.qheads; a.dbg = nullptr;
    match_hard_body<WIDE, false>(a, cells_all[threadIdx.x >> 6], [&](uint32_t slot, int32_t *&idx, float *&d2) {
        const uint32_t s_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);  // one point per wave: uniform
        idx = b.d[s_].nn_idx;
        d2 = b.d[s_].nn_d2;
    });
}

template <int G>
static void launch_easy(const MatchArgs &a, bool wide, bool cells, int nb, hipStream_t st)
{
    const int64_t threads = (int64_t)a.n * G;
    const int blocks = (int)((threads + 255) / 256);
    // 32-bit byte offsets into the sorted point array unless the map is too large for them
    if (cells) {  // the per-cell form (S2M_EASY_CELLS=1), kept for A/B measurements
        if (!wide) hipLaunchKernelGGL((match_easy<G, false>), dim3(blocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((match_easy<G, true>), dim3(blocks), dim3(256), 0, st, a);
    } else if (nb == 1) {
        if (!wide) hipLaunchKernelGGL((match_rows<G, false, 1>), dim3(blocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((match_rows<G, true, 1>), dim3(blocks), dim3(256), 0, st, a);
    } else if (nb == 2) {
        if (!wide) hipLaunchKernelGGL((match_rows<G, false, 2>), dim3(blocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((match_rows<G, true, 2>), dim3(blocks), dim3(256), 0, st, a);
    } else {  // default: three batches (24 point loads) per trip
        if (!wide) hipLaunchKernelGGL((match_rows<G, false, 3>), dim3(blocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((match_rows<G, true, 3>), dim3(blocks), dim3(256), 0, st, a);
    }
}

// Lanes per far point.  Measured (gpurun_out/r03g): with ONE scan in flight a wave per point is faster (C3 0.150 vs
// 0.159 ms/step, C4 0.208 vs 0.212: the two halves of a wave share one instruction stream, so their load chains run one
// after the other and a point's latency nearly doubles -- and a single scan's launch is as long as its slowest points);
// with K scans in one grid the half-wave form wins (18.7 -> 20.4 k scans/s at K = 8, 21.1 -> 23.3 k at K = 16: the list
// is long, only throughput counts, and 8,192 points in flight hide more of each other's waits).  So: half-waves in
// batched launches, waves otherwise; S2M_HARD_LANES=32|64 forces one form everywhere (A/B).
static bool hard_half_waves(bool batched)
{
    static const int v = std::getenv("S2M_HARD_LANES") ? std::atoi(std::getenv("S2M_HARD_LANES")) : 0;
    return v == 32 ? true : (v == 64 ? false : batched);
}

// maximum over the ACTIVE lanes of the wave (cold path: plain shuffles; inactive lanes contribute 0)
__device__ __forceinline__ uint32_t wave_max_u32_slow(uint32_t v)
{
    const unsigned long long act = __ballot(1);
    uint32_t m = v;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int src = (int)(threadIdx.x & 63) ^ off;
        const uint32_t o = (uint32_t)__shfl((int)m, src, 64);
        if ((act >> src) & 1ull) m = max(m, o);
    }
    return m;
}

// ---- completion of the lists that ended short at the gate (s2m_complete_neighbors) -------------------------
// A list is the exact, final answer when it holds five neighbours whose 5th distance is inside the radius the search
// was allowed (a.gates.knn_d2_gate: the gate, or the larger radius of the last completion round) -- both search
// kernels guarantee that much and no more: beyond it a list may be short, or full of whatever the last band happened
// to see.  Every other scan point goes to the far-point list again, as a point without a radius (its world-frame
// query is the one of the rematch pass that produced the list).
__global__ __launch_bounds__(256) void collect_short_kernel(MatchArgs a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool want = false;
    uint32_t far_bits = 0u;
    HardRec rec = {0.f, 0.f, 0.f, 0u, 0.f, 0u, 0u, 0u};
    if (i < a.n && !(a.nn_idx[(int64_t)i * kK + (kK - 1)] >= 0 && a.nn_d2[(int64_t)i * kK + (kK - 1)] <= a.gates.knn_d2_gate)) {
        body_to_world(a.pose, a.sx[i], a.sy[i], a.sz[i], rec.wx, rec.wy, rec.wz);
        rec.qi = (uint32_t)i;
        // squared distance to the centre of the grid: the host derives from its maximum the radius at which every
        // map point has been seen (a query may lie far outside the grid).  A non-finite query can have no neighbours:
        // it is left as it is.
        const Grid &g = a.grid;
        const float cx = g.ox + 0.5f * (float)g.ncx * g.c, cy = g.oy + 0.5f * (float)g.ncy * g.c, cz = g.oz + 0.5f * (float)g.ncz * g.c;
        const float d2c = ((rec.wx - cx) * (rec.wx - cx) + (rec.wy - cy) * (rec.wy - cy)) + (rec.wz - cz) * (rec.wz - cz);
        want = d2c < 3.0e38f;  // false for NaN and +inf
        if (want) far_bits = __float_as_uint(d2c);  // >= 0: the bit pattern orders like the value
    }
    // the wave's maximum, taken by ALL lanes (a lane with a complete list contributes 0) and reported by whichever lane
    // comes first: short lists are sparse, so the wave's lane 0 usually is not one of them
    const uint32_t mx = wave_max_u32_slow(far_bits);
    if (mx != 0u && (threadIdx.x & 63) == 0) atomicMax(a.hard_count + 2, mx);
    append_rec(a.hard_rec, a.hard_count, want, rec);
}

void launch_collect_short(const MatchArgs &a, hipStream_t st)
{
    if (a.n <= 0) return;
    hipLaunchKernelGGL(collect_short_kernel, dim3((a.n + 255) / 256), dim3(256), 0, st, a);
}

static void launch_hard(const MatchArgs &a, bool wide, hipStream_t st)
{
    // the rest: one wave per point (narrower groups measured slower: the far tail is latency-bound)
    const int hg = 64;
    // as many waves as stay resident together (116 VGPRs: 4 per SIMD, 4,096 on the chip); the rest of the list is
    // pulled through the queue heads
    if (a.qheads && !a.dbg && hard_half_waves(false)) {
        // two points per wave: as many half-waves as stay resident together (8,192), at most one per scan point
        const int64_t halves = std::min<int64_t>(a.n, 1024 * S2M_HARD_OCC * 2);
        const int blocks = (int)((halves * 32 + 255) / 256);
        if (!wide) hipLaunchKernelGGL(match_hard32<false>, dim3(blocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(match_hard32<true>, dim3(blocks), dim3(256), 0, st, a);
        return;
    }
    const int64_t groups = std::min<int64_t>(a.n, (a.qheads ? 1024 * S2M_HARD_OCC : 8192) * (64 / hg));
    const int blocks = (int)((groups * hg + 255) / 256);
    if (!wide) hipLaunchKernelGGL(match_hard<false>, dim3(blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(match_hard<true>, dim3(blocks), dim3(256), 0, st, a);
}

void launch_match_hard_only(const MatchArgs &a, hipStream_t st)
{
    if (a.n <= 0) return;
    const bool wide = a.grid.sent_off == 0 && a.grid.m != 0;
    const int64_t groups = std::min<int64_t>(a.n, 1024 * S2M_HARD_OCC);
    const int blocks = (int)((groups * 64 + 255) / 256);
    if (!wide) hipLaunchKernelGGL((match_hard<false, true>), dim3(blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((match_hard<true, true>), dim3(blocks), dim3(256), 0, st, a);
}

template <bool WIDE>
__global__ __launch_bounds__(256, S2M_HARD_OCC) void match_hard32_batch(BatchArgs b)
{
    __shared__ uint2 cells_all[8][kHalfCells];
    MatchArgs a;
    a.grid = b.grid; a.gates = b.gates;
    a.sx = a.sy = a.sz = nullptr; a.n = b.n_max;
    a.nn_idx = nullptr; a.nn_d2 = nullptr;
    a.hard_rec = b.hard_rec; a.hard_off1 = b.hard_off1; a.slot = 0;
    a.hard_count = b.hard_count; a.qheads = b.qheads; a.dbg = nullptr;
    match_hard32_body<WIDE>(a, cells_all[threadIdx.x >> 5], [&](uint32_t slot, int32_t *&idx, float *&d2) {
        idx = b.d[slot].nn_idx;   // the two halves of a wave may serve different scans: a per-lane look-up in the table
        d2 = b.d[slot].nn_d2;
    });
}

// search kernels of one batched pass (G = 2 lanes per point, two point batches per trip: the chip is shared by K scans)
void launch_match_batch(const BatchArgs &b, hipStream_t st)
{
    if (b.n_max <= 0 || b.k <= 0) return;
    const bool wide = b.grid.sent_off == 0 && b.grid.m != 0;
    static const int nb_env = std::getenv("S2M_BATCH_NB") ? std::atoi(std::getenv("S2M_BATCH_NB")) : 0;  // dev knobs
    static const int g_env = std::getenv("S2M_BATCH_G") ? std::atoi(std::getenv("S2M_BATCH_G")) : 0;
    const int G = g_env == 1 ? 1 : (g_env == 4 ? 4 : 2);
    const int64_t threads = (int64_t)b.n_max * G;
    const dim3 grid((unsigned)((threads + 255) / 256), (unsigned)b.k);
    if (wide) hipLaunchKernelGGL((match_rows_batch<2, true, 2>), dim3((unsigned)(((int64_t)b.n_max * 2 + 255) / 256), (unsigned)b.k), dim3(256), 0, st, b);
    else if (G == 1 && nb_env == 1) hipLaunchKernelGGL((match_rows_batch<1, false, 1>), grid, dim3(256), 0, st, b);
    else if (G == 1 && nb_env == 3) hipLaunchKernelGGL((match_rows_batch<1, false, 3>), grid, dim3(256), 0, st, b);
    else if (G == 1) hipLaunchKernelGGL((match_rows_batch<1, false, 2>), grid, dim3(256), 0, st, b);
    else if (G == 4 && nb_env == 1) hipLaunchKernelGGL((match_rows_batch<4, false, 1>), grid, dim3(256), 0, st, b);
    else if (G == 4) hipLaunchKernelGGL((match_rows_batch<4, false, 2>), grid, dim3(256), 0, st, b);
    else if (nb_env == 1) hipLaunchKernelGGL((match_rows_batch<2, false, 1>), grid, dim3(256), 0, st, b);
    else if (nb_env == 3) hipLaunchKernelGGL((match_rows_batch<2, false, 3>), grid, dim3(256), 0, st, b);
    else hipLaunchKernelGGL((match_rows_batch<2, false, 2>), grid, dim3(256), 0, st, b);
    const int blocks = 1024 * S2M_HARD_OCC * 64 / 256;  // the resident waves; the rest of the list comes through the heads
    if (hard_half_waves(true)) {
        if (!wide) hipLaunchKernelGGL(match_hard32_batch<false>, dim3(blocks), dim3(256), 0, st, b);
        else hipLaunchKernelGGL(match_hard32_batch<true>, dim3(blocks), dim3(256), 0, st, b);
    } else if (!wide) hipLaunchKernelGGL(match_hard_batch<false>, dim3(blocks), dim3(256), 0, st, b);
    else hipLaunchKernelGGL(match_hard_batch<true>, dim3(blocks), dim3(256), 0, st, b);
}

void launch_match_far_points(const MatchArgs &a, int group, hipStream_t st)
{
    if (a.n <= 0) return;
    launch_hard(a, (a.grid.sent_off == 0 && a.grid.m != 0) || (group & 0x10000), st);
}

// group bit 0x40000: the first-shell kernel only (the host bets that it resolves every point; s2m_engine.cpp, run_pass)
void launch_match(const MatchArgs &a, int group, hipStream_t st)
{
    if (a.n <= 0) return;
    // 64-bit point addresses when the sentinel block is out of reach of a 32-bit byte offset (or on request:
    // bit 16 of `group`, S2M_WIDE_ADDR=1, so the tests can cover that path on a small map)
    const bool wide = (a.grid.sent_off == 0 && a.grid.m != 0) || (group & 0x10000);
    const bool cells = (group & 0x20000) != 0;
    const int nb = (group >> 8) & 0xf;  // batches per trip of the row-run kernel (0 = default)
    switch (group & 0xff) {
        case 1: launch_easy<1>(a, wide, cells, nb, st); break;
        case 4: launch_easy<4>(a, wide, cells, nb, st); break;
        case 8: launch_easy<8>(a, wide, cells, nb, st); break;
        default: launch_easy<2>(a, wide, cells, nb, st); break;
    }
    if (!(group & 0x40000)) launch_hard(a, wide, st);
}

}  // namespace s2m
