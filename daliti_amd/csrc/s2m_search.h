// s2m_search.h -- what the two search kernels share (s2m_match.hip: first shell; s2m_match_far.hip: far points):
// the 64-bit candidate key and its top-5 network, DPP minima and prefix sums, the query's home cell and the
// termination bounds, the 8-point batch scan, the far-point record.
//
// Candidates are ranked by the 64-bit key (float bits of d2) << 32 | sorted position: d2 >= 0 so the
// bit pattern orders like the value, the position makes keys unique, and a top-5 insertion is a
// handful of 64-bit min/max with no tie branches.  Sorted position = (brick, cell, caller index), a
// total order the oracle computes from s2m_map_info (ikd-Tree ranks by d2, then x, ikd_Tree.h:102-108, with
// a traversal-dependent choice at the 5th place; exact ties are ~1e-7 of queries and every choice is a valid
// exact 5-NN).  The position doubles as the gather address of the plane fit (s2m_reduce.hip).
// Padding slots of a point batch load a sentinel point beyond the array (distance +inf), so the inner loops
// have no predicate; keys go into the top-5 eight at a time through a pruned sorting network.
//
// Arithmetic contract: compiled with -ffp-contract=off; d2 = ((dx*dx + dy*dy) + dz*dz) in float
// exactly like calc_dist (ikd_Tree.cpp:1682-1688) and oracle/s2m_oracle.c.
#pragma once
#include <cfloat>
#include <cmath>

#include "s2m_device.h"
#include "s2m_kernels.h"

namespace s2m {

typedef unsigned long long u64;
// empty slot: +infinity as a double (orders after every key)
constexpr u64 kEmptyKey = 0x7ff0000000000000ull;
// A slot is empty when the float in its high word is not a finite distance: the initial +inf pattern
// above, or the key of the sentinel point pts[m] = (3e38, 3e38, 3e38) that padding lanes of a batch load
// (its squared distance overflows to +inf), so padding needs no predicate anywhere after the address.
__device__ __forceinline__ bool is_empty(u64 k) { return (uint32_t)(k >> 32) >= 0x7f800000u; }
constexpr int kBatch = 8;  // point loads in flight per lane and batch = the width of the top-5 network
#ifndef S2M_HARD_BAND
#define S2M_HARD_BAND 1.7f  // first band of the far-point search in cells (measured sweeps: NOTEBOOK.md)
#endif

__device__ __forceinline__ u64 make_key(float d2, uint32_t orig)
{
    return ((u64)__float_as_uint(d2) << 32) | (u64)orig;
}

// v_min_f64 / v_max_f64 issued directly: fmin()/fmax() make the compiler canonicalise every operand first
// (one extra v_max_f64 x, x, x per key -- 13 of ~61 operations per batch); the keys are never NaN as
// doubles (a float's bits in the high word give an exponent below 0x7fd, see cex), so the result is
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
// A key with a non-negative float in its high word is a finite positive double whose IEEE order equals the unsigned
// order of the bits (float exponent 0xFF maps to double exponent <= 0x7FC, still finite), so one compare-exchange is
// v_min_f64 + v_max_f64: branch-free, no SGPR / exec traffic.
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
        static_assert(B == 8, "the top-5 network takes eight keys");
        insert_batch8(t, key);
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
    // cell and in-cell position in double from the float operands (cell_pos, s2m_device.h): the map was binned with this
    // expression, and the fraction's only error is its rounding to float (6e-8 of a cell), wherever the origin is
    const double fx = cell_pos(q.wx, g.ox, g.inv_c), fy = cell_pos(q.wy, g.oy, g.inv_c), fz = cell_pos(q.wz, g.oz, g.inv_c);
    const double flx = floor(fx), fly = floor(fy), flz = floor(fz);
    // clamp far-away queries so the int conversion is defined; the bound stays valid because the
    // clamped cells lie outside the bricks in use and hold no points
    const double lim = 1.0e9;
    q.cx = (int)fmin(fmax(flx, -lim), lim);
    q.cy = (int)fmin(fmax(fly, -lim), lim);
    q.cz = (int)fmin(fmax(flz, -lim), lim);
    q.frx = (float)(fx - flx); q.fry = (float)(fy - fly); q.frz = (float)(fz - flz);
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

}  // namespace s2m
