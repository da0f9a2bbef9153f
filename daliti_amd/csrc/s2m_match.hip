// s2m_match.hip -- rematch pass, part 1: exact 5-NN on the brick grid + 5-point plane fit.
//
// Replaces, per scan point (eskf_lio/src/laserMapping.cpp:835-863):
//   body->world transform (:835-841), ikdtree.Nearest_Search(point_world, 5, ...) (:850;
//   ikd-Tree/ikd_Tree.cpp:425-461, 1061-1244), the neighbour gate (:852-854) and
//   esti_plane(pabcd, points_near, 0.1f) (:863; eskf_lio/include/common_lib.h:267-299).
// The plane only depends on the (world-frame, constant) neighbours, so it is fitted once per
// rematch and cached; the reference re-fits the identical plane every iteration.
//
// Execution model (gfx950, wave64): a group of G lanes owns one scan point.  The lanes split the
// (2r+1)^2 x-rows of the cube of cells around the query, each lane streams its rows' candidate
// points (one 16-byte load each) through a private sorted top-5 in registers, and the group
// merges the private lists with shuffle min-reductions.  The cube grows (r = 1, 2, ...) until the
// 5th-best distance is provably inside the visited cube, so the result is the exact 5-NN; the
// search stops as soon as the bound passes the reference's d2 <= 5 gate.
//
// Arithmetic contract: this file is compiled with -ffp-contract=off; every float expression below
// is evaluated in the same order as oracle/s2m_oracle.c so per-point results are bit-identical.
#include <cfloat>
#include <cmath>

#include "s2m_device.h"
#include "s2m_kernels.h"

namespace s2m {

struct Cand {
    float d2, x, y, z;
    uint32_t w;
};

// strict total order (d2, x, y, z): the reference orders by d2 and breaks d2 ties by x
// (ikd-Tree/ikd_Tree.h:102-108); y, z make the order total so the result is layout independent
__device__ __forceinline__ bool cand_less(const Cand &a, const Cand &b)
{
    if (a.d2 != b.d2) return a.d2 < b.d2;
    if (a.x != b.x) return a.x < b.x;
    if (a.y != b.y) return a.y < b.y;
    return a.z < b.z;
}

__device__ __forceinline__ void cand_swap_if(bool p, Cand &a, Cand &b)
{
    const Cand ta = a, tb = b;
    a.d2 = p ? tb.d2 : ta.d2; a.x = p ? tb.x : ta.x; a.y = p ? tb.y : ta.y; a.z = p ? tb.z : ta.z;
    a.w = p ? tb.w : ta.w;
    b.d2 = p ? ta.d2 : tb.d2; b.x = p ? ta.x : tb.x; b.y = p ? ta.y : tb.y; b.z = p ? ta.z : tb.z;
    b.w = p ? ta.w : tb.w;
}

__device__ __forceinline__ void offer(Cand (&t)[kK], const Cand &c)
{
    if (c.d2 > t[kK - 1].d2) return;  // common case
    if (!cand_less(c, t[kK - 1])) return;
    t[kK - 1] = c;
#pragma unroll
    for (int k = kK - 2; k >= 0; --k) cand_swap_if(cand_less(t[k + 1], t[k]), t[k], t[k + 1]);
}

template <int G>
__device__ __forceinline__ float group_min(float v)
{
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, G));
    return v;
}

// esti_plane<float>: column-pivoted Householder QR least squares of A x = -1, same operation
// order as orc_esti_plane (oracle/s2m_oracle.c).  Returns the inlier verdict.
__device__ bool fit_plane(const Cand (&nb)[kK], float thr, float4 &pl)
{
    float A[kK][3], c[kK];
    float tau[3], nu[3], nd[3];
    int trans[3];
#pragma unroll
    for (int i = 0; i < kK; ++i) {
        A[i][0] = nb[i].x; A[i][1] = nb[i].y; A[i][2] = nb[i].z;
        c[i] = -1.0f;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < kK; ++i) s = s + A[i][k] * A[i][k];
        nd[k] = __builtin_sqrtf(s);
        nu[k] = nd[k];
    }
    float nmax = nu[0];
    if (nu[1] > nmax) nmax = nu[1];
    if (nu[2] > nmax) nmax = nu[2];
    const float th = nmax * FLT_EPSILON;
    const float threshold_helper = (th * th) / (float)kK;
    const float downdate_thr = __builtin_sqrtf(FLT_EPSILON);
    int nonzero = 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int big = k;
        float bigv = nu[k];
#pragma unroll
        for (int j = k + 1; j < 3; ++j)
            if (nu[j] > bigv) { big = j; bigv = nu[j]; }
        const float big_sq = bigv * bigv;
        if (nonzero == 3 && big_sq < threshold_helper * (float)(kK - k)) nonzero = k;
        trans[k] = big;
        // column swap k <-> big with static indices (big is k, k+1 or 2)
#pragma unroll
        for (int j = k + 1; j < 3; ++j) {
            const bool sw = (big == j);
#pragma unroll
            for (int i = 0; i < kK; ++i) {
                const float a = A[i][k], b = A[i][j];
                A[i][k] = sw ? b : a;
                A[i][j] = sw ? a : b;
            }
            const float u0 = nu[k], u1 = nu[j], d0 = nd[k], d1 = nd[j];
            nu[k] = sw ? u1 : u0; nu[j] = sw ? u0 : u1;
            nd[k] = sw ? d1 : d0; nd[j] = sw ? d0 : d1;
        }
        float tail = 0.0f;
#pragma unroll
        for (int i = k + 1; i < kK; ++i) tail = tail + A[i][k] * A[i][k];
        const float c0 = A[k][k];
        float beta;
        if (tail <= FLT_MIN) {
            tau[k] = 0.0f;
            beta = c0;
#pragma unroll
            for (int i = k + 1; i < kK; ++i) A[i][k] = 0.0f;
        } else {
            beta = __builtin_sqrtf(c0 * c0 + tail);
            if (c0 >= 0.0f) beta = -beta;
            const float den = c0 - beta;
#pragma unroll
            for (int i = k + 1; i < kK; ++i) A[i][k] = A[i][k] / den;
            tau[k] = (beta - c0) / beta;
        }
        A[k][k] = beta;
        if (tau[k] != 0.0f) {
#pragma unroll
            for (int j = k + 1; j < 3; ++j) {
                float tmp = 0.0f;
#pragma unroll
                for (int i = k + 1; i < kK; ++i) tmp = tmp + A[i][k] * A[i][j];
                tmp = tmp + A[k][j];
                A[k][j] = A[k][j] - tau[k] * tmp;
#pragma unroll
                for (int i = k + 1; i < kK; ++i) A[i][j] = A[i][j] - (tau[k] * A[i][k]) * tmp;
            }
        }
#pragma unroll
        for (int j = k + 1; j < 3; ++j) {
            if (nu[j] != 0.0f) {
                float t = fabsf(A[k][j]) / nu[j];
                t = (1.0f + t) * (1.0f - t);
                if (t < 0.0f) t = 0.0f;
                const float q = nu[j] / nd[j];
                const float t2 = t * (q * q);
                if (t2 <= downdate_thr) {
                    float s = 0.0f;
#pragma unroll
                    for (int i = k + 1; i < kK; ++i) s = s + A[i][j] * A[i][j];
                    nd[j] = __builtin_sqrtf(s);
                    nu[j] = nd[j];
                } else {
                    nu[j] = nu[j] * __builtin_sqrtf(t);
                }
            }
        }
    }
    // permutation = identity with the transpositions applied on the right (trans[k] >= k)
    int p0 = 0, p1 = 1, p2 = 2;
    if (trans[0] == 1) { const int t = p0; p0 = p1; p1 = t; }
    else if (trans[0] == 2) { const int t = p0; p0 = p2; p2 = t; }
    if (trans[1] == 2) { const int t = p1; p1 = p2; p2 = t; }
    const int perm[3] = {p0, p1, p2};
    float xs[3] = {0.0f, 0.0f, 0.0f};
    if (nonzero > 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (k < nonzero && tau[k] != 0.0f) {
                float tmp = 0.0f;
#pragma unroll
                for (int i = k + 1; i < kK; ++i) tmp = tmp + A[i][k] * c[i];
                tmp = tmp + c[k];
                c[k] = c[k] - tau[k] * tmp;
#pragma unroll
                for (int i = k + 1; i < kK; ++i) c[i] = c[i] - (tau[k] * A[i][k]) * tmp;
            }
        }
#pragma unroll
        for (int i = 2; i >= 0; --i) {
            if (i < nonzero) {
                c[i] = c[i] / A[i][i];
#pragma unroll
                for (int r = 0; r < i; ++r) c[r] = c[r] - c[i] * A[r][i];
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (i < nonzero) {
                if (perm[i] == 0) xs[0] = c[i];
                else if (perm[i] == 1) xs[1] = c[i];
                else xs[2] = c[i];
            }
        }
    }
    const float n = __builtin_sqrtf((xs[0] * xs[0] + xs[1] * xs[1]) + xs[2] * xs[2]);
    pl.x = xs[0] / n;
    pl.y = xs[1] / n;
    pl.z = xs[2] / n;
    pl.w = (float)(1.0 / (double)n);
    bool ok = true;
#pragma unroll
    for (int j = 0; j < kK; ++j) {
        const float v = ((pl.x * nb[j].x + pl.y * nb[j].y) + pl.z * nb[j].z) + pl.w;
        if (fabsf(v) > thr) ok = false;
    }
    return ok;
}

// Streams the points of cells [xa, xb] of one x-row (cell row yy, zz) through the private top-5.
// A row crosses at most a few bricks; per brick: one top-level load, two table loads, then the
// candidates in batches of four independent 16-byte loads.
__device__ __forceinline__ void scan_row(const Grid &g, int yy, int zz, int xa, int xb, float wx, float wy,
                                         float wz, Cand (&t)[kK])
{
    const int by = yy >> 3, bz = zz >> 3;
    const int rowoff = (((zz & 7) << 3) | (yy & 7)) << 3;
    const int64_t toprow = ((int64_t)bz * g.nby + by) * g.nbx;
    for (int bx = xa >> 3; bx <= (xb >> 3); ++bx) {
        const uint32_t b = g.top[toprow + bx];
        if (b == 0) continue;
        const int l0 = max(xa, bx << 3) & 7, l1 = min(xb, (bx << 3) + 7) & 7;
        const uint32_t *tb = g.tab + (int64_t)(b - 1) * kBrickStride + rowoff;
        const uint32_t s = tb[l0], e = tb[l1 + 1];
        for (uint32_t i = s; i < e; i += 4) {
            float4 p[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) p[u] = g.pts[min(i + u, e - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i + u < e) {
                    Cand c;
                    const float dx = wx - p[u].x, dy = wy - p[u].y, dz = wz - p[u].z;
                    float d = dx * dx + dy * dy;
                    d = d + dz * dz;
                    c.d2 = d; c.x = p[u].x; c.y = p[u].y; c.z = p[u].z; c.w = __float_as_uint(p[u].w);
                    offer(t, c);
                }
            }
        }
    }
}

// Group-wide sorted top-5 of the G private lists (non-destructive: works on a copy).
template <int G>
__device__ __forceinline__ int merge_lists(const Cand (&priv)[kK], Cand (&best)[kK], int j, int gbase, uint64_t gmask)
{
    Cand t[kK];
#pragma unroll
    for (int k = 0; k < kK; ++k) t[k] = priv[k];
    int found = 0;
#pragma unroll
    for (int k = 0; k < kK; ++k) {
        const float key = t[0].d2;
        const float m = group_min<G>(key);
        uint64_t bal = __ballot(key == m && m < INFINITY);
        uint64_t gb = (bal >> gbase) & gmask;
        best[k].d2 = INFINITY; best[k].x = 0.f; best[k].y = 0.f; best[k].z = 0.f; best[k].w = 0xffffffffu;
        if (gb != 0) {
            int win;
            if (__popcll(gb) == 1) {
                win = __ffsll((unsigned long long)gb) - 1;
            } else {  // equal d2 in several lanes: smallest (x, y, z) wins
                bool cnd = (gb >> j) & 1ull;
                const float mx = group_min<G>(cnd ? t[0].x : INFINITY);
                cnd = cnd && (t[0].x == mx);
                const float my = group_min<G>(cnd ? t[0].y : INFINITY);
                cnd = cnd && (t[0].y == my);
                const float mz = group_min<G>(cnd ? t[0].z : INFINITY);
                cnd = cnd && (t[0].z == mz);
                bal = __ballot(cnd);
                gb = (bal >> gbase) & gmask;
                win = __ffsll((unsigned long long)gb) - 1;
            }
            best[k].d2 = m;
            best[k].x = __shfl(t[0].x, win, G);
            best[k].y = __shfl(t[0].y, win, G);
            best[k].z = __shfl(t[0].z, win, G);
            best[k].w = __shfl(t[0].w, win, G);
            ++found;
            if (j == win) {
#pragma unroll
                for (int s = 0; s < kK - 1; ++s) t[s] = t[s + 1];
                t[kK - 1].d2 = INFINITY;
            }
        }
    }
    return found;
}

template <int G>
__global__ __launch_bounds__(256) void match_kernel(Grid g, Pose pose, Gates gates,
                                                    const float *__restrict__ sx,
                                                    const float *__restrict__ sy,
                                                    const float *__restrict__ sz, int n,
                                                    float4 *__restrict__ plane_out,
                                                    uint8_t *__restrict__ flags_out,
                                                    uint8_t *__restrict__ sel_out,
                                                    int32_t *__restrict__ nn_idx,
                                                    float *__restrict__ nn_d2, uint32_t *__restrict__ dbg)
{
    const long long dbg_t0 = dbg ? wall_clock64() : 0;
    uint32_t dbg_rounds = 0;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int q = tid / G;
    const int j = threadIdx.x & (G - 1);
    if (q >= n) return;  // group-uniform
    const int lane = threadIdx.x & 63;
    const int gbase = lane & ~(G - 1);
    const uint64_t gmask = (G == 64) ? ~0ull : ((1ull << G) - 1ull);

    float wx, wy, wz;
    body_to_world(pose, sx[q], sy[q], sz[q], wx, wy, wz);

    // home cell and the query's position inside it, in cell units
    const float fx = (wx - g.ox) * g.inv_c, fy = (wy - g.oy) * g.inv_c, fz = (wz - g.oz) * g.inv_c;
    const float flx = floorf(fx), fly = floorf(fy), flz = floorf(fz);
    // clamp far-away queries so the int conversion is defined; the bound below stays valid
    // because clamped cells lie outside the grid and hold no points
    const float lim = 1.0e9f;
    const int cx = (int)fminf(fmaxf(flx, -lim), lim), cy = (int)fminf(fmaxf(fly, -lim), lim),
              cz = (int)fminf(fmaxf(flz, -lim), lim);
    float fmin_ = fminf(fminf(fx - flx, 1.0f - (fx - flx)), fminf(fy - fly, 1.0f - (fy - fly)));
    fmin_ = fminf(fmin_, fminf(fz - flz, 1.0f - (fz - flz)));

    Cand t[kK], best[kK];
#pragma unroll
    for (int k = 0; k < kK; ++k) { t[k].d2 = INFINITY; t[k].x = 0.f; t[k].y = 0.f; t[k].z = 0.f; t[k].w = 0xffffffffu; }
    int found = 0;
    const int rcap = max(max(g.ncx, g.ncy), g.ncz);
    // smallest radius whose bound passes the d2 gate: beyond it the 5th neighbour cannot matter
    const int rgate = (int)ceilf(sqrtf(gates.knn_d2_gate) * g.inv_c * 1.000002f - fmin_ + g.slop) + 1;
    int rdone = -1;  // cube of radius rdone around the home cell is fully scanned (-1: nothing)
    int r = 1;
    for (;;) {
        // scan the shell (rdone, r]: new rows completely, old rows only their two new ends
        const int side = 2 * r + 1;
        const int nrows = side * side;
        const int xlo = max(cx - r, 0), xhi = min(cx + r, g.ncx - 1);
        for (int row = j; row < nrows; row += G) {
            const int dy = (row % side) - r, dz = (row / side) - r;
            const int yy = cy + dy, zz = cz + dz;
            if (yy < 0 || yy >= g.ncy || zz < 0 || zz >= g.ncz) continue;
            if (max(abs(dy), abs(dz)) > rdone) {
                if (xlo <= xhi) scan_row(g, yy, zz, xlo, xhi, wx, wy, wz, t);
            } else {
                const int a1 = min(cx - rdone - 1, g.ncx - 1), b0 = max(cx + rdone + 1, 0);
                if (xlo <= a1) scan_row(g, yy, zz, xlo, a1, wx, wy, wz, t);
                if (b0 <= xhi) scan_row(g, yy, zz, b0, xhi, wx, wy, wz, t);
            }
        }
        rdone = r;
        ++dbg_rounds;
        found = merge_lists<G>(t, best, j, gbase, gmask);
        // every point outside the scanned cube is at least lb away from the query
        float lb = ((float)r + fmin_ - g.slop) * g.c;
        lb = fmaxf(lb, 0.0f) * 0.999999f;
        const float lb2 = lb * lb;
        if (found == kK && best[kK - 1].d2 <= lb2) break;  // exact 5-NN found
        if (lb2 > gates.knn_d2_gate) break;                // 5th neighbour is beyond the gate
        if (r >= rcap) break;                              // whole grid scanned
        int rn;
        if (found == kK) {
            // all better candidates lie within sqrt(d5): jump straight to the radius covering it
            rn = (int)ceilf(sqrtf(best[kK - 1].d2) * g.inv_c * 1.000002f - fmin_ + g.slop);
        } else {
            rn = 2 * r;
        }
        r = min(max(rn, r + 1), max(rgate, r + 1));
        r = min(r, rcap);
    }

    const bool gate = (found == kK) && !(best[kK - 1].d2 > gates.knn_d2_gate);
    float4 pl = make_float4(0.f, 0.f, 0.f, 0.f);
    bool plane_ok = false;
    if (gate) plane_ok = fit_plane(best, gates.plane_thr, pl);
    if (j == 0) {
        plane_out[q] = pl;
        flags_out[q] = (uint8_t)((gate ? kFlagGate : 0) | (plane_ok ? kFlagPlane : 0));
        sel_out[q] = gate ? 1 : 0;  // point_selected_surf after the gate (:852-854)
    }
    if (dbg != nullptr && j == 0) {
        dbg[4 * (int64_t)q + 0] = (uint32_t)(wall_clock64() - dbg_t0);
        dbg[4 * (int64_t)q + 1] = (uint32_t)rdone;
        dbg[4 * (int64_t)q + 2] = 0;
        dbg[4 * (int64_t)q + 3] = dbg_rounds;
    }
    if (nn_idx != nullptr && j < kK) {
        // lane k of the group stores neighbour k
        int32_t idx = -1;
        float d2v = INFINITY;
#pragma unroll
        for (int k = 0; k < kK; ++k)
            if (j == k) { idx = (int32_t)best[k].w; d2v = best[k].d2; }
        nn_idx[(int64_t)q * kK + j] = idx;
        nn_d2[(int64_t)q * kK + j] = d2v;
    }
}

template <int G>
static void launch_g(const MatchArgs &a, hipStream_t st)
{
    const int64_t threads = (int64_t)a.n * G;
    const int blocks = (int)((threads + 255) / 256);
    if (blocks == 0) return;
    hipLaunchKernelGGL(match_kernel<G>, dim3(blocks), dim3(256), 0, st, a.grid, a.pose, a.gates, a.sx, a.sy,
                       a.sz, a.n, a.plane, a.flags, a.sel, a.nn_idx, a.nn_d2, a.dbg);
}

void launch_match(const MatchArgs &a, int group, hipStream_t st)
{
    switch (group) {
        case 8: launch_g<8>(a, st); break;
        case 32: launch_g<32>(a, st); break;
        default: launch_g<16>(a, st); break;
    }
}

}  // namespace s2m
