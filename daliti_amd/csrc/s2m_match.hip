// s2m_match.hip -- rematch pass, part 1: exact 5 nearest map points of every scan point, first shell.
//
// Replaces, per scan point (eskf_lio/src/laserMapping.cpp:835-850): the body->world transform
// (:835-841) and ikdtree.Nearest_Search(point_world, 5, points_near, pointSearchSqDis) (:850;
// eskf_lio/include/ikd-Tree/ikd_Tree.cpp:425-461, 1061-1244, 1682-1709).  Output is Nearest_Points as
// sorted positions in the map's point array plus the ascending squared distances; the neighbour gate
// (:852-854) and esti_plane (:863) run in the thread-per-point kernel of s2m_reduce.hip.
//
// Two kernels, because measured cost is a long tail of far queries on top of the first-shell work:
//   match_rows<G> (here): G = 2 lanes per scan point scan the 3x3x3 cells around it as nine x-ROWS; if the
//                   5th-best distance is provably inside the cube the result is final; otherwise the point is
//                   appended to the far-point list with everything the second kernel needs (HardRec).
//   match_hard (s2m_match_far.hip): one wave (32 lanes in batched launches) per far point.
#include <algorithm>

#include "s2m_search.h"

namespace s2m {

// ---- first shell: 3x3x3 cells as nine x-rows, G lanes per scan point --------------------------------------
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
template <bool WIDE, int S = 1>
__device__ __forceinline__ void load_batch(const Grid &g, uint32_t i, uint32_t e, float4 (&p)[8], uint32_t j = 0)
{
    static_assert(8 * S <= kSentinelPoints, "sentinel block too short");
    const float4 *__restrict__ pts = g.pts;
    const uint32_t left = e > i + j ? e - i - j : 0u;  // slot u is real iff S u < left
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
__device__ __forceinline__ void consume_batch(const float4 (&p)[8], float wx, float wy, float wz, u64 (&t)[kK])
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
        key[u] = make_key(d, map_point_pos(p[u]));
    }
    insert_batch8(t, key);
}

constexpr int kRunSlots = 16;  // 8 rows x 2 segments besides the home row

// position in a lane's chain of runs: run k of nr, next batch at i, run end e (i = e = 0 once exhausted)
struct RunCursor {
    uint32_t k, i, e;
};


template <int G, bool WIDE, int NB>
__device__ __forceinline__ void match_rows_body(const MatchArgs &a, uint2 *__restrict__ runs)
{
    const Grid &g = a.grid;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int qi = tid / G;
    const int j = tid % G;
    if (qi >= a.n) return;  // group-uniform
    const Query q = make_query(g, a.pose, a.sx[qi], a.sy[qi], a.sz[qi]);
    // x extent of the neighbourhood inside the bricks in use (signed cell coordinates: >> floors, & 7 is the cell inside
    // its brick); segment A lies in brick bA, segment B (if any) in bA + 1
    const int x_lo = max(q.cx - 1, g.blo[0] * 8), x_hi = min(q.cx + 1, g.bhi[0] * 8 + 7);
    const bool xok = x_lo <= x_hi;
    const int bA = x_lo >> 3, bB = x_hi >> 3;
    const bool split = xok && bB != bA;
    const int ax1 = split ? bA * 8 + 7 : x_hi;  // last cell of segment A
    const int bx0 = bB * 8;                      // first cell of segment B
    // phase 1: brick ids of the nine rows (two per row where the row is split).  The rows cy-1..cy+1 x cz-1..cz+1 touch at
    // most 2 x 2 bricks in (y, z): four top entries per x-brick are loaded and the nine rows select among them (nine loads
    // per x-brick before: at saturation the kernel is bound by vector-memory instructions, scripts/ta_lines.hip).  The row
    // masks are not consulted: an empty row's prefix words are equal, i.e. an empty run.
    uint32_t idA[9], idB[9];
    int rowbit[9];
    {
        const int yb0 = (q.cy - 1) >> 3, yb1 = (q.cy + 1) >> 3, zb0 = (q.cz - 1) >> 3, zb1 = (q.cz + 1) >> 3;
        uint32_t tA[2][2] = {{0u, 0u}, {0u, 0u}}, tB[2][2] = {{0u, 0u}, {0u, 0u}};
#pragma unroll
        for (int zi = 0; zi < 2; ++zi)
#pragma unroll
            for (int yi = 0; yi < 2; ++yi) {
                const int yb = yi ? yb1 : yb0, zb = zi ? zb1 : zb0;
                const bool okb = xok && yb >= g.blo[1] && yb <= g.bhi[1] && zb >= g.blo[2] && zb <= g.bhi[2];
                const uint32_t toprow = top_row(g, yb, zb);
                if (okb) tA[zi][yi] = g.top[toprow | ((uint32_t)bA & g.tmx)].x;
                if (okb && split) tB[zi][yi] = g.top[toprow | ((uint32_t)bB & g.tmx)].x;
            }
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const int yy = q.cy + (r % 3) - 1, zz = q.cz + (r / 3) - 1;
            // (a row outside the bricks in use selects one of the zeros above)
            rowbit[r] = ((zz & 7) << 3) | (yy & 7);
            const bool ysel = (yy >> 3) != yb0, zsel = (zz >> 3) != zb0;
            const uint32_t a = zsel ? (ysel ? tA[1][1] : tA[1][0]) : (ysel ? tA[0][1] : tA[0][0]);
            const uint32_t b = zsel ? (ysel ? tB[1][1] : tB[1][0]) : (ysel ? tB[0][1] : tB[0][0]);
            idA[r] = a;
            idB[r] = b;
        }
    }
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
    // in steps of 8 G points
    auto push_run = [&](uint32_t s, uint32_t e) {
        if (s >= e) return;
        runs[nr * 256 + threadIdx.x] = make_uint2(s, e); ++nr;
    };
    // the lane's batches, NB at a time: all 8 * NB point loads of a trip are in flight together (an exhausted cursor
    // loads the sentinel block -- distance +inf, nothing is inserted -- so the trip has no branches around its loads)
    auto walk_runs = [&]() {
        if (nr == 0) return;
        RunCursor c;
        c.k = 0;
        { const uint2 r0 = runs[threadIdx.x]; c.i = r0.x; c.e = r0.y; }
        auto advance = [&]() {
            c.i += 8u * G;
            if (c.i >= c.e) {
                ++c.k;
                c.i = 0u; c.e = 0u;
                if (c.k < nr) { const uint2 rk = runs[c.k * 256 + threadIdx.x]; c.i = rk.x; c.e = rk.y; }
            }
        };
        while (c.k < nr) {
            float4 p[NB][8];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                load_batch<WIDE, G>(g, c.i, c.e, p[b], (uint32_t)j);
                advance();
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) consume_batch(p[b], q.wx, q.wy, q.wz, t);
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
            sA[r] = tb[la & 7]; eA[r] = tb[(ha & 7) + 1];
        }
        const int lb = max(xa, bx0), hb = min(xb, x_hi);  // piece inside segment B
        if (idB[r] && lb <= hb) {
            const uint32_t *tb = g.tab + (int64_t)(idB[r] - 1) * kBrickStride + (rowbit[r] << 3);
            sB[r] = tb[lb & 7]; eB[r] = tb[(hb & 7) + 1];
        }
    }
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        if (r == 4) continue;
        push_run(sA[r], eA[r]);
        push_run(sB[r], eB[r]);
    }
    walk_runs();
    merge_lists<G>(t, best);
    const bool found5 = !is_empty(best[kK - 1]);
    const float d5 = __uint_as_float((uint32_t)(best[kK - 1] >> 32));
    const bool done = found5 && d5 <= cube_bound2(g, q, 1);
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
    }
}

template <int G, bool WIDE, int NB>
__global__ __launch_bounds__(256, NB == 2 ? 4 : 3) void match_rows(MatchArgs a)
{
    __shared__ uint2 runs[kRunSlots * 256];  // [slot][thread]: conflict-free whatever the per-lane fill
    match_rows_body<G, WIDE, NB>(a, runs);
}

// the same inside the device-resident loop (s2m_loop.h): the first kernel of a scan runs at the pose in its arguments
// and has workgroup 0 bring the init record over; a later one runs only if the device decided to search again, at the
// pose the last update left
template <int G, bool WIDE, int NB>
__global__ __launch_bounds__(256, NB == 2 ? 4 : 3) void match_rows_loop(MatchArgs a)
{
    __shared__ uint2 runs[kRunSlots * 256];
    if (a.loop.init) {
        if (blockIdx.x == 0) loop_copy_init(a.loop);
    } else {
        int rematch_now = 0;
        if (!loop_enter(a.loop, a.pose, rematch_now) || !rematch_now) return;
    }
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
    m.hard_count = b.hard_count; m.qheads = b.qheads;
    return m;
}
// K scans, one grid: blockIdx.y = scan.  The scans that do not search in this pass leave at once.
template <int G, bool WIDE, int NB>
__global__ __launch_bounds__(256, NB == 2 ? 4 : 3) void match_rows_batch(BatchArgs b)
{
    __shared__ uint2 runs[kRunSlots * 256];
    const ScanDesc &d = b.d[blockIdx.y];
    if (!d.active) return;
    MatchArgs a = batch_match_args(b, blockIdx.y);
    if (d.loop.state) {  // device-resident loop: the device knows whether this scan searches in this pass
        if (d.loop.init) {
            if (blockIdx.x == 0) loop_copy_init(d.loop);
        } else {
            int rematch_now = 0;
            if (!loop_enter(d.loop, a.pose, rematch_now) || !rematch_now) return;
        }
    } else if (!d.rematch) {
        return;
    }
    match_rows_body<G, WIDE, NB>(a, runs);
}

// group bit 0x10000: 64-bit point addresses on request (S2M_WIDE_ADDR=1: the tests cover that path on a small map);
// bits 8..11: point batches per trip (0 = default); bit 0x40000: the first-shell kernel only (the host bets that it
// resolves every point; s2m_engine.cpp, run_pass)
void launch_match(const MatchArgs &a, int group, hipStream_t st)
{
    if (a.n <= 0) return;
    const bool wide = (a.grid.sent_off == 0 && a.grid.m != 0) || (group & 0x10000);
    const int nb = (group >> 8) & 0xf;
    const int blocks = (int)(((int64_t)a.n * 2 + 255) / 256);
    if (a.loop.state) {
        if (nb == 2) {
            if (!wide) hipLaunchKernelGGL((match_rows_loop<2, false, 2>), dim3(blocks), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((match_rows_loop<2, true, 2>), dim3(blocks), dim3(256), 0, st, a);
        } else {
            if (!wide) hipLaunchKernelGGL((match_rows_loop<2, false, 3>), dim3(blocks), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((match_rows_loop<2, true, 3>), dim3(blocks), dim3(256), 0, st, a);
        }
    } else if (nb == 2) {
        if (!wide) hipLaunchKernelGGL((match_rows<2, false, 2>), dim3(blocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((match_rows<2, true, 2>), dim3(blocks), dim3(256), 0, st, a);
    } else {  // default: three batches (24 point loads) per trip
        if (!wide) hipLaunchKernelGGL((match_rows<2, false, 3>), dim3(blocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((match_rows<2, true, 3>), dim3(blocks), dim3(256), 0, st, a);
    }
    if (!(group & 0x40000)) launch_far_points(a, wide, st);
}

// search kernels of one batched pass (two point batches per trip: the chip is shared by K scans)
void launch_match_batch(const BatchArgs &b, hipStream_t st)
{
    if (b.n_max <= 0 || b.k <= 0) return;
    const bool wide = b.grid.sent_off == 0 && b.grid.m != 0;
    const dim3 grid((unsigned)(((int64_t)b.n_max * 2 + 255) / 256), (unsigned)b.k);
    if (wide) hipLaunchKernelGGL((match_rows_batch<2, true, 2>), grid, dim3(256), 0, st, b);
    else hipLaunchKernelGGL((match_rows_batch<2, false, 2>), grid, dim3(256), 0, st, b);
    launch_far_points_batch(b, wide, st);
}

}  // namespace s2m
