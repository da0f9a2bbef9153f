// s2m_match_far.hip -- rematch pass, part 2: the scan points whose 5th neighbour is not provably inside the first
// shell (ikdtree.Nearest_Search, eskf_lio/include/ikd-Tree/ikd_Tree.cpp:425-461, 1061-1244, for the queries that need
// more than 3x3x3 cells), and the completion of the lists beyond the gate (s2m_complete_neighbors: the reference's
// search is unbounded, max_dist = INFINITY, ikd_Tree.cpp:425).
//
//   match_hard    : one wave per far point.  The x-rows that can hold a point within the current radius
//                   (the first shell's 5th distance, else a growing band) are found either directly -- the
//                   7x7 rows around the home row while the radius is within three cells -- or from the
//                   row masks of the surrounding bricks; their cells are expanded into an LDS cell list by
//                   a DPP prefix sum and dealt to the 64 lanes, so the point loads of all rows are in
//                   flight together.  Stops once the bound passes the d2 <= 5 gate (laserMapping.cpp:853).
//   match_hard32_batch : the same search with 32 lanes per point, two points per wave, for the batched launches.
#include <algorithm>
#include <cstdlib>

#include "s2m_search.h"

namespace s2m {

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

constexpr int kHardChunks = 4;  // chunks of 64 listed cells whose table words are fetched per round trip
#ifndef S2M_HARD_BAND_EMPTY
#define S2M_HARD_BAND_EMPTY 2.8f  // first band (cells) of a far point whose first shell held nothing
#endif
constexpr int kHardOcc = 4;  // waves per SIMD the far-point kernels are compiled for (<= 128 VGPRs) = resident waves / 1024
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
        // the point's record: query, index, and the radius when the first shell found five (then one round is exact)
        const uint4 *rp = reinterpret_cast<const uint4 *>(h < c0 ? a.hard_rec + h : a.hard_rec + (a.hard_off1 + (h - c0)));
        const uint4 r0 = rp[0], r1 = rp[1];
        const int qi = (int)r0.w;
        const Query q = query_at(g, __uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r0.z));
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
        // the completion of lists beyond the gate knows that an earlier search found nothing closer than its radius
        if (FAR && a.band0 > 0.0f) band = a.band0;
        band = fminf(band, sqrtf(a.gates.knn_d2_gate * 1.0001f));  // no first band beyond the gate either (coarse grids)
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
            for (int jb = 0; jb < nc; jb += 64 * kHardChunks) {  // wave-uniform trip count
                uint32_t rs[kHardChunks], re[kHardChunks];
#pragma unroll
                for (int u = 0; u < kHardChunks; ++u) {
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
                for (int u = 0; u < kHardChunks; ++u) {
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
            if (no > 0 && no <= 32) {  // wave-uniform
                const int P = no <= 16 ? 4 : 2;
                const int piece = lane / no;
                if (piece < P) {
                    const uint2 run = cells[lane - piece * no];
                    const uint32_t s0 = run.x + 8u * (uint32_t)piece;
                    const uint32_t e0 = (piece == P - 1) ? run.y : min(s0 + 8u, run.y);
                    if (s0 < e0) scan_points<kBatch, WIDE>(g, s0, e0, q.wx, q.wy, q.wz, t);
                }
            } else
            for (int jb = 0; jb < no; jb += 64) {  // wave-uniform trip count
                const int j = jb + lane;
                if (j < no) {
                    const uint2 run = cells[j];
                    scan_points<kBatch, WIDE>(g, run.x, run.y, q.wx, q.wy, q.wz, t);
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
                if (lane < 49 && (yy >> 3) >= g.blo[1] && (yy >> 3) <= g.bhi[1] && (zz >> 3) >= g.blo[2] && (zz >> 3) <= g.bhi[2]) {
                    const float gy = ndy > 0 ? (float)ndy - q.fry : (ndy < 0 ? q.fry - (float)(ndy + 1) : 0.0f);
                    const float gz = ndz > 0 ? (float)ndz - q.frz : (ndz < 0 ? q.frz - (float)(ndz + 1) : 0.0f);
                    const float ay = fmaxf(gy - g.slop, 0.0f), az = fmaxf(gz - g.slop, 0.0f);
                    const float b2 = (ay * ay + az * az) * (g.c * g.c) * 0.99999f;
                    if (b2 <= r2) {
                        const float reach = sqrtf(fmaxf(r2 - b2, 0.0f)) * g.inv_c * 1.00001f + g.slop;
                        // (relative to the home cell: exact however large the cell coordinates are)
                        const int xa = max(q.cx + (int)floorf(q.frx - reach), g.blo[0] * 8), xb = min(q.cx + (int)floorf(q.frx + reach), g.bhi[0] * 8 + 7);
                        nrow = ((zz & 7) << 3) | (yy & 7);
                        const uint32_t toprow = top_row(g, yy >> 3, zz >> 3);
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const int bx = (xa >> 3) + k;
                            if (xa > xb || bx > (xb >> 3)) continue;
                            const uint4 te = g.top[toprow | ((uint32_t)bx & g.tmx)];
                            const uint32_t mword = (nrow & 32) ? te.w : te.z;
                            if (te.x == 0 || ((mword >> (nrow & 31)) & 1u) == 0) continue;
                            nid[k] = te.x;
                            nxa[k] = max(xa, bx * 8);
                            ncl[k] = min(xb, bx * 8 + 7) - nxa[k] + 1;
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) append_cells(nid[k], nrow, nxa[k], ncl[k]);  // <= 2 x 49 x 7 cells
            } else {
            // FAR: the neighbourhood clipped to the grid (empty when the point lies further outside than the radius) -- and to
            // the rings THIS round's radius needs: the launch's own radius may be that of the whole map (one launch finishes
            // every list, s2m_engine_map.cpp), a round must not pay for more bricks than its band reaches
            const int NBr = FAR ? min(NB, max(1, (int)fminf(ceilf(sqrtf(r2) * g.inv_c * 0.125f + 1e-3f), 1048576.0f))) : NB;
            const int flx = max(hbx - NBr, g.blo[0]), fly = max(hby - NBr, g.blo[1]), flz = max(hbz - NBr, g.blo[2]);
            const int fsx = FAR ? max(min(hbx + NBr, g.bhi[0]) - flx + 1, 0) : 0;
            const int fsy = FAR ? max(min(hby + NBr, g.bhi[1]) - fly + 1, 0) : 0;
            const int fsz = FAR ? max(min(hbz + NBr, g.bhi[2]) - flz + 1, 0) : 0;
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
                    if (brick_in_bounds(g, bx, by, bz)) {
                        const uint4 te = g.top[top_slot(g, bx, by, bz)];
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
                            const int yy = oby * 8 + (rowbit & 7), zz = obz * 8 + (rowbit >> 3);
                            // lower bound of the (y,z) distance from the query to this row, in cells
                            const int dy = yy - q.cy, dz = zz - q.cz;
                            const float gy = dy > 0 ? (float)dy - q.fry : (dy < 0 ? q.fry - (float)(dy + 1) : 0.0f);
                            const float gz = dz > 0 ? (float)dz - q.frz : (dz < 0 ? q.frz - (float)(dz + 1) : 0.0f);
                            const float ay = fmaxf(gy - g.slop, 0.0f), az = fmaxf(gz - g.slop, 0.0f);
                            const float b2 = (ay * ay + az * az) * (g.c * g.c) * 0.99999f;  // metres^2
                            if (b2 <= r2) {
                                // x cells the radius can reach in this row: |x - qx| <= sqrt(r2 - b2)
                                const float reach = sqrtf(fmaxf(r2 - b2, 0.0f)) * g.inv_c * 1.00001f + g.slop;
                                xa = max(q.cx + (int)floorf(q.frx - reach), obx * 8);
                                const int xb = min(q.cx + (int)floorf(q.frx + reach), obx * 8 + 7);
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
            merge_lists<G>(t, best);
            // (the completion may ask for the NEAREST neighbour only: the list is settled once its short_k-th entry is)
            const u64 kth = (FAR && a.short_k == 1) ? best[0] : best[kK - 1];
            const bool found5 = !is_empty(kth);
            const float d5 = __uint_as_float((uint32_t)(kth >> 32));
            if (have_tau) break;  // every point within tau was visited: exact
            // band mode: rows with bound <= band^2 were scanned over their whole reach of this band only,
            // so restart the private lists when the radius changes (rows are rescanned with the new reach)
            if (found5 && d5 <= band * band) break;          // five found inside the fully scanned band
            if (band * band > a.gates.knn_d2_gate) {         // beyond the gate: result is "not five within it"
                if (FAR && a.open_count && lane == 0) atomicAdd(a.open_count, 1u);  // (the completion counts the lists it leaves open)
                break;
            }
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
    float tau = 0.0f, band = 0.0f;
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
            for (int jb = 0; jb < nc; jb += GL * kHardChunks) {
                uint32_t rs[kHardChunks], re[kHardChunks];
#pragma unroll
                for (int u = 0; u < kHardChunks; ++u) {
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
                for (int u = 0; u < kHardChunks; ++u) {
                    const bool holds = rs[u] < re[u];
                    const uint32_t m = half_ballot(holds, half);
                    if (holds) cells[no + __popc(m & ((1u << lane) - 1u))] = make_uint2(rs[u], re[u]);
                    no += __popc(m);
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            }
            if (no > 0 && no <= 16) {  // idle lanes take the later pieces of the runs (see match_hard_body)
                const int P = no <= 8 ? 4 : 2;
                const int piece = lane / no;
                if (piece < P) {
                    const uint2 run = cells[lane - piece * no];
                    const uint32_t s0 = run.x + 8u * (uint32_t)piece;
                    const uint32_t e0 = (piece == P - 1) ? run.y : min(s0 + 8u, run.y);
                    if (s0 < e0) scan_points<kBatch, WIDE>(g, s0, e0, q.wx, q.wy, q.wz, t);
                }
            } else
            for (int jb = 0; jb < no; jb += GL) {
                const int j = jb + lane;
                if (j < no) {
                    const uint2 run = cells[j];
                    scan_points<kBatch, WIDE>(g, run.x, run.y, q.wx, q.wy, q.wz, t);
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
                if (rl < 49 && (yy >> 3) >= g.blo[1] && (yy >> 3) <= g.bhi[1] && (zz >> 3) >= g.blo[2] && (zz >> 3) <= g.bhi[2]) {
                    const float gy = ndy > 0 ? (float)ndy - q.fry : (ndy < 0 ? q.fry - (float)(ndy + 1) : 0.0f);
                    const float gz = ndz > 0 ? (float)ndz - q.frz : (ndz < 0 ? q.frz - (float)(ndz + 1) : 0.0f);
                    const float ay = fmaxf(gy - g.slop, 0.0f), az = fmaxf(gz - g.slop, 0.0f);
                    const float b2 = (ay * ay + az * az) * (g.c * g.c) * 0.99999f;
                    if (b2 <= r2) {
                        const float reach = sqrtf(fmaxf(r2 - b2, 0.0f)) * g.inv_c * 1.00001f + g.slop;
                        // (relative to the home cell: exact however large the cell coordinates are)
                        const int xa = max(q.cx + (int)floorf(q.frx - reach), g.blo[0] * 8), xb = min(q.cx + (int)floorf(q.frx + reach), g.bhi[0] * 8 + 7);
                        nrow = ((zz & 7) << 3) | (yy & 7);
                        const uint32_t toprow = top_row(g, yy >> 3, zz >> 3);
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const int bx = (xa >> 3) + k;
                            if (xa > xb || bx > (xb >> 3)) continue;
                            const uint4 te = g.top[toprow | ((uint32_t)bx & g.tmx)];
                            const uint32_t mword = (nrow & 32) ? te.w : te.z;
                            if (te.x == 0 || ((mword >> (nrow & 31)) & 1u) == 0) continue;
                            nid[k] = te.x;
                            nxa[k] = max(xa, bx * 8);
                            ncl[k] = min(xb, bx * 8 + 7) - nxa[k] + 1;
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
                    if (brick_in_bounds(g, bx, by, bz)) {
                        const uint4 te = g.top[top_slot(g, bx, by, bz)];
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
                            const int yy = oby * 8 + (rowbit & 7), zz = obz * 8 + (rowbit >> 3);
                            const int dy = yy - q.cy, dz = zz - q.cz;
                            const float gy = dy > 0 ? (float)dy - q.fry : (dy < 0 ? q.fry - (float)(dy + 1) : 0.0f);
                            const float gz = dz > 0 ? (float)dz - q.frz : (dz < 0 ? q.frz - (float)(dz + 1) : 0.0f);
                            const float ay = fmaxf(gy - g.slop, 0.0f), az = fmaxf(gz - g.slop, 0.0f);
                            const float b2 = (ay * ay + az * az) * (g.c * g.c) * 0.99999f;
                            if (b2 <= r2) {
                                const float reach = sqrtf(fmaxf(r2 - b2, 0.0f)) * g.inv_c * 1.00001f + g.slop;
                                xa = max(q.cx + (int)floorf(q.frx - reach), obx * 8);
                                const int xb = min(q.cx + (int)floorf(q.frx + reach), obx * 8 + 7);
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

template <bool WIDE, bool FAR = false>
__global__ __launch_bounds__(256, kHardOcc) void match_hard(MatchArgs a)
{
    __shared__ uint2 cells_all[4][kMaxCells];  // one cell list per wave of the workgroup
    // device-resident loop: a launch of an iteration in which the device did not search finds an empty list anyway (every
    // reduce launch zeroes the counters); leaving here saves the waves the look
    if (!FAR && a.loop.state && (!loop_launch_due(a.loop) || !a.loop.state->rematch_now)) return;
    match_hard_body<WIDE, FAR>(a, cells_all[threadIdx.x >> 6], [&](uint32_t, int32_t *&idx, float *&d2) {
        idx = a.nn_idx;
        d2 = a.nn_d2;
    });
}

template <bool WIDE>
__global__ __launch_bounds__(256, kHardOcc) void match_hard32(MatchArgs a)
{
    __shared__ uint2 cells_all[8][kHalfCells];  // one cell list per half-wave of the workgroup
    match_hard32_body<WIDE>(a, cells_all[threadIdx.x >> 5], [&](uint32_t, int32_t *&idx, float *&d2) {
        idx = a.nn_idx;
        d2 = a.nn_d2;
    });
}

template <bool WIDE>
__global__ __launch_bounds__(256, kHardOcc) void match_hard32_batch(BatchArgs b)
{
    __shared__ uint2 cells_all[8][kHalfCells];
    MatchArgs a;
    a.grid = b.grid; a.gates = b.gates;
    a.sx = a.sy = a.sz = nullptr; a.n = b.n_max;
    a.nn_idx = nullptr; a.nn_d2 = nullptr;
    a.hard_rec = b.hard_rec; a.hard_off1 = b.hard_off1; a.slot = 0;
    a.hard_count = b.hard_count; a.qheads = b.qheads;
    match_hard32_body<WIDE>(a, cells_all[threadIdx.x >> 5], [&](uint32_t slot, int32_t *&idx, float *&d2) {
        idx = b.d[slot].nn_idx;   // the two halves of a wave may serve different scans: a per-lane look-up in the table
        d2 = b.d[slot].nn_d2;
    });
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
    if (i == 0 && a.open_count) *a.open_count = 0u;  // (the far-point launch behind this one counts the lists it leaves open)
    bool want = false;
    uint32_t far_bits = 0u;
    HardRec rec = {0.f, 0.f, 0.f, 0u, 0.f, 0u, 0u, 0u};
    const int last = a.short_k - 1;  // the entry that must be proven: the 5th for a complete list, the first for the nearest neighbour only
    if (i < a.n && !(a.nn_idx[(int64_t)i * kK + last] >= 0 && a.nn_d2[(int64_t)i * kK + last] <= a.gates.knn_d2_gate)) {
        body_to_world(a.pose, a.sx[i], a.sy[i], a.sz[i], rec.wx, rec.wy, rec.wz);
        rec.qi = (uint32_t)i;
        // squared distance to the centre of the grid: the host derives from its maximum the radius at which every
        // map point has been seen (a query may lie far outside the grid).  A non-finite query can have no neighbours:
        // it is left as it is.
        const Grid &g = a.grid;
        // (centre of the box of bricks in use)
        const float cx = g.ox + 4.0f * (float)(g.blo[0] + g.bhi[0] + 1) * g.c, cy = g.oy + 4.0f * (float)(g.blo[1] + g.bhi[1] + 1) * g.c,
                    cz = g.oz + 4.0f * (float)(g.blo[2] + g.bhi[2] + 1) * g.c;
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

__global__ __launch_bounds__(256) void far_reset_kernel(uint32_t *__restrict__ hard_count, uint32_t *__restrict__ qheads, int words)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < words) hard_count[i] = 0u;
    if (i < kQueueWords) qheads[i] = 0u;
}
void launch_far_reset(uint32_t *hard_count, uint32_t *qheads, hipStream_t st, bool keep_open)
{
    hipLaunchKernelGGL(far_reset_kernel, dim3((kQueueWords + 255) / 256), dim3(256), 0, st, hard_count, qheads, keep_open ? 3 : 4);
}

void launch_match_hard_only(const MatchArgs &a, hipStream_t st)
{
    if (a.n <= 0) return;
    const bool wide = a.grid.sent_off == 0 && a.grid.m != 0;
    const int64_t groups = std::min<int64_t>(a.n, 1024 * kHardOcc);
    const int blocks = (int)((groups * 64 + 255) / 256);
    if (!wide) hipLaunchKernelGGL((match_hard<false, true>), dim3(blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((match_hard<true, true>), dim3(blocks), dim3(256), 0, st, a);
}

// Lanes per far point.  Measured (NOTEBOOK.md, round 3): with ONE scan in flight a wave per point is faster (the two
// halves of a wave share one instruction stream, so their load chains run one after the other and a single scan's
// launch is as long as its slowest points); with K scans in one grid the half-wave form wins (the list is long, only
// throughput counts, and 8,192 points in flight hide more of each other's waits).
void launch_far_points(const MatchArgs &a, bool wide, hipStream_t st)
{
    if (a.n <= 0) return;
    // test hook (S2M_HARD_LANES=32): the half-wave form for single scans too, so that the exactness stress tests reach it
    static const bool half = std::getenv("S2M_HARD_LANES") && std::atoi(std::getenv("S2M_HARD_LANES")) == 32;
    if (half && a.qheads) {
        const int64_t halves = std::min<int64_t>(a.n, 1024 * kHardOcc * 2);
        const int blocks = (int)((halves * 32 + 255) / 256);
        if (!wide) hipLaunchKernelGGL(match_hard32<false>, dim3(blocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(match_hard32<true>, dim3(blocks), dim3(256), 0, st, a);
        return;
    }
    // as many waves as stay resident together (4 per SIMD, 4,096 on the chip); the rest of the list is pulled through
    // the queue heads
    int64_t groups = std::min<int64_t>(a.n, a.qheads ? 1024 * kHardOcc : 8192);
    if (a.far_waves > 0 && a.qheads) groups = std::min<int64_t>(groups, a.far_waves);
    const int blocks = (int)((groups * 64 + 255) / 256);
    if (!wide) hipLaunchKernelGGL(match_hard<false>, dim3(blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(match_hard<true>, dim3(blocks), dim3(256), 0, st, a);
}

void launch_far_points_batch(const BatchArgs &b, bool wide, hipStream_t st)
{
    const int blocks = 1024 * kHardOcc * 64 / 256;  // the resident waves; the rest of the list comes through the heads
    if (!wide) hipLaunchKernelGGL(match_hard32_batch<false>, dim3(blocks), dim3(256), 0, st, b);
    else hipLaunchKernelGGL(match_hard32_batch<true>, dim3(blocks), dim3(256), 0, st, b);
}

void launch_match_far_points(const MatchArgs &a, int group, hipStream_t st)
{
    launch_far_points(a, (a.grid.sent_off == 0 && a.grid.m != 0) || (group & 0x10000), st);
}

}  // namespace s2m
