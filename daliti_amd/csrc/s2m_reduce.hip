// s2m_reduce.hip -- [gate + plane fit,] residual, gates, Jacobian row and the fused H^T H / H^T z
// contraction.
//
// On a rematch pass (FIT) each lane first applies the neighbour gate (laserMapping.cpp:852-854) to
// the fresh Nearest_Points of its scan point and fits the plane (esti_plane, :863;
// common_lib.h:267-299) from the five neighbours gathered by sorted position; the plane only depends on the
// (world-frame, constant) neighbours, so it is cached and the reuse passes skip the fit -- the
// reference re-fits the identical plane every iteration.
//
// Replaces, per scan point, eskf_lio/src/laserMapping.cpp:857-881 (point-to-plane residual,
// s-gate, sticky selection), :887-896 (effective set, total_residual), :948-979 (Jacobian row,
// meas_vec) and the m x 12 GEMMs of :1015 / :1019-1032, of which only Hsub^T*Hsub (12x12) and
// Hsub^T*meas_vec (12) are needed:  K*z = K_1[:, :12]*(H^T z),  K*H = K_1[:, :12]*(H^T H).
// No per-point H is written; the reduction is a halving shuffle butterfly per wave -> LDS -> one
// fp64 partial row per workgroup (write-through) -> a fixed-order final sum by the last workgroup
// to arrive, so the block is deterministic for a given launch shape and needs no second launch.
//
// One lane per scan point, SoA loads (x[], y[], z[], float4 plane) are fully coalesced.
// Compiled with -ffp-contract=off; the row arithmetic follows oracle/s2m_oracle.c:jac_row.
#include <algorithm>
#include <cmath>

#include "s2m_device.h"
#include "s2m_kernels.h"
#include "s2m_loop.h"
#include "s2m_plane.h"
#include "s2m_point.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "s2m_reduce.hip is written for gfx950: v_permlane32_swap / v_permlane16_swap, f64 MFMA, 79 KB of LDS per workgroup"
#endif

namespace s2m {

// Term layout of a partial row for NC Jacobian columns (NC = 6 without extrinsic estimation, 12
// with): [0, NC(NC+1)/2) upper triangle of H^T H (row-major, r <= c), then NC terms of H^T z, then
// total_residual, the effective count and (rematch passes) the number of neighbour lists that did not fill inside the
// gate; padded to a multiple of 32 (32 or 96 slots).
template <int NC>
struct Terms {
    static constexpr int kTri = NC * (NC + 1) / 2;
    static constexpr int kHtz = kTri;
    static constexpr int kRes = kTri + NC;
    static constexpr int kCnt = kTri + NC + 1;
    static constexpr int kShort = kTri + NC + 2;            // rematch pass: lists that did not fill inside the gate
    static constexpr int kUsed = kTri + NC + 3;             // 30 or 93
    static constexpr int kSlots = ((kUsed + 31) / 32) * 32;  // 32 or 96
    __host__ __device__ static constexpr int tri(int r, int c) { return r * NC - (r * (r - 1)) / 2 + (c - r); }
};

// Sum of 32 values per lane across the 64 lanes of a wave with a halving butterfly: at each of
// the five halving steps a lane keeps one half of its values and hands the other half to its
// partner, so 32 + 1 exchanges replace 32 x 6.  On return v[0] of lane l is the wave-wide sum of
// value (l >> 1) & 31 (both lanes of a pair hold it).  Fixed order, hence deterministic.
// The two widest steps (partner lane ^ 32 and lane ^ 16: 24 of the 32 exchanges) with gfx950's v_permlane32_swap /
// v_permlane16_swap: one VALU instruction exchanges the upper half (the odd 16-lane rows) of one register with the
// lower half (the even rows) of another, which IS the keep / send pattern of a halving step -- no selects, no trip
// through the LDS crossbar (ds_bpermute), and the sums are the same two operands as before (bit-identical results).
// Measured on the last workgroup of the reuse-pass kernel: butterfly + LDS stage 2.4 us with shuffles.
template <int M>
__device__ __forceinline__ double swap_add(double x, double y)
{
    const uint32_t xl = (uint32_t)__double_as_longlong(x), xh = (uint32_t)((unsigned long long)__double_as_longlong(x) >> 32);
    const uint32_t yl = (uint32_t)__double_as_longlong(y), yh = (uint32_t)((unsigned long long)__double_as_longlong(y) >> 32);
    uint32_t al, bl, ah, bh;
    if (M == 32) {
        const auto l = __builtin_amdgcn_permlane32_swap(xl, yl, false, false);
        const auto h = __builtin_amdgcn_permlane32_swap(xh, yh, false, false);
        al = l[0]; bl = l[1]; ah = h[0]; bh = h[1];
    } else {
        const auto l = __builtin_amdgcn_permlane16_swap(xl, yl, false, false);
        const auto h = __builtin_amdgcn_permlane16_swap(xh, yh, false, false);
        al = l[0]; bl = l[1]; ah = h[0]; bh = h[1];
    }
    const double a = __longlong_as_double((long long)(((unsigned long long)ah << 32) | al));
    const double b = __longlong_as_double((long long)(((unsigned long long)bh << 32) | bl));
    return a + b;  // lanes with bit M clear: x[l] + x[l ^ M]; with bit M set: y[l ^ M] + y[l]
}
template <int HALF>
__device__ __forceinline__ void halve_step_swap(double (&v)[32])
{
#pragma unroll
    for (int i = 0; i < HALF; ++i) v[i] = swap_add<HALF * 2>(v[i], v[i + HALF]);
}

// The narrow steps (partner lane ^ 8, ^ 4, ^ 2, ^ 1) stay inside a 16-lane row, where DPP moves reach: row_ror:8 is
// lane ^ 8; lane ^ 4 is row_ror:4 for the banks whose lanes have bit 2 set and row_ror:12 for the others (bank masks);
// quad_perm covers ^ 2 and ^ 1.  Same operands and sums as the shuffle form.
template <int M>
__device__ __forceinline__ uint32_t dpp_xor(uint32_t v)
{
    if (M == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);          // row_ror:8
    if (M == 4) {
        const int a = __builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xa, false);                      // row_ror:4 -> banks 1, 3: lane - 4
        return (uint32_t)__builtin_amdgcn_update_dpp(a, (int)v, 0x12c, 0xf, 0x5, false);                    // row_ror:12 -> banks 0, 2: lane + 4
    }
    if (M == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4e, 0xf, 0xf, false);            // quad_perm [2,3,0,1]
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1, 0xf, 0xf, false);                          // quad_perm [1,0,3,2]
}
template <int M>
__device__ __forceinline__ double dpp_xor(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = dpp_xor<M>((uint32_t)b), hi = dpp_xor<M>((uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int HALF>
__device__ __forceinline__ void halve_step_dpp(double (&v)[32], int lane)
{
    constexpr int m = HALF * 2;
    const bool hi = (lane & m) != 0;
#pragma unroll
    for (int i = 0; i < HALF; ++i) {
        const double send = hi ? v[i] : v[i + HALF];
        const double keep = hi ? v[i + HALF] : v[i];
        v[i] = keep + dpp_xor<m>(send);
    }
}

__device__ __forceinline__ void wave_sum32(double (&v)[32], int lane)
{
    halve_step_swap<16>(v);
    halve_step_swap<8>(v);
    halve_step_dpp<4>(v, lane);
    halve_step_dpp<2>(v, lane);
    halve_step_dpp<1>(v, lane);
    v[0] += dpp_xor<1>(v[0]);
}

// compile-time (row, column) of upper-triangle slot t
template <int NC>
__host__ __device__ constexpr int tri_row(int t)
{
    int r = 0;
    for (int rr = 0; rr < NC; ++rr)
        if (t >= Terms<NC>::tri(rr, rr)) r = rr;
    return r;
}

// Hand-off of the 160-double block to the spinning host thread.  A system-scope release fence (or release
// store) makes the compiler write back the whole L2 first (buffer_wbl2) -- microseconds, with the per-point
// outputs of this very launch dirty in it -- although only these 160 stores have to be visible.  So: the
// payload goes out as write-through system-scope stores, every storing wave drains them (vmcnt(0): the
// fabric has accepted them), the workgroup meets, and the flag follows as one more such store on the same
// ordered path to the pinned host page.
__device__ __forceinline__ void publish_store(double *dst, double v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(dst), (unsigned long long)__double_as_longlong(v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void publish_flag(unsigned long long *flag, unsigned long long seq)  // whole workgroup
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// an empty asm that "uses" a loaded value: the load has to be issued before this point
template <class T>
__device__ __forceinline__ void pin(T &v)
{
    asm volatile("" : "+v"(v));
}


// the workgroup's shared memory, declared once per kernel so that the two per-pass forms of a device-loop launch
// (rematch / reuse, chosen at run time by a uniform branch) use the same storage
template <bool EXT>
struct ReduceShared {
    using T = Terms<EXT ? 12 : 6>;
    static constexpr int kLanes = kRedBlock / T::kSlots;  // lanes available per term in the final sum: 16 (NC = 6) or 5 (NC = 12)
    static constexpr int kTree = kLanes >= 16 ? 16 : 4;   // ... of which a power of two takes part
    double red[kRedBlock / 64][T::kSlots];
    double part[kTree][T::kSlots];
    double tot[T::kSlots];
    double stage[EXT ? kRedBlock / 64 : 1][EXT ? 64 * 17 : 1];  // MFMA operand staging; rows padded to 17 doubles: conflict-free column reads
    LoopScratch loop;
    uint32_t last;
};

// system-scope store of a 32-bit word into the pinned record
__device__ __forceinline__ void publish_word(int32_t *dst, int32_t v)
{
    __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// bx / nblk: this workgroup's index among the workgroups of ITS scan and their number (blockIdx.x / gridDim.x for one
// scan per launch; in a batched launch the grid is sized for the largest scan).  FIT: rematch pass (gate + plane fit
// first); LOOP: device-resident loop -- the last workgroup goes on with the Kalman update and the judgement (s2m_loop.h)
template <bool EXT, bool FIT, bool LOOP>
__device__ __forceinline__ void reduce_body(const ReduceArgs &a, ReduceShared<EXT> &sh, const uint32_t bx, const uint32_t nblk)
{
    constexpr int NC = EXT ? 12 : 6;
    using T = Terms<NC>;
    auto &red = sh.red;
    LoopScratch &loop_scratch = sh.loop;
    const int i = (int)bx * kRedBlock + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    bool eff = false, short_list = false;
    double h[12], z = 0.0, absr = 0.0;
#pragma unroll
    for (int k = 0; k < 12; ++k) h[k] = 0.0;
    if (i < a.n) {
        // Everything a lane may need is requested up front, in one round trip: left to itself the compiler
        // sinks each load into the branch that uses it (selected? -> flags -> plane and coordinates), i.e.
        // three dependent trips of ~0.7 us each in a kernel whose whole per-point phase is 3.5 us.
        float bx = a.sx[i], by = a.sy[i], bz = a.sz[i];
        uint8_t sel, fl;
        float4 pl;
        if (FIT) {
            // neighbour gate: five neighbours and d2[4] <= 5 (:852-854)
            int32_t ni[kK];
#pragma unroll
            for (int k = 0; k < kK; ++k) ni[k] = a.nn_idx[(int64_t)i * kK + k];
            float d4 = a.nn_d2[(int64_t)i * kK + (kK - 1)];
            pin(bx); pin(by); pin(bz); pin(d4);  // all issued above; pinned only now (a pin waits for its value)
#pragma unroll
            for (int k = 0; k < kK; ++k) pin(ni[k]);
            const bool gate = (ni[kK - 1] >= 0) && !(d4 > a.gates.knn_d2_gate);
            bool plane_ok = false;
            pl = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gate) {
                float nx[kK], ny[kK], nz[kK];
#pragma unroll
                for (int k = 0; k < kK; ++k) {
                    // neighbours are sorted positions: the five of a query sit in the same or adjacent cells of
                    // the cell-ordered array, i.e. in a few cache lines (round 2 gathered from a copy in caller
                    // order: five random lines per point, 2 x 12.6 MB fetched per pass to read 5.2 MB)
                    const float4 p = a.pts[ni[k]];
                    nx[k] = p.x; ny[k] = p.y; nz[k] = map_point_z(p);
                }
                plane_ok = fit_plane(nx, ny, nz, a.gates.plane_thr, pl);
            }
            short_list = !gate;  // the unbounded search of the reference would go on for this point (s2m_complete_neighbors)
            sel = gate ? 1 : 0;  // point_selected_surf after the gate
            fl = (uint8_t)((gate ? kFlagGate : 0) | (plane_ok ? kFlagPlane : 0));
            a.plane[i] = pl;
            a.flags[i] = fl;
        } else {
            uint32_t s32 = a.sel[i], f32 = a.flags[i];
            pl = a.plane[i];
            pin(bx); pin(by); pin(bz);  // all issued above; pinned only now (a pin waits for its value)
            pin(s32); pin(f32); pin(pl.x); pin(pl.y); pin(pl.z); pin(pl.w);
            sel = (uint8_t)s32;
            fl = (uint8_t)f32;
        }
        if (sel) {
            uint8_t sel_new = 0;  // sticky: only fit-ok + s-gate re-selects (:862,:873)
            if (fl & kFlagPlane) {
                bool keep = false;
                const float pd2 = point_residual(a.pose, a.gates, bx, by, bz, pl, keep, eff);  // :866-889 (s2m_point.h)
                a.pd2[i] = pd2;
                if (keep) sel_new = 1;
                if (eff) {
                    jac_row<EXT>(a.pose, bx, by, bz, pl, pd2, h, z);
                    absr = fabs((double)pd2);
                }
            }
            a.sel[i] = sel_new;
        } else if (FIT) {
            a.sel[i] = 0;
        }
        a.eff[i] = eff ? 1 : 0;
    }
    // ineffective lanes carry h = 0, z = 0 and contribute exact zeros.
    if constexpr (EXT) {
        // Twelve Jacobian columns: the wave's 92 sums as ONE matrix product.  The 64 x 16 matrix [h | z | |r| | eff | 0]
        // goes through LDS and is multiplied with its own transpose by sixteen v_mfma_f64_16x16x4_f64 (four points per
        // instruction; the same register is the A and the B operand: A[i][k] = B[k][i] = M[point 4 s + lane / 16][column
        // lane % 16]); H^T H, H^T z, sum |r| and the count are entries of the 16 x 16 result, of which lane l holds
        // D[l / 16 + 4 r][l % 16], r = 0..3.  Measured against three halving butterflies over 96 terms
        // (scripts/hth_mfma_probe.hip, two waves per SIMD as here): 1.10 vs 2.08 us per wave -- and 0.72 us for the ONE
        // butterfly of the six-column case, which therefore keeps the shuffle form (north star: "MFMA ... taken only if
        // rocprof shows it beating the shuffle reduce").  Still a fixed function of the wave's 64 points: the
        // partition-consistency of the sums above the wave level is untouched.
        typedef double double4_t __attribute__((ext_vector_type(4)));
        double *my = sh.stage[wave];
#pragma unroll
        for (int k = 0; k < 12; ++k) my[lane * 17 + k] = h[k];
        my[lane * 17 + 12] = z;
        my[lane * 17 + 13] = absr;
        my[lane * 17 + 14] = eff ? 1.0 : 0.0;
        my[lane * 17 + 15] = short_list ? 1.0 : 0.0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const double m = my[(4 * s + (lane >> 4)) * 17 + (lane & 15)];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(m, m, acc, 0, 0, 0);
        }
        const int j = lane & 15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = (lane >> 4) + 4 * r;
            int slot = -1;
            if (i <= j && j < 12) slot = T::tri(i, j);
            else if (j == 12 && i < 12) slot = T::kHtz + i;
            else if (i == 13 && j == 14) slot = T::kRes;
            else if (i == 14 && j == 14) slot = T::kCnt;
            else if (i == 15 && j == 15) slot = T::kShort;
            if (slot >= 0) red[wave][slot] = acc[r];
        }
        if (lane < T::kSlots - T::kUsed) red[wave][T::kUsed + lane] = 0.0;  // padding slots of the row
    } else {
    double hz[12];
#pragma unroll
    for (int r = 0; r < 12; ++r) hz[r] = h[r] * z;
#pragma unroll
    for (int chunk = 0; chunk < T::kSlots / 32; ++chunk) {
        double v[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const int t = chunk * 32 + k;  // compile-time after unrolling
            double val = 0.0;
            if (t < T::kTri) {
                const int r = tri_row<NC>(t), c = r + (t - T::tri(r, r));
                val = h[r] * h[c];
            } else if (t < T::kTri + NC) {
                val = hz[t - T::kTri];
            } else if (t == T::kRes) {
                val = absr;
            } else if (t == T::kCnt) {
                val = eff ? 1.0 : 0.0;
            } else if (t == T::kShort) {
                val = short_list ? 1.0 : 0.0;
            }
            v[k] = val;
        }
        wave_sum32(v, lane);
        if ((lane & 1) == 0) red[wave][chunk * 32 + ((lane >> 1) & 31)] = v[0];
    }
    }
    __syncthreads();
    // one fp64 row per workgroup, published write-through (agent-scope 8-byte stores) so the last
    // workgroup to arrive can read every row without a release/acquire fence pair
    if (threadIdx.x < T::kSlots) {
        const int t = threadIdx.x;
        static_assert(kRedBlock / 64 == 8, "the wave sums of a workgroup go through a three-level tree");
        // a tree over the wave index too (not a running sum): the partition-consistency of the final sum below then
        // reaches down to aligned 64-point pieces of the scan
        const double s = ((red[0][t] + red[1][t]) + (red[2][t] + red[3][t])) + ((red[4][t] + red[5][t]) + (red[6][t] + red[7][t]));
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(a.partials) + (int64_t)bx * T::kSlots + t,
                           (unsigned long long)__double_as_longlong(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains before the ticket
    __syncthreads();
    uint32_t &s_last = sh.last;
    if (threadIdx.x == 0) {
        // Two-level arrival count: same-address atomics serialise in L2 (~11 ns each), so 128 workgroups on
        // one counter keep the last one waiting 1.4 us.  Sixteen group counters on separate cache lines take
        // the arrivals in parallel; the last of each group takes the top-level ticket.
        const uint32_t groups = min(nblk, (uint32_t)kTicketGroups);
        const uint32_t grp = bx % groups;
        const uint32_t gsize = (nblk - grp + groups - 1) / groups;
        uint32_t *gt = a.ticket + kTicketStride * (1 + grp);
        uint32_t last = 0;
        if (__hip_atomic_fetch_add(gt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1) {
            __hip_atomic_store(gt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
            last = (__hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == groups - 1) ? 1u : 0u;
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    // device-resident loop: the scan's state (4 KB) is requested now and lands in shared memory after the final sum -- its
    // round trip overlaps that of the partial rows
    double pre0 = 0.0, pre1 = 0.0;
    if constexpr (LOOP) {
        const double *src = reinterpret_cast<const double *>(&a.loop.state->in);
        pre0 = src[threadIdx.x];
        if (threadIdx.x + kRedBlock < kLoopInitDoubles) pre1 = src[threadIdx.x + kRedBlock];
    }

    // ---- last workgroup: fixed-order sum of the rows (independent of arrival order) -----------------
    // The order is a perfect BINARY TREE over the workgroup index (rows beyond the last one count as +0.0): lane ch
    // of kTree takes the contiguous rows [ch * R, (ch + 1) * R), R = P / kTree with P the power of two >= blocks,
    // eight at a time through a three-level tree, the groups of eight through a binary counter (the carry chain of
    // pairwise summation), and the kTree lane sums go through the same tree in LDS.  A tree over the index is what
    // makes the sum PARTITION-CONSISTENT: a handle that holds an aligned power-of-two piece of the scan (rows
    // [r * S, (r + 1) * S) of the whole) computes exactly the node of this tree that covers those rows, so n such
    // blocks combined pairwise by the host (s2m_iterated_update_multi) reproduce the single-handle block bit for bit.
    constexpr int kTree = ReduceShared<EXT>::kTree;     // lanes per term that take part: 16 (NC = 6) or 4 (NC = 12)
    constexpr int kDepth = 8;                           // independent loads in flight per lane
    auto &part = sh.part;
    auto &tot = sh.tot;
    const int blocks = (int)nblk;
    int P = kTree;
    while (P < blocks) P <<= 1;
    const int R = P / kTree;                            // rows per lane: a power of two
    if (threadIdx.x < kTree * T::kSlots) {
        const int t = threadIdx.x % T::kSlots, ch = threadIdx.x / T::kSlots;
        const unsigned long long *pp = reinterpret_cast<const unsigned long long *>(a.partials) + t;
        // binary counter of pending subtree sums: level l holds the sum of 8 * 2^l rows, or nothing
        constexpr int kLevels = 14;  // 8 * 2^14 rows per lane: more than a 2^28-point scan has
        double stack[kLevels];
        uint32_t occupied = 0u;
        double lane_sum = 0.0;
        for (int r0 = 0; r0 < R; r0 += kDepth) {
            unsigned long long raw[kDepth];
#pragma unroll
            for (int k = 0; k < kDepth; ++k) {
                const int b = ch * R + r0 + k;
                raw[k] = (r0 + k < R && b < blocks) ? __hip_atomic_load(pp + (int64_t)b * T::kSlots, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                    : 0ull;  // +0.0
            }
            double v[kDepth];
#pragma unroll
            for (int k = 0; k < kDepth; ++k) v[k] = __longlong_as_double((long long)raw[k]);
            // R < 8: the rows sit in the first R slots and the rest are +0.0 -- ((a0 + a1) + (0 + 0)) + ... is the tree over R
            double s = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            if (R <= kDepth) {
                lane_sum = s;
            } else {
                // push: merge with the pending sums of equal size, older (lower rows) on the left
#pragma unroll
                for (int l = 0; l < kLevels; ++l) {
                    const bool carry = (occupied >> l) & 1u;
                    const bool low_all = (occupied & ((1u << l) - 1u)) == ((1u << l) - 1u);  // every level below carried
                    if (carry && low_all) s = stack[l] + s;
                    else if (!carry && low_all) stack[l] = s;
                }
                occupied += 1u;  // the counter itself: bit l set <=> level l pending
                lane_sum = s;    // after the last group every level has carried: s is the whole tree
            }
        }
        part[ch][t] = lane_sum;
    }
    __syncthreads();
    if (threadIdx.x < T::kSlots) {
        const int t = threadIdx.x;
        double s;
        if (kTree == 16) {
            const double q0 = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
            const double q1 = (part[4 % kTree][t] + part[5 % kTree][t]) + (part[6 % kTree][t] + part[7 % kTree][t]);
            const double q2 = (part[8 % kTree][t] + part[9 % kTree][t]) + (part[10 % kTree][t] + part[11 % kTree][t]);
            const double q3 = (part[12 % kTree][t] + part[13 % kTree][t]) + (part[14 % kTree][t] + part[15 % kTree][t]);
            s = (q0 + q1) + (q2 + q3);
        } else {
            s = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
        }
        tot[t] = s;
    }
    __syncthreads();
    // far points of this pass (the two list lengths): published as block[158]; a pass that ran WITHOUT the far-point
    // kernel on the host's bet that the list would be empty is void when it is not -- the counters then stay for the redo
    const uint32_t far_points = a.hard_count ? a.hard_count[0] + a.hard_count[1] : 0u;
    const bool void_pass = a.spec != 0 && far_points != 0u;
    if (threadIdx.x < 160) {
        const int o = threadIdx.x;
        double v = 0.0;
        if (o < 144) {
            const int r = o / 12, c = o % 12;
            const int lo = r <= c ? r : c, hi_ = r <= c ? c : r;
            if (hi_ < NC) v = tot[T::tri(lo, hi_)];
        } else if (o < 156) {
            if (o - 144 < NC) v = tot[T::kHtz + (o - 144)];
        } else if (o == 156) {
            v = tot[T::kCnt];
        } else if (o == 157) {
            v = tot[T::kRes];
        } else if (o == 158) {
            v = (double)far_points;
        } else if (o == 159) {
            v = tot[T::kShort];  // rematch passes: neighbour lists short of the gate (0 on a reuse pass)
        }
        a.block[o] = v;
        if (a.host_block) publish_store(a.host_block + o, v);
        if (LOOP) loop_scratch.blk[o] = v;
    }
    if (threadIdx.x == 0) {
        __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    }
    __syncthreads();  // every thread has read the list lengths before they are reset
    if (threadIdx.x == 0 && a.hard_count && !void_pass) { a.hard_count[0] = 0u; a.hard_count[1] = 0u; a.hard_count[2] = 0u; }
    if (a.qheads && threadIdx.x < kQueueShards && !void_pass) a.qheads[threadIdx.x * kQueueStride] = 0u;
    if (a.host_flag) publish_flag(a.host_flag, a.seq);
    if constexpr (LOOP) {
        // ---- device-resident loop: the Kalman update, the judgement and the hand-over to the next pass (s2m_loop.h) ----
        LoopState *ls = a.loop.state;
        LoopRecord *rec = a.loop.record;
        {   // the scan's state into shared memory (its loads have been in flight since before the final sum)
            double *dst = reinterpret_cast<double *>(&loop_scratch.in);
            dst[threadIdx.x] = pre0;
            if (threadIdx.x + kRedBlock < kLoopInitDoubles) dst[threadIdx.x + kRedBlock] = pre1;
        }
        __syncthreads();
        if (threadIdx.x >= 64) return;  // one wave goes on
        const int t = threadIdx.x;
        const int it = ls->it;
        double old;
        loop_step_wave<NC>(ls, loop_scratch, FIT, old);
        // this iteration's log row (what Log/mat_out.txt records, :936-937): kept on the device until the loop ends
        const int32_t conv_now = ls->conv;
        if (t < S2M_DIM) ls->log_sol[it][t] = loop_scratch.sol[t];
        if (t == 32) {
            ls->log_res[it] = loop_scratch.blk[157];
            ls->log_effct[it] = (int32_t)loop_scratch.blk[156];
            ls->log_rematch[it] = FIT ? 1 : 0;
            ls->log_conv[it] = conv_now;
            ls->log_far[it] = (int32_t)loop_scratch.blk[158];
        }
        const bool finished = loop_scratch.finished != 0;
        if (!finished && a.loop.last_of_chunk) {  // the host enqueued no further: tell it where the loop stands
            if (t == 0) {
                publish_word(&rec->iters, it + 1);
                publish_word(&rec->finished, 0);
                publish_word(&rec->abort, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (t == 0) __hip_atomic_store(&rec->flag, a.loop.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (finished) {
            for (int q = 0; q < it; ++q) {  // the earlier passes' rows (written by earlier launches), then this pass's
                if (t < S2M_DIM) publish_store(&rec->solution[q][t], ls->log_sol[q][t]);
                if (t == 32) {
                    publish_store(&rec->total_residual[q], ls->log_res[q]);
                    publish_word(&rec->effct[q], ls->log_effct[q]);
                    publish_word(&rec->rematch[q], ls->log_rematch[q]);
                    publish_word(&rec->conv_it[q], ls->log_conv[q]);
                    publish_word(&rec->far_points[q], ls->log_far[q]);
                }
            }
            if (t < S2M_DIM) publish_store(&rec->solution[it][t], loop_scratch.sol[t]);
            if (t == 32) {
                publish_store(&rec->total_residual[it], loop_scratch.blk[157]);
                publish_word(&rec->effct[it], (int32_t)loop_scratch.blk[156]);
                publish_word(&rec->rematch[it], FIT ? 1 : 0);
                publish_word(&rec->conv_it[it], conv_now);
                publish_word(&rec->far_points[it], (int32_t)loop_scratch.blk[158]);
            }
            if (t < S2M_STATE_DOUBLES) publish_store(&rec->x[t], loop_scratch.in.x[t]);
            for (int o = t; o < S2M_BLOCK_DOUBLES; o += 64) publish_store(&rec->block[o], loop_scratch.blk[o]);
            if (t < 24) {
                publish_store(&rec->pose_last[t], old);
                publish_store(&rec->pose_rematch[t], FIT ? old : ls->pose_rematch[t]);
            }
            if (t < S2M_FEAT_QUEUE + 2) publish_word(&rec->queue[t], loop_scratch.in.queue[t]);
            if (t == 0) {
                publish_word(&rec->queue_len, loop_scratch.in.queue_len);
                publish_word(&rec->iters, it + 1);
                publish_word(&rec->passes, ls->passes);
                publish_word(&rec->conv, ls->conv);
                publish_word(&rec->stop, ls->stop);
                publish_word(&rec->finished, 1);
                publish_word(&rec->abort, 0);
                publish_word(&rec->numeric, ls->numeric);
                publish_word(&rec->update_cov, ls->update_cov);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (t == 0) __hip_atomic_store(&rec->flag, a.loop.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

template <bool EXT, bool FIT>
__global__ __launch_bounds__(kRedBlock) void reduce_kernel(ReduceArgs a)
{
    __shared__ ReduceShared<EXT> sh;
    reduce_body<EXT, FIT, false>(a, sh, blockIdx.x, gridDim.x);
}

// A reduce launch that finds the device wanting a rematch pass although the host enqueued no search kernels in front of
// it (the schedule the host predicted from the previous scan did not hold): nothing is computed, the record tells the
// host where to resume.  (The opposite mismatch is harmless: the search kernels left at once and this pass reuses.)
__device__ __forceinline__ void loop_abort(const LoopLaunch &l, uint32_t bx)
{
    if (bx != 0) return;
    if (threadIdx.x == 0) {
        l.state->abort = l.gen;
        publish_word(&l.record->iters, l.state->it);
        publish_word(&l.record->finished, 0);
        publish_word(&l.record->abort, 1);
    }
    publish_flag(&l.record->flag, l.seq);
}

template <bool EXT>
__global__ __launch_bounds__(kRedBlock, EXT ? 2 : 6) void reduce_kernel_loop(ReduceArgs a)
{
    if (a.loop.init && a.n == 0) {  // an empty scan has no search kernel in front: its one workgroup brings the init record over
        loop_copy_init(a.loop);
        __threadfence();
        __syncthreads();
    }
    int rematch_now = 0;
    if (!loop_enter(a.loop, a.pose, rematch_now)) return;
    if (rematch_now && !a.loop.kind) { loop_abort(a.loop, blockIdx.x); return; }
    a.fit = rematch_now;
    __shared__ ReduceShared<EXT> sh;
    if (a.fit) reduce_body<EXT, true, true>(a, sh, blockIdx.x, gridDim.x);
    else reduce_body<EXT, false, true>(a, sh, blockIdx.x, gridDim.x);
}

// K scans, one grid: blockIdx.y = scan; FIT serves the scans that searched in this pass, the plain form the others
// (two launches when a pass holds both kinds).  Same per-point code, same workgroup -> points mapping and the same
// fixed-order final sum per scan as the single-scan kernel: every scan's block is bit-identical to what
// s2m_iterated_update produces.  The last workgroup of every scan also zeroes the OTHER set of far-point counters and
// queue heads (the set the next launch will use; the current one may still be read by nobody -- match_hard is done --
// but the next launch's search kernels must find theirs zero).
__device__ __forceinline__ ReduceArgs batch_reduce_args(const BatchArgs &b, const ScanDesc &d)
{
    ReduceArgs a;
    a.pose = d.pose; a.gates = b.gates;
    a.sx = d.sx; a.sy = d.sy; a.sz = d.sz; a.n = d.n;
    a.fit = d.rematch;
    a.nn_idx = d.nn_idx; a.nn_d2 = d.nn_d2; a.pts = b.grid.pts;
    a.plane = d.plane; a.flags = d.flags; a.sel = d.sel; a.eff = d.eff; a.pd2 = d.pd2;
    a.partials = d.partials; a.block = d.block; a.ticket = d.ticket;
    a.hard_count = b.hard_count_next; a.qheads = b.qheads_next;
    a.host_block = d.host_block; a.host_flag = d.host_flag; a.seq = d.seq;
    a.loop = d.loop;
    return a;
}

template <bool EXT, bool FIT>
__global__ __launch_bounds__(kRedBlock) void reduce_kernel_batch(BatchArgs b)
{
    const ScanDesc &d = b.d[blockIdx.y];
    if (!d.active || (d.rematch != 0) != FIT) return;
    const uint32_t nblk = (uint32_t)max((d.n + kRedBlock - 1) / kRedBlock, 1);
    if (blockIdx.x >= nblk) return;
    const ReduceArgs a = batch_reduce_args(b, d);
    __shared__ ReduceShared<EXT> sh;
    reduce_body<EXT, FIT, false>(a, sh, blockIdx.x, nblk);
}

// the same with the loop on the device: ONE launch per pass serves the rematching and the reusing scans alike
template <bool EXT>
__global__ __launch_bounds__(kRedBlock, EXT ? 2 : 6) void reduce_kernel_batch_loop(BatchArgs b)
{
    const ScanDesc &d = b.d[blockIdx.y];
    if (!d.active) return;
    const uint32_t nblk = (uint32_t)max((d.n + kRedBlock - 1) / kRedBlock, 1);
    if (blockIdx.x >= nblk) return;
    if (d.loop.init && b.n_max <= 0) {  // a group of empty scans: no search kernel ran in front to bring the init records over
        loop_copy_init(d.loop);
        __threadfence();
        __syncthreads();
    }
    ReduceArgs a = batch_reduce_args(b, d);
    int rematch_now = 0;
    if (!loop_enter(d.loop, a.pose, rematch_now)) {
        // a scan whose loop has ended still keeps the hand-over of the far-point counters going: the set the NEXT launch
        // will use must be zero whoever is left to run (every scan does it, redundantly)
        if (blockIdx.x == 0) {
            if (threadIdx.x < 3) b.hard_count_next[threadIdx.x] = 0u;
            if (threadIdx.x < kQueueShards) b.qheads_next[threadIdx.x * kQueueStride] = 0u;
        }
        return;
    }
    if (rematch_now && !d.loop.kind) { loop_abort(d.loop, blockIdx.x); return; }
    a.fit = rematch_now;
    a.host_block = nullptr; a.host_flag = nullptr;
    __shared__ ReduceShared<EXT> sh;
    if (a.fit) reduce_body<EXT, true, true>(a, sh, blockIdx.x, nblk);
    else reduce_body<EXT, false, true>(a, sh, blockIdx.x, nblk);
}

void launch_reduce_batch_loop(const BatchArgs &b, hipStream_t st)
{
    const dim3 grid((unsigned)std::max(reduce_blocks(b.n_max), 1), (unsigned)b.k);
    if (b.gates.extrinsic) hipLaunchKernelGGL((reduce_kernel_batch_loop<true>), grid, dim3(kRedBlock), 0, st, b);
    else hipLaunchKernelGGL((reduce_kernel_batch_loop<false>), grid, dim3(kRedBlock), 0, st, b);
}

void launch_reduce_batch(const BatchArgs &b, bool any_fit, bool any_plain, hipStream_t st)
{
    const dim3 grid((unsigned)std::max(reduce_blocks(b.n_max), 1), (unsigned)b.k);
    if (b.gates.extrinsic) {
        if (any_fit) hipLaunchKernelGGL((reduce_kernel_batch<true, true>), grid, dim3(kRedBlock), 0, st, b);
        if (any_plain) hipLaunchKernelGGL((reduce_kernel_batch<true, false>), grid, dim3(kRedBlock), 0, st, b);
    } else {
        if (any_fit) hipLaunchKernelGGL((reduce_kernel_batch<false, true>), grid, dim3(kRedBlock), 0, st, b);
        if (any_plain) hipLaunchKernelGGL((reduce_kernel_batch<false, false>), grid, dim3(kRedBlock), 0, st, b);
    }
}

// after a collective: copy the summed block to pinned host memory and raise the sequence flag
__global__ __launch_bounds__(192) void publish_kernel(const double *__restrict__ block, double *__restrict__ host_block,
                                                      unsigned long long *__restrict__ host_flag, unsigned long long seq)
{
    if (threadIdx.x < 160) publish_store(host_block + threadIdx.x, block[threadIdx.x]);
    publish_flag(host_flag, seq);
}
void launch_publish(const double *block, double *host_block, unsigned long long *host_flag, unsigned long long seq,
                    hipStream_t st)
{
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(192), 0, st, block, host_block, host_flag, seq);
}

int reduce_blocks(int n) { return (n + kRedBlock - 1) / kRedBlock; }
int rows_blocks(int n) { return (n + kRowsBlock - 1) / kRowsBlock; }

void launch_reduce(const ReduceArgs &a, hipStream_t st)
{
    // an empty scan still produces a (zero) block: one workgroup with no points
    const int blocks = std::max(reduce_blocks(a.n), 1);
    if (a.loop.state) {
        if (a.gates.extrinsic) hipLaunchKernelGGL((reduce_kernel_loop<true>), dim3(blocks), dim3(kRedBlock), 0, st, a);
        else hipLaunchKernelGGL((reduce_kernel_loop<false>), dim3(blocks), dim3(kRedBlock), 0, st, a);
        return;
    }
    if (a.gates.extrinsic) {
        if (a.fit) hipLaunchKernelGGL((reduce_kernel<true, true>), dim3(blocks), dim3(kRedBlock), 0, st, a);
        else hipLaunchKernelGGL((reduce_kernel<true, false>), dim3(blocks), dim3(kRedBlock), 0, st, a);
    } else {
        if (a.fit) hipLaunchKernelGGL((reduce_kernel<false, true>), dim3(blocks), dim3(kRedBlock), 0, st, a);
        else hipLaunchKernelGGL((reduce_kernel<false, false>), dim3(blocks), dim3(kRedBlock), 0, st, a);
    }
}

// ---- dense rows in index order (Hsub / meas_vec / laserCloudOri order), on request --------------
__global__ __launch_bounds__(kRowsBlock) void rows_count_kernel(const uint8_t *__restrict__ eff, int n,
                                                               uint32_t *__restrict__ block_cnt)
{
    __shared__ uint32_t wc[kRowsBlock / 64];
    const int i = blockIdx.x * kRowsBlock + threadIdx.x;
    const bool e = (i < n) && eff[i];
    const uint64_t b = __ballot(e);
    if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = (uint32_t)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
}

// exclusive scan of block counts in place, single workgroup; block_off[blocks] = total
__global__ __launch_bounds__(1024) void rows_scan_kernel(uint32_t *__restrict__ block_off, int blocks)
{
    __shared__ uint32_t carry;
    __shared__ uint32_t ws[16];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < blocks; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t v = (i < blocks) ? block_off[i] : 0u;
        uint32_t x = v;  // inclusive scan inside the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(x, off, 64);
            if ((int)(threadIdx.x & 63) >= off) x += y;
        }
        if ((threadIdx.x & 63) == 63) ws[threadIdx.x >> 6] = x;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += ws[w];
        const uint32_t c = carry;
        if (i < blocks) block_off[i] = c + woff + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + woff + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) block_off[blocks] = carry;
}

template <bool EXT>
__global__ __launch_bounds__(kRowsBlock) void rows_emit_kernel(RowsArgs a)
{
    __shared__ uint32_t wc[kRowsBlock / 64];
    const int i = blockIdx.x * kRowsBlock + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool e = (i < a.n) && a.eff[i];
    const uint64_t b = __ballot(e);
    if (lane == 0) wc[wave] = (uint32_t)__popcll(b);
    __syncthreads();
    if (!e) return;
    uint32_t pos = a.block_off[blockIdx.x] + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) pos += wc[w];
    double h[12], z;
    jac_row<EXT>(a.pose, a.sx[i], a.sy[i], a.sz[i], a.plane[i], a.pd2[i], h, z);
    if (a.h_x) {
#pragma unroll
        for (int k = 0; k < 12; ++k) a.h_x[(int64_t)pos * 12 + k] = h[k];
    }
    if (a.h) a.h[pos] = z;
    if (a.scan_index) a.scan_index[pos] = i;
}

void launch_rows(const RowsArgs &a, hipStream_t st)
{
    const int blocks = rows_blocks(a.n);
    if (blocks > 0) hipLaunchKernelGGL(rows_count_kernel, dim3(blocks), dim3(kRowsBlock), 0, st, a.eff, a.n, a.block_off);
    hipLaunchKernelGGL(rows_scan_kernel, dim3(1), dim3(1024), 0, st, a.block_off, blocks);
    if (blocks > 0) {
        if (a.gates.extrinsic)
            hipLaunchKernelGGL(rows_emit_kernel<true>, dim3(blocks), dim3(kRowsBlock), 0, st, a);
        else
            hipLaunchKernelGGL(rows_emit_kernel<false>, dim3(blocks), dim3(kRowsBlock), 0, st, a);
    }
}

}  // namespace s2m
