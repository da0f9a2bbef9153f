// s2m_loop.h -- the iterated update with the state on the device (round 4).
//
// The host-stepped loop (s2m_engine.cpp) turns round after every pass: block to pinned memory, a 24x24 fp64 solve on
// the host, the next launch -- five round trips per scan, a quarter of a single scan's latency with the GPU idle.
// Here the last workgroup of the reduce kernel does what laserMapping.cpp:899-918 and :1012-1101 do with the block
// (degeneracy queue, Kalman update, [+], convergence test, rematch judgement, exit test) and leaves the new pose and
// the control words in device memory for the kernels of the next iteration, which the host has already enqueued;
// the host reads ONE record at the end and applies the covariance update (:1084-1085).
//
// Algebra.  The reference inverts two 24x24 matrices per iteration: K_1 = (H^T H (+) 0 + (P/R)^-1)^-1 (:1017-1018).
// With P' = P / R, U = [I_nc; 0] (nc = 6 Jacobian columns, 12 with extrinsic estimation), A = H^T H (nc x nc) and
// C = U^T P' U the matrix-inversion lemma gives  K_1 U = P' U (I + A C)^-1 = (P' U C^-1) (C^-1 + A)^-1 = G M^-1,
// G = P' U C^-1 (24 x nc) and C^-1 constant for the scan (host, once), M = C^-1 + A symmetric positive definite:
//     solution = K z + vec - K H vec[0:nc] = vec + G M^-1 (H^T z - A vec[0:nc])            (:1019-1032)
//     P <- (I - K H (+) 0) P = P - G M^-1 (A P[0:nc, :])                                    (:1084-1085)
// -- one nc x nc Cholesky solve per iteration instead of two 24x24 LU inverses; same mathematics, agreement with the
// literal form ~1e-12 (tests compare every entry point with the oracle's two-inverse form at 1e-9).
#pragma once
#include <stdint.h>

#include "../../include/daliti_s2m.h"
#include "s2m_device.h"
#include "s2m_trig.h"

namespace s2m {

constexpr int kLoopMaxIter = 64;

// what the host hands over per scan (pinned host memory, copied to the device by workgroup 0 of the first kernel)
struct LoopInit {
    double x[S2M_STATE_DOUBLES];       // state the first pass runs at
    double x_prop[S2M_STATE_DOUBLES];
    double G[S2M_DIM * 12];            // row-major 24 x nc
    double Cinv[12 * 12];              // row-major nc x nc
    double conv_rot_deg, conv_pos_cm;
    int32_t max_iter, feat_threshold, nc, queue_len;
    int32_t queue[S2M_FEAT_QUEUE + 2];
};
static_assert(sizeof(LoopInit) % 8 == 0, "copied as doubles");

// device-resident loop state of one scan
struct LoopState {
    LoopInit in;               // x evolves; the rest is constant for the scan
    double pose_last[24];      // pose the last executed pass ran at (dense rows on request are evaluated there)
    double pose_rematch[24];   // pose of the last rematch pass (the world-frame queries Nearest_Points belong to)
    // control words, written by the last workgroup of every reduce launch
    int32_t it;                // iteration the NEXT pass is (0 before the first)
    int32_t rematch_now;       // the next pass searches
    int32_t rematch_num, rematch_en, conv, stop;
    int32_t finished;          // the loop has ended: every later kernel of the chain leaves at once
    int32_t abort;             // generation of the plan that did not hold (0 = none): a reduce launch found the device wanting
                               // a rematch pass where the host had enqueued none; the rest of that plan leaves at once and
                               // the host re-plans from `it` under a new generation
    int32_t passes;            // rematch passes so far
    int32_t numeric;           // the nc x nc system was not positive definite
    int32_t update_cov;        // the loop ended through the exit test without EKF_stop: the covariance is updated (:1084)
    int32_t pad;
    // per-iteration log rows, kept here until the loop ends (a store to pinned host memory holds the kernel's end back by
    // a PCIe round trip: once per scan, not once per pass)
    double log_res[kLoopMaxIter];
    double log_sol[kLoopMaxIter][S2M_DIM];
    int32_t log_effct[kLoopMaxIter], log_rematch[kLoopMaxIter], log_conv[kLoopMaxIter], log_far[kLoopMaxIter];
};

// the record the host reads at the end (pinned host memory; the flag is written last)
struct LoopRecord {
    double x[S2M_STATE_DOUBLES];
    double block[S2M_BLOCK_DOUBLES];   // normal block of the last pass (covariance update, rows on request)
    double pose_last[24];
    double pose_rematch[24];
    double total_residual[kLoopMaxIter];
    double solution[kLoopMaxIter][S2M_DIM];
    int32_t effct[kLoopMaxIter], rematch[kLoopMaxIter], conv_it[kLoopMaxIter], far_points[kLoopMaxIter];
    int32_t queue[S2M_FEAT_QUEUE + 2];
    int32_t queue_len, iters, passes, conv, stop, finished, abort, numeric, update_cov, pad;
    unsigned long long flag;           // sequence number of the call, written after everything else
};

// one launch's place in the chain the host enqueued
struct LoopLaunch {
    LoopState *state = nullptr;        // nullptr: host-stepped pass (pose in the kernel arguments)
    const LoopInit *init = nullptr;    // first kernel of a scan: copy this into *state (device-visible pinned pointer)
    LoopRecord *record = nullptr;      // device-visible pinned pointer
    unsigned long long seq = 0;
    int32_t expect_it = 0;             // iteration this launch belongs to
    int32_t kind = 0;                  // 1: the host enqueued the search kernels in front of this reduce launch
    int32_t gen = 0;                   // generation of the host's plan (a re-plan after an abort takes a new one)
    int32_t last_of_chunk = 0;         // reduce launch: the host has enqueued nothing behind this one -- report even if the loop goes on
};

#if defined(__HIPCC__)
// Is this launch the one the device-resident loop is waiting for?  The host enqueues the kernels of several iterations
// ahead; the control words the previous reduce launch left decide what each of them does.  False when the launch has
// nothing to do (the loop has ended, or the host is re-planning after a schedule it predicted wrongly).
__device__ __forceinline__ bool loop_launch_due(const LoopLaunch &l)
{
    const LoopState *ls = l.state;
    return !(ls->finished || ls->abort == l.gen || ls->it != l.expect_it);
}
// What a launch of the chain needs to know, requested in ONE round trip: the control words the previous reduce launch
// left and the pose it left (the loads are issued before the first branch; the state is written by this very kernel's
// last workgroup, so the compiler must treat them as divergent -- 48 VGPRs for a value every lane shares -- unless the
// values are moved to scalar registers by hand).  Returns whether the launch is due; rematch_now: the pass searches.
__device__ __forceinline__ bool loop_enter(const LoopLaunch &l, Pose &pose, int &rematch_now)
{
    const LoopState *ls = l.state;
    int fin = ls->finished, ab = ls->abort, it = ls->it, rn = ls->rematch_now;
    double px[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) px[i] = ls->in.x[i];
    asm volatile("" : "+v"(fin), "+v"(ab), "+v"(it), "+v"(rn));
#pragma unroll
    for (int i = 0; i < 24; ++i) asm volatile("" : "+v"(px[i]));
    fin = __builtin_amdgcn_readfirstlane(fin); ab = __builtin_amdgcn_readfirstlane(ab);
    it = __builtin_amdgcn_readfirstlane(it); rn = __builtin_amdgcn_readfirstlane(rn);
    rematch_now = rn;
    if (fin || ab == l.gen || it != l.expect_it) return false;
    double *o = reinterpret_cast<double *>(&pose);
#pragma unroll
    for (int i = 0; i < 24; ++i) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(px[i]);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32));
        o[i] = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    return true;
}
// First kernel of a scan's chain, workgroup 0: the init record from pinned host memory into the device state.  The
// kernel itself runs at the pose in its arguments; everything later in the stream reads the state.
__device__ __forceinline__ void loop_copy_init(const LoopLaunch &l)
{
    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(l.init);
    unsigned long long *dst = reinterpret_cast<unsigned long long *>(&l.state->in);
    for (int i = threadIdx.x; i < (int)(sizeof(LoopInit) / 8); i += blockDim.x)
        dst[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x == 0) {
        LoopState *ls = l.state;
        ls->it = 0; ls->rematch_now = 1; ls->rematch_num = 0; ls->rematch_en = 0; ls->conv = 0; ls->stop = 0;
        ls->finished = 0; ls->abort = 0; ls->passes = 0; ls->numeric = 0; ls->update_cov = 0;
    }
}

// ---- small fp64 algebra, one thread ------------------------------------------------------------------------------
// R <- R * Exp(v) in place (so3_math.h:55-72: Exp is the identity unless |v| > 1e-5, else Rodrigues on the unit axis k:
// Exp = I + sin(t) [k]x + (1 - cos(t)) [k]x^2 with [k]x^2 = k k^T - I).  Written row by row with nine scalars of Exp live:
// this runs in the last workgroup of a kernel that is compiled for 64 registers.
__device__ inline void loop_rot_times_exp(double *R, double v1, double v2, double v3)
{
    const double norm = sqrt(v1 * v1 + v2 * v2 + v3 * v3);
    if (!(norm > 0.00001)) return;
    const double k0 = v1 / norm, k1 = v2 / norm, k2 = v3 / norm;
    const double s = trig_sin(norm), c1 = 1.0 - trig_cos(norm);
    const double d = 1.0 - c1;
    const double e00 = d + c1 * (k0 * k0), e01 = c1 * (k0 * k1) - s * k2, e02 = c1 * (k0 * k2) + s * k1;
    const double e10 = c1 * (k1 * k0) + s * k2, e11 = d + c1 * (k1 * k1), e12 = c1 * (k1 * k2) - s * k0;
    const double e20 = c1 * (k2 * k0) - s * k1, e21 = c1 * (k2 * k1) + s * k0, e22 = d + c1 * (k2 * k2);
#pragma unroll 1
    for (int r = 0; r < 3; ++r) {
        const double a0 = R[r * 3 + 0], a1 = R[r * 3 + 1], a2 = R[r * 3 + 2];
        R[r * 3 + 0] = (a0 * e00 + a1 * e10) + a2 * e20;
        R[r * 3 + 1] = (a0 * e01 + a1 * e11) + a2 * e21;
        R[r * 3 + 2] = (a0 * e02 + a1 * e12) + a2 * e22;
    }
}
// out = Log(A^T B) (so3_math.h:76-81; the rotation part of a [-] b, common_lib.h:173-187), scalars only
__device__ inline void loop_log_atb(const double *A, const double *B, double *out)
{
    auto el = [&](int r, int c) { return (A[0 * 3 + r] * B[0 * 3 + c] + A[1 * 3 + r] * B[1 * 3 + c]) + A[2 * 3 + r] * B[2 * 3 + c]; };
    const double tr = el(0, 0) + el(1, 1) + el(2, 2);
    const double k0 = el(2, 1) - el(1, 2), k1 = el(0, 2) - el(2, 0), k2 = el(1, 0) - el(0, 1);
    // The reference: theta = acos((tr - 1) / 2) (0 when tr > 3 - 1e-6), factor 0.5 when |theta| < 0.001, else
    // 0.5 theta / sin(theta).  The vector (k0, k1, k2) / 2 has length sin(theta): for the small rotations this is called
    // with (the state against its own prediction) theta / sin(theta) = asin(s) / s is a short series in s^2 -- no acos, no
    // square root, no division in the chain of the one wave that runs this (each costs ~150 cycles there); agreement with
    // the acos form ~1e-16 (theta / sin(theta) is flat in theta).  Larger rotations take the reference's form as written.
    const double s2 = 0.25 * ((k0 * k0 + k1 * k1) + k2 * k2);
    double f;
    if (tr > 2.8 && s2 < 0.01) {                      // cos(theta) > 0.9, sin^2(theta) < 0.01
        const double lim = 9.999996666667e-07;         // sin^2(0.001)
        f = 0.5;
        if (!(s2 < lim))
            f = 0.5 * (1.0 + s2 * (1.0 / 6 + s2 * (3.0 / 40 + s2 * (5.0 / 112 + s2 * (35.0 / 1152 + s2 * (63.0 / 2816 +
                       s2 * (231.0 / 13312 + s2 * (143.0 / 10240))))))));
    } else {
        const double theta = (tr > 3.0 - 1e-6) ? 0.0 : trig_acos(0.5 * (tr - 1));
        f = (fabs(theta) < 0.001) ? 0.5 : (0.5 * theta / trig_sin(theta));
    }
    out[0] = f * k0; out[1] = f * k1; out[2] = f * k2;
}
// offset in the 36-double state of the linear block that error-state entries e .. e + 2 belong to (common_lib.h:146-157)
__device__ inline int loop_linear_offset(int e) { return e == 3 ? 9 : (e == 9 ? 21 : 12 + e); }

// shared-memory work area of the step
struct LoopScratch {
    double blk[S2M_BLOCK_DOUBLES];   // the pass's block
    LoopInit in;                     // the scan's state, staged from device memory by the whole workgroup (one round trip)
    double vec[S2M_DIM], sol[S2M_DIM];
    int32_t finished;
};
constexpr int kLoopInitDoubles = (int)(sizeof(LoopInit) / 8);

__device__ __forceinline__ double loop_readlane(double v, int lane)  // lane: compile-time constant after unrolling
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// LDS writes of this wave before LDS reads by other lanes of the same wave
__device__ __forceinline__ void loop_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// What the reference does with the result of one pass (laserMapping.cpp:899-918, 1012-1101), by ONE WAVE of the last
// workgroup of the reduce kernel (threadIdx.x < 64); s.blk holds the pass's block and s.in the staged state.  The
// nc x nc system M w = b (M = C^-1 + H^T H, symmetric positive definite: elimination without pivoting) is solved with a
// row per lane in registers and v_readlane broadcasts of the pivot row -- no shared-memory round trips in the chain; the
// 24-wide parts take a lane per row, the two rotations a lane each.  On return the state and the control words in *ls
// describe the next pass, s.sol / s.finished are set, and `old` holds the pose this pass ran at (lanes < 24).
template <int NC>
__device__ inline void loop_step_wave(LoopState *ls, LoopScratch &s, bool was_rematch, double &old)
{
    const int lane = threadIdx.x;
    LoopInit &in = s.in;
    const double *A = s.blk, *Htz = s.blk + 144;  // 12 x 12 block layout, zeros beyond nc
    // control words of the previous pass (one round trip, overlapped with everything up to the judgement)
    const int it = ls->it;
    int rematch_num = ls->rematch_num, conv = ls->conv;
    const int passes = ls->passes;
    old = lane < 24 ? in.x[lane] : 0.0;
    // effct_feat_numQueue / EKF_stop_flg (:899-918): the queue entry a lane holds after the push
    const int32_t effct = (int32_t)s.blk[156];
    int len = in.queue_len;
    int32_t qv = 0x7fffffff;
    if (len < S2M_FEAT_QUEUE) {
        if (lane < len) qv = in.queue[lane];
        if (lane == len) qv = effct;
        len += 1;
    } else {
        if (lane < S2M_FEAT_QUEUE - 1) qv = in.queue[lane + 1];
        if (lane == S2M_FEAT_QUEUE - 1) qv = effct;
        len = S2M_FEAT_QUEUE;
    }
    const bool stop = __ballot(lane < len && qv <= in.feat_threshold) != 0ull;
    loop_wave_sync();  // every lane has read its old entry
    if (lane < S2M_FEAT_QUEUE + 1) in.queue[lane] = lane < len ? qv : 0;
    if (lane == 0) in.queue_len = len;
    double sol = 0.0;
    bool ok = true;
    if (!stop) {  // wave-uniform
        // lane i < NC: row i of M = C^-1 + A (requested before the chain below needs it)
        const int i = lane < NC ? lane : 0;
        double Mr[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) Mr[c] = in.Cinv[i * NC + c] + A[i * 12 + c];
        const double hz = Htz[i];
        // vec = x_prop [-] x (:1028; common_lib.h:173-187)
        if (lane < 2) {
            const int o = lane == 0 ? 0 : 12;
            loop_log_atb(in.x + o, in.x_prop + o, s.vec + (lane == 0 ? 0 : 6));
        } else if (lane >= 8 && lane < 8 + 18) {
            const int k = lane - 8, e = (k < 3) ? 3 + k : ((k < 6) ? 9 + (k - 3) : 12 + (k - 6));
            const int base = e - (e % 3);
            s.vec[e] = in.x_prop[loop_linear_offset(base) + e % 3] - in.x[loop_linear_offset(base) + e % 3];
        }
        loop_wave_sync();
        // b_i = (H^T z)_i - (A vec[0:nc])_i
        double bi;
        {
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c < NC; ++c) acc += A[i * 12 + c] * s.vec[c];
            bi = hz - acc;
        }
        const double vr = lane < S2M_DIM ? s.vec[lane] : 0.0;
        // forward elimination, pivot row broadcast from lane k (one reciprocal per pivot, reused by the back substitution)
        double rp[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const double pk = loop_readlane(Mr[k], k);
            ok = ok && (pk > 0.0);
            rp[k] = 1.0 / pk;
            const double f = Mr[k] * rp[k];
            const double bk = loop_readlane(bi, k);
#pragma unroll
            for (int c = k + 1; c < NC; ++c) {
                const double rkc = loop_readlane(Mr[c], k);
                if (lane > k) Mr[c] -= f * rkc;
            }
            if (lane > k) bi -= f * bk;
        }
        // back substitution: w_k from lane k, every lane above folds it into its own right-hand side
        double w[NC];
#pragma unroll
        for (int k = NC - 1; k >= 0; --k) {
            const double wk = loop_readlane(bi, k) * rp[k];
            w[k] = wk;
            if (lane < k) bi -= Mr[k] * wk;
        }
        if (ok) {
            // solution = vec + G w (:1032), a lane per row
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c < NC; ++c) acc += in.G[(lane < S2M_DIM ? lane : 0) * NC + c] * w[c];
            sol = lane < S2M_DIM ? vr + acc : 0.0;
        }
    }
    if (lane < S2M_DIM) s.sol[lane] = sol;
    loop_wave_sync();
    if (!stop && ok) {
        // x [+]= solution (:1033; common_lib.h:146-157), on the staged copy
        if (lane < 2) {
            const int o = lane == 0 ? 0 : 12, e = lane == 0 ? 0 : 6;
            loop_rot_times_exp(in.x + o, s.sol[e], s.sol[e + 1], s.sol[e + 2]);
        } else if (lane >= 8 && lane < 8 + 18) {
            const int k = lane - 8, e = (k < 3) ? 3 + k : ((k < 6) ? 9 + (k - 3) : 12 + (k - 6));
            const int base = e - (e % 3);
            in.x[loop_linear_offset(base) + e % 3] += s.sol[e];
        }
        loop_wave_sync();
    }
    // back to device memory: the state the next pass runs at, the queue, this pass's poses
    if (lane < S2M_STATE_DOUBLES) ls->in.x[lane] = in.x[lane];
    if (lane < S2M_FEAT_QUEUE + 1) ls->in.queue[lane] = in.queue[lane];
    if (lane < 24) {
        ls->pose_last[lane] = old;
        if (was_rematch) ls->pose_rematch[lane] = old;
    }
    if (lane == 0) {
        ls->in.queue_len = len;
        if (!stop && ok) {
            const double rn = sqrt(s.sol[0] * s.sol[0] + s.sol[1] * s.sol[1] + s.sol[2] * s.sol[2]);
            const double tn = sqrt(s.sol[3] * s.sol[3] + s.sol[4] * s.sol[4] + s.sol[5] * s.sol[5]);
            conv = ((rn * 57.3 < in.conv_rot_deg) && (tn * 100 < in.conv_pos_cm)) ? 1 : 0;  // :1040
        }
        // rematch judgement and exit test (:1070-1101; s2m_iterctl.h)
        int32_t rematch_en = 0;
        if (conv || (rematch_num == 0 && it == in.max_iter - 2)) { rematch_en = 1; rematch_num++; }
        bool finished = false, update_cov = false;
        if (rematch_num >= 2 || it == in.max_iter - 1) { finished = true; update_cov = !stop; }
        else if (stop) finished = true;
        if (!ok) { finished = true; update_cov = false; }
        ls->conv = conv;
        ls->stop = stop ? 1 : 0;
        ls->rematch_en = rematch_en;
        ls->rematch_num = rematch_num;
        ls->passes = passes + (was_rematch ? 1 : 0);
        ls->it = it + 1;
        ls->rematch_now = rematch_en;
        ls->finished = finished ? 1 : 0;
        ls->numeric = ok ? 0 : 1;
        ls->update_cov = update_cov ? 1 : 0;
        s.finished = finished ? 1 : 0;
    }
    loop_wave_sync();
}
#endif  // __HIPCC__

}  // namespace s2m
