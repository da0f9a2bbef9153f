// bench_loop.cpp -- the timed region of bench.py as a C++ caller of the C ABI (include/daliti_s2m.h).
//
// The reference's caller is C++ (eskf_lio/src/laserMapping.cpp:731, one iterated update per scan inside
// `while (sync_packages(Measures))`); a step of the benchmark is exactly what that loop body does with the
// engine: hand it the propagated state and covariance, run s2m_iterated_update, read the result.  Driving
// the K timed steps from here instead of from Python keeps the interpreter's ~10 us per call out of a
// 150 us step.  Plain g++, links only libdaliti_s2m.so; built by __graft_entry__.build() as
// daliti_amd/_lib/libs2m_benchloop.so and loaded by bench.py with ctypes.
#include <chrono>
#include <cstdint>
#include <cstring>

#include "daliti_s2m.h"

extern "C" {

// `steps` steps over `k` scans (handles[i] holds scan i).  Every step is the same scan arriving fresh: the
// degeneracy queue is cleared, x := x_prop, P := P0 with its first element perturbed in the last bit on odd
// steps (a real stream hands in a different covariance every scan, so the engine's cache of (P/R)^-1 must
// miss once per scan).  mode 0: the scans one after the other (s2m_iterated_update); mode 1: all k in flight
// (s2m_iterated_update_batch); mode 2: the k handles hold the shards of ONE scan (s2m_iterated_update_multi).
// x / P (k x 36, k x 576) hold the last step's results on return, logs (k) its per-iteration logs; iters /
// rematch accumulate iterations and kNN passes over all steps and scans.  step_us (optional, `steps` doubles): the
// wall time of every step on the host's steady clock (one clock read per step; diagnostics of short runs).
int s2m_bench_loop(s2m_engine *const *handles, int32_t k, int32_t steps, int32_t step0, int32_t mode,
                   const double *x_prop, const double *P0, double *x, double *P, s2m_iter_log *logs, int64_t *iters,
                   int64_t *rematch, double *step_us)
{
    if (!handles || k < 1 || steps < 0 || !x_prop || !P0 || !x || !P || !logs || !iters || !rematch) return S2M_ERR_ARG;
    const size_t xs = S2M_STATE_DOUBLES, ps = (size_t)S2M_DIM * S2M_DIM;
    const int ns = mode == 2 ? 1 : k;  // independent scans per step
    auto t_prev = std::chrono::steady_clock::now();
    for (int s = 0; s < steps; ++s) {
        for (int i = 0; i < k; ++i) {
            int rc = s2m_feat_queue_set(handles[i], nullptr, 0);
            if (rc) return rc;
        }
        for (int i = 0; i < ns; ++i) {
            std::memcpy(x + i * xs, x_prop + i * xs, xs * sizeof(double));
            std::memcpy(P + i * ps, P0 + i * ps, ps * sizeof(double));
            P[i * ps] += (double)((step0 + s) & 1) * 1e-15;
        }
        int rc = S2M_OK;
        if (mode == 1) {
            rc = s2m_iterated_update_batch(handles, k, x, x_prop, P, logs);
        } else if (mode == 2) {
            rc = s2m_iterated_update_multi(handles, k, x, x_prop, P, logs);
        } else {
            for (int i = 0; i < k && rc == S2M_OK; ++i)
                rc = s2m_iterated_update(handles[i], x + i * xs, x_prop + i * xs, P + i * ps, logs + i);
        }
        if (rc) return rc;
        for (int i = 0; i < ns; ++i) {
            *iters += logs[i].iters;
            *rematch += logs[i].rematch_passes;
        }
        if (step_us) {
            const auto t_now = std::chrono::steady_clock::now();
            step_us[s] = std::chrono::duration<double, std::micro>(t_now - t_prev).count();
            t_prev = t_now;
        }
    }
    return S2M_OK;
}

}  // extern "C"
