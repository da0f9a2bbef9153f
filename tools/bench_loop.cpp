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

// The frame leg of bench.py as a C++ caller: `frames` whole frames back to back, one in flight -- raw sweep records on
// the host -> s2m_scan_set_from_raw (undistort + voxel grid) -> s2m_iterated_update -> s2m_map_incremental ->
// s2m_fov_segment, i.e. what the reference's node does per scan (laserMapping.cpp:731-1175) with the engine in place of
// its CPU stages.  frame_us[f] = wall time of frame f on the host's steady clock, merged[f] = how its map update was
// produced (s2m_map_last_update).  x (36) holds the last frame's state on return.
// A different scan every step through ONE handle (vary != 0: scan s % k; vary == 0: always scan 0): what a live stream
// does to the handle's histories -- the bet on an empty far-point list (s2m_config.far_point_bet) is decided by what
// the LAST scan's pass in the same position found, and a benchmark that replays one scan makes that a perfect predictor.
// Every step hands the scan over with s2m_scan_set (device pointer), so both settings pay the same hand-over.
int s2m_bench_loop_varying(s2m_engine *e, int32_t k, const float *const *scans_dev, const int64_t *n, int32_t steps,
                           int32_t vary, const double *x_prop, const double *P0, double *x, double *P, s2m_iter_log *log,
                           int64_t *iters, int64_t *rematch)
{
    if (!e || k < 1 || !scans_dev || !n || steps < 0 || !x_prop || !P0 || !x || !P || !log || !iters || !rematch) return S2M_ERR_ARG;
    const size_t xs = S2M_STATE_DOUBLES, ps = (size_t)S2M_DIM * S2M_DIM;
    for (int s = 0; s < steps; ++s) {
        const int j = vary ? s % k : 0;
        int rc = s2m_scan_set(e, scans_dev[j], 3, n[j], 1);
        if (rc) return rc;
        rc = s2m_feat_queue_set(e, nullptr, 0);
        if (rc) return rc;
        std::memcpy(x, x_prop + j * xs, xs * sizeof(double));
        std::memcpy(P, P0 + j * ps, ps * sizeof(double));
        P[0] += (double)(s & 1) * 1e-15;
        rc = s2m_iterated_update(e, x, x_prop + j * xs, P, log);
        if (rc) return rc;
        *iters += log->iters;
        *rematch += log->rematch_passes;
    }
    return S2M_OK;
}

int s2m_bench_frames(s2m_engine *e, int32_t frames, const float *records, int64_t stride_floats, int64_t n,
                     int32_t time_off_a, int32_t time_off_b, const s2m_imu_pose *poses, int32_t n_poses,
                     const double *state_end, float leaf, const double *x_prop, const double *P0, double filter_size_map,
                     double cube_len, int32_t prefetch, double *x, double *frame_us, int32_t *merged, double *pose_us)
{
    // prefetch: 0 = every frame on its own; 1 = the next sweep's records cross PCIe while this one is registered
    // (s2m_scan_prefetch_raw); 2 = that, and the next frame's undistortion and voxel grid run beside this frame's map
    // update (s2m_scan_prepare_raw, called where the reference's loop has the next message's IMU poses: after the update,
    // before map_incremental).  pose_us[f] (optional) = records in -> pose out of frame f.
    if (!e || frames < 0 || !records || !poses || !state_end || !x_prop || !P0 || !x || !frame_us || !merged) return S2M_ERR_ARG;
    double P[S2M_DIM * S2M_DIM];
    s2m_iter_log log;
    // two untimed frames first (frame index < 0): the handle's side thread, its stream and the host's polling loops are
    // warm when the timed frames start, as in any steady stream
    for (int f = -2; f < frames; ++f) {
        const auto t0 = std::chrono::steady_clock::now();
        int64_t n_out = 0, na = 0, nb = 0;
        int rc = s2m_scan_set_from_raw(e, records, stride_floats, n, time_off_a, time_off_b, poses, n_poses, state_end, leaf, 0, &n_out);
        if (rc) return rc;
        if (prefetch >= 1 && f + 1 < frames) {
            rc = s2m_scan_prefetch_raw(e, records, stride_floats, n, time_off_a, time_off_b);
            if (rc) return rc;
        }
        std::memcpy(x, x_prop, S2M_STATE_DOUBLES * sizeof(double));
        std::memcpy(P, P0, sizeof(P));
        rc = s2m_iterated_update(e, x, x_prop, P, &log);
        if (rc) return rc;
        if (pose_us && f >= 0) pose_us[f] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (prefetch == 2 && f + 1 < frames) {
            rc = s2m_scan_prepare_raw(e, records, stride_floats, n, time_off_a, time_off_b, poses, n_poses, state_end, leaf);
            if (rc) return rc;
        }
        rc = s2m_map_incremental(e, x, filter_size_map, 1, &na, &nb);
        if (rc) return rc;
        rc = s2m_fov_segment(e, x + 9, cube_len, nullptr, nullptr, nullptr);
        if (rc) return rc;
        if (f < 0) continue;
        frame_us[f] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        rc = s2m_map_last_update(e, &merged[f]);
        if (rc) return rc;
    }
    return S2M_OK;
}

}  // extern "C"
