// bench_loop.cpp -- the timed region of bench.py as a C++ caller of the C ABI (include/daliti_s2m.h).
//
// The reference's caller is C++ (eskf_lio/src/laserMapping.cpp:731, one iterated update per scan inside
// `while (sync_packages(Measures))`); a step of the benchmark is exactly what that loop body does with the
// engine: hand it the propagated state and covariance, run s2m_iterated_update, read the result.  Driving
// the K timed steps from here instead of from Python keeps the interpreter's ~10 us per call out of a
// 150 us step.  Plain g++, links only libdaliti_s2m.so; built by __graft_entry__.build() as
// daliti_amd/_lib/libs2m_benchloop.so and loaded by bench.py with ctypes.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include "daliti_s2m.h"
#include "daliti_s2m_mirror.hpp"
#include "world.h"

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


// ---- the moving-trajectory workload (tools/world.h) ----------------------------------------------------------------------
// `count` consecutive sweeps starting at frame f0, generated on `threads` host threads: sweep k's records at
// rec + k * rec_stride_floats (room for beams * az records of 12 floats), its record count in n_out[k], its IMU poses at
// poses + k * n_poses, its predicted / true end-of-sweep states at x_prop / x_true + 36 k.
int s2m_world_sweeps(const s2m_world *w, int32_t f0, int32_t count, int32_t beams, int32_t az, float *rec, int64_t rec_stride_floats,
                     int64_t *n_out, s2m_imu_pose *poses, int32_t n_poses, double *x_prop, double *x_true, int32_t threads)
{
    if (!w || count < 0 || beams < 1 || az < 1 || !rec || !n_out || !poses || n_poses < 1 || !x_prop || !x_true) return S2M_ERR_ARG;
    if (rec_stride_floats < (int64_t)beams * az * 12) return S2M_ERR_ARG;
    const int nt = threads < 1 ? 1 : (threads > 64 ? 64 : threads);
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; ++t)
        pool.emplace_back([=] {
            for (int k = t; k < count; k += nt)
                n_out[k] = s2m_world_sweep_impl(*w, f0 + k, beams, az, rec + (int64_t)k * rec_stride_floats, poses + (int64_t)k * n_poses, n_poses,
                                                x_prop + (int64_t)k * S2M_STATE_DOUBLES, x_true + (int64_t)k * S2M_STATE_DOUBLES);
        });
    for (auto &th : pool) th.join();
    return S2M_OK;
}
int s2m_world_seed(const s2m_world *w, double span, int64_t m, float *xyz)
{
    if (!w || m < 0 || !xyz || !(span > 0.0)) return S2M_ERR_ARG;
    s2m_world_seed_impl(*w, span, m, xyz);
    return S2M_OK;
}

// A stall reporter for the frame loops: the loop leaves its frame and call numbers in two atomics, a side thread looks once a
// second and says on stderr where the loop stands -- and what the handle says it is doing (s2m_debug_state: streams, side
// thread, every hand-back word against the number the host expects) -- when nothing has moved for two and for ten seconds (a
// frame takes well under a millisecond; a stall seen once on a shared box left no trace of where it was).  Since round 6 every
// wait inside the engine ends at its deadline with S2M_ERR_TIMEOUT, so the loop itself returns; the reporter stays for waits
// outside the engine.  Costs the loop two relaxed stores per call.
namespace {
struct StallReporter {
    std::atomic<int64_t> beat{0};
    std::atomic<int32_t> frame{-1}, call{-1};
    std::mutex mu;
    std::condition_variable cv;
    bool stop = false;
    std::thread th;
    const char *const *names;
    s2m_engine *eng;
    StallReporter(const char *const *call_names, s2m_engine *e) : names(call_names), eng(e)
    {
        th = std::thread([this] {
            int64_t seen = -1;
            int idle = 0;
            std::unique_lock<std::mutex> lk(mu);
            while (!cv.wait_for(lk, std::chrono::seconds(1), [this] { return stop; })) {
                const int64_t b = beat.load(std::memory_order_relaxed);
                idle = b == seen ? idle + 1 : 0;
                seen = b;
                if (idle == 2 || idle == 10 || (idle > 10 && idle % 60 == 0)) {
                    const int c = call.load(std::memory_order_relaxed);
                    char st[2048] = "";
                    if (eng) (void)s2m_debug_state(eng, st, (int64_t)sizeof(st));  // (reads only: callable beside the loop's thread)
                    std::fprintf(stderr, "[bench_loop] no progress for %d s: frame %d, inside %s; %s\n", idle, frame.load(std::memory_order_relaxed),
                                 c >= 0 ? names[c] : "(start)", st);
                    std::fflush(stderr);
                }
            }
        });
    }
    void at(int32_t f, int32_t c)
    {
        frame.store(f, std::memory_order_relaxed);
        call.store(c, std::memory_order_relaxed);
        beat.fetch_add(1, std::memory_order_relaxed);
    }
    ~StallReporter()
    {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        th.join();
    }
};
}  // namespace

// What the reference's node does per scan (laserMapping.cpp:731-1175) along a trajectory: every frame its own sweep, poses
// and predicted state -- raw records on the host -> s2m_scan_set_from_raw (undistort + voxel grid) -> s2m_iterated_update
// from the frame's predicted state -> s2m_map_incremental at the updated state -> s2m_fov_segment at the LiDAR position.
// Frames 0 .. warm - 1 are part of the drive but not timed.  frame_us[f], how[f] (0 = map rebuilt, 1 = map re-laid by a
// merge, 2 = updated in place), deleted[f] (points the field-of-view trim removed), n_scan[f] (points after the voxel
// grid), allocs[f] (device buffers the map code (re)allocated during the frame), x_out (36 per frame: the updated state)
// for f = 0 .. warm + frames - 1.  prefetch as in s2m_bench_frames.
int s2m_bench_frames_moving(s2m_engine *e, int32_t frames, int32_t warm, const float *rec, int64_t rec_stride_floats, const int64_t *n,
                            int32_t time_off_a, int32_t time_off_b, const s2m_imu_pose *poses, int32_t n_poses, const double *x_prop,
                            const double *P0, float leaf, double filter_size_map, double cube_len, int32_t prefetch, double *x_out,
                            double *frame_us, int32_t *how, int64_t *deleted, int64_t *n_scan, s2m_iter_log *logs, int32_t *allocs,
                            int32_t publish, double *publish_us, int64_t *mirror_stats, double *stage_us, double *fetch_us)
{
    if (!e || frames < 0 || warm < 0 || !rec || !n || !poses || !x_prop || !P0 || !x_out || !frame_us || !how) return S2M_ERR_ARG;
    double P[S2M_DIM * S2M_DIM];
    s2m_iter_log log;
    const int total = warm + frames;
    int64_t inplace_before = 0;
    int rc = s2m_map_inplace_updates(e, &inplace_before);
    if (rc) return rc;
    int64_t st_prev[12] = {0};
    (void)s2m_map_update_stats(e, st_prev);
    // publish != 0: the node also keeps /Laser_map up to date every frame (laserMapping.cpp:1170-1175, 1229-1235) -- a host
    // mirror fed by s2m_map_get_changes (daliti_s2m_mirror.hpp); publish_us[f] = that call's share of the frame
    s2m_map_mirror mirror;
    // publish: 1 = the map as it is now (the call waits for the device), 2 = one call behind (nobody waits), 3 = one call behind
    // and the report applied by a publisher thread (the node's publishing thread): the frame pays for the fetch only
    mirror.lag = publish >= 2 ? 1 : 0;
    struct Publisher {   // applies the reports the frame loop hands over, one at a time
        s2m_map_mirror &m;
        std::mutex mu;
        std::condition_variable cv;
        int job = -1;
        bool busy = false, stop = false;
        std::thread th;
        explicit Publisher(s2m_map_mirror &mm) : m(mm)
        {
            th = std::thread([this] {
                std::unique_lock<std::mutex> lk(mu);
                for (;;) {
                    cv.wait(lk, [this] { return stop || job >= 0; });
                    if (stop) return;
                    const int k = job;
                    job = -1;
                    lk.unlock();
                    m.apply_report(k);
                    lk.lock();
                    busy = false;
                    cv.notify_all();
                }
            });
        }
        void idle() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [this] { return !busy; }); }
        void give(int k) { { std::lock_guard<std::mutex> lk(mu); job = k; busy = true; } cv.notify_all(); }
        ~Publisher() { idle(); { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); th.join(); }
    };
    std::unique_ptr<Publisher> pub;
    if (publish == 3) pub.reset(new Publisher(mirror));
    if (publish) {
        rc = mirror.update(e);  // the one whole-map fetch, before the drive
        if (rc) return rc;
    }
    static const char *const call_names[7] = {"s2m_scan_set_from_raw", "s2m_scan_prefetch_raw", "s2m_iterated_update", "s2m_scan_prepare_raw",
                                              "s2m_map_incremental", "s2m_fov_segment", "the map mirror"};
    StallReporter stall(call_names, e);
    for (int f = 0; f < total; ++f) {
        const auto t0 = std::chrono::steady_clock::now();
        const float *r = rec + (int64_t)f * rec_stride_floats;
        const s2m_imu_pose *ps = poses + (int64_t)f * n_poses;
        const double *xp = x_prop + (int64_t)f * S2M_STATE_DOUBLES;
        double *x = x_out + (int64_t)f * S2M_STATE_DOUBLES;
        int64_t n_out = 0, na = 0, nb = 0, nd = 0;
        static const bool progress = std::getenv("S2M_PROGRESS") != nullptr;  // (hunting a stall: which frame, which call)
        auto say = [&](const char *what) { if (progress) { std::fprintf(stderr, "[frame %d] %s\n", f, what); std::fflush(stderr); } };
        say("begin");
        auto lap = [&](int k) {  // host wall time of the frame's calls, in order (stage_us: 6 per frame, optional)
            static const char *names[6] = {"set_from_raw", "prefetch", "iterated_update", "prepare", "map_incremental", "fov"};
            say(names[k]);
            if (stage_us) stage_us[(int64_t)f * 6 + k] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        };
        stall.at(f, 0);
        rc = s2m_scan_set_from_raw(e, r, 12, n[f], time_off_a, time_off_b, ps, n_poses, xp, leaf, 0, &n_out);
        if (rc) return rc;
        lap(0);
        if (prefetch >= 1 && f + 1 < total) {
            stall.at(f, 1);
            rc = s2m_scan_prefetch_raw(e, r + rec_stride_floats, 12, n[f + 1], time_off_a, time_off_b);
            if (rc) return rc;
        }
        std::memcpy(x, xp, S2M_STATE_DOUBLES * sizeof(double));
        std::memcpy(P, P0, sizeof(P));
        P[0] += (double)(f & 1) * 1e-15;
        lap(1);
        stall.at(f, 2);
        rc = s2m_iterated_update(e, x, xp, P, &log);
        if (rc) return rc;
        lap(2);
        if (logs) logs[f] = log;
        if (prefetch == 2 && f + 1 < total) {
            stall.at(f, 3);
            rc = s2m_scan_prepare_raw(e, r + rec_stride_floats, 12, n[f + 1], time_off_a, time_off_b, ps + n_poses, n_poses,
                                      xp + S2M_STATE_DOUBLES, leaf);
            if (rc) return rc;
        }
        lap(3);
        stall.at(f, 4);
        rc = s2m_map_incremental(e, x, filter_size_map, 1, &na, &nb);
        if (rc) return rc;
        lap(4);
        int32_t merged = 0;
        rc = s2m_map_last_update(e, &merged);
        if (rc) return rc;
        int64_t inplace_now = 0;
        rc = s2m_map_inplace_updates(e, &inplace_now);
        if (rc) return rc;
        int32_t h = !merged ? 0 : (inplace_now > inplace_before ? 2 : 1);
        inplace_before = inplace_now;
        // pos_lid = pos_end + rot_end * T_L_I (laserMapping.cpp:753)
        double lid[3];
        for (int k = 0; k < 3; ++k) lid[k] = x[9 + k] + (x[3 * k] * x[21] + x[3 * k + 1] * x[22] + x[3 * k + 2] * x[23]);
        stall.at(f, 5);
        rc = s2m_fov_segment(e, lid, cube_len, nullptr, nullptr, &nd);
        if (rc) return rc;
        if (nd > 0) {  // the trim changed the map too: how that update was produced counts for the frame
            rc = s2m_map_last_update(e, &merged);
            if (rc) return rc;
            rc = s2m_map_inplace_updates(e, &inplace_now);
            if (rc) return rc;
            const int32_t h2 = !merged ? 0 : (inplace_now > inplace_before ? 2 : 1);
            inplace_before = inplace_now;
            if (h2 < h) h = h2;
        }
        lap(5);
        if (publish) {
            const auto tp = std::chrono::steady_clock::now();
            stall.at(f, 6);
            if (pub) pub->idle();    // (the previous report has been applied: its buffer is free again in two fetches' time, this waits for nothing)
            rc = mirror.fetch(e);    // the engine's side of it: what changed, out of pinned memory into the mirror's arrays
            if (rc) return rc;
            if (fetch_us) fetch_us[f] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tp).count();
            const int64_t missed_before = mirror.missed;
            if (pub) pub->give(mirror.hand_over());   // the follower's side on the publisher's thread ...
            else mirror.apply();                      // ... or here
            if (mirror.missed != missed_before) {
                std::fprintf(stderr, "[bench_loop] frame %d: the follower missed %lld removals (added %lld removed %lld boxes %lld in this report)\n", f,
                             (long long)(mirror.missed - missed_before), (long long)mirror.last_added, (long long)mirror.last_removed, (long long)mirror.last_boxes);
                for (size_t k = (size_t)std::min<int64_t>(missed_before, 64); k < mirror.missed_ids.size(); ++k) {
                    float at[3] = {0, 0, 0};
                    const bool have = mirror.find_anywhere(mirror.missed_ids[k], at);
                    std::fprintf(stderr, "    id %u reported at (%.4f %.4f %.4f); the mirror holds it %s (%.4f %.4f %.4f)\n", mirror.missed_ids[k], mirror.missed_xyz[3 * k],
                                 mirror.missed_xyz[3 * k + 1], mirror.missed_xyz[3 * k + 2], have ? "at" : "nowhere", at[0], at[1], at[2]);
                }
            }
            if (publish_us) publish_us[f] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tp).count();
        }
        frame_us[f] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        how[f] = h;
        if (deleted) deleted[f] = nd;
        if (n_scan) n_scan[f] = n_out;
        if (allocs) {  // device buffers (re)allocated by the map code during this frame (each one stalls the stream)
            int64_t st_now[12] = {0};
            (void)s2m_map_update_stats(e, st_now);
            allocs[f] = (int32_t)(st_now[3] - st_prev[3]);
            st_prev[3] = st_now[3];
        }
    }
    if (pub) pub.reset();
    if (publish && mirror.lag != 0) {  // (untimed: a follower that is one call behind catches up before it is compared with the map)
        mirror.lag = 0;
        rc = mirror.update(e);
        if (rc) return rc;
    }
    if (publish && mirror_stats) {
        int64_t m = 0;
        (void)s2m_map_size(e, &m);
        mirror_stats[0] = mirror.size();
        mirror_stats[1] = m;
        mirror_stats[2] = mirror.resyncs;
        mirror_stats[3] = mirror.missed;
        if (mirror.missed > 0) {   // (the follower is out of step: say which removals found nothing)
            std::fprintf(stderr, "[bench_loop] the map's follower missed %lld removals; first ids:", (long long)mirror.missed);
            for (uint32_t id : mirror.missed_ids) std::fprintf(stderr, " %u", id);
            std::fprintf(stderr, "\n");
        }
    }
    return S2M_OK;
}

}  // extern "C"
