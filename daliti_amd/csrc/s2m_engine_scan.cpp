// s2m_engine_scan.cpp -- the front half of a frame: the scan handed over as it is, through the voxel grid (pcl::VoxelGrid,
// laserMapping.cpp:775-776) or from raw records through undistortion (IMU_Processing.hpp:333-370) and the voxel grid; the
// worker thread and side stream that bring the NEXT sweep over, and prepare it, while the current one is registered.
#include "s2m_engine_internal.h"

using namespace s2m;
using namespace s2m_eng;

extern "C" {

namespace {
int pf_drain(s2m_engine *e);  // the side thread (s2m_scan_prefetch_raw / s2m_scan_prepare_raw) is idle
int pf_invalidate(s2m_engine *e);

int scan_reserve(s2m_engine *e, int64_t n)
{
    if (n <= e->n_cap && e->n_cap > 0) return S2M_OK;
    int rc = pf_drain(e);  // (a prepared scan was laid out for the old capacity: it is recognised as stale when it is picked up)
    if (rc) return rc;
    // (an eighth more than asked: a stream's sweeps differ by a few hundred points, and a larger one must not re-allocate
    // fourteen arrays in the middle of a frame; an empty first scan still gets buffers)
    // ... and a stream whose sweeps keep growing (a sensor driving towards a wall) gets half as much again the second time: the
    // re-allocation is a device-wide stall, 0.3 - 0.6 ms of the frame it falls into (frame 1067 of the bench's drive)
    const int64_t cap = ((std::max<int64_t>(std::max<int64_t>(n + n / 8, e->n_cap + e->n_cap / 2), 1) + 255) / 256) * 256;
    rc = rc ? rc : grow(e, &e->d_scan, 3 * cap);
    rc = rc ? rc : grow(e, &e->d_plane, cap);
    rc = rc ? rc : grow(e, &e->d_flags, cap);
    rc = rc ? rc : grow(e, &e->d_sel, cap);
    rc = rc ? rc : grow(e, &e->d_eff, cap);
    rc = rc ? rc : grow(e, &e->d_pd2, cap);
    rc = rc ? rc : grow(e, &e->d_nn_idx, cap * S2M_K);
    rc = rc ? rc : grow(e, &e->d_nn_d2, cap * S2M_K);
    rc = rc ? rc : grow(e, &e->d_partials, (int64_t)std::max(reduce_blocks((int)cap), 1) * kRedTerms);
    rc = rc ? rc : grow(e, &e->d_block_off, (int64_t)rows_blocks((int)cap) + 1);
    rc = rc ? rc : grow(e, &e->d_hard, 3 * cap + 16);
    rc = rc ? rc : grow(e, &e->d_hrec, 2 * cap);
    if (!rc) S2M_HIP(e, hipMemsetAsync(e->d_hard + 3 * cap, 0, 16 * sizeof(uint32_t), e->stream));
    if (rc) return rc;
    e->n_cap = cap;
    e->rows_cap = 0;
    return S2M_OK;
}

// point_selected_surf(feats_down_size, true) (:812); neighbours invalid until the first rematch
int scan_reset(s2m_engine *e, int64_t n, bool wait = true)
{
    launch_scan_reset(n, e->d_sel, e->d_eff, e->d_flags, e->stream);
    e->step = "the scan's hand-over";
    if (wait) S2M_HIP(e, mail_wait(e->mail, e->stream));  // the host buffer may be reused by the caller now
    e->n = n;
    e->scan_ready = true;
    e->pass_done = false;
    e->nn_valid = false;
    return relay_poll(e);   // (a layout produced beside the frames takes the live map's place between two scans)
}
}  // namespace

int s2m_scan_set(s2m_engine *e, const float *xyz, int64_t stride, int64_t n, int on_device)
{
    if (!e || n < 0 || stride < 3 || (n > 0 && !xyz)) return fail(e, S2M_ERR_ARG, "s2m_scan_set: bad argument");
    if (n > (int64_t)1 << 28) return fail(e, S2M_ERR_CAPACITY, "scan too large");
    S2M_ENTER(e);
    int rc = pf_invalidate(e);  // whatever the side thread holds was meant for a sweep that is not coming by this road
    if (rc) return rc;
    rc = scan_reserve(e, n);
    if (rc) return rc;
    const float *dev = nullptr;
    rc = stage_cloud(e, xyz, stride, n, on_device, &dev);
    if (rc) return rc;
    if (n > 0) launch_deinterleave(dev, stride, n, e->d_scan, e->d_scan + e->n_cap, e->d_scan + 2 * e->n_cap, e->stream);
    return scan_reset(e, n);
}

int s2m_scan_set_downsampled(s2m_engine *e, const float *xyz, int64_t stride, int64_t n, float leaf, int on_device,
                             int64_t *n_out)
{
    if (!e || n < 0 || stride < 3 || (n > 0 && !xyz) || !(leaf > 0.0f)) return fail(e, S2M_ERR_ARG, "s2m_scan_set_downsampled: bad argument");
    if (n > (int64_t)1 << 28) return fail(e, S2M_ERR_CAPACITY, "scan too large");
    S2M_ENTER(e);
    int rc = pf_invalidate(e);  // the voxel-grid buffers are shared with the side thread; what it holds is stale
    if (rc) return rc;
    rc = scan_reserve(e, n);  // the output cannot be larger than the input
    if (rc) return rc;
    const float *dev = nullptr;
    rc = stage_cloud(e, xyz, stride, n, on_device, &dev);
    if (rc) return rc;
    int64_t m = 0;
    bool too_fine = false;
    S2M_HIP(e, voxel_downsample(e->vox, dev, stride, n, leaf, e->d_scan, e->d_scan + e->n_cap, e->d_scan + 2 * e->n_cap,
                                &m, &too_fine, e->stream));
    if (too_fine) return fail(e, S2M_ERR_CAPACITY, "leaf size too small for the cloud extent (voxel index overflows int32)");
    if (n_out) *n_out = m;
    return scan_reset(e, m, n == 0);  // (the voxel count came back through the mailbox: the caller's buffer has been read)
}

namespace {
int check_undistort_args(s2m_engine *e, const float *points, int64_t stride, int64_t n, int32_t oa, int32_t ob,
                         const s2m_imu_pose *poses, int32_t np, const double *state_end)
{
    if (!e || n < 0 || stride < 3 || (n > 0 && !points) || !poses || np < 1 || !state_end)
        return fail(e, S2M_ERR_ARG, "undistort: bad argument");
    if (oa < 0 || oa >= stride || ob >= stride) return fail(e, S2M_ERR_ARG, "undistort: time offsets outside the record");
    if (n > (int64_t)1 << 28) return fail(e, S2M_ERR_CAPACITY, "scan too large");
    static_assert(sizeof(s2m_imu_pose) == 22 * sizeof(double), "s2m_imu_pose must be 22 packed doubles");
    return S2M_OK;
}
}  // namespace

int s2m_undistort(s2m_engine *e, const float *points, int64_t stride, int64_t n, int32_t oa, int32_t ob,
                  const s2m_imu_pose *poses, int32_t np, const double state_end[S2M_STATE_DOUBLES], int sort_by_time,
                  int on_device, float *out_xyz, uint32_t *perm)
{
    int rc = check_undistort_args(e, points, stride, n, oa, ob, poses, np, state_end);
    if (rc) return rc;
    if (n > 0 && !out_xyz) return fail(e, S2M_ERR_ARG, "undistort: null output");
    if (n == 0) return S2M_OK;
    S2M_ENTER(e);
    rc = pf_drain(e);  // the undistortion buffers are shared with the side thread
    if (rc) return rc;
    e->pf.ordered = false;  // (a time order the side thread left there is about to be overwritten)
    const float *dev = nullptr;
    // stage whole records (the time fields may sit anywhere in the record)
    if (on_device) {
        dev = points;
    } else {
        const int64_t floats = n * stride;
        if (floats > e->stage_cap) {
            rc = grow(e, &e->d_stage, floats + floats / 8);  // (room for the next, slightly larger sweep)
            if (rc) return rc;
            e->stage_cap = floats + floats / 8;
        }
        S2M_HIP(e, hipMemcpyAsync(e->d_stage, points, (size_t)floats * sizeof(float), hipMemcpyHostToDevice, e->stream));
        dev = e->d_stage;
    }
    uint32_t *d_perm = nullptr;
    if (perm) {  // kept across calls (sized with the other undistort buffers)
        if (e->und.perm_cap < n) {
            if (e->und.perm) S2M_HIP(e, hipFree(e->und.perm));
            e->und.perm = nullptr;
            e->und.perm_cap = 0;
            S2M_HIP(e, hipMalloc((void **)&e->und.perm, (size_t)n * sizeof(uint32_t)));
            e->und.perm_cap = n;
        }
        d_perm = e->und.perm;
    }
    hipError_t he = undistort(e->und, dev, stride, n, oa, ob, reinterpret_cast<const double *>(poses), np,
                              pose_of(state_end), sort_by_time != 0, d_perm, e->stream);
    if (he == hipSuccess)
        he = hipMemcpyAsync(out_xyz, e->und.out, (size_t)n * 3 * sizeof(float),
                            on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, e->stream);
    if (he == hipSuccess && perm) he = hipMemcpyAsync(perm, d_perm, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream);
    if (he == hipSuccess) he = wait_stream(&e->wait, e->stream, "the undistorted points on their way to the caller");
    if (he != hipSuccess) return fail(e, S2M_ERR_HIP, "undistort", he);
    return S2M_OK;
}

namespace {
// The side thread of a handle: brings the next sweep's records over (s2m_scan_prefetch_raw) and, when asked
// (s2m_scan_prepare_raw), also undistorts and down-samples them into the spare scan arrays -- on its own stream, with the
// undistortion / voxel-grid buffers and mailboxes the main thread only touches through the scan_set entry points, which
// wait for this thread first (pf_drain).
void prefetch_worker(s2m_engine *e)
{
    auto &p = e->pf;
    (void)hipSetDevice(e->device);
    tl_wait = &e->wait;  // this thread's waits (the voxel grid's and the time sort's hand-backs) end at the handle's deadline too
    struct Exited {
        std::atomic<int> &f;
        ~Exited() { f.store(1, std::memory_order_release); }
    } exited{p.exited};
    std::unique_lock<std::mutex> lk(p.mu);
    for (;;) {
        if (!(p.quit || p.busy)) {  // the next job usually follows within a frame: poll for it before going to sleep
            lk.unlock();
            for (int spin = 0; spin < 20000 && p.busy_a.load(std::memory_order_acquire) == 0; ++spin) __builtin_ia32_pause();
            lk.lock();
        }
        p.cv.wait(lk, [&] { return p.quit || p.busy; });  // (idle: the only wait of the handle without a deadline)
        if (p.quit) return;
        const float *src = p.src;
        const int64_t floats = p.floats;
        const bool prepare = p.prepare;
        lk.unlock();
        if (e->wait.withhold(kStallWorker)) {  // fault injection: this job is never finished (the thread still leaves when told to)
            while (p.quit_a.load(std::memory_order_acquire) == 0) {
                struct timespec ts = {0, 1000000L};
                nanosleep(&ts, nullptr);
            }
            return;
        }
        hipError_t he = p.copied ? hipSuccess
                                 : hipMemcpyAsync(p.d_buf, src, (size_t)floats * sizeof(float), hipMemcpyHostToDevice, p.stream);
        int64_t m = p.n;
        bool ok = he == hipSuccess;
        bool ordered = p.copied && p.ordered;
        if (ok && !prepare && p.want_order) {  // the time order needs the records only: it is ready when the poses arrive
            he = undistort_order(e->und, p.d_buf, p.stride, p.n, p.order_oa, p.order_ob, p.stream);
            ok = he == hipSuccess;
            ordered = ok;
        }
        if (ok && prepare) {
            const bool have_order = ordered && p.order_oa == p.oa && p.order_ob == p.ob;
            he = undistort(e->und, p.d_buf, p.stride, p.n, p.oa, p.ob, p.poses.data(), (int)(p.poses.size() / 22), pose_of(p.state_end),
                           true, nullptr, p.stream, have_order);
            ordered = false;  // (the voxel grid and the next sort reuse the buffers)
            ok = he == hipSuccess;
            float *sx = e->d_scan_alt, *sy = e->d_scan_alt + e->scan_alt_cap, *sz = e->d_scan_alt + 2 * e->scan_alt_cap;
            if (ok && p.leaf > 0.0f) {
                bool too_fine = false;
                he = voxel_downsample(e->vox, e->und.out, 3, p.n, p.leaf, sx, sy, sz, &m, &too_fine, p.stream);
                ok = he == hipSuccess && !too_fine;  // a refusal is reported by the synchronous path, which runs instead
            } else if (ok) {
                launch_deinterleave(e->und.out, 3, p.n, sx, sy, sz, p.stream);
            }
        }
        if (he == hipSuccess) he = hipEventRecord(p.done, p.stream);
        lk.lock();
        p.err = he;
        p.m = m;
        p.ordered = ordered && he == hipSuccess;
        p.busy = false;
        p.busy_a.store(0, std::memory_order_release);
        p.ready = he == hipSuccess && !prepare;
        p.prepared = ok && he == hipSuccess && prepare;
        p.gpu_pending = he == hipSuccess;
        p.cv.notify_all();
    }
}

// the side thread is idle AND whatever its last job enqueued on the side stream is ordered in front of everything the
// caller enqueues on the main stream from here on: the undistortion / voxel-grid scratch (e->und, e->vox) is shared by the
// two streams, so a caller that goes on to sort in it must not overtake a job's kernels that are still running
int pf_drain(s2m_engine *e)
{
    auto &p = e->pf;
    if (!p.worker.joinable()) return S2M_OK;
    // (busy_a mirrors `busy`, both written under the mutex: polled under the handle's wait policy and deadline -- a job is a
    // copy and a dozen launches; a side thread that does not come back is stuck in the runtime or behind a dead device, and
    // the caller must hear of it)
    if (p.busy_a.load(std::memory_order_acquire) != 0) {
        e->step = "the side thread's job";
        if (!wait_until(&e->wait, [&] { return p.busy_a.load(std::memory_order_acquire) == 0; },
                        "the handle's side thread to finish its job (copy, undistortion, voxel grid of the next sweep)"))
            return fail(e, S2M_ERR_HIP, "pf_drain", kWaitTimedOut);
    }
    std::unique_lock<std::mutex> lk(p.mu);  // (orders this thread behind the job's last writes)
    if (p.busy) return fail(e, S2M_ERR_STATE, "pf_drain: the side thread took a job nobody gave it");
    if (p.err == kWaitTimedOut) return fail(e, S2M_ERR_HIP, "the side thread's job", kWaitTimedOut);  // a wait of the side thread itself expired
    if (p.gpu_pending && p.done) {
        (void)hipStreamWaitEvent(e->stream, p.done, 0);
        p.gpu_pending = false;
    }
    return S2M_OK;
}
// a scan arrives by another road than the one the side thread prepared for: what it holds is stale (a node that recycles
// its host buffers may present NEW records at the address and size of a sweep that was prefetched and then dropped)
int pf_invalidate(s2m_engine *e)
{
    int rc = pf_drain(e);
    if (rc) return rc;
    e->pf.ready = false;
    e->pf.prepared = false;
    e->pf.ordered = false;
    return S2M_OK;
}

int pf_start(s2m_engine *e, const float *points, int64_t floats)
{
    auto &p = e->pf;
    int rc = pf_drain(e);
    if (rc) return rc;
    p.ready = false;
    p.prepared = false;
    if (!p.stream) {
        // The side stream must run BESIDE the main stream.  The runtime maps every new stream onto the least used of a few hardware
        // queues per priority; a side stream of the main stream's priority, created when the first sweep is announced -- i.e. after
        // whatever other handles and streams the process has made in between --, has been seen to land on the main stream's own
        // queue and to turn the frame pipeline into a sequence (0.30 -> 0.44 ms per frame); created together with the main stream
        // it took the queues the launch groups of a batch of handles need to overlap (25.9 k -> 22.1 k scans/s at 24 in flight).
        // A stream of another priority lives in another pool of queues: it can share one with neither.  (S2M_PF_STREAM=normal: the
        // main stream's priority, for A/B runs; profiles/r06_side_stream_ab.txt.)
        static const bool normal = [] { const char *g = std::getenv("S2M_PF_STREAM"); return g && std::string(g) == "normal"; }();
        int least = 0, greatest = 0;
        if (normal || hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess ||
            hipStreamCreateWithPriority(&p.stream, hipStreamNonBlocking, least) != hipSuccess) {
            (void)hipGetLastError();
            S2M_HIP(e, hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking));
        }
    }
    if (!p.done) S2M_HIP(e, hipEventCreateWithFlags(&p.done, hipEventDisableTiming));
    if (floats > p.cap) {
        rc = sync_stream(e, p.stream, "the side stream, before its buffer is reallocated");
        if (rc) return rc;
        if (p.d_buf) S2M_HIP(e, hipFree(p.d_buf));
        p.d_buf = nullptr;
        // (an eighth more than asked: sweeps differ by a few hundred returns, and a reallocation -- a device-wide stall of
        // ~0.3 ms -- every time a slightly larger one arrives showed up as the worst frame of a drive)
        const int64_t want = std::max<int64_t>(floats + floats / 8, p.cap + p.cap / 2);   // (half as much again when it has to grow a second time)
        S2M_HIP(e, hipMalloc((void **)&p.d_buf, (size_t)want * sizeof(float)));
        p.cap = want;
    }
    if (!p.worker.joinable()) p.worker = std::thread(prefetch_worker, e);
    p.src = points;
    p.floats = floats;
    return S2M_OK;
}
}  // namespace

int s2m_scan_prefetch_raw(s2m_engine *e, const float *points, int64_t stride, int64_t n, int32_t oa, int32_t ob)
{
    if (!e) return S2M_ERR_ARG;
    if (!points) {  // cancel: the sweep that was announced is not coming (dropped, skipped): forget its copy
        if (e->poisoned) return refuse_poisoned(e);
        tl_wait = &e->wait;
        return pf_invalidate(e);
    }
    if (n < 0 || stride < 3 || oa >= stride || ob >= stride) return fail(e, S2M_ERR_ARG, "s2m_scan_prefetch_raw: bad argument");
    if (n == 0) return S2M_OK;
    S2M_ENTER(e);
    int rc = pf_start(e, points, n * stride);
    if (rc) return rc;
    {
        std::lock_guard<std::mutex> lk(e->pf.mu);
        e->pf.prepare = false;
        e->pf.copied = false;
        e->pf.want_order = oa >= 0;
        e->pf.ordered = false;
        e->pf.order_oa = oa; e->pf.order_ob = ob;
        e->pf.stride = stride; e->pf.n = n;
        e->pf.busy = true;
        e->pf.busy_a.store(1, std::memory_order_release);
    }
    e->pf.cv.notify_all();
    return S2M_OK;
}

int s2m_scan_prepare_raw(s2m_engine *e, const float *points, int64_t stride, int64_t n, int32_t oa, int32_t ob,
                         const s2m_imu_pose *poses, int32_t np, const double state_end[S2M_STATE_DOUBLES], float leaf)
{
    int rc = check_undistort_args(e, points, stride, n, oa, ob, poses, np, state_end);
    if (rc) return rc;
    if (n == 0) return S2M_OK;
    S2M_ENTER(e);
    // the prepared scan must fit the arrays of the current one (they are swapped, not copied); a first or larger sweep is
    // left to the synchronous call
    if (e->n_cap < n) return S2M_OK;
    rc = pf_drain(e);
    if (rc) return rc;
    // the records may be on the device already: s2m_scan_prefetch_raw at the start of the frame, this call once the poses exist
    const bool have = e->pf.ready && e->pf.src == points && e->pf.floats == n * stride;
    const bool have_order = have && e->pf.ordered;
    rc = pf_start(e, points, n * stride);
    if (rc) return rc;
    e->pf.ordered = have_order;
    if (e->scan_alt_cap != e->n_cap) {
        if (e->d_scan_alt) S2M_HIP(e, hipFree(e->d_scan_alt));
        e->d_scan_alt = nullptr;
        e->scan_alt_cap = 0;
        S2M_HIP(e, hipMalloc((void **)&e->d_scan_alt, (size_t)3 * e->n_cap * sizeof(float)));
        e->scan_alt_cap = e->n_cap;
    }
    auto &p = e->pf;
    p.stride = stride; p.n = n; p.oa = oa; p.ob = ob; p.leaf = leaf;
    p.poses.assign(reinterpret_cast<const double *>(poses), reinterpret_cast<const double *>(poses) + (size_t)np * 22);
    std::memcpy(p.state_end, state_end, sizeof(p.state_end));
    {
        std::lock_guard<std::mutex> lk(p.mu);
        p.prepare = true;
        p.copied = have;
        p.busy = true;
        p.busy_a.store(1, std::memory_order_release);
    }
    p.cv.notify_all();
    return S2M_OK;
}

int s2m_scan_set_from_raw(s2m_engine *e, const float *points, int64_t stride, int64_t n, int32_t oa, int32_t ob,
                          const s2m_imu_pose *poses, int32_t np, const double state_end[S2M_STATE_DOUBLES], float leaf,
                          int on_device, int64_t *n_out)
{
    int rc = check_undistort_args(e, points, stride, n, oa, ob, poses, np, state_end);
    if (rc) return rc;
    S2M_ENTER(e);
    rc = pf_drain(e);
    if (rc) return rc;
    if (n > 0 && !on_device && e->pf.prepared) {  // has s2m_scan_prepare_raw done exactly this call already?
        auto &p = e->pf;
        const bool same = p.src == points && p.stride == stride && p.n == n && p.oa == oa && p.ob == ob && p.leaf == leaf &&
                          p.poses.size() == (size_t)np * 22 && std::memcmp(p.poses.data(), poses, p.poses.size() * sizeof(double)) == 0 &&
                          std::memcmp(p.state_end, state_end, sizeof(p.state_end)) == 0 && e->scan_alt_cap == e->n_cap && p.m <= e->n_cap;
        p.prepared = false;  // consumed or stale
        if (same) {
            S2M_HIP(e, hipStreamWaitEvent(e->stream, p.done, 0));
            std::swap(e->d_scan, e->d_scan_alt);
            if (n_out) *n_out = p.m;
            return scan_reset(e, p.m, false);  // no host buffer is in flight: nothing to wait for
        }
    }
    rc = scan_reserve(e, n);
    if (rc) return rc;
    if (n > 0) {
        const float *dev = points;
        bool prefetched = false, order_ready = false;
        if (!on_device && e->pf.worker.joinable()) {  // has s2m_scan_prefetch_raw brought exactly these records over already?
            auto &p = e->pf;
            std::unique_lock<std::mutex> lk(p.mu);
            if (p.src == points && p.floats == n * stride && !p.busy) {  // (pf_drain above: the side thread is idle)
                if (p.ready) {
                    p.ready = false;  // consumed
                    order_ready = p.ordered && p.order_oa == oa && p.order_ob == ob;
                    p.ordered = false;
                    lk.unlock();
                    S2M_HIP(e, hipStreamWaitEvent(e->stream, p.done, 0));
                    dev = p.d_buf;
                    prefetched = true;
                }
            }
        }
        if (!prefetched) {  // other records, or records on the device: a copy the side thread still holds is stale, and
            e->pf.ordered = false;  // this call's own sort overwrites the order it may have left
            e->pf.ready = false;
        }
        if (!on_device && !prefetched) {
            const int64_t floats = n * stride;
            if (floats > e->stage_cap) {
                rc = grow(e, &e->d_stage, floats + floats / 8);  // (room for the next, slightly larger sweep)
                if (rc) return rc;
                e->stage_cap = floats + floats / 8;
            }
            S2M_HIP(e, hipMemcpyAsync(e->d_stage, points, (size_t)floats * sizeof(float), hipMemcpyHostToDevice, e->stream));
            dev = e->d_stage;
        }
        S2M_HIP(e, undistort(e->und, dev, stride, n, oa, ob, reinterpret_cast<const double *>(poses), np,
                             pose_of(state_end), true, nullptr, e->stream, order_ready));
    }
    int64_t m = n;
    float *sx = e->d_scan, *sy = e->d_scan + e->n_cap, *sz = e->d_scan + 2 * e->n_cap;
    bool synced = false;  // the host has already waited for something behind the copy of the caller's buffer
    if (leaf > 0.0f && n > 0) {
        bool too_fine = false;
        S2M_HIP(e, voxel_downsample(e->vox, e->und.out, 3, n, leaf, sx, sy, sz, &m, &too_fine, e->stream));
        if (too_fine) return fail(e, S2M_ERR_CAPACITY, "leaf size too small for the cloud extent (voxel index overflows int32)");
        synced = true;  // (the voxel count came back through the mailbox: everything before it has finished)
    } else if (n > 0) {
        launch_deinterleave(e->und.out, 3, n, sx, sy, sz, e->stream);
    }
    if (n_out) *n_out = m;
    return scan_reset(e, m, !synced);
}

int s2m_scan_get(s2m_engine *e, float *xyz, int64_t capacity, int64_t *n)
{
    if (!e || !n) return fail(e, S2M_ERR_ARG, "null argument");
    if (!e->scan_ready) return fail(e, S2M_ERR_STATE, "no scan");
    *n = e->n;
    if (!xyz || e->n == 0) return S2M_OK;
    if (capacity < e->n) return fail(e, S2M_ERR_CAPACITY, "scan buffer too small");
    S2M_ENTER(e);
    int rc = sync_stream(e, e->stream, "the stream, before the scan is copied out");
    if (rc) return rc;
    // three strided copies SoA -> packed AoS
    for (int k = 0; k < 3; ++k)
        S2M_HIP(e, hipMemcpy2D(xyz + k, 3 * sizeof(float), e->d_scan + k * e->n_cap, sizeof(float), sizeof(float),
                               (size_t)e->n, hipMemcpyDeviceToHost));
    return S2M_OK;
}


}  // extern "C"
