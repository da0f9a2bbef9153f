// s2m_engine_loop.cpp -- residual / Jacobian passes (laserMapping.cpp:829-979 fused with :1015-1032) and the iterated ESKF
// update (:820-1102) in its four forms: one scan on one handle, K scans in one grid per pass, one scan on n handles with
// the blocks summed by the host, one scan per rank with a collective between the ranks.
#include "s2m_engine_internal.h"

using namespace s2m;
using namespace s2m_eng;

namespace {

// defer_publish: the caller sums d_out over the ranks first and publishes the result itself (launch_publish);
// every other caller gets the block and the flag in pinned host memory straight from the reduce kernel, with or
// without a communicator attached to the handle.
int run_pass(s2m_engine *e, const double state[S2M_STATE_DOUBLES], int rematch, double *d_out, bool defer_publish = false,
             bool skip_far = false)
{
    if (!e || !state) return fail(e, S2M_ERR_ARG, "null argument");
    if (!e->map_ready) return fail(e, S2M_ERR_STATE, "no map: call s2m_map_build first");
    if (!e->scan_ready) return fail(e, S2M_ERR_STATE, "no scan: call s2m_scan_set first");
    if (!rematch && !e->nn_valid) return fail(e, S2M_ERR_STATE, "first pass of a scan must be a rematch pass");
    S2M_INNER(e);
    const Pose pose = pose_of(state);
    const Gates gates = gates_of(e->cfg);
    const int n = (int)e->n;
    float *sx = e->d_scan, *sy = e->d_scan + e->n_cap, *sz = e->d_scan + 2 * e->n_cap;
    e->last_rematch = rematch != 0;
    // HIP events on the engine's stream around the search kernels and around the reduce kernel of every
    // stride-th pass (a stride coprime to the passes per scan walks through every kind of pass)
    bool timed = e->timing;
    if (timed && e->timing_stride > 1) timed = (e->timing_phase++ % e->timing_stride) == 0;
    e->timed_this_pass = timed;
    if (timed) S2M_HIP(e, hipEventRecord(e->ev[0], e->stream));
    if (rematch) {
        MatchArgs m;
        m.grid = e->grid; m.pose = pose; m.gates = gates;
        m.sx = sx; m.sy = sy; m.sz = sz; m.n = n;
        m.nn_idx = e->d_nn_idx; m.nn_d2 = e->d_nn_d2;
        m.hard_rec = e->d_hrec; m.hard_off1 = n; m.hard_count = e->d_hard + 3 * e->n_cap;
        m.qheads = e->d_qheads;
        // Point batches per trip of the first-shell kernel (unless S2M_EASY_NB fixed it): three (24 loads in flight, 160
        // VGPRs, 3 waves/SIMD) is fastest while all of the launch's waves are resident anyway -- up to 98 k points; beyond
        // that, or when several scans are in flight on the chip (the batch entry), two (128 VGPRs, 4 waves/SIMD) wins:
        // C4 0.225 -> 0.214 ms/step, C5 batch 13.3 -> 14.2 k scans/s, against C3 20.3 -> 21.1 us the other way.
        int group = e->match_group;
        if (((group >> 8) & 0xf) == 0 && ((int64_t)n * 2 > 3072 * 64 || e->in_batch)) group |= 2 << 8;
        if (skip_far) group |= 0x40000;     // the first-shell kernel only: the reduce kernel reports whether the bet held
        launch_match(m, group, e->stream);  // hard_count is zero: reset by every reduce launch
        e->nn_valid = true;
        e->nn_complete = false;
        e->nn_nearest = false;
        e->rematch_pose = pose;
    }
    if (timed) S2M_HIP(e, hipEventRecord(e->ev[1], e->stream));
    ReduceArgs r;
    r.pose = pose; r.gates = gates;
    r.sx = sx; r.sy = sy; r.sz = sz; r.n = n;
    r.fit = rematch ? 1 : 0;
    r.nn_idx = e->d_nn_idx; r.nn_d2 = e->d_nn_d2; r.pts = e->grid.pts;
    r.plane = e->d_plane; r.flags = e->d_flags; r.sel = e->d_sel; r.eff = e->d_eff; r.pd2 = e->d_pd2;
    r.partials = e->d_partials; r.block = d_out;
    r.ticket = e->d_ticket; r.hard_count = e->d_hard + 3 * e->n_cap;
    r.qheads = e->d_qheads;
    r.spec = (rematch && skip_far) ? 1 : 0;
    // (fault injection, S2M_TEST_STALL=reduce: the kernel runs but its block never reaches the host)
    const bool publish = e->host_poll && d_out == e->d_block && !defer_publish && !e->wait.withhold(kStallReduce);
    r.host_block = publish ? e->h_block_dev : nullptr;
    r.host_flag = publish ? reinterpret_cast<unsigned long long *>(e->h_block_dev + S2M_BLOCK_DOUBLES) : nullptr;
    r.seq = ++e->seq;
    launch_reduce(r, e->stream);
    if (timed) S2M_HIP(e, hipEventRecord(e->ev[2], e->stream));
    S2M_HIP(e, hipGetLastError());
    e->last_pose = pose;
    e->pass_done = true;
    return S2M_OK;
}

// The pass just waited for ran without the far-point kernel and its list turned out not to be empty: run that kernel
// now (the list and its counters are still in place) and the reduce kernel again -- every per-point output of a rematch
// pass is a function of the neighbour lists alone, so the second run overwrites the first completely.
int redo_with_far_points(s2m_engine *e, const double state[S2M_STATE_DOUBLES], double *d_out)
{
    const Pose pose = pose_of(state);
    const Gates gates = gates_of(e->cfg);
    const int n = (int)e->n;
    float *sx = e->d_scan, *sy = e->d_scan + e->n_cap, *sz = e->d_scan + 2 * e->n_cap;
    MatchArgs m;
    m.grid = e->grid; m.pose = pose; m.gates = gates;
    m.sx = sx; m.sy = sy; m.sz = sz; m.n = n;
    m.nn_idx = e->d_nn_idx; m.nn_d2 = e->d_nn_d2;
    m.hard_rec = e->d_hrec; m.hard_off1 = n; m.hard_count = e->d_hard + 3 * e->n_cap;
    m.qheads = e->d_qheads;
    launch_match_far_points(m, e->match_group, e->stream);
    ReduceArgs r;
    r.pose = pose; r.gates = gates;
    r.sx = sx; r.sy = sy; r.sz = sz; r.n = n;
    r.fit = 1;
    r.nn_idx = e->d_nn_idx; r.nn_d2 = e->d_nn_d2; r.pts = e->grid.pts;
    r.plane = e->d_plane; r.flags = e->d_flags; r.sel = e->d_sel; r.eff = e->d_eff; r.pd2 = e->d_pd2;
    r.partials = e->d_partials; r.block = d_out;
    r.ticket = e->d_ticket; r.hard_count = e->d_hard + 3 * e->n_cap;
    r.qheads = e->d_qheads;
    const bool publish = e->host_poll && d_out == e->d_block;
    r.host_block = publish ? e->h_block_dev : nullptr;
    r.host_flag = publish ? reinterpret_cast<unsigned long long *>(e->h_block_dev + S2M_BLOCK_DOUBLES) : nullptr;
    r.seq = ++e->seq;
    launch_reduce(r, e->stream);
    S2M_HIP(e, hipGetLastError());
    return S2M_OK;
}

int finish_timing(s2m_engine *e)
{
    if (!e->timing || !e->timed_this_pass) return S2M_OK;
    float a = 0.f, b = 0.f;
    S2M_HIP(e, wait_event(&e->wait, e->ev[2], "the timing event behind a pass"));
    S2M_HIP(e, hipEventElapsedTime(&a, e->ev[0], e->ev[1]));
    S2M_HIP(e, hipEventElapsedTime(&b, e->ev[1], e->ev[2]));
    e->last_ms[0] = e->last_rematch ? a : 0.0;
    e->last_ms[1] = b;
    e->last_ms[2] = a + b;
    if (e->last_rematch) {
        e->tstats[0] += a; e->tstats[1] += 1;
        e->tstats[2] += b; e->tstats[3] += 1;
    } else {
        e->tstats[4] += b; e->tstats[5] += 1;
    }
    return S2M_OK;
}


}  // namespace

extern "C" {

// Wait for the block of the pass just enqueued and return a host pointer to it.  Fast path: the
// reduce kernel writes the block and a sequence flag straight into pinned host memory and the host
// spins on the flag (no D2H copy, no driver sync).  Fallback: D2H copy + stream synchronise.
static int wait_block(s2m_engine *e, const double *d_src, const double **host)
{
    e->step = "the block of a pass";
    if (e->host_poll && d_src == e->d_block) {
        const volatile unsigned long long *flag = reinterpret_cast<const volatile unsigned long long *>(e->h_block + S2M_BLOCK_DOUBLES);
        const hipError_t he = wait_word(&e->wait, flag, e->seq, e->stream, "the block of a pass (published by the reduce kernel)");
        if (he != hipSuccess) return fail(e, S2M_ERR_HIP, "reduce kernel did not publish its block", he);
        *host = e->h_block;
        return S2M_OK;
    }
    S2M_HIP(e, hipMemcpyAsync(e->h_block, d_src, S2M_BLOCK_DOUBLES * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    int rc = sync_stream(e, e->stream, "the block of a pass (copied)");
    if (rc) return rc;
    *host = e->h_block;
    return S2M_OK;
}

int s2m_residual_pass(s2m_engine *e, const double state[S2M_STATE_DOUBLES], int rematch, s2m_pass_out *out)
{
    if (!out) return fail(e, S2M_ERR_ARG, "null output");
    if (e) { e->where = __func__; e->step = ""; }
    int rc = run_pass(e, state, rematch, e ? e->d_block : nullptr);
    if (rc) return rc;
    const double *hb = nullptr;
    rc = wait_block(e, e->d_block, &hb);
    if (rc) return rc;
    rc = finish_timing(e);
    if (rc) return rc;
    if (rematch) e->short_lists = (int64_t)hb[159];
    std::memcpy(out->HtH, hb, 144 * sizeof(double));
    std::memcpy(out->Htz, hb + 144, 12 * sizeof(double));
    out->effct_feat_num = (int32_t)hb[156];
    out->total_residual = hb[157];
    out->rematch = rematch ? 1 : 0;
    return S2M_OK;
}

int s2m_residual_pass_device(s2m_engine *e, const double state[S2M_STATE_DOUBLES], int rematch, double *d_block)
{
    if (!d_block) return fail(e, S2M_ERR_ARG, "null device block");
    if (e && rematch) e->short_lists = -1;  // the block stays on the device
    return run_pass(e, state, rematch, d_block);
}

int s2m_get_rows(s2m_engine *e, double *h_x, double *h, int32_t *scan_index, int64_t capacity, int64_t *m_out)
{
    if (!e || !m_out) return fail(e, S2M_ERR_ARG, "null argument");
    if (!e->pass_done) return fail(e, S2M_ERR_STATE, "no pass yet");
    S2M_ENTER(e);
    if (e->rows_cap < e->n) {
        int rc = 0;
        rc = rc ? rc : grow(e, &e->d_hx, e->n_cap * 12);
        rc = rc ? rc : grow(e, &e->d_h, e->n_cap);
        rc = rc ? rc : grow(e, &e->d_rowidx, e->n_cap);
        if (rc) return rc;
        e->rows_cap = e->n_cap;
    }
    RowsArgs a;
    a.pose = e->last_pose; a.gates = gates_of(e->cfg);
    a.sx = e->d_scan; a.sy = e->d_scan + e->n_cap; a.sz = e->d_scan + 2 * e->n_cap; a.n = (int)e->n;
    a.plane = e->d_plane; a.pd2 = e->d_pd2; a.eff = e->d_eff;
    a.block_off = e->d_block_off; a.h_x = e->d_hx; a.h = e->d_h; a.scan_index = e->d_rowidx;
    launch_rows(a, e->stream);
    uint32_t m = 0;
    S2M_HIP(e, hipMemcpyAsync(&m, e->d_block_off + rows_blocks((int)e->n), sizeof(uint32_t), hipMemcpyDeviceToHost,
                              e->stream));
    int rc = sync_stream(e, e->stream, "the row count");
    if (rc) return rc;
    *m_out = m;
    if ((h_x || h || scan_index) && capacity < (int64_t)m) return fail(e, S2M_ERR_CAPACITY, "row buffers too small");
    if (h_x && m) S2M_HIP(e, hipMemcpyAsync(h_x, e->d_hx, (size_t)m * 12 * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    if (h && m) S2M_HIP(e, hipMemcpyAsync(h, e->d_h, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    if (scan_index && m) S2M_HIP(e, hipMemcpyAsync(scan_index, e->d_rowidx, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    return sync_stream(e, e->stream, "the rows on their way to the caller");
}

int s2m_get_point_state(s2m_engine *e, uint8_t *selected, uint8_t *effective, float *plane, float *pd2)
{
    if (!e) return S2M_ERR_ARG;
    if (!e->pass_done) return fail(e, S2M_ERR_STATE, "no pass yet");
    S2M_ENTER(e);
    const size_t n = (size_t)e->n;
    if (n == 0) return S2M_OK;
    int rc = sync_stream(e, e->stream, "the last pass");  // (a copy into pageable memory waits for the stream inside the runtime: wait here, with the deadline)
    if (rc) return rc;
    if (selected) S2M_HIP(e, hipMemcpyAsync(selected, e->d_sel, n, hipMemcpyDeviceToHost, e->stream));
    if (effective) S2M_HIP(e, hipMemcpyAsync(effective, e->d_eff, n, hipMemcpyDeviceToHost, e->stream));
    if (plane) S2M_HIP(e, hipMemcpyAsync(plane, e->d_plane, n * sizeof(float4), hipMemcpyDeviceToHost, e->stream));
    if (pd2) S2M_HIP(e, hipMemcpyAsync(pd2, e->d_pd2, n * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    return sync_stream(e, e->stream, "the per-point state on its way to the caller");
}

int s2m_get_neighbors(s2m_engine *e, int32_t *idx, float *d2)
{
    if (!e) return S2M_ERR_ARG;
    if (!e->nn_valid) return fail(e, S2M_ERR_STATE, "no rematch pass yet");
    S2M_ENTER(e);
    const size_t n = (size_t)e->n;
    if (n == 0) return S2M_OK;
    int rc = sync_stream(e, e->stream, "the last rematch pass");
    if (rc) return rc;
    if (idx) {  // the engine identifies a neighbour by its sorted position; the caller's indices are looked up on request
        const int64_t words = (int64_t)n * S2M_K;
        if (words > e->stage_cap) {
            rc = grow(e, &e->d_stage, words);
            if (rc) return rc;
            e->stage_cap = words;
        }
        int32_t *tmp = reinterpret_cast<int32_t *>(e->d_stage);
        const uint32_t *rank = nullptr;
        rc = caller_index_table(e, &rank);
        if (rc) return rc;
        launch_positions_to_indices(e->d_nn_idx, rank, words, tmp, e->stream);
        rc = sync_stream(e, e->stream, "the neighbour indices in caller order");
        if (rc) return rc;
        S2M_HIP(e, hipMemcpyAsync(idx, tmp, (size_t)words * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    }
    if (d2) S2M_HIP(e, hipMemcpyAsync(d2, e->d_nn_d2, n * S2M_K * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    return sync_stream(e, e->stream, "the neighbour lists on their way to the caller");
}


int s2m_eskf_update(s2m_engine *e, double x[S2M_STATE_DOUBLES], const double x_prop[S2M_STATE_DOUBLES],
                    const double P[S2M_DIM * S2M_DIM], const double HtH[144], const double Htz[12],
                    double solution[S2M_DIM], int32_t *converged)
{
    if (!e || !x || !x_prop || !P || !HtH || !Htz || !solution || !converged) return fail(e, S2M_ERR_ARG, "null argument");
    State xs, xp;
    Mat24 Pm;
    std::memcpy(&xs, x, sizeof(xs));
    std::memcpy(&xp, x_prop, sizeof(xp));
    std::memcpy(Pm.data(), P, sizeof(double) * S2M_DIM * S2M_DIM);
    EskfParams prm;
    prm.laser_point_cov = e->cfg.laser_point_cov;
    prm.conv_rot_deg = e->cfg.conv_rot_deg;
    prm.conv_pos_cm = e->cfg.conv_pos_cm;
    Vec24 sol{};
    bool conv = false;
    if (!eskf_update(prm, xs, xp, Pm, HtH, Htz, sol, conv, e->work)) return fail(e, S2M_ERR_NUMERIC, "singular matrix in eskf update");
    std::memcpy(x, &xs, sizeof(xs));
    std::memcpy(solution, sol.data(), sizeof(double) * S2M_DIM);
    *converged = conv ? 1 : 0;
    return S2M_OK;
}

int s2m_cov_update(s2m_engine *e, double P[S2M_DIM * S2M_DIM])
{
    if (!e || !P) return fail(e, S2M_ERR_ARG, "null argument");
    if (!e->work.valid) return fail(e, S2M_ERR_STATE, "no eskf update yet");
    Mat24 Pm;
    std::memcpy(Pm.data(), P, sizeof(double) * S2M_DIM * S2M_DIM);
    cov_update(e->work, Pm);
    std::memcpy(P, Pm.data(), sizeof(double) * S2M_DIM * S2M_DIM);
    return S2M_OK;
}

namespace {
// Everything the reference does with the result of one pass (:899-918, 1012-1101): degeneracy queue, Kalman update,
// log row, rematch judgement, exit test + covariance update.  finished = the loop ends after this iteration.
int consume_block(s2m_engine *e, const double *hb, IterCtl &c, double x[S2M_STATE_DOUBLES],
                  const double x_prop[S2M_STATE_DOUBLES], double P[S2M_DIM * S2M_DIM], s2m_iter_log *log, bool &finished)
{
    const int max_iter = e->cfg.max_iter;
    const double *HtH = hb, *Htz = hb + 144;
    const int32_t effct = (int32_t)hb[156];
    const double total_res = hb[157];
    c.stop = degeneracy_push(e->queue, e->queue_len, effct, e->cfg.feat_threshold);  // :899-918 (s2m_iterctl.h)
    double sol[S2M_DIM] = {0};
    if (!c.stop) {  // flg_EKF_inited is always true (INIT_TIME == 0, :75,:762)
        int rc = s2m_eskf_update(e, x, x_prop, P, HtH, Htz, sol, &c.conv);
        if (rc) return rc;
    }
    if (log) {
        log->effct[c.it] = effct;
        log->rematch[c.it] = c.rematch;
        log->conv[c.it] = c.conv;
        log->total_residual[c.it] = total_res;
        std::memcpy(log->solution[c.it], sol, sizeof(sol));
    }
    bool update_cov = false;
    iter_judge(c, max_iter, finished, update_cov);  // rematch judgement and exit test (:1070-1101, s2m_iterctl.h)
    if (update_cov) {
        int rc = s2m_cov_update(e, P);
        if (rc) return rc;
    }
    return S2M_OK;
}

void reset_log(s2m_iter_log *log, int max_iter)
{
    if (!log) return;  // header fields + the rows this call can write (the struct holds 64 rows, 13 KB)
    log->iters = log->rematch_passes = log->converged = log->ekf_stop = 0;
    const size_t rows = (size_t)std::min(max_iter, 64);
    std::memset(log->effct, 0, rows * sizeof(log->effct[0]));
    std::memset(log->rematch, 0, rows * sizeof(log->rematch[0]));
    std::memset(log->conv, 0, rows * sizeof(log->conv[0]));
    std::memset(log->total_residual, 0, rows * sizeof(log->total_residual[0]));
    std::memset(log->solution, 0, rows * sizeof(log->solution[0]));
}
}  // namespace

// Publish this rank's block to the shared segment, collect everybody's and sum them pairwise over the rank index (the
// same perfect binary tree as s2m_iterated_update_multi: with aligned power-of-two shards the sum equals the unsplit
// scan's block bit for bit, and every rank computes the identical sum, so the redundant fp64 updates stay in step).
// `bet`: this rank skipped the far-point kernel in this pass (word 159 of the published copy says so); *any_void =
// some rank bet and found far points, i.e. published a void block -- every rank learns it from the same data, so all of
// them enter the second exchange together even if their bets differed (handles with different histories).
static int shm_sum(s2m_engine *e, const double **hb, bool bet, bool *any_void)
{
    const int n = e->shm.nranks;
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    e->shm_blocks.resize((size_t)np2 * S2M_BLOCK_DOUBLES);
    double *sum = e->shm_blocks.data();
    double mine[S2M_BLOCK_DOUBLES];
    std::memcpy(mine, *hb, sizeof(mine));
    mine[159] = bet ? 1.0 : 0.0;
    std::string err;
    if (!shm_exchange(e->shm, mine, S2M_BLOCK_DOUBLES, sum, err)) return fail(e, S2M_ERR_HIP, err.c_str());
    bool v = false;
    for (int r = 0; r < n; ++r) {
        double *b = sum + (size_t)r * S2M_BLOCK_DOUBLES;
        v = v || (b[159] != 0.0 && b[158] != 0.0);
        b[159] = 0.0;
    }
    if (any_void) *any_void = v;
    std::fill(e->shm_blocks.begin() + (size_t)n * S2M_BLOCK_DOUBLES, e->shm_blocks.end(), 0.0);
    for (int w = 1; w < np2; w <<= 1)
        for (int i = 0; i + w < np2; i += 2 * w)
            for (int k = 0; k < S2M_BLOCK_DOUBLES; ++k) sum[(size_t)i * S2M_BLOCK_DOUBLES + k] += sum[(size_t)(i + w) * S2M_BLOCK_DOUBLES + k];
    *hb = sum;
    return S2M_OK;
}

namespace {
constexpr int kLoopChunk = 6;  // iterations enqueued ahead (the reference's yaml runs 10 at most and leaves after ~5)

int ensure_loop(s2m_engine *e)
{
    if (e->d_loop) return S2M_OK;
    S2M_HIP(e, hipMalloc((void **)&e->d_loop, sizeof(LoopState)));
    S2M_HIP(e, hipMemsetAsync(e->d_loop, 0, sizeof(LoopState), e->stream));
    S2M_HIP(e, hipHostMalloc((void **)&e->h_init, sizeof(LoopInit), hipHostMallocMapped));
    S2M_HIP(e, hipHostGetDevicePointer((void **)&e->h_init_dev, e->h_init, 0));
    S2M_HIP(e, hipHostMalloc((void **)&e->h_rec, sizeof(LoopRecord), hipHostMallocMapped));
    S2M_HIP(e, hipHostGetDevicePointer((void **)&e->h_rec_dev, e->h_rec, 0));
    std::memset(e->h_rec, 0, sizeof(LoopRecord));
    return S2M_OK;
}

// can this handle's update run with the state on the device?
bool loop_eligible(const s2m_engine *e)
{
    return e->cfg.device_loop != 0 && !e->comm.handle && !e->shm.base && e->host_poll && !e->timing && e->scan_ready &&
           e->cfg.max_iter <= kLoopMaxIter;
}

// the init record of a scan: state, G and C^-1 (s2m_loop.h), thresholds, degeneracy queue.  false: P[0:nc, 0:nc] is not
// positive definite -- the caller takes the host-stepped loop, whose LU form does not need that
bool loop_fill_init(s2m_engine *e, const double *x, const double *x_prop, const double *P)
{
    LoopInit &in = *e->h_init;
    const int nc = e->cfg.extrinsic_est_en ? 12 : 6;
    if (!loop_prepare(e->cfg.laser_point_cov, P, nc, in.G, in.Cinv)) return false;
    std::memcpy(in.x, x, sizeof(in.x));
    std::memcpy(in.x_prop, x_prop, sizeof(in.x_prop));
    in.conv_rot_deg = e->cfg.conv_rot_deg;
    in.conv_pos_cm = e->cfg.conv_pos_cm;
    in.max_iter = e->cfg.max_iter;
    in.feat_threshold = e->cfg.feat_threshold;
    in.nc = nc;
    in.queue_len = e->queue_len;
    std::memset(in.queue, 0, sizeof(in.queue));
    std::memcpy(in.queue, e->queue, sizeof(int32_t) * (S2M_FEAT_QUEUE + 1));
    return true;
}

// st: the stream the loop's kernels were enqueued on (a batch's launch group runs on its lead's stream)
int loop_wait(s2m_engine *e, unsigned long long seq, hipStream_t st = nullptr)
{
    e->step = "the record of the device-resident loop";
    const hipError_t he = wait_word(&e->wait, (const volatile unsigned long long *)&e->h_rec->flag, seq, st ? st : e->stream,
                                    "the record of the device-resident loop");
    if (he != hipSuccess) return fail(e, S2M_ERR_HIP, "the device-resident loop did not publish its record", he);
    return S2M_OK;
}

// what the host does with the record of a finished loop: state, log, degeneracy queue, covariance update (:1084-1085),
// and the handle's bookkeeping of the last pass
int loop_finish(s2m_engine *e, double *x, double *P, s2m_iter_log *log)
{
    const LoopRecord &rec = *e->h_rec;
    if (rec.numeric) return fail(e, S2M_ERR_NUMERIC, "singular matrix in eskf update");
    const int iters = rec.iters;
    std::memcpy(x, rec.x, sizeof(rec.x));
    if (log) {
        log->iters = iters;
        log->rematch_passes = rec.passes;
        log->converged = rec.conv;
        log->ekf_stop = rec.stop;
        for (int i = 0; i < iters && i < 64; ++i) {
            log->effct[i] = rec.effct[i];
            log->rematch[i] = rec.rematch[i];
            log->conv[i] = rec.conv_it[i];
            log->total_residual[i] = rec.total_residual[i];
            std::memcpy(log->solution[i], rec.solution[i], sizeof(rec.solution[i]));
        }
    }
    std::memcpy(e->queue, rec.queue, sizeof(int32_t) * (S2M_FEAT_QUEUE + 1));
    e->queue_len = rec.queue_len;
    if (rec.update_cov && !loop_cov_update(e->h_init->G, e->h_init->Cinv, rec.block, e->h_init->nc, P))
        return fail(e, S2M_ERR_NUMERIC, "singular matrix in the covariance update");
    double st[S2M_STATE_DOUBLES] = {0};
    std::memcpy(st, rec.pose_last, sizeof(rec.pose_last));
    e->last_pose = pose_of(st);
    std::memcpy(st, rec.pose_rematch, sizeof(rec.pose_rematch));
    e->rematch_pose = pose_of(st);
    e->last_rematch = iters > 0 && rec.rematch[iters - 1] != 0;
    e->nn_valid = true;
    e->nn_complete = false;
    e->nn_nearest = false;
    e->pass_done = true;
    e->short_lists = -1;
    e->sched_hist.assign((size_t)e->cfg.max_iter, 0);
    for (int i = 0; i < iters; ++i) {
        e->sched_hist[i] = (int8_t)(rec.rematch[i] != 0);
        if (rec.rematch[i]) { if (i == 0) e->far_first = rec.far_points[i]; else e->far_later = rec.far_points[i]; }
    }
    return S2M_OK;
}

// One scan's device-resident update as three steps, so that several handles can have their chains in flight at once:
// loop_begin (init record, plan), then loop_enqueue (the kernels of up to kLoopChunk iterations, following the schedule
// of the previous scan on this handle: which iterations searched) and loop_collect (wait for the record; finish, or go on
// from where the chunk ended or the plan did not hold) until run.done.  Every kernel looks at the control words the
// previous pass left and leaves at once when it is not due; a pass that needs a search the host did not enqueue stops
// the chain and reports.
struct LoopRun {
    std::vector<int8_t> kinds;
    Pose pose0;
    int it0 = 0;
    bool first = true, done = false;
    unsigned long long seq = 0;
    int guard = 0;
};

int loop_begin(s2m_engine *e, const double *x, const double *x_prop, const double *P, s2m_iter_log *log, LoopRun &run, bool &used)
{
    used = false;
    if (!loop_eligible(e)) return S2M_OK;
    if (!e->map_ready) return fail(e, S2M_ERR_STATE, "no map: call s2m_map_build first");
    S2M_INNER(e);
    int rc = ensure_loop(e);
    if (rc) return rc;
    if (!loop_fill_init(e, x, x_prop, P)) return S2M_OK;
    used = true;
    const int max_iter = e->cfg.max_iter;
    reset_log(log, max_iter);
    e->nn_valid = false;
    run.kinds.assign((size_t)max_iter, 1);   // no history: search kernels in front of every pass (never wrong)
    if ((int)e->sched_hist.size() == max_iter) run.kinds = e->sched_hist;
    run.kinds[0] = 1;
    run.pose0 = pose_of(x);
    return S2M_OK;
}

int loop_enqueue(s2m_engine *e, LoopRun &run)
{
    S2M_INNER(e);
    const int max_iter = e->cfg.max_iter;
    const Gates gates = gates_of(e->cfg);
    const int n = (int)e->n;
    float *sx = e->d_scan, *sy = e->d_scan + e->n_cap, *sz = e->d_scan + 2 * e->n_cap;
    run.seq = ++e->loop_seq;
    if (++e->loop_gen <= 0) e->loop_gen = 1;
    const int it_end = std::min(max_iter, run.it0 + kLoopChunk);
    for (int it = run.it0; it < it_end; ++it) {
        LoopLaunch l;
        l.state = e->d_loop; l.record = e->h_rec_dev; l.seq = run.seq;
        l.expect_it = it; l.kind = run.kinds[it]; l.gen = e->loop_gen; l.last_of_chunk = it == it_end - 1 ? 1 : 0;
        l.init = (run.first && it == run.it0) ? e->h_init_dev : nullptr;
        if (run.kinds[it]) {
            MatchArgs m;
            m.grid = e->grid; m.pose = run.pose0; m.gates = gates;
            m.sx = sx; m.sy = sy; m.sz = sz; m.n = n;
            m.nn_idx = e->d_nn_idx; m.nn_d2 = e->d_nn_d2;
            m.hard_rec = e->d_hrec; m.hard_off1 = n; m.hard_count = e->d_hard + 3 * e->n_cap;
            m.qheads = e->d_qheads;
            m.loop = l;
            // few far points expected (the pass in this position of the last scan had none): a small far-point grid -- the
            // queue serves any number, a wrong guess only costs time
            const int64_t hist = it == 0 ? e->far_first : e->far_later;
            m.far_waves = (e->spec_mode != 0 && hist == 0) ? 256 : 0;
            int group = e->match_group;
            if (((group >> 8) & 0xf) == 0 && ((int64_t)n * 2 > 3072 * 64 || e->in_batch)) group |= 2 << 8;
            launch_match(m, group, e->stream);
        }
        ReduceArgs r;
        r.pose = run.pose0; r.gates = gates;
        r.sx = sx; r.sy = sy; r.sz = sz; r.n = n;
        r.fit = 0;
        r.nn_idx = e->d_nn_idx; r.nn_d2 = e->d_nn_d2; r.pts = e->grid.pts;
        r.plane = e->d_plane; r.flags = e->d_flags; r.sel = e->d_sel; r.eff = e->d_eff; r.pd2 = e->d_pd2;
        r.partials = e->d_partials; r.block = e->d_block;
        r.ticket = e->d_ticket; r.hard_count = e->d_hard + 3 * e->n_cap;
        r.qheads = e->d_qheads;
        r.spec = 0;
        r.host_block = nullptr; r.host_flag = nullptr; r.seq = 0;
        r.loop = l;
        launch_reduce(r, e->stream);
    }
    S2M_HIP(e, hipGetLastError());
    run.first = false;
    return S2M_OK;
}

int loop_collect(s2m_engine *e, LoopRun &run, double *x, double *P, s2m_iter_log *log)
{
    int rc = loop_wait(e, run.seq);
    if (rc) return rc;
    const LoopRecord &rec = *e->h_rec;
    if (rec.finished) {
        run.done = true;
        return loop_finish(e, x, P, log);
    }
    run.it0 = rec.iters;                 // the chunk ended, or the plan did not hold at this iteration: go on from here
    if (run.it0 < 0 || run.it0 >= e->cfg.max_iter || ++run.guard > 4 * kLoopMaxIter)
        return fail(e, S2M_ERR_HIP, "the device-resident loop did not end");
    if (rec.abort) run.kinds[run.it0] = 1;
    return S2M_OK;
}

int iterated_update_loop(s2m_engine *e, double *x, const double *x_prop, double *P, s2m_iter_log *log, bool &used)
{
    LoopRun run;
    int rc = loop_begin(e, x, x_prop, P, log, run, used);
    if (rc || !used) return rc;
    while (!run.done) {
        rc = loop_enqueue(e, run);
        if (rc) return rc;
        rc = loop_collect(e, run, x, P, log);
        if (rc) return rc;
    }
    return S2M_OK;
}
}  // namespace

int s2m_iterated_update_sharded(s2m_engine *e, double x[S2M_STATE_DOUBLES], const double x_prop[S2M_STATE_DOUBLES],
                                double P[S2M_DIM * S2M_DIM], s2m_iter_log *log, double *d_block,
                                s2m_allreduce_fn reduce, void *user)
{
    if (!e || !x || !x_prop || !P) return fail(e, S2M_ERR_ARG, "null argument");
    if (reduce && !d_block) return fail(e, S2M_ERR_ARG, "sharded update needs a device block");
    if (!e->scan_ready) return fail(e, S2M_ERR_STATE, "no scan: call s2m_scan_set first");
    if (e->poisoned) return refuse_poisoned(e);
    e->where = "s2m_iterated_update";
    e->step = "";
    e->nn_valid = false;   // a new scan's first pass searches: the lists of the last one are over
    {
        tl_wait = &e->wait;
        int rc = relay_poll(e);   // (... which is when a layout produced beside the frames may take the live map's place)
        if (rc) return rc;
    }
    if (!reduce && (!d_block || d_block == e->d_block)) {  // state on the device where the form allows it (s2m_loop.h)
        bool used = false;
        int rc = iterated_update_loop(e, x, x_prop, P, log, used);
        if (rc || used) return rc;
    }
    if (!d_block) d_block = e->d_block;
    // a new scan starts with every point selected and no neighbours (laserMapping.cpp:810-818);
    // iteration 0 is always a rematch pass, whose gate rewrites point_selected_surf for every point
    S2M_INNER(e);
    e->nn_valid = false;
    const int max_iter = e->cfg.max_iter;
    int rematch_num = 0, rematch_en = 0, it = 0, passes = 0;
    int32_t conv = 0, stop = 0;
    reset_log(log, max_iter);
    for (it = 0; it < max_iter; ++it) {
        const int rematch = (it == 0) || rematch_en;  // :847
        passes += rematch;
        const bool collective = !reduce && e->comm.handle;  // built-in RCCL sum of the block before the hand-off
        const bool shm = !reduce && !collective && e->shm.base && d_block == e->d_block;  // host shared-memory sum after it
        // Bet on "no far points" when the last rematch pass in this position (first of a scan / later) had none: at a
        // converged pose the first shell resolves every point (measured: 0 of 65,536 at C3, 0 of 131,072 at C4), and
        // the far-point kernel -- a launch, a kernel boundary and 4,096 waves that find an empty list -- is ~4 us.
        // Plain single-handle loop only (no collective: every rank would have to lose the bet together).
        const int64_t hist = it == 0 ? e->far_first : e->far_later;
        const bool spec = rematch && !reduce && !collective && d_block == e->d_block && e->host_poll &&
                          (e->spec_mode == 2 || (e->spec_mode == 1 && hist == 0));
        int rc = run_pass(e, x, rematch, d_block, collective, spec);
        if (rc) return rc;
        if (it == 0) {
            // (state.cov / LASER_POINT_COV).inverse() (:1017) depends on the covariance alone: 8 us of host LU that
            // run here, behind the launch of the first pass, instead of after its block has arrived
            Mat24 Pm;
            std::memcpy(Pm.data(), P, sizeof(double) * S2M_DIM * S2M_DIM);
            EskfParams prm;
            prm.laser_point_cov = e->cfg.laser_point_cov;
            (void)eskf_prepare(prm, Pm, e->work);  // a singular P is reported by the update itself
        }
        if (reduce && reduce(user) != 0) return fail(e, S2M_ERR_HIP, "all-reduce callback failed");
        const double *hb = nullptr;
        if (collective) {
            // built-in collective: sum the block over the ranks on this stream, then publish it to the host
            std::string cerr_;
            if (!comm_allreduce_sum_f64(e->comm, d_block, S2M_BLOCK_DOUBLES, e->stream, cerr_)) return fail(e, S2M_ERR_HIP, cerr_.c_str());
            if (e->host_poll && d_block == e->d_block)
                launch_publish(d_block, e->h_block_dev, reinterpret_cast<unsigned long long *>(e->h_block_dev + S2M_BLOCK_DOUBLES),
                               e->seq, e->stream);
        }
        // after a collective the summed block only exists in d_block: copy it; otherwise poll
        if (reduce) {
            S2M_HIP(e, hipMemcpyAsync(e->h_block, d_block, S2M_BLOCK_DOUBLES * sizeof(double), hipMemcpyDeviceToHost,
                                      e->stream));
            rc = sync_stream(e, e->stream, "the summed block of a sharded pass");
            hb = e->h_block;
        } else {
            rc = wait_block(e, d_block, &hb);
        }
        if (rc) return rc;
        rc = finish_timing(e);
        if (rc) return rc;
        const double *own = hb;  // this rank's block in its pinned page
        bool any_void = false;
        if (shm) {  // every rank's block, summed in rank order: the far-point count below is then the job's, not the rank's
            rc = shm_sum(e, &hb, spec, &any_void);
            if (rc) return rc;
        }
        if (rematch) e->short_lists = (!reduce && !collective) ? (int64_t)own[159] : -1;
        if (rematch && !reduce && !collective) {
            const int64_t far_points = (int64_t)hb[158];
            if (it == 0) e->far_first = far_points; else e->far_later = far_points;
            const bool lost = spec && (int64_t)own[158] != 0;  // this rank's bet is lost: its block is void (s2m_reduce.hip)
            if (spec) { if (lost) ++e->bets_lost; else ++e->bets_won; }
            if (lost) {
                rc = redo_with_far_points(e, x, d_block);
                if (rc) return rc;
                rc = wait_block(e, d_block, &hb);
                if (rc) return rc;
                own = hb;
                e->short_lists = (int64_t)own[159];
            }
            if (shm && any_void) {  // somebody's block was void: everybody publishes again (the unchanged block where it was valid)
                hb = own;
                rc = shm_sum(e, &hb, false, nullptr);
                if (rc) return rc;
            }
        }
        IterCtl ctl{it, rematch, rematch_num, rematch_en, conv, stop};
        bool finished = false;
        rc = consume_block(e, hb, ctl, x, x_prop, P, log, finished);
        rematch_num = ctl.rematch_num; rematch_en = ctl.rematch_en; conv = ctl.conv; stop = ctl.stop;
        if (rc) return rc;
        if (finished) { ++it; break; }
    }
    if (log) {
        log->iters = it;
        log->rematch_passes = passes;
        log->converged = conv;
        log->ekf_stop = stop;
    }
    return S2M_OK;
}

int s2m_iterated_update(s2m_engine *e, double x[S2M_STATE_DOUBLES], const double x_prop[S2M_STATE_DOUBLES],
                        double P[S2M_DIM * S2M_DIM], s2m_iter_log *log)
{
    return s2m_iterated_update_sharded(e, x, x_prop, P, log, nullptr, nullptr, nullptr);
}

// K scans in flight on one GPU from ONE host thread (BASELINE configs[4] on a single device): every handle keeps its
// own stream and per-scan state and the same loop as s2m_iterated_update, but the host never sits in one handle's
// wait -- it goes round the handles, picks up whichever block has arrived, solves, and launches that handle's next
// pass, so the kernels of different scans fill each other's latency gaps (a single scan in flight leaves the GPU
// idle during every host turn-around and most of every latency-bound kernel).
namespace {
int batch_fused(s2m_engine *const *handles, int32_t k, double *x, const double *x_prop, double *P, s2m_iter_log *logs);
int batch_fused_loop(s2m_engine *const *handles, int32_t k, double *x, const double *x_prop, double *P, s2m_iter_log *logs,
                     bool &used);

// the K scans can go through ONE grid per pass when they search the same map on the same device with the same gates
bool batch_can_fuse(s2m_engine *const *handles, int32_t k)
{
    if (k < 2) return false;
    const s2m_engine *a = handles[0];
    for (int i = 0; i < k; ++i) {
        const s2m_engine *e = handles[i];
        if (!e->map_ready || e->grid.pts != a->grid.pts || e->grid.tab != a->grid.tab || e->grid.m != a->grid.m) return false;
        if (e->timing || e->match_group != 0) return false;
        if (e->cfg.max_iter != a->cfg.max_iter || e->cfg.extrinsic_est_en != a->cfg.extrinsic_est_en ||
            e->cfg.plane_thr != a->cfg.plane_thr || e->cfg.knn_d2_gate != a->cfg.knn_d2_gate ||
            e->cfg.s_gate != a->cfg.s_gate || e->cfg.res_gate != a->cfg.res_gate)
            return false;
    }
    return true;
}
}  // namespace

int s2m_iterated_update_batch(s2m_engine *const *handles, int32_t k, double *x, const double *x_prop, double *P,
                              s2m_iter_log *logs)
{
    if (!handles || k < 1 || k > 256 || !x || !x_prop || !P) return S2M_ERR_ARG;
    for (int i = 0; i < k; ++i)
        if (handles[i]) {
            if (handles[i]->poisoned) return refuse_poisoned(handles[i]);
            handles[i]->where = "s2m_iterated_update_batch";
            handles[i]->step = "";
        }
    {
        bool ok = true;
        for (int i = 0; i < k && ok; ++i) {
            ok = handles[i] != nullptr && handles[i]->scan_ready && !handles[i]->comm.handle && !handles[i]->shm.base && handles[i]->host_poll &&
                 handles[i]->device == handles[0]->device;
            for (int j = 0; j < i && ok; ++j) ok = handles[j] != handles[i];
        }
        if (ok && batch_can_fuse(handles, k)) {
            bool used = false;
            int rc = batch_fused_loop(handles, k, x, x_prop, P, logs, used);   // state on the device where every scan allows it
            if (rc || used) return rc;
            return batch_fused(handles, k, x, x_prop, P, logs);
        }
    }
    {   // handles that cannot share a launch (different maps or gates): every handle's chain on its own stream, state on the
        // device, all K in flight at once; the host only collects.  (Handles that cannot take the device-resident loop
        // fall through to the host-stepped form below.)
        bool all = true;
        for (int i = 0; i < k && all; ++i) all = handles[i] != nullptr && loop_eligible(handles[i]) && handles[i]->map_ready;
        for (int i = 0; i < k && all; ++i)
            for (int j = 0; j < i && all; ++j) all = handles[j] != handles[i];
        if (all) {
            std::vector<LoopRun> runs((size_t)k);
            std::vector<char> used((size_t)k, 0);
            bool every = true;
            for (int i = 0; i < k; ++i) {
                bool u = false;
                handles[i]->in_batch = k >= 4;
                int rc = loop_begin(handles[i], x + (size_t)i * S2M_STATE_DOUBLES, x_prop + (size_t)i * S2M_STATE_DOUBLES,
                                    P + (size_t)i * S2M_DIM * S2M_DIM, logs ? logs + i : nullptr, runs[i], u);
                if (rc) { for (int j = 0; j <= i; ++j) handles[j]->in_batch = false; return rc; }
                used[i] = u;
                every = every && u;
            }
            if (every) {
                int left = k, rc = S2M_OK;
                while (left > 0 && rc == S2M_OK) {
                    for (int i = 0; i < k && rc == S2M_OK; ++i)
                        if (!runs[i].done) rc = loop_enqueue(handles[i], runs[i]);
                    for (int i = 0; i < k && rc == S2M_OK; ++i)
                        if (!runs[i].done) {
                            rc = loop_collect(handles[i], runs[i], x + (size_t)i * S2M_STATE_DOUBLES, P + (size_t)i * S2M_DIM * S2M_DIM,
                                              logs ? logs + i : nullptr);
                            if (runs[i].done) --left;
                        }
                }
                for (int i = 0; i < k; ++i) handles[i]->in_batch = false;
                return rc;
            }
            for (int i = 0; i < k; ++i) handles[i]->in_batch = false;   // nothing was enqueued: the host-stepped form takes over
        }
    }
    struct Slot {
        IterCtl c{0, 1, 0, 0, 0, 0};
        int passes = 0;
        bool active = true;
        unsigned long long seq = 0;
        bool timed = false;
        int evset = 0;
    };
    Slot slots[256];
    auto xk = [&](int i) { return x + (size_t)i * S2M_STATE_DOUBLES; };
    auto xpk = [&](int i) { return x_prop + (size_t)i * S2M_STATE_DOUBLES; };
    auto Pk = [&](int i) { return P + (size_t)i * S2M_DIM * S2M_DIM; };
    for (int i = 0; i < k; ++i) {
        s2m_engine *e = handles[i];
        if (!e) return S2M_ERR_ARG;
        for (int j = 0; j < i; ++j)
            if (handles[j] == e) return fail(e, S2M_ERR_ARG, "s2m_iterated_update_batch: a handle appears twice");
        if (!e->scan_ready) return fail(e, S2M_ERR_STATE, "no scan: call s2m_scan_set first");
        if (e->comm.handle || e->shm.base || !e->host_poll) return fail(e, S2M_ERR_STATE, "s2m_iterated_update_batch: single-GPU handles with the host-polled block only");
        if (e->device != handles[0]->device) return fail(e, S2M_ERR_ARG, "s2m_iterated_update_batch: handles on different devices");
    }
    auto launch = [&](int i) -> int {
        s2m_engine *e = handles[i];
        Slot &s = slots[i];
        s.c.rematch = (s.c.it == 0) || s.c.rematch_en;  // :847
        s.passes += s.c.rematch;
        int rc = run_pass(e, xk(i), s.c.rematch, e->d_block);
        if (rc) return rc;
        s.seq = e->seq;
        s.timed = e->timed_this_pass;
        if (s.c.it == 0) {  // (P/R)^-1 behind the launch of the first pass, see s2m_iterated_update_sharded
            Mat24 Pm;
            std::memcpy(Pm.data(), Pk(i), sizeof(double) * S2M_DIM * S2M_DIM);
            EskfParams prm;
            prm.laser_point_cov = e->cfg.laser_point_cov;
            (void)eskf_prepare(prm, Pm, e->work);
        }
        return S2M_OK;
    };
    for (int i = 0; i < k; ++i) handles[i]->in_batch = k >= 4;
    struct ClearHint {  // whatever way the call ends
        s2m_engine *const *h; int k;
        ~ClearHint() { for (int i = 0; i < k; ++i) h[i]->in_batch = false; }
    } clear_hint{handles, k};
    for (int i = 0; i < k; ++i) {
        s2m_engine *e = handles[i];
        S2M_INNER(e);
        e->nn_valid = false;
        reset_log(logs ? logs + i : nullptr, e->cfg.max_iter);
        int rc = launch(i);
        if (rc) return rc;
    }
    int active = k;
    IdleWait idle(&handles[0]->wait);
    handles[0]->step = "the blocks of a batch";
    while (active > 0) {
        bool progress = false;
        for (int i = 0; i < k; ++i) {
            Slot &s = slots[i];
            if (!s.active) continue;
            s2m_engine *e = handles[i];
            volatile unsigned long long *flag = reinterpret_cast<volatile unsigned long long *>(e->h_block + S2M_BLOCK_DOUBLES);
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != s.seq) continue;
            progress = true;
            int rc = finish_timing(e);
            if (rc) return rc;
            bool finished = false;
            if (s.c.rematch) e->short_lists = (int64_t)e->h_block[159];
            rc = consume_block(e, e->h_block, s.c, xk(i), xpk(i), Pk(i), logs ? logs + i : nullptr, finished);
            if (rc) return rc;
            ++s.c.it;
            if (finished) {
                s.active = false;
                --active;
                if (logs) {
                    logs[i].iters = s.c.it;
                    logs[i].rematch_passes = s.passes;
                    logs[i].converged = s.c.conv;
                    logs[i].ekf_stop = s.c.stop;
                }
            } else {
                rc = launch(i);
                if (rc) return rc;
            }
        }
        if (progress) { idle.progress(); continue; }
        if (idle.idle()) {  // no block for a whole deadline: a runtime error is reported as such, anything else as the timeout it is
            for (int i = 0; i < k; ++i)
                if (slots[i].active) {
                    const hipError_t q = hipStreamQuery(handles[i]->stream);
                    if (q != hipSuccess && q != hipErrorNotReady) return fail(handles[i], S2M_ERR_HIP, "s2m_iterated_update_batch: a pass failed", q);
                }
            wait_expired(&handles[0]->wait, "the blocks of a batch of scans", idle.us);
            return fail(handles[0], S2M_ERR_HIP, "s2m_iterated_update_batch: a pass did not publish its block", kWaitTimedOut);
        }
    }
    return S2M_OK;
}

// ONE scan split over n handles, driven by ONE host thread, no collective library ("a single-process
// peer-copy gather ... whichever measures lower"): handles[i] holds shard i (contiguous ranges in handle order) and
// the map, on any mix of devices -- n GPUs of a node, or n shards on one GPU.  Every pass is launched on all handles;
// each reduce kernel publishes its 160-double block straight into that handle's pinned host page (as in the
// single-handle loop: no D2H copy, no driver sync, no publish kernel); the host picks the n blocks up as they land,
// sums them in handle order (fixed order: the result is deterministic for a given n) and runs ONE fp64 update.
// The degeneracy queue and the Kalman work area are those of handles[0].
int s2m_iterated_update_multi(s2m_engine *const *handles, int32_t n, double x[S2M_STATE_DOUBLES],
                              const double x_prop[S2M_STATE_DOUBLES], double P[S2M_DIM * S2M_DIM], s2m_iter_log *log)
{
    if (!handles || n < 1 || n > 256 || !x || !x_prop || !P) return S2M_ERR_ARG;
    for (int i = 0; i < n; ++i) {
        s2m_engine *e = handles[i];
        if (!e) return S2M_ERR_ARG;
        for (int j = 0; j < i; ++j)
            if (handles[j] == e) return fail(e, S2M_ERR_ARG, "s2m_iterated_update_multi: a handle appears twice");
        if (!e->scan_ready) return fail(e, S2M_ERR_STATE, "no scan: call s2m_scan_set first");
        if (e->comm.handle || e->shm.base || !e->host_poll)
            return fail(e, S2M_ERR_STATE, "s2m_iterated_update_multi: handles without a communicator, host-polled block only");
        if (e->poisoned) return refuse_poisoned(e);
        e->where = "s2m_iterated_update_multi";
        e->step = "";
        e->nn_valid = false;
    }
    s2m_engine *e0 = handles[0];
    const int max_iter = e0->cfg.max_iter;
    reset_log(log, max_iter);
    IterCtl c{0, 1, 0, 0, 0, 0};
    int passes = 0, it = 0;
    // the n blocks are combined pairwise over the handle index (a perfect binary tree, missing handles count as
    // +0.0): with the tree-shaped final sum of the reduce kernel, n aligned power-of-two pieces of a scan give the
    // single-handle block bit for bit
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    std::vector<double> tree((size_t)np2 * S2M_BLOCK_DOUBLES);
    double *sum = tree.data();
    for (it = 0; it < max_iter; ++it) {
        c.it = it;
        c.rematch = (it == 0) || c.rematch_en;  // :847
        passes += c.rematch;
        for (int i = 0; i < n; ++i) {
            int rc = run_pass(handles[i], x, c.rematch, handles[i]->d_block);
            if (rc) return rc;
        }
        if (c.rematch)
            for (int i = 0; i < n; ++i) handles[i]->short_lists = -1;  // (each handle's own count is in its block; not tracked here)
        if (it == 0) {  // (P/R)^-1 behind the launches of the first pass, see s2m_iterated_update_sharded
            Mat24 Pm;
            std::memcpy(Pm.data(), P, sizeof(double) * S2M_DIM * S2M_DIM);
            EskfParams prm;
            prm.laser_point_cov = e0->cfg.laser_point_cov;
            (void)eskf_prepare(prm, Pm, e0->work);
        }
        for (int i = 0; i < n; ++i) {  // handle order: the sum below must not depend on the arrival order
            const double *hb = nullptr;
            int rc = wait_block(handles[i], handles[i]->d_block, &hb);
            if (rc) return rc;
            rc = finish_timing(handles[i]);
            if (rc) return rc;
            std::memcpy(sum + (size_t)i * S2M_BLOCK_DOUBLES, hb, S2M_BLOCK_DOUBLES * sizeof(double));
        }
        std::fill(tree.begin() + (size_t)n * S2M_BLOCK_DOUBLES, tree.end(), 0.0);
        for (int w = 1; w < np2; w <<= 1)
            for (int i = 0; i + w < np2; i += 2 * w)
                for (int k = 0; k < S2M_BLOCK_DOUBLES; ++k) sum[(size_t)i * S2M_BLOCK_DOUBLES + k] += sum[(size_t)(i + w) * S2M_BLOCK_DOUBLES + k];
        bool finished = false;
        int rc = consume_block(e0, sum, c, x, x_prop, P, log, finished);
        if (rc) return rc;
        if (finished) { ++it; break; }
    }
    if (log) {
        log->iters = it;
        log->rematch_passes = passes;
        log->converged = c.conv;
        log->ekf_stop = c.stop;
    }
    return S2M_OK;
}

namespace {
// One grid per pass for all scans of a launch group (s2m_kernels.h, BatchArgs).  The K scans are dealt to G groups of
// at most kBatchMax; every group runs its scans in lock step -- pass w of all its active scans is one set of launches
// on the group's stream -- while the host consumes the blocks as they land (one fp64 solve each) and launches the
// group's next pass once the last of them is in.  With two groups the kernels of one cover the host turn-around of the
// other.  Per scan the loop is exactly that of s2m_iterated_update: results are bit-identical.
int batch_fused(s2m_engine *const *handles, int32_t k, double *x, const double *x_prop, double *P, s2m_iter_log *logs)
{
    struct Slot {
        IterCtl c{0, 1, 0, 0, 0, 0};
        int passes = 0;
        bool active = true, waiting = false;
        unsigned long long seq = 0;
    };
    struct Group {
        int first = 0, count = 0;   // slots [first, first + count)
        int waiting = 0, active = 0;
        s2m_engine *lead = nullptr;
        BatchArgs args;
    };
    int ng = k >= 4 ? 2 : 1;
    // launch groups of eight or four scans where the count allows it (measured, gpurun_out: K = 12 as 3 x 4 21.3 k scans/s,
    // as 2 x 6 19.1 k; K = 16 as 2 x 8 24.0 k, as 3 groups 22.4 k; K = 8 as 2 x 4 20.5 k, as 3 groups 19.0 k)
    for (int g : {8, 4})
        if (k % g == 0 && k / g >= 2) { ng = k / g; break; }
    while ((k + ng - 1) / ng > kBatchMax) ++ng;
    std::vector<Slot> slots((size_t)k);
    std::vector<Group> groups((size_t)ng);  // a group carries its kernel-argument table (2.9 KB): not on the stack
    auto xk = [&](int i) { return x + (size_t)i * S2M_STATE_DOUBLES; };
    auto xpk = [&](int i) { return x_prop + (size_t)i * S2M_STATE_DOUBLES; };
    auto Pk = [&](int i) { return P + (size_t)i * S2M_DIM * S2M_DIM; };
    s2m_engine *e0 = handles[0];
    S2M_INNER(e0);
    for (int g = 0, at = 0; g < ng; ++g) {
        Group &G = groups[g];
        G.first = at;
        G.count = (k - at + (ng - g) - 1) / (ng - g);
        at += G.count;
        G.active = G.count;
        G.lead = handles[G.first];
        // the group's shared far-point lists: room for every scan's points in each of the two lists
        int64_t total = 0;
        int n_max = 0;
        for (int i = G.first; i < G.first + G.count; ++i) {
            total += handles[i]->n;
            n_max = std::max<int>(n_max, (int)handles[i]->n);
        }
        s2m_engine *L = G.lead;
        if (L->brec_cap < total || !L->d_brec) {
            int rc = grow(L, &L->d_brec, 2 * total);
            if (rc) return rc;
            L->brec_cap = total;
        }
        if (!L->d_bcnt) {
            const size_t words = 2 * (16 + kQueueWords);
            S2M_HIP(L, hipMalloc((void **)&L->d_bcnt, words * sizeof(uint32_t)));
            S2M_HIP(L, hipMemsetAsync(L->d_bcnt, 0, words * sizeof(uint32_t), L->stream));
        }
        BatchArgs &b = G.args;
        b.grid = L->grid;
        b.gates = gates_of(L->cfg);
        b.k = G.count;
        b.n_max = n_max;
        b.hard_rec = L->d_brec;
        b.hard_off1 = L->brec_cap;
        for (int j = 0; j < kBatchMax; ++j) std::memset(&b.d[j], 0, sizeof(ScanDesc));
        for (int j = 0; j < G.count; ++j) {
            s2m_engine *e = handles[G.first + j];
            ScanDesc &d = b.d[j];
            d.sx = e->d_scan; d.sy = e->d_scan + e->n_cap; d.sz = e->d_scan + 2 * e->n_cap;
            d.nn_idx = e->d_nn_idx; d.nn_d2 = e->d_nn_d2;
            d.plane = e->d_plane; d.flags = e->d_flags; d.sel = e->d_sel; d.eff = e->d_eff; d.pd2 = e->d_pd2;
            d.partials = e->d_partials; d.block = e->d_block; d.ticket = e->d_ticket;
            d.host_block = e->h_block_dev;
            d.host_flag = reinterpret_cast<unsigned long long *>(e->h_block_dev + S2M_BLOCK_DOUBLES);
            d.n = (int32_t)e->n;
            d.active = 1;
            e->nn_valid = false;
            reset_log(logs ? logs + G.first + j : nullptr, e->cfg.max_iter);
            // the handle's own stream may still hold its scan hand-over: the group's launches go to the lead's stream
            if (e != L && e->stream != L->stream) {
                int rc = sync_stream(e, e->stream, "a scan's own stream, before the group's launches");
                if (rc) return rc;
            }
        }
    }
    // one pass of every active scan of the group: the table's poses and flags, then the launches
    auto launch = [&](Group &G) -> int {
        s2m_engine *L = G.lead;
        BatchArgs &b = G.args;
        uint32_t *set0 = L->d_bcnt, *set1 = L->d_bcnt + (16 + kQueueWords);
        const bool odd = (L->bwave++ & 1ull) != 0;
        b.hard_count = odd ? set1 : set0;            b.qheads = b.hard_count + 16;
        b.hard_count_next = odd ? set0 : set1;       b.qheads_next = b.hard_count_next + 16;
        bool any_rematch = false, any_plain = false;
        G.waiting = 0;
        for (int j = 0; j < G.count; ++j) {
            const int i = G.first + j;
            Slot &s = slots[i];
            ScanDesc &d = b.d[j];
            d.active = s.active ? 1 : 0;
            if (!s.active) continue;
            s2m_engine *e = handles[i];
            s.c.rematch = (s.c.it == 0) || s.c.rematch_en;  // :847
            s.passes += s.c.rematch;
            d.rematch = s.c.rematch;
            d.pose = pose_of(xk(i));
            d.seq = s.seq = ++e->seq;
            const bool lost = e->wait.withhold(kStallReduce);  // (fault injection: this scan's block never reaches the host)
            d.host_block = lost ? nullptr : e->h_block_dev;
            d.host_flag = lost ? nullptr : reinterpret_cast<unsigned long long *>(e->h_block_dev + S2M_BLOCK_DOUBLES);
            any_rematch = any_rematch || s.c.rematch;
            any_plain = any_plain || !s.c.rematch;
            s.waiting = true;
            ++G.waiting;
            e->last_rematch = s.c.rematch != 0;
            e->last_pose = d.pose;
            if (s.c.rematch) { e->nn_valid = true; e->nn_complete = false; e->nn_nearest = false; e->rematch_pose = d.pose; }
            e->pass_done = true;
            e->timed_this_pass = false;
        }
        if (any_rematch) launch_match_batch(b, L->stream);
        launch_reduce_batch(b, any_rematch, any_plain, L->stream);
        S2M_HIP(L, hipGetLastError());
        for (int j = 0; j < G.count; ++j) {  // (P/R)^-1 behind the launches of the first pass
            const int i = G.first + j;
            if (!slots[i].active || slots[i].c.it != 0) continue;
            Mat24 Pm;
            std::memcpy(Pm.data(), Pk(i), sizeof(double) * S2M_DIM * S2M_DIM);
            EskfParams prm;
            prm.laser_point_cov = handles[i]->cfg.laser_point_cov;
            (void)eskf_prepare(prm, Pm, handles[i]->work);
        }
        return S2M_OK;
    };
    for (int g = 0; g < ng; ++g) {
        int rc = launch(groups[g]);
        if (rc) return rc;
    }
    int groups_left = ng;
    IdleWait idle(&e0->wait);
    e0->step = "the blocks of a batch (one grid per pass)";
    while (groups_left > 0) {
        bool progress = false;
        for (int g = 0; g < ng; ++g) {
            Group &G = groups[g];
            if (G.active == 0) continue;
            for (int j = 0; j < G.count; ++j) {
                const int i = G.first + j;
                Slot &s = slots[i];
                if (!s.waiting) continue;
                s2m_engine *e = handles[i];
                volatile unsigned long long *flag = reinterpret_cast<volatile unsigned long long *>(e->h_block + S2M_BLOCK_DOUBLES);
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != s.seq) continue;
                progress = true;
                s.waiting = false;
                --G.waiting;
                bool finished = false;
                if (s.c.rematch) e->short_lists = (int64_t)e->h_block[159];
                int rc = consume_block(e, e->h_block, s.c, xk(i), xpk(i), Pk(i), logs ? logs + i : nullptr, finished);
                if (rc) return rc;
                ++s.c.it;
                if (finished) {
                    s.active = false;
                    --G.active;
                    if (logs) {
                        logs[i].iters = s.c.it;
                        logs[i].rematch_passes = s.passes;
                        logs[i].converged = s.c.conv;
                        logs[i].ekf_stop = s.c.stop;
                    }
                }
            }
            if (G.waiting == 0) {
                if (G.active > 0) {
                    int rc = launch(G);
                    if (rc) return rc;
                } else {
                    --groups_left;
                    G.active = 0;
                    G.waiting = -1;  // done: never looked at again
                }
            }
        }
        if (progress) { idle.progress(); continue; }
        if (idle.idle()) {  // no block for a whole deadline: a runtime error is reported as such, anything else as the timeout it is
            for (int g = 0; g < ng; ++g) {
                const hipError_t q = hipStreamQuery(groups[g].lead->stream);
                if (q != hipSuccess && q != hipErrorNotReady) return fail(groups[g].lead, S2M_ERR_HIP, "s2m_iterated_update_batch: a pass failed", q);
            }
            wait_expired(&e0->wait, "the blocks of a batch of scans", idle.us);
            return fail(e0, S2M_ERR_HIP, "s2m_iterated_update_batch: a pass did not publish its block", kWaitTimedOut);
        }
    }
    return S2M_OK;
}
}  // namespace

namespace {
// The batched form with the loops on the device: the K scans are dealt to launch groups as in batch_fused, but a group's
// passes are enqueued kLoopChunk iterations ahead -- search kernels in front of every pass (the scans of a group do not
// rematch in the same iterations; a scan that does not search leaves those kernels at once), ONE reduce launch per pass
// for the rematching and the reusing scans alike -- and the host only collects the K records at the end.  No host turn-
// around between the passes, no lock step with the host: the groups' chains fill the chip side by side.
int batch_fused_loop(s2m_engine *const *handles, int32_t k, double *x, const double *x_prop, double *P, s2m_iter_log *logs,
                     bool &used)
{
    used = false;
    bool any_points = false;
    for (int i = 0; i < k; ++i) {
        if (!loop_eligible(handles[i])) return S2M_OK;
        any_points = any_points || handles[i]->n > 0;
    }
    if (!any_points) return S2M_OK;
    struct Group {
        int first = 0, count = 0, passes = 0;
        s2m_engine *lead = nullptr;
        BatchArgs args;
    };
    int ng = k >= 4 ? 2 : 1;
    for (int g : {8, 4})
        if (k % g == 0 && k / g >= 2) { ng = k / g; break; }
    while ((k + ng - 1) / ng > kBatchMax) ++ng;
    std::vector<Group> groups((size_t)ng);
    auto xk = [&](int i) { return x + (size_t)i * S2M_STATE_DOUBLES; };
    auto xpk = [&](int i) { return x_prop + (size_t)i * S2M_STATE_DOUBLES; };
    auto Pk = [&](int i) { return P + (size_t)i * S2M_DIM * S2M_DIM; };
    s2m_engine *e0 = handles[0];
    S2M_INNER(e0);
    for (int i = 0; i < k; ++i) {
        int rc = ensure_loop(handles[i]);
        if (rc) return rc;
    }
    for (int i = 0; i < k; ++i)   // before anything is enqueued: a covariance the Cholesky form cannot take sends the call back
        if (!loop_fill_init(handles[i], xk(i), xpk(i), Pk(i))) return S2M_OK;
    used = true;
    const int max_iter = e0->cfg.max_iter;
    for (int g = 0, at = 0; g < ng; ++g) {
        Group &G = groups[g];
        G.first = at;
        G.count = (k - at + (ng - g) - 1) / (ng - g);
        at += G.count;
        G.lead = handles[G.first];
        int64_t total = 0;
        int n_max = 0;
        for (int i = G.first; i < G.first + G.count; ++i) {
            total += handles[i]->n;
            n_max = std::max<int>(n_max, (int)handles[i]->n);
        }
        s2m_engine *L = G.lead;
        if (L->brec_cap < total || !L->d_brec) {
            int rc = grow(L, &L->d_brec, 2 * total);
            if (rc) return rc;
            L->brec_cap = total;
        }
        if (!L->d_bcnt) {
            const size_t words = 2 * (16 + kQueueWords);
            S2M_HIP(L, hipMalloc((void **)&L->d_bcnt, words * sizeof(uint32_t)));
            S2M_HIP(L, hipMemsetAsync(L->d_bcnt, 0, words * sizeof(uint32_t), L->stream));
        }
        BatchArgs &b = G.args;
        b.grid = L->grid;
        b.gates = gates_of(L->cfg);
        b.k = G.count;
        b.n_max = n_max;
        b.hard_rec = L->d_brec;
        b.hard_off1 = L->brec_cap;
        for (int j = 0; j < kBatchMax; ++j) b.d[j] = ScanDesc{};
        for (int j = 0; j < G.count; ++j) {
            s2m_engine *e = handles[G.first + j];
            ScanDesc &d = b.d[j];
            d.pose = pose_of(xk(G.first + j));
            d.sx = e->d_scan; d.sy = e->d_scan + e->n_cap; d.sz = e->d_scan + 2 * e->n_cap;
            d.nn_idx = e->d_nn_idx; d.nn_d2 = e->d_nn_d2;
            d.plane = e->d_plane; d.flags = e->d_flags; d.sel = e->d_sel; d.eff = e->d_eff; d.pd2 = e->d_pd2;
            d.partials = e->d_partials; d.block = e->d_block; d.ticket = e->d_ticket;
            d.host_block = nullptr; d.host_flag = nullptr; d.seq = 0;
            d.n = (int32_t)e->n;
            d.rematch = 1;
            d.active = 1;
            d.loop.state = e->d_loop;
            d.loop.record = e->h_rec_dev;
            e->nn_valid = false;
            reset_log(logs ? logs + G.first + j : nullptr, e->cfg.max_iter);
            if (e != L && e->stream != L->stream) {
                int rc = sync_stream(e, e->stream, "a scan's own stream, before the group's launches");
                if (rc) return rc;
            }
        }
    }
    // A group's search kernels are enqueued in front of a pass only where some scan of the group is expected to search:
    // every scan's plan is the schedule of the previous scan on its handle (all passes, without one).  A scan that wants
    // to search where nothing was enqueued stops its own chain and reports (loop_abort); it is resumed in the next round
    // from that iteration -- the scans of a group need not be at the same iteration.
    std::vector<char> done((size_t)k, 0);
    std::vector<int> it0((size_t)k, 0);
    std::vector<std::vector<int8_t>> kinds((size_t)k);
    for (int i = 0; i < k; ++i) {
        kinds[i].assign((size_t)max_iter, 1);
        if ((int)handles[i]->sched_hist.size() == max_iter) kinds[i] = handles[i]->sched_hist;
        kinds[i][0] = 1;
    }
    int left = k;
    bool first = true;
    for (int round = 0; left > 0; ++round) {
        if (round > 4 * kLoopMaxIter) return fail(e0, S2M_ERR_HIP, "s2m_iterated_update_batch: a device-resident loop did not end");
        for (int i = 0; i < k; ++i) {
            s2m_engine *e = handles[i];
            ++e->loop_seq;
            if (++e->loop_gen <= 0) e->loop_gen = 1;
        }
        for (int g = 0; g < ng; ++g) {
            Group &G = groups[g];
            G.passes = 0;
            for (int j = 0; j < G.count; ++j)
                if (!done[G.first + j]) G.passes = std::max(G.passes, std::min(kLoopChunk, max_iter - it0[G.first + j]));
        }
        for (int p = 0; p < kLoopChunk; ++p)
            for (int g = 0; g < ng; ++g) {
                Group &G = groups[g];
                if (p >= G.passes) continue;
                s2m_engine *L = G.lead;
                BatchArgs &b = G.args;
                bool search = false;
                for (int j = 0; j < G.count; ++j) {
                    const int i = G.first + j, it = it0[i] + p;
                    if (!done[i] && it < max_iter) search = search || kinds[i][it] != 0;
                }
                for (int j = 0; j < G.count; ++j) {
                    const int i = G.first + j, it = it0[i] + p;
                    s2m_engine *e = handles[i];
                    LoopLaunch &l = b.d[j].loop;
                    b.d[j].active = (!done[i] && it < max_iter) ? 1 : 0;
                    l.seq = e->loop_seq; l.gen = e->loop_gen;
                    l.expect_it = it; l.kind = search ? 1 : 0;
                    l.last_of_chunk = (p == std::min(kLoopChunk, max_iter - it0[i]) - 1) ? 1 : 0;
                    l.init = (first && p == 0) ? e->h_init_dev : nullptr;
                }
                if (search) {
                    uint32_t *set0 = L->d_bcnt, *set1 = L->d_bcnt + (16 + kQueueWords);
                    const bool odd = (L->bwave++ & 1ull) != 0;
                    b.hard_count = odd ? set1 : set0;            b.qheads = b.hard_count + 16;
                    b.hard_count_next = odd ? set0 : set1;       b.qheads_next = b.hard_count_next + 16;
                    launch_match_batch(b, L->stream);
                }
                launch_reduce_batch_loop(b, L->stream);
                S2M_HIP(L, hipGetLastError());
            }
        first = false;
        for (int i = 0; i < k; ++i) {
            if (done[i]) continue;
            s2m_engine *e = handles[i];
            hipStream_t st = nullptr;
            for (int g = 0; g < ng; ++g)
                if (i >= groups[g].first && i < groups[g].first + groups[g].count) st = groups[g].lead->stream;
            int rc = loop_wait(e, e->loop_seq, st);
            if (rc) return rc;
            const LoopRecord &rec = *e->h_rec;
            if (rec.finished) {
                rc = loop_finish(e, xk(i), Pk(i), logs ? logs + i : nullptr);
                if (rc) return rc;
                done[i] = 1;
                --left;
            } else {
                it0[i] = rec.iters;   // the chunk ended, or this scan's plan did not hold at this iteration
                if (it0[i] < 0 || it0[i] >= max_iter) return fail(e, S2M_ERR_HIP, "a device-resident loop reported an impossible iteration");
                if (rec.abort) kinds[i][it0[i]] = 1;
            }
        }
    }
    return S2M_OK;
}
}  // namespace


int s2m_feat_queue_get(const s2m_engine *e, int32_t q[S2M_FEAT_QUEUE], int32_t *len)
{
    if (!e || !q || !len) return S2M_ERR_ARG;
    std::memcpy(q, e->queue, sizeof(int32_t) * S2M_FEAT_QUEUE);
    *len = e->queue_len;
    return S2M_OK;
}

int s2m_feat_queue_set(s2m_engine *e, const int32_t *q, int32_t len)
{
    if (!e || len < 0 || len > S2M_FEAT_QUEUE || (len > 0 && !q)) return S2M_ERR_ARG;
    if (len) std::memcpy(e->queue, q, sizeof(int32_t) * (size_t)len);
    e->queue_len = len;
    return S2M_OK;
}

int s2m_h_share_model(s2m_engine *e, const double state[S2M_STATE_DOUBLES], int first_iteration,
                      s2m_dyn_share *d)
{
    if (!e || !state || !d) return fail(e, S2M_ERR_ARG, "null argument");
    s2m_pass_out out;
    const int rematch = first_iteration || d->converge || !e->nn_valid;
    int rc = s2m_residual_pass(e, state, rematch, &out);
    if (rc) return rc;
    d->rows = out.effct_feat_num;
    d->total_residual = out.total_residual;
    d->valid = out.effct_feat_num >= 1;
    if (!d->valid) return S2M_OK;
    int64_t m = 0;
    return s2m_get_rows(e, d->h_x, d->h, nullptr, d->capacity, &m);
}

int s2m_set_timing(s2m_engine *e, int enabled)
{
    if (!e) return S2M_ERR_ARG;
    e->timing = enabled != 0;
    e->timing_stride = enabled > 2 ? enabled : 1;  // n > 2: time every n-th pass
    e->timing_phase = 0;
    for (double &t : e->tstats) t = 0;
    return S2M_OK;
}

int s2m_get_timing(const s2m_engine *e, double ms[3])
{
    if (!e || !ms) return S2M_ERR_ARG;
    ms[0] = e->last_ms[0]; ms[1] = e->last_ms[1]; ms[2] = e->last_ms[2];
    return S2M_OK;
}

int s2m_bet_stats(const s2m_engine *e, int64_t stats[2])
{
    if (!e || !stats) return S2M_ERR_ARG;
    stats[0] = e->bets_won;
    stats[1] = e->bets_lost;
    return S2M_OK;
}

int s2m_get_timing_stats(const s2m_engine *e, double stats[6])
{
    if (!e || !stats) return S2M_ERR_ARG;
    for (int i = 0; i < 6; ++i) stats[i] = e->tstats[i];
    return S2M_OK;
}

}  // extern "C"
