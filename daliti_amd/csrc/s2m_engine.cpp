// s2m_engine.cpp -- host engine and the C ABI of include/daliti_s2m.h.
//
// The engine owns what laserMapping.cpp keeps in globals for this path (ikdtree :164,
// Nearest_Points :578, point_selected_surf :812, effct_feat_numQueue :193, K / H_T_H :696,983) and
// drives the HIP kernels; there is no CPU fallback for any compute entry point.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <new>
#include <string>
#include <vector>

#include "../../include/daliti_s2m.h"
#include "s2m_comm.h"
#include "s2m_eskf.h"
#include "s2m_fov.h"
#include "s2m_iterctl.h"
#include "s2m_kernels.h"

namespace s2m {
void launch_deinterleave(const float *src, int64_t stride, int64_t n, float *sx, float *sy, float *sz,
                         hipStream_t st);
void launch_scan_reset(int64_t n, uint8_t *sel, uint8_t *eff, uint8_t *flags, hipStream_t st);
}

using namespace s2m;

struct s2m_engine {
    s2m_config cfg{};
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    bool timing = false;
    int timing_stride = 1, timing_phase = 0;   // time every stride-th pass (sampling keeps the probe cheap)
    bool timed_this_pass = false;
    double last_ms[3] = {0, 0, 0};
    double tstats[6] = {0, 0, 0, 0, 0, 0};     // {match ms, n, reduce<FIT> ms, n, reduce (reuse pass) ms, n}
    bool last_rematch = false;
    int match_group = 0;  // launch_match flags (s2m_kernels.h): wide addresses, point batches per trip
    std::string err;

    MapBuffers map;
    UpdateBuffers upd;
    VoxelBuffers vox;
    UndistBuffers und;
    // lasermap_fov_segment state (laserMapping.cpp:311-312)
    float local_map[6] = {0, 0, 0, 0, 0, 0};
    bool local_map_init = false;
    float built_cell = 0.0f;  // cell size of the current grid (kept across incremental rebuilds)
    Mailbox mail;                   // stream waits of the per-frame entry points (polled, not hipStreamSynchronize)
    bool in_batch = false;          // set while the handle is served by s2m_iterated_update_batch with several scans
    bool no_merge = false;          // S2M_NO_MERGE=1: every update rebuilds the grid from scratch (A/B and tests)
    bool no_slab = false;           // S2M_NO_SLAB=1: no in-place update of the touched bricks, every update merges (A/B and tests)
    int64_t n_inplace = 0;          // updates applied in place (counted among the merged ones too)
    ChangeLog log;                  // what the updates added / removed since the last s2m_map_get_changes
    uint64_t log_seq = 0;
    bool last_update_merged = false;
    int64_t n_merged = 0, n_rebuilt = 0, n_regrid = 0;  // how this handle's map updates were produced (s2m_map_update_stats)
    std::mutex stats_mu;            // the lazily fetched counts of a merged update may be asked for by borrowers' threads
    Grid grid{};
    MapStats stats;
    bool map_ready = false;
    bool map_borrowed = false;  // grid points into another handle's buffers (s2m_map_share)

    // staging for host inputs
    float *d_stage = nullptr;
    int64_t stage_cap = 0;  // floats
    // s2m_scan_prefetch_raw: the NEXT sweep's records cross PCIe on a side stream, driven by a worker thread (a copy out of
    // pageable memory blocks the thread that issues it), while the caller's thread registers the current sweep
    struct Prefetch {
        std::thread worker;
        std::mutex mu;
        std::condition_variable cv;
        bool quit = false, busy = false, ready = false;
        bool gpu_pending = false;        // the last job's work on `stream` has not been ordered in front of the main stream yet
        std::atomic<int> busy_a{0};      // mirror of `busy` for the short spins in front of the condition-variable waits: a futex
                                         // sleep / wake is tens of microseconds at best and has been seen to cost 10 ms once
        const float *src = nullptr;      // host records the job copies / the copy in d_buf belongs to
        int64_t floats = 0;
        hipError_t err = hipSuccess;
        float *d_buf = nullptr;
        int64_t cap = 0;                 // floats
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
        // s2m_scan_prepare_raw: the job also undistorts and down-samples into the spare scan arrays (d_scan_alt) on
        // the side stream; what it was asked for is kept so that s2m_scan_set_from_raw can recognise the same call
        bool prepare = false, prepared = false;
        bool copied = false;             // the records of this job are in d_buf already (a prefetch that was not consumed)
        // the time order of the records in d_buf (und.val2) has been computed for these time fields (a prefetch with offsets)
        bool want_order = false, ordered = false;
        int32_t order_oa = 0, order_ob = 0;
        int64_t stride = 0, n = 0, m = 0;
        int32_t oa = 0, ob = 0;
        float leaf = 0.0f;
        std::vector<double> poses;       // 22 doubles per pose
        double state_end[S2M_STATE_DOUBLES] = {0};
    } pf;
    float *d_scan_alt = nullptr;  // sx | sy | sz of the scan being prepared, laid out like d_scan (same n_cap)
    int64_t scan_alt_cap = 0;     // the n_cap it was allocated for

    // scan + per-point state
    int64_t n = 0, n_cap = 0;
    bool scan_ready = false, pass_done = false;
    float *d_scan = nullptr;  // sx | sy | sz, each n_cap floats
    float4 *d_plane = nullptr;
    uint8_t *d_flags = nullptr, *d_sel = nullptr, *d_eff = nullptr;
    float *d_pd2 = nullptr;
    int32_t *d_nn_idx = nullptr;
    float *d_nn_d2 = nullptr;
    double *d_partials = nullptr;
    double *d_block = nullptr;
    double *h_block = nullptr;  // pinned host: 160 doubles + completion flag, written by the reduce kernel
    double *h_block_dev = nullptr;          // the same memory as seen from the device
    unsigned long long seq = 0;             // pass sequence number published through the flag
    uint32_t *d_ticket = nullptr;
    bool host_poll = true;                  // the reduce kernel publishes the block to pinned host memory (else: D2H copy + sync)
    // rows on request
    uint32_t *d_block_off = nullptr;
    double *d_hx = nullptr, *d_h = nullptr;
    int32_t *d_rowidx = nullptr;
    int64_t rows_cap = 0;

    Pose last_pose{};
    Pose rematch_pose{};          // pose of the last rematch pass: the world-frame queries Nearest_Points belong to
    uint32_t *d_hard = nullptr;   // the far-point lists' counters sit behind 3 x n_cap words (the words themselves are free)
    uint32_t *d_qheads = nullptr; // match_hard's dequeue heads (kQueueWords)
    HardRec *d_hrec = nullptr;    // the far points' records: 2 x n_cap (without / with a radius)
    bool nn_valid = false;
    bool nn_complete = false;     // s2m_complete_neighbors has run on the current lists
    bool nn_nearest = false;      // ... or at least every list's nearest neighbour is proven (what s2m_map_incremental needs)
    int blind_rounds = 1;         // completion rounds that s2m_map_incremental enqueues without asking whether anything is open
    float first_round_gain = 16.0f; // ... and the factor on the gate (squared radius) of their first round: radius x 4 (measured on
                                    // the moving-trajectory leg, x2 / x4 with 1 / 2 blind rounds: median frame 0.51 / 0.50 ms, p99 0.82 / 0.63)
    // far points (scan points the first-shell kernel could not resolve) of the last FIRST rematch pass of a scan and of
    // the last LATER one; -1 = unknown.  A pass whose predecessor in the same position had none runs without the
    // far-point kernel on that bet (spec_mode: 0 never, 1 by history, 2 always -- the last two for tests)
    int64_t far_first = -1, far_later = -1;
    int64_t bets_won = 0, bets_lost = 0;   // passes that ran without the far-point kernel and were right / had to be redone
    // neighbour lists of the last rematch pass that did not fill inside the gate (block[159]); -1 = not known for this
    // handle (forms that only see the sum over shards, the device-resident loop): s2m_map_incremental then asks the device
    int64_t short_lists = -1;
    int spec_mode = 1;
    bool spec_env = false;  // S2M_SPEC set: the environment overrides the config (A/B runs)

    // far-point lists, counters and queue heads shared by the scans of a batched launch; owned by the first handle of
    // a launch group of s2m_iterated_update_batch
    HardRec *d_brec = nullptr;
    int64_t brec_cap = 0;          // records per list
    uint32_t *d_bcnt = nullptr;    // two sets of {16 counter words, kQueueWords queue heads}, used alternately
    unsigned long long bwave = 0;  // launches so far (selects the set)

    // device-resident loop (s2m_loop.h): state on the device, init record and result record in pinned host memory
    LoopState *d_loop = nullptr;
    LoopInit *h_init = nullptr, *h_init_dev = nullptr;
    LoopRecord *h_rec = nullptr, *h_rec_dev = nullptr;
    unsigned long long loop_seq = 0;   // one per enqueued chunk: the value the record's flag takes
    int32_t loop_gen = 0;              // generation of the enqueued plan (a re-plan after a wrong prediction takes a new one)
    std::vector<int8_t> sched_hist;    // which iterations of the last scan searched: the plan for the next one

    EskfWork work;
    Comm comm;  // attached RCCL communicator (multi-GPU form), handle == nullptr when single GPU
    ShmExchange shm;               // or: host shared-memory exchange between the processes of one node (s2m_comm_init_shm)
    std::vector<double> shm_blocks;  // the ranks' blocks of one exchange, padded to a power of two for the tree sum
    std::vector<float> h_changes;    // s2m_map_get_changes: the added points on their way to the caller's arrays
    int32_t queue[S2M_FEAT_QUEUE + 1] = {0};
    int32_t queue_len = 0;
};

namespace {

int complete_lists(s2m_engine *e, int k, int blind, int64_t *n_completed);  // (defined with s2m_complete_neighbors)

int fail(s2m_engine *e, int code, const char *what, hipError_t he = hipSuccess)
{
    if (e) {
        e->err = what;
        if (he != hipSuccess) {
            e->err += ": ";
            e->err += hipGetErrorString(he);
        }
    }
    return code;
}

#define S2M_HIP(e, call)                                                 \
    do {                                                                 \
        hipError_t he_ = (call);                                         \
        if (he_ != hipSuccess) return fail((e), S2M_ERR_HIP, #call, he_); \
    } while (0)

template <class T>
int grow(s2m_engine *e, T **p, int64_t count)
{
    if (*p) S2M_HIP(e, hipFree(*p));
    *p = nullptr;
    S2M_HIP(e, hipMalloc((void **)p, (size_t)std::max<int64_t>(count, 1) * sizeof(T)));
    return S2M_OK;
}

Gates gates_of(const s2m_config &c)
{
    Gates g;
    g.plane_thr = c.plane_thr;
    g.knn_d2_gate = c.knn_d2_gate;
    g.s_gate = c.s_gate;
    g.res_gate = c.res_gate;
    g.extrinsic = c.extrinsic_est_en ? 1 : 0;
    return g;
}

Pose pose_of(const double s[S2M_STATE_DOUBLES])
{
    Pose p;
    std::memcpy(p.R, s + 0, 9 * sizeof(double));
    std::memcpy(p.t, s + 9, 3 * sizeof(double));
    std::memcpy(p.RLI, s + 12, 9 * sizeof(double));
    std::memcpy(p.TLI, s + 21, 3 * sizeof(double));
    return p;
}

int check_config(const s2m_config *c)
{
    if (!c) return S2M_ERR_ARG;
    if (!(c->plane_thr >= 0.0f) || !(c->knn_d2_gate > 0.0f) || !(c->laser_point_cov > 0.0)) return S2M_ERR_ARG;
    if (c->max_iter < 1 || c->max_iter > 64) return S2M_ERR_ARG;
    return S2M_OK;
}

// stage a host or device AoS cloud; returns a device pointer usable until the next stage call
int stage_cloud(s2m_engine *e, const float *xyz, int64_t stride, int64_t count, int on_device, const float **dev)
{
    if (on_device || count == 0) {
        *dev = xyz;
        return S2M_OK;
    }
    const int64_t floats = (count - 1) * stride + 3;
    if (floats > e->stage_cap) {
        int rc = grow(e, &e->d_stage, floats);
        if (rc) return rc;
        e->stage_cap = floats;
    }
    S2M_HIP(e, hipMemcpyAsync(e->d_stage, xyz, (size_t)floats * sizeof(float), hipMemcpyHostToDevice, e->stream));
    *dev = e->d_stage;
    return S2M_OK;
}

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
    S2M_HIP(e, hipSetDevice(e->device));
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
    const bool publish = e->host_poll && d_out == e->d_block && !defer_publish;
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
    S2M_HIP(e, hipEventSynchronize(e->ev[2]));
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

int s2m_abi_version(void) { return S2M_ABI_VERSION; }

int s2m_config_default(s2m_config *c)
{
    if (!c) return S2M_ERR_ARG;
    std::memset(c, 0, sizeof(*c));
    c->plane_thr = 0.1f;
    c->knn_d2_gate = 5.0f;
    c->s_gate = 0.9;
    c->res_gate = 2.0;
    c->laser_point_cov = 0.0015;
    c->conv_rot_deg = 0.01;
    c->conv_pos_cm = 0.015;
    c->extrinsic_est_en = 0;
    c->max_iter = 10;  // mapping/max_iteration default, laserMapping.cpp:656
    c->feat_threshold = 100;
    c->cell_size = 0.0f;
    c->device = -1;
    c->far_point_bet = 1;
    c->device_loop = 0;
    return S2M_OK;
}

const char *s2m_strerror(int code)
{
    switch (code) {
        case S2M_OK: return "ok";
        case S2M_ERR_ARG: return "invalid argument";
        case S2M_ERR_NO_DEVICE: return "no gfx950 HIP device";
        case S2M_ERR_HIP: return "HIP runtime error";
        case S2M_ERR_STATE: return "call order error";
        case S2M_ERR_CAPACITY: return "capacity exceeded";
        case S2M_ERR_NUMERIC: return "singular matrix";
        default: return "unknown error";
    }
}

int s2m_create(const s2m_config *cfg, s2m_engine **out)
{
    if (!out) return S2M_ERR_ARG;
    *out = nullptr;
    int rc = check_config(cfg);
    if (rc) return rc;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return S2M_ERR_NO_DEVICE;
    int dev = cfg->device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return S2M_ERR_NO_DEVICE;
    if (dev >= count) return S2M_ERR_ARG;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return S2M_ERR_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return S2M_ERR_NO_DEVICE;  // kernels are gfx950-only
    s2m_engine *e = new (std::nothrow) s2m_engine();
    if (!e) return S2M_ERR_CAPACITY;
    e->cfg = *cfg;
    e->device = dev;
    if (const char *g = std::getenv("S2M_WIDE_ADDR"))  // test hook: 64-bit point addresses on a small map
        if (std::atoi(g) != 0) e->match_group |= 0x10000;
    if (const char *g = std::getenv("S2M_EASY_NB")) {  // test hook: both first-shell instantiations on any scan size
        const int v = std::atoi(g);
        if (v == 2 || v == 3) e->match_group |= v << 8;
    }
    if (cfg->far_point_bet >= 0 && cfg->far_point_bet <= 2) e->spec_mode = cfg->far_point_bet;
    if (const char *g = std::getenv("S2M_SPEC")) {  // overrides the config: 0 never bet, 1 by history, 2 always (A/B runs, tests)
        const int v = std::atoi(g);
        if (v >= 0 && v <= 2) { e->spec_mode = v; e->spec_env = true; }
    }
    e->no_merge = std::getenv("S2M_NO_MERGE") != nullptr;
    e->no_slab = std::getenv("S2M_NO_SLAB") != nullptr;
    if (const char *g = std::getenv("S2M_FIRST_GAIN")) e->first_round_gain = std::max(1.5f, std::min(256.0f, (float)std::atof(g)));
    if (const char *g = std::getenv("S2M_BLIND_ROUNDS")) e->blind_rounds = std::max(0, std::min(8, std::atoi(g)));  // (A/B runs)
    bool ok = hipSetDevice(dev) == hipSuccess && hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; ok && i < 3; ++i) ok = hipEventCreate(&e->ev[i]) == hipSuccess;
    ok = ok && hipMalloc((void **)&e->d_block, S2M_BLOCK_DOUBLES * sizeof(double)) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&e->h_block, (S2M_BLOCK_DOUBLES + 8) * sizeof(double), hipHostMallocMapped) == hipSuccess;
    ok = ok && hipHostGetDevicePointer((void **)&e->h_block_dev, e->h_block, 0) == hipSuccess;
    ok = ok && hipMalloc((void **)&e->d_ticket, kTicketWords * sizeof(uint32_t)) == hipSuccess &&
         hipMemset(e->d_ticket, 0, kTicketWords * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMalloc((void **)&e->d_qheads, kQueueWords * sizeof(uint32_t)) == hipSuccess &&
         hipMemset(e->d_qheads, 0, kQueueWords * sizeof(uint32_t)) == hipSuccess;
    if (ok) std::memset(e->h_block, 0, (S2M_BLOCK_DOUBLES + 8) * sizeof(double));
    if (!ok) {
        s2m_destroy(e);
        return S2M_ERR_HIP;
    }
    e->stream = e->own_stream;
    *out = e;
    return S2M_OK;
}

int s2m_destroy(s2m_engine *e)
{
    if (!e) return S2M_ERR_ARG;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->pf.worker.joinable()) {
        { std::lock_guard<std::mutex> lk(e->pf.mu); e->pf.quit = true; }
        e->pf.cv.notify_all();
        e->pf.worker.join();
    }
    if (e->pf.stream) { (void)hipStreamSynchronize(e->pf.stream); (void)hipStreamDestroy(e->pf.stream); }
    if (e->pf.done) (void)hipEventDestroy(e->pf.done);
    if (e->pf.d_buf) (void)hipFree(e->pf.d_buf);
    if (e->d_scan_alt) (void)hipFree(e->d_scan_alt);
    free_map(e->map);
    free_update(e->upd);
    free_changelog(e->log);
    free_mailbox(e->mail);
    free_voxel(e->vox);
    free_undist(e->und);
    comm_destroy(e->comm);
    shm_exchange_destroy(e->shm);
    if (e->h_init) (void)hipHostFree(e->h_init);
    if (e->h_rec) (void)hipHostFree(e->h_rec);
    void *ptrs[] = {e->d_loop, e->d_brec, e->d_bcnt, e->d_stage, e->d_scan, e->d_plane, e->d_flags, e->d_sel, e->d_eff, e->d_pd2, e->d_nn_idx,
                    e->d_nn_d2, e->d_hard, e->d_qheads, e->d_hrec, e->d_ticket, e->d_partials, e->d_block, e->d_block_off, e->d_hx, e->d_h, e->d_rowidx};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (e->h_block) (void)hipHostFree(e->h_block);
    for (auto &ev : e->ev)
        if (ev) (void)hipEventDestroy(ev);
    if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
    delete e;
    return S2M_OK;
}

const char *s2m_last_error(const s2m_engine *e) { return e ? e->err.c_str() : "null handle"; }

int s2m_set_config(s2m_engine *e, const s2m_config *cfg)
{
    if (!e) return S2M_ERR_ARG;
    int rc = check_config(cfg);
    if (rc) return fail(e, rc, "invalid config");
    const float cell = e->cfg.cell_size;
    const int dev = e->cfg.device;
    e->cfg = *cfg;
    e->cfg.cell_size = cell;
    e->cfg.device = dev;
    if (e->cfg.far_point_bet >= 0 && e->cfg.far_point_bet <= 2 && !e->spec_env) e->spec_mode = e->cfg.far_point_bet;
    return S2M_OK;
}

int s2m_set_stream(s2m_engine *e, void *hip_stream)
{
    if (!e) return S2M_ERR_ARG;
    S2M_HIP(e, hipStreamSynchronize(e->stream));
    e->stream = hip_stream ? (hipStream_t)hip_stream : e->own_stream;
    return S2M_OK;
}

int s2m_map_build(s2m_engine *e, const float *xyz, int64_t stride, int64_t m, int on_device)
{
    if (!e || m < 0 || stride < 3 || (m > 0 && !xyz)) return fail(e, S2M_ERR_ARG, "s2m_map_build: bad argument");
    if (m >= ((int64_t)1 << 31)) return fail(e, S2M_ERR_CAPACITY, "map too large (>= 2^31 points)");
    S2M_HIP(e, hipSetDevice(e->device));
    const float *dev = nullptr;
    int rc = stage_cloud(e, xyz, stride, m, on_device, &dev);
    if (rc) return rc;
    e->map_ready = false;
    e->map_borrowed = false;
    bool too_large = false;
    hipError_t he = build_map(dev, stride, m, e->cfg.cell_size, e->map, e->grid, e->stats, too_large, e->stream);
    if (he != hipSuccess) return fail(e, S2M_ERR_HIP, "build_map", he);
    if (too_large) return fail(e, S2M_ERR_CAPACITY, "map bounding box too large for the cell size");
    e->map_ready = true;
    e->nn_valid = false;
    e->built_cell = e->grid.c;
    e->log.token = 0;  // (a follower of the old map starts over)
    return S2M_OK;
}

int s2m_map_share(s2m_engine *e, const s2m_engine *owner)
{
    if (!e || !owner || e == owner) return fail(e, S2M_ERR_ARG, "s2m_map_share: bad argument");
    if (!owner->map_ready) return fail(e, S2M_ERR_STATE, "s2m_map_share: the owner has no map");
    if (owner->device != e->device) return fail(e, S2M_ERR_ARG, "s2m_map_share: handles on different devices");
    S2M_HIP(e, hipSetDevice(e->device));
    S2M_HIP(e, hipStreamSynchronize(e->stream));
    S2M_HIP(e, hipStreamSynchronize(owner->stream));  // the owner's build has finished
    {   // the counts of the owner's last merged update arrive lazily; several borrowers may ask at once
        s2m_engine *o = const_cast<s2m_engine *>(owner);
        std::lock_guard<std::mutex> lk(o->stats_mu);
        (void)resolve_stats(o->map, o->stats);
        e->stats = o->stats;
    }
    e->grid = owner->grid;
    e->map.ids_dense = owner->map.ids_dense;  // (the borrower's own map buffers stay empty; the getters ask this flag)
    e->map.next_id = owner->map.next_id;
    e->built_cell = owner->built_cell;
    e->map_ready = true;
    e->map_borrowed = true;
    e->nn_valid = false;
    return S2M_OK;
}

namespace {
// what the update kernels need to know about the map's layout (s2m_kernels.h, UpdateBuffers)
void bind_update(s2m_engine *e)
{
    e->upd.bmark = e->map.bmark;
    e->upd.layout_gen = e->map.layout_gen;
    e->upd.reserve_hint = std::max(e->upd.reserve_hint, e->n_cap);
}

// the map after an update: merged into the sorted arrays when possible (s2m_mapedit.hip: in place, else merge_update), else rebuilt
// from upd.list (survivors in index order, then the staged points) -- the same caller order either way
int commit_update(s2m_engine *e)
{
    bool merged = false;
    e->map_ready = false;
    hipError_t he;
    {
        std::lock_guard<std::mutex> lk(e->stats_mu);
        he = resolve_stats(e->map, e->stats);  // counts of the previous build / merge
    }
    if (he != hipSuccess) return fail(e, S2M_ERR_HIP, "resolve_stats", he);
    // The cell size is kept across updates (a stable grid) unless the density has drifted by more than 2x from
    // the ~11 points per occupied cell it was chosen for -- e.g. a map seeded from a handful of points and then
    // grown, or a dense seed thinned by the voxel rule: then it is chosen again from the density.  A merged update
    // does not wait for its own counts, so the drift it causes is seen when the next update begins.
    // Judged from the counts of the last build or merge and THAT layout's point count (in-place updates change the number
    // of points and of occupied cells alike, and only a layout counts the cells).
    auto drifted = [&]() {
        if (e->cfg.cell_size > 0.0f || e->stats.occupied_cells <= 0 || e->stats.layout_points <= 0) return false;
        const double mean = (double)e->stats.layout_points / (double)e->stats.occupied_cells;
        return mean < 5.5 || mean > 22.0;
    };
    const bool drift_before = drifted();
    const int64_t id0 = e->map.next_id;  // the first id this update hands out
    if (e->log.on && e->log.token != 0 && e->grid.m > 0)   // somebody follows the map: the ids about to disappear, before anything moves
        launch_log_removed(e->log, e->map.counters + kBricksWord, e->stats.bricks, e->map.bmark, e->grid.tab, e->upd.alive_s, e->grid.pidx, e->stream);
    if (!e->no_merge && !e->no_slab && !drift_before) {  // in place when every touched brick fits where it stands
        he = slab_update(e->map, e->grid, e->stats, e->upd.alive_s, e->upd.stage, e->upd.stage_n, e->upd.counters + kUpdSlabWord, merged, e->stream);
        if (he != hipSuccess) return fail(e, S2M_ERR_HIP, "slab_update", he);
        if (merged) ++e->n_inplace;
    }
    if (!e->no_merge && !drift_before && !merged) {
        he = merge_update(e->map, e->grid, e->stats, e->upd.alive_s, e->upd.stage, e->upd.stage_n, merged, e->stream, !e->no_slab);
        if (he != hipSuccess) return fail(e, S2M_ERR_HIP, "merge_update", he);
    }
    e->last_update_merged = merged;
    if (merged) ++e->n_merged; else ++e->n_rebuilt;
    if (e->log.on && e->log.token != 0) {
        if (merged) launch_log_added(e->log, e->upd.stage, e->upd.stage_n, (uint32_t)id0, e->stream);
        else e->log.token = 0;  // a rebuild numbers the points anew: whoever follows the map starts over
    }
    if (!merged) {
        int64_t m_new = 0;
        bool too_large = false;
        he = update_finish(e->upd, e->grid, &m_new, e->stream);
        if (he != hipSuccess) return fail(e, S2M_ERR_HIP, "update_finish", he);
        if (m_new >= ((int64_t)1 << 31)) return fail(e, S2M_ERR_CAPACITY, "map too large (>= 2^31 points)");
        const float cell = e->cfg.cell_size > 0.0f ? e->cfg.cell_size : e->built_cell;
        // the cells stay where they are (same origin) unless the map was empty or has wandered beyond the representable range
        const float origin[3] = {e->grid.ox, e->grid.oy, e->grid.oz};
        const bool keep = e->grid.m > 0 && cell == e->grid.c;
        he = build_map(reinterpret_cast<const float *>(e->upd.list), 4, m_new, cell, e->map, e->grid, e->stats, too_large,
                       e->stream, keep ? origin : nullptr);
        if (he == hipSuccess && too_large && keep)   // beyond the range of the old origin: a new one
            he = build_map(reinterpret_cast<const float *>(e->upd.list), 4, m_new, cell, e->map, e->grid, e->stats, too_large, e->stream);
        if (he != hipSuccess) return fail(e, S2M_ERR_HIP, "build_map", he);
        if (too_large) return fail(e, S2M_ERR_CAPACITY, "map bounding box too large for the cell size");
        if (drifted()) {
            he = build_map(reinterpret_cast<const float *>(e->upd.list), 4, m_new, 0.0f, e->map, e->grid, e->stats, too_large,
                           e->stream);
            if (he != hipSuccess) return fail(e, S2M_ERR_HIP, "build_map", he);
            if (too_large) return fail(e, S2M_ERR_CAPACITY, "map bounding box too large for the cell size");
            e->built_cell = e->grid.c;
            ++e->n_regrid;
        }
    }
    e->map_ready = true;
    e->nn_valid = false;  // neighbour indices referred to the old point list
    if (e->built_cell <= 0.0f) e->built_cell = e->grid.c;
    return S2M_OK;
}
}  // namespace

int s2m_map_add(s2m_engine *e, const float *xyz, int64_t stride, int64_t n, int downsample_on, float downsample_size,
                int on_device, int64_t *n_added)
{
    if (!e || n < 0 || stride < 3 || (n > 0 && !xyz)) return fail(e, S2M_ERR_ARG, "s2m_map_add: bad argument");
    if (downsample_on && !(downsample_size > 0.0f)) return fail(e, S2M_ERR_ARG, "s2m_map_add: downsample size must be > 0");
    if (!e->map_ready) return fail(e, S2M_ERR_STATE, "no map: call s2m_map_build first");
    if (e->map_borrowed) return fail(e, S2M_ERR_STATE, "the map belongs to another handle (s2m_map_share)");
    S2M_HIP(e, hipSetDevice(e->device));
    const float *dev = nullptr;
    int rc = stage_cloud(e, xyz, stride, n, on_device, &dev);
    if (rc) return rc;
    float4 *np = nullptr;
    S2M_HIP(e, xyz_to_float4(e->upd, dev, stride, n, &np, e->stream));
    bind_update(e);
    S2M_HIP(e, update_begin(e->upd, e->grid, e->stream));
    int64_t added = 0;
    S2M_HIP(e, update_add(e->upd, e->grid, np, n, downsample_on != 0, downsample_size, &added, e->stream));
    if (n_added) *n_added = added;
    return commit_update(e);
}

int s2m_map_delete_boxes(s2m_engine *e, const float *boxes, int64_t n, int64_t *n_deleted)
{
    if (!e || n < 0 || (n > 0 && !boxes) || n > 4096) return fail(e, S2M_ERR_ARG, "s2m_map_delete_boxes: bad argument");
    if (!e->map_ready) return fail(e, S2M_ERR_STATE, "no map: call s2m_map_build first");
    if (e->map_borrowed) return fail(e, S2M_ERR_STATE, "the map belongs to another handle (s2m_map_share)");
    S2M_HIP(e, hipSetDevice(e->device));
    bind_update(e);
    S2M_HIP(e, update_begin(e->upd, e->grid, e->stream));
    int64_t del = 0;
    S2M_HIP(e, update_delete(e->upd, e->grid, boxes, (int)n, &del, e->stream));
    if (n_deleted) *n_deleted = del;
    if (del == 0) return S2M_OK;  // nothing changed: keep the grid and the neighbour indices
    return commit_update(e);
}

int s2m_fov_reset(s2m_engine *e)
{
    if (!e) return S2M_ERR_ARG;
    e->local_map_init = false;
    return S2M_OK;
}

int s2m_fov_segment(s2m_engine *e, const double pos_lid[3], double cube_len, float local_map[6], int32_t *n_boxes,
                    int64_t *n_deleted)
{
    if (!e || !pos_lid || !(cube_len > 0.0)) return fail(e, S2M_ERR_ARG, "s2m_fov_segment: bad argument");
    if (n_boxes) *n_boxes = 0;
    if (n_deleted) *n_deleted = 0;
    float boxes[3][6];
    const int nb = fov_step(e->local_map, e->local_map_init, pos_lid, cube_len, boxes);  // :313-366 (s2m_fov.h)
    if (local_map) std::memcpy(local_map, e->local_map, sizeof(e->local_map));
    if (n_boxes) *n_boxes = nb;
    if (nb > 0 && e->map_ready) return s2m_map_delete_boxes(e, &boxes[0][0], nb, n_deleted);  // :367-368
    return S2M_OK;
}

int s2m_map_incremental(s2m_engine *e, const double state[S2M_STATE_DOUBLES], double filter_size_map,
                        int32_t ekf_inited, int64_t *n_to_add, int64_t *n_no_downsample)
{
    if (!e || !state || !(filter_size_map > 0.0)) return fail(e, S2M_ERR_ARG, "s2m_map_incremental: bad argument");
    if (!e->map_ready) return fail(e, S2M_ERR_STATE, "no map: call s2m_map_build first");
    if (e->map_borrowed) return fail(e, S2M_ERR_STATE, "the map belongs to another handle (s2m_map_share)");
    if (!e->scan_ready) return fail(e, S2M_ERR_STATE, "no scan: call s2m_scan_set first");
    S2M_HIP(e, hipSetDevice(e->device));
    const Pose pose = pose_of(state);
    float4 *la = nullptr, *lb = nullptr;
    int64_t na = 0, nb = 0;
    // Nearest_Points[i] of the reference is never short (unbounded search): finish the lists that ended at the gate -- when
    // the last rematch pass reported any (block[159]; a scan inside the mapped area has none: no launch, no round trip)
    if (e->nn_valid && ekf_inited != 0 && e->short_lists != 0 && !e->nn_complete && !e->nn_nearest && e->n > 0 && e->grid.m > 0) {
        int rc = complete_lists(e, 1, e->blind_rounds, nullptr);
        if (rc) return rc;
        e->nn_nearest = true;
    }
    VoxBox vox;
    bind_update(e);
    S2M_HIP(e, incr_classify(e->upd, pose, e->d_scan, e->d_scan + e->n_cap, e->d_scan + 2 * e->n_cap, (int)e->n,
                             e->d_nn_idx, e->grid, e->nn_valid && ekf_inited != 0, filter_size_map, &la, &na, &lb, &nb, e->stream,
                             &vox, true));   // (update_begin runs inside, while the counts travel to the host)
    if (n_to_add) *n_to_add = na;
    if (n_no_downsample) *n_no_downsample = nb;
    S2M_HIP(e, update_add(e->upd, e->grid, la, na, true, (float)filter_size_map, nullptr, e->stream, &vox));   // :627
    S2M_HIP(e, update_add(e->upd, e->grid, lb, nb, false, 0.0f, nullptr, e->stream));                    // :628
    return commit_update(e);
}

namespace {
// rank[position] = caller index of every sorted position: the point ids themselves while no point has been removed since
// the last build, else their ranks (a sort of the ids: the getters that answer in caller indices are not on any hot path)
int caller_index_table(s2m_engine *e, const uint32_t **rank)
{
    *rank = e->grid.pidx;
    if (e->map.ids_dense) return S2M_OK;
    int64_t live = 0;
    S2M_HIP(e, caller_ranks(e->upd, e->grid, nullptr, rank, &live, e->stream));
    if (live != e->grid.live) return fail(e, S2M_ERR_STATE, "map ids out of step with the map size");
    return S2M_OK;
}
}  // namespace

int s2m_map_get_points(s2m_engine *e, float *xyz, int64_t capacity, int64_t *m)
{
    if (!e || !m) return fail(e, S2M_ERR_ARG, "null argument");
    if (!e->map_ready) return fail(e, S2M_ERR_STATE, "no map");
    *m = e->grid.live;
    if (!xyz) return S2M_OK;
    if (capacity < e->grid.live) return fail(e, S2M_ERR_CAPACITY, "point buffer too small");
    if (e->grid.live == 0) return S2M_OK;
    S2M_HIP(e, hipSetDevice(e->device));
    const int64_t floats = e->grid.live * 3;
    if (floats > e->stage_cap) {
        int rc = grow(e, &e->d_stage, floats);
        if (rc) return rc;
        e->stage_cap = floats;
    }
    const uint32_t *rank = nullptr;
    int rc = caller_index_table(e, &rank);
    if (rc) return rc;
    launch_map_to_xyz(e->grid.pts, rank, e->grid.m, e->d_stage, e->stream);  // caller order
    S2M_HIP(e, hipMemcpyAsync(xyz, e->d_stage, (size_t)floats * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    S2M_HIP(e, hipStreamSynchronize(e->stream));
    return S2M_OK;
}

int s2m_map_size(const s2m_engine *e, int64_t *m)
{
    if (!e || !m) return S2M_ERR_ARG;
    *m = e->map_ready ? e->grid.live : 0;
    return S2M_OK;
}

int s2m_map_last_update(const s2m_engine *e, int32_t *merged)
{
    if (!e || !merged) return S2M_ERR_ARG;
    *merged = e->last_update_merged ? 1 : 0;
    return S2M_OK;
}

int s2m_map_info(const s2m_engine *ce, double info[8])
{
    if (!ce || !info) return S2M_ERR_ARG;
    if (!ce->map_ready) return S2M_ERR_STATE;
    s2m_engine *e = const_cast<s2m_engine *>(ce);  // the counts of a merged update are fetched on demand
    if (!e->map_borrowed) {
        std::lock_guard<std::mutex> lk(e->stats_mu);
        if (resolve_stats(e->map, e->stats) != hipSuccess) return S2M_ERR_HIP;
    }
    info[0] = e->grid.c;
    info[1] = e->grid.ox; info[2] = e->grid.oy; info[3] = e->grid.oz;
    info[4] = (double)e->stats.bricks;
    info[5] = (double)e->stats.top_entries;
    info[6] = (double)e->stats.occupied_cells;
    info[7] = e->stats.occupied_cells ? (double)e->grid.live / (double)e->stats.occupied_cells : 0.0;
    return S2M_OK;
}

namespace {
void pf_drain(s2m_engine *e);  // the side thread (s2m_scan_prefetch_raw / s2m_scan_prepare_raw) is idle
void pf_invalidate(s2m_engine *e);

int scan_reserve(s2m_engine *e, int64_t n)
{
    if (n <= e->n_cap && e->n_cap > 0) return S2M_OK;
    pf_drain(e);  // (a prepared scan was laid out for the old capacity: it is recognised as stale when it is picked up)
    const int64_t cap = ((std::max<int64_t>(n, 1) + 255) / 256) * 256;  // an empty first scan still gets buffers
    int rc = 0;
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
    if (wait) S2M_HIP(e, mail_wait(e->mail, e->stream));  // the host buffer may be reused by the caller now
    e->n = n;
    e->scan_ready = true;
    e->pass_done = false;
    e->nn_valid = false;
    return S2M_OK;
}
}  // namespace

int s2m_scan_set(s2m_engine *e, const float *xyz, int64_t stride, int64_t n, int on_device)
{
    if (!e || n < 0 || stride < 3 || (n > 0 && !xyz)) return fail(e, S2M_ERR_ARG, "s2m_scan_set: bad argument");
    if (n > (int64_t)1 << 28) return fail(e, S2M_ERR_CAPACITY, "scan too large");
    S2M_HIP(e, hipSetDevice(e->device));
    pf_invalidate(e);  // whatever the side thread holds was meant for a sweep that is not coming by this road
    int rc = scan_reserve(e, n);
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
    S2M_HIP(e, hipSetDevice(e->device));
    pf_invalidate(e);  // the voxel-grid buffers are shared with the side thread; what it holds is stale
    int rc = scan_reserve(e, n);  // the output cannot be larger than the input
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
    S2M_HIP(e, hipSetDevice(e->device));
    pf_drain(e);  // the undistortion buffers are shared with the side thread
    e->pf.ordered = false;  // (a time order the side thread left there is about to be overwritten)
    const float *dev = nullptr;
    // stage whole records (the time fields may sit anywhere in the record)
    if (on_device) {
        dev = points;
    } else {
        const int64_t floats = n * stride;
        if (floats > e->stage_cap) {
            rc = grow(e, &e->d_stage, floats);
            if (rc) return rc;
            e->stage_cap = floats;
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
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
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
    std::unique_lock<std::mutex> lk(p.mu);
    for (;;) {
        if (!(p.quit || p.busy)) {  // the next job usually follows within a frame: poll for it before going to sleep
            lk.unlock();
            for (int spin = 0; spin < 20000 && p.busy_a.load(std::memory_order_acquire) == 0; ++spin) __builtin_ia32_pause();
            lk.lock();
        }
        p.cv.wait(lk, [&] { return p.quit || p.busy; });
        if (p.quit) return;
        const float *src = p.src;
        const int64_t floats = p.floats;
        const bool prepare = p.prepare;
        lk.unlock();
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
void pf_drain(s2m_engine *e)
{
    if (!e->pf.worker.joinable()) return;
    for (int spin = 0; spin < 40000 && e->pf.busy_a.load(std::memory_order_acquire) != 0; ++spin) __builtin_ia32_pause();
    std::unique_lock<std::mutex> lk(e->pf.mu);
    e->pf.cv.wait(lk, [&] { return !e->pf.busy; });
    if (e->pf.gpu_pending && e->pf.done) {
        (void)hipStreamWaitEvent(e->stream, e->pf.done, 0);
        e->pf.gpu_pending = false;
    }
}
// a scan arrives by another road than the one the side thread prepared for: what it holds is stale (a node that recycles
// its host buffers may present NEW records at the address and size of a sweep that was prefetched and then dropped)
void pf_invalidate(s2m_engine *e)
{
    pf_drain(e);
    e->pf.ready = false;
    e->pf.prepared = false;
    e->pf.ordered = false;
}

int pf_start(s2m_engine *e, const float *points, int64_t floats)
{
    auto &p = e->pf;
    pf_drain(e);
    p.ready = false;
    p.prepared = false;
    if (!p.stream) S2M_HIP(e, hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking));
    if (!p.done) S2M_HIP(e, hipEventCreateWithFlags(&p.done, hipEventDisableTiming));
    if (floats > p.cap) {
        S2M_HIP(e, hipStreamSynchronize(p.stream));
        if (p.d_buf) S2M_HIP(e, hipFree(p.d_buf));
        p.d_buf = nullptr;
        S2M_HIP(e, hipMalloc((void **)&p.d_buf, (size_t)floats * sizeof(float)));
        p.cap = floats;
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
        pf_invalidate(e);
        return S2M_OK;
    }
    if (n < 0 || stride < 3 || oa >= stride || ob >= stride) return fail(e, S2M_ERR_ARG, "s2m_scan_prefetch_raw: bad argument");
    if (n == 0) return S2M_OK;
    S2M_HIP(e, hipSetDevice(e->device));
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
    S2M_HIP(e, hipSetDevice(e->device));
    // the prepared scan must fit the arrays of the current one (they are swapped, not copied); a first or larger sweep is
    // left to the synchronous call
    if (e->n_cap < n) return S2M_OK;
    pf_drain(e);
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
    S2M_HIP(e, hipSetDevice(e->device));
    pf_drain(e);
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
            if (p.src == points && p.floats == n * stride && (p.busy || p.ready)) {
                p.cv.wait(lk, [&] { return !p.busy; });
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
                rc = grow(e, &e->d_stage, floats);
                if (rc) return rc;
                e->stage_cap = floats;
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
    S2M_HIP(e, hipSetDevice(e->device));
    S2M_HIP(e, hipStreamSynchronize(e->stream));
    // three strided copies SoA -> packed AoS
    for (int k = 0; k < 3; ++k)
        S2M_HIP(e, hipMemcpy2D(xyz + k, 3 * sizeof(float), e->d_scan + k * e->n_cap, sizeof(float), sizeof(float),
                               (size_t)e->n, hipMemcpyDeviceToHost));
    return S2M_OK;
}

// Wait for the block of the pass just enqueued and return a host pointer to it.  Fast path: the
// reduce kernel writes the block and a sequence flag straight into pinned host memory and the host
// spins on the flag (no D2H copy, no driver sync).  Fallback: D2H copy + stream synchronise.
static int wait_block(s2m_engine *e, const double *d_src, const double **host)
{
    if (e->host_poll && d_src == e->d_block) {
        volatile unsigned long long *flag = reinterpret_cast<volatile unsigned long long *>(e->h_block + S2M_BLOCK_DOUBLES);
        bool seen = false;
        for (long spin = 0; spin < 20000000L; ++spin) {
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == e->seq) { seen = true; break; }
            __builtin_ia32_pause();
        }
        if (!seen) {  // kernel slow or failed: let the runtime tell us
            S2M_HIP(e, hipStreamSynchronize(e->stream));
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != e->seq) return fail(e, S2M_ERR_HIP, "reduce kernel did not publish its block");
        }
        *host = e->h_block;
        return S2M_OK;
    }
    S2M_HIP(e, hipMemcpyAsync(e->h_block, d_src, S2M_BLOCK_DOUBLES * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    S2M_HIP(e, hipStreamSynchronize(e->stream));
    *host = e->h_block;
    return S2M_OK;
}

int s2m_residual_pass(s2m_engine *e, const double state[S2M_STATE_DOUBLES], int rematch, s2m_pass_out *out)
{
    if (!out) return fail(e, S2M_ERR_ARG, "null output");
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
    S2M_HIP(e, hipSetDevice(e->device));
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
    S2M_HIP(e, hipStreamSynchronize(e->stream));
    *m_out = m;
    if ((h_x || h || scan_index) && capacity < (int64_t)m) return fail(e, S2M_ERR_CAPACITY, "row buffers too small");
    if (h_x && m) S2M_HIP(e, hipMemcpy(h_x, e->d_hx, (size_t)m * 12 * sizeof(double), hipMemcpyDeviceToHost));
    if (h && m) S2M_HIP(e, hipMemcpy(h, e->d_h, (size_t)m * sizeof(double), hipMemcpyDeviceToHost));
    if (scan_index && m) S2M_HIP(e, hipMemcpy(scan_index, e->d_rowidx, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost));
    return S2M_OK;
}

int s2m_get_point_state(s2m_engine *e, uint8_t *selected, uint8_t *effective, float *plane, float *pd2)
{
    if (!e) return S2M_ERR_ARG;
    if (!e->pass_done) return fail(e, S2M_ERR_STATE, "no pass yet");
    S2M_HIP(e, hipSetDevice(e->device));
    S2M_HIP(e, hipStreamSynchronize(e->stream));
    const size_t n = (size_t)e->n;
    if (n == 0) return S2M_OK;
    if (selected) S2M_HIP(e, hipMemcpy(selected, e->d_sel, n, hipMemcpyDeviceToHost));
    if (effective) S2M_HIP(e, hipMemcpy(effective, e->d_eff, n, hipMemcpyDeviceToHost));
    if (plane) S2M_HIP(e, hipMemcpy(plane, e->d_plane, n * sizeof(float4), hipMemcpyDeviceToHost));
    if (pd2) S2M_HIP(e, hipMemcpy(pd2, e->d_pd2, n * sizeof(float), hipMemcpyDeviceToHost));
    return S2M_OK;
}

int s2m_get_neighbors(s2m_engine *e, int32_t *idx, float *d2)
{
    if (!e) return S2M_ERR_ARG;
    if (!e->nn_valid) return fail(e, S2M_ERR_STATE, "no rematch pass yet");
    S2M_HIP(e, hipSetDevice(e->device));
    S2M_HIP(e, hipStreamSynchronize(e->stream));
    const size_t n = (size_t)e->n;
    if (n == 0) return S2M_OK;
    if (idx) {  // the engine identifies a neighbour by its sorted position; the caller's indices are looked up on request
        const int64_t words = (int64_t)n * S2M_K;
        if (words > e->stage_cap) {
            int rc = grow(e, &e->d_stage, words);
            if (rc) return rc;
            e->stage_cap = words;
        }
        int32_t *tmp = reinterpret_cast<int32_t *>(e->d_stage);
        const uint32_t *rank = nullptr;
        int rc = caller_index_table(e, &rank);
        if (rc) return rc;
        launch_positions_to_indices(e->d_nn_idx, rank, words, tmp, e->stream);
        S2M_HIP(e, hipMemcpyAsync(idx, tmp, (size_t)words * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
        S2M_HIP(e, hipStreamSynchronize(e->stream));
    }
    if (d2) S2M_HIP(e, hipMemcpy(d2, e->d_nn_d2, n * S2M_K * sizeof(float), hipMemcpyDeviceToHost));
    return S2M_OK;
}

int s2m_map_get_ids(s2m_engine *e, uint32_t *ids, int64_t capacity, int64_t *m)
{
    if (!e || !m) return fail(e, S2M_ERR_ARG, "null argument");
    if (!e->map_ready) return fail(e, S2M_ERR_STATE, "no map");
    *m = e->grid.live;
    if (!ids || e->grid.live == 0) return S2M_OK;
    if (capacity < e->grid.live) return fail(e, S2M_ERR_CAPACITY, "id buffer too small");
    S2M_HIP(e, hipSetDevice(e->device));
    if (e->grid.live > e->stage_cap) {
        int rc = grow(e, &e->d_stage, e->grid.live);
        if (rc) return rc;
        e->stage_cap = e->grid.live;
    }
    const uint32_t *rank = nullptr;
    int rc = caller_index_table(e, &rank);
    if (rc) return rc;
    launch_ids_by_rank(e->grid.pidx, rank, e->grid.m, reinterpret_cast<uint32_t *>(e->d_stage), e->stream);
    S2M_HIP(e, hipMemcpyAsync(ids, e->d_stage, (size_t)e->grid.live * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    S2M_HIP(e, hipStreamSynchronize(e->stream));
    return S2M_OK;
}

int s2m_map_get_changes(s2m_engine *e, uint64_t *token, float *added_xyz, uint32_t *added_ids, int64_t cap_added, int64_t *n_added,
                        uint32_t *removed_ids, int64_t cap_removed, int64_t *n_removed, int32_t *resync)
{
    if (!e || !token || !n_added || !n_removed || !resync) return fail(e, S2M_ERR_ARG, "null argument");
    if (!e->map_ready) return fail(e, S2M_ERR_STATE, "no map");
    if (e->map_borrowed) return fail(e, S2M_ERR_STATE, "the map belongs to another handle (s2m_map_share)");
    S2M_HIP(e, hipSetDevice(e->device));
    *n_added = 0; *n_removed = 0; *resync = 0;
    auto fresh = [&]() {
        e->log.on = true;
        e->log.token = (++e->log_seq << 8) | 1u;
        *token = e->log.token;
    };
    if (!e->log.on || e->log.token == 0 || *token != e->log.token) {
        // the caller does not hold the state the log starts from (first call, a rebuild in between, another follower's token)
        // (room for a field-of-view trim of a few million points: 20 bytes per entry; beyond that the follower fetches the map)
        S2M_HIP(e, changelog_ensure(e->log, std::max<int64_t>((int64_t)1 << 22, 4 * e->n_cap), e->stream));
        launch_log_reset(e->log, e->stream);
        fresh();
        *resync = 1;
        return S2M_OK;
    }
    const uint32_t *src[3] = {e->log.counts, e->log.counts + 1, e->log.counts + 2};
    uint32_t v[3] = {0, 0, 0};
    S2M_HIP(e, mail_fetch(e->mail, src, 3, v, e->stream));
    if (v[2] != 0u) {  // more changes than the log holds: start over
        launch_log_reset(e->log, e->stream);
        fresh();
        *resync = 1;
        return S2M_OK;
    }
    *n_added = v[0];
    *n_removed = v[1];
    if ((v[0] > 0 && (!added_xyz || !added_ids || cap_added < (int64_t)v[0])) || (v[1] > 0 && (!removed_ids || cap_removed < (int64_t)v[1])))
        return fail(e, S2M_ERR_CAPACITY, "s2m_map_get_changes: buffers too small (the changes are kept)");
    if (v[0] > 0) {
        e->h_changes.resize((size_t)v[0] * 4);
        S2M_HIP(e, hipMemcpyAsync(e->h_changes.data(), e->log.added, (size_t)v[0] * sizeof(float4), hipMemcpyDeviceToHost, e->stream));
    }
    if (v[1] > 0) S2M_HIP(e, hipMemcpyAsync(removed_ids, e->log.removed, (size_t)v[1] * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    launch_log_reset(e->log, e->stream);
    S2M_HIP(e, hipStreamSynchronize(e->stream));
    for (uint32_t i = 0; i < v[0]; ++i) {
        const float *p = e->h_changes.data() + (size_t)i * 4;
        added_xyz[3 * (size_t)i] = p[0]; added_xyz[3 * (size_t)i + 1] = p[1]; added_xyz[3 * (size_t)i + 2] = p[2];
        std::memcpy(&added_ids[i], &p[3], sizeof(uint32_t));
    }
    fresh();
    return S2M_OK;
}

int s2m_map_get_order(s2m_engine *e, uint32_t *order, int64_t capacity, int64_t *m)
{
    if (!e || !m) return fail(e, S2M_ERR_ARG, "null argument");
    if (!e->map_ready) return fail(e, S2M_ERR_STATE, "no map");
    *m = e->grid.m;
    if (!order || e->grid.m == 0) return S2M_OK;
    if (capacity < e->grid.m) return fail(e, S2M_ERR_CAPACITY, "order buffer too small");
    S2M_HIP(e, hipSetDevice(e->device));
    const uint32_t *rank = nullptr;
    int rc = caller_index_table(e, &rank);
    if (rc) return rc;
    S2M_HIP(e, hipStreamSynchronize(e->stream));
    S2M_HIP(e, hipMemcpy(order, rank, (size_t)e->grid.m * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return S2M_OK;
}

int s2m_map_grid(const s2m_engine *e, int32_t bricks[6])
{
    if (!e || !bricks) return S2M_ERR_ARG;
    if (!e->map_ready) return S2M_ERR_STATE;
    for (int k = 0; k < 3; ++k) { bricks[k] = e->grid.blo[k]; bricks[3 + k] = e->grid.bhi[k]; }
    return S2M_OK;
}

int s2m_map_inplace_updates(const s2m_engine *e, int64_t *n)
{
    if (!e || !n) return S2M_ERR_ARG;
    *n = e->n_inplace;
    return S2M_OK;
}

int s2m_map_update_stats(const s2m_engine *e, int64_t stats[6])
{
    if (!e || !stats) return S2M_ERR_ARG;
    stats[0] = e->n_merged;
    stats[1] = e->n_rebuilt;
    stats[2] = e->n_regrid;
    stats[3] = map_allocations();
    stats[4] = e->map.n_relaid;
    stats[5] = e->map.n_big_slab;
    return S2M_OK;
}

// Nearest_Points beyond the gate.  ikdtree.Nearest_Search is unbounded (max_dist = INFINITY, ikd_Tree.cpp:425): every
// scan point gets its five nearest map points however far they are, and map_incremental reads points_near[0] of
// exactly those far points when the sensor enters new territory (laserMapping.cpp:593-607).  The per-iteration search
// stops at the d2 <= 5 gate (:853) -- nothing beyond it can enter the update -- so a point whose neighbourhood is
// emptier than that ends the pass with a list that is short, or not proven beyond the gate.  This call completes those
// lists: the points whose 5th distance is not inside the radius searched so far are collected and handed to the
// far-point kernel again with the radius doubled per round, until every one has its exact five (or the radius exceeds
// the grid).  Cold path: nothing to do for a scan inside the mapped area.
namespace {
// k = 5: every list complete (s2m_complete_neighbors).  k = 1: only the NEAREST neighbour of every scan point proven -- all
// that map_incremental reads of a list beyond the gate (laserMapping.cpp:603 tests points_near[0]; the other entries of a
// list that ends at the gate lie more than sqrt(5) m from the point and cannot pass the test of :612-616, see
// incr_classify_kernel) -- a radius that a frontier point reaches one or two doublings earlier than the radius that holds
// five.  blind: that many rounds are enqueued without asking the device whether anything is still open (a round over an
// empty list costs a few microseconds, a question ~15): the sensor at the edge of the mapped area always has open lists.
int complete_lists(s2m_engine *e, int k, int blind, int64_t *n_completed)
{
    const int n = (int)e->n;
    MatchArgs m;
    m.grid = e->grid; m.pose = e->rematch_pose; m.gates = gates_of(e->cfg);
    m.sx = e->d_scan; m.sy = e->d_scan + e->n_cap; m.sz = e->d_scan + 2 * e->n_cap; m.n = n;
    m.nn_idx = e->d_nn_idx; m.nn_d2 = e->d_nn_d2;
    m.hard_rec = e->d_hrec; m.hard_off1 = n; m.hard_count = e->d_hard + 3 * e->n_cap;
    m.qheads = e->d_qheads;
    m.short_k = k;
    const double c = e->grid.c;
    double diag2 = 0.0;
    for (int q = 0; q < 3; ++q) {  // (the box of the bricks in use, in cells)
        const double cells = 8.0 * ((double)e->grid.bhi[q] - (double)e->grid.blo[q] + 1.0);
        diag2 += cells * cells;
    }
    const double half_diag = 0.5 * c * std::sqrt(diag2);
    int64_t first = -1;
    for (int round = 0; round < 64; ++round) {
        // hard_count / qheads are zero here: every reduce launch and every round below leaves them so
        launch_collect_short(m, e->stream);
        const bool ask = round >= blind;
        bool last = false;
        double reach = 0.0;
        if (ask) {
            const uint32_t *src[2] = {m.hard_count, m.hard_count + 2};
            uint32_t v[2] = {0, 0};
            S2M_HIP(e, mail_fetch(e->mail, src, 2, v, e->stream));
            if (first < 0) first = v[0];
            last = v[0] == 0;
            // the farthest of the open queries from the grid centre, plus half the grid's diagonal: a radius beyond
            // that has seen every map point (a list still short then belongs to a map of fewer than five points)
            float far2;
            std::memcpy(&far2, &v[1], sizeof(far2));
            reach = std::sqrt((double)far2) + half_diag + c;
        }
        if (!last) {
            m.gates.knn_d2_gate *= (round == 0 && k == 1) ? e->first_round_gain : 4.0f;  // radius x 2
            launch_match_hard_only(m, e->stream);
            if (ask) last = (double)m.gates.knn_d2_gate > reach * reach || !(m.gates.knn_d2_gate < 1.0e37f);
        }
        launch_far_reset(m.hard_count, e->d_qheads, e->stream);
        if (last) break;
    }
    S2M_HIP(e, hipGetLastError());
    if (n_completed) *n_completed = first < 0 ? 0 : first;
    return S2M_OK;
}
}  // namespace

int s2m_complete_neighbors(s2m_engine *e, int64_t *n_completed)
{
    if (n_completed) *n_completed = 0;
    if (!e) return S2M_ERR_ARG;
    if (!e->map_ready || !e->nn_valid) return fail(e, S2M_ERR_STATE, "no rematch pass yet");
    S2M_HIP(e, hipSetDevice(e->device));
    if (e->n == 0 || e->grid.m == 0 || e->nn_complete) return S2M_OK;
    if (e->short_lists == 0) { e->nn_complete = true; return S2M_OK; }  // the last rematch pass counted them: none
    int rc = complete_lists(e, kK, 0, n_completed);
    if (rc) return rc;
    e->nn_complete = true;
    e->nn_nearest = true;
    return S2M_OK;
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

int loop_wait(s2m_engine *e, unsigned long long seq)
{
    volatile unsigned long long *flag = &e->h_rec->flag;
    for (long spin = 0; spin < 40000000L; ++spin) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return S2M_OK;
        __builtin_ia32_pause();
    }
    S2M_HIP(e, hipStreamSynchronize(e->stream));
    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) return fail(e, S2M_ERR_HIP, "the device-resident loop did not publish its record");
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
    S2M_HIP(e, hipSetDevice(e->device));
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
    S2M_HIP(e, hipSetDevice(e->device));
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
    if (!reduce && (!d_block || d_block == e->d_block)) {  // state on the device where the form allows it (s2m_loop.h)
        bool used = false;
        int rc = iterated_update_loop(e, x, x_prop, P, log, used);
        if (rc || used) return rc;
    }
    if (!d_block) d_block = e->d_block;
    // a new scan starts with every point selected and no neighbours (laserMapping.cpp:810-818);
    // iteration 0 is always a rematch pass, whose gate rewrites point_selected_surf for every point
    S2M_HIP(e, hipSetDevice(e->device));
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
            S2M_HIP(e, hipStreamSynchronize(e->stream));
            hb = e->h_block;
            rc = S2M_OK;
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
        S2M_HIP(e, hipSetDevice(e->device));
        e->nn_valid = false;
        reset_log(logs ? logs + i : nullptr, e->cfg.max_iter);
        int rc = launch(i);
        if (rc) return rc;
    }
    int active = k;
    long idle_spins = 0;
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
        if (progress) { idle_spins = 0; continue; }
        __builtin_ia32_pause();
        if (++idle_spins > 200000000L) {  // a kernel failed: let the runtime say which
            for (int i = 0; i < k; ++i)
                if (slots[i].active) S2M_HIP(handles[i], hipStreamSynchronize(handles[i]->stream));
            return fail(handles[0], S2M_ERR_HIP, "s2m_iterated_update_batch: a pass did not publish its block");
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
    S2M_HIP(e0, hipSetDevice(e0->device));
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
            if (e != L && e->stream != L->stream) S2M_HIP(e, hipStreamSynchronize(e->stream));
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
    long idle_spins = 0;
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
        if (progress) { idle_spins = 0; continue; }
        __builtin_ia32_pause();
        if (++idle_spins > 200000000L) {  // a kernel failed: let the runtime say which
            for (int g = 0; g < ng; ++g) S2M_HIP(groups[g].lead, hipStreamSynchronize(groups[g].lead->stream));
            return fail(handles[0], S2M_ERR_HIP, "s2m_iterated_update_batch: a pass did not publish its block");
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
    S2M_HIP(e0, hipSetDevice(e0->device));
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
            if (e != L && e->stream != L->stream) S2M_HIP(e, hipStreamSynchronize(e->stream));
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
            int rc = loop_wait(e, e->loop_seq);
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

int s2m_comm_unique_id(uint8_t id[S2M_COMM_ID_BYTES])
{
    if (!id) return S2M_ERR_ARG;
    std::string err;
    return comm_unique_id(id, err) ? S2M_OK : S2M_ERR_HIP;
}

int s2m_comm_init(s2m_engine *e, const uint8_t id[S2M_COMM_ID_BYTES], int32_t nranks, int32_t rank)
{
    if (!e || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(e, S2M_ERR_ARG, "s2m_comm_init: bad argument");
    S2M_HIP(e, hipSetDevice(e->device));
    comm_destroy(e->comm);
    shm_exchange_destroy(e->shm);   // one exchange at a time
    std::string err;
    if (!comm_init(e->comm, id, nranks, rank, err)) return fail(e, S2M_ERR_HIP, err.c_str());
    return S2M_OK;
}

int s2m_comm_init_shm(s2m_engine *e, const char *name, int32_t nranks, int32_t rank)
{
    if (!e || !name || nranks < 1 || nranks > 256 || rank < 0 || rank >= nranks) return fail(e, S2M_ERR_ARG, "s2m_comm_init_shm: bad argument");
    if (!e->host_poll) return fail(e, S2M_ERR_STATE, "s2m_comm_init_shm: needs the host-polled block");
    comm_destroy(e->comm);
    std::string err;
    if (!shm_exchange_init(e->shm, name, nranks, rank, err)) return fail(e, S2M_ERR_HIP, err.c_str());
    return S2M_OK;
}

int s2m_comm_destroy(s2m_engine *e)
{
    if (!e) return S2M_ERR_ARG;
    comm_destroy(e->comm);
    shm_exchange_destroy(e->shm);
    return S2M_OK;
}

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
