// s2m_engine.cpp -- the handle: lifetime, configuration, exchange attachment; and the helpers the other translation
// units of the engine share (s2m_engine_internal.h).  There is no CPU fallback for any compute entry point.
#include "s2m_engine_internal.h"

using namespace s2m;
using namespace s2m_eng;

namespace s2m_eng {

int fail(s2m_engine *e, int code, const char *what, hipError_t he)
{
    if (he == kWaitTimedOut) {  // a wait of s2m_wait.h expired somewhere below `what`
        code = S2M_ERR_TIMEOUT;
        if (e && !e->poisoned) {
            const char *w = e->wait.expired.load();
            char head[256];
            std::snprintf(head, sizeof(head), "no answer after %.0f ms while waiting for %s (in %s); ",
                          1e-3 * (double)e->wait.waited_us.load(), w ? w : what, what);
            e->err = head + debug_state(e);
            e->poisoned = true;
        }
        return code;
    }
    if (e) {
        e->err = what;
        if (he != hipSuccess) {
            e->err += ": ";
            e->err += hipGetErrorString(he);
        }
    }
    return code;
}

int refuse_poisoned(s2m_engine *e)
{
    if (e->err.compare(0, 9, "given up:") != 0) e->err = "given up: " + e->err;
    return S2M_ERR_TIMEOUT;
}

int sync_stream(s2m_engine *e, hipStream_t st, const char *what)
{
    e->step = what;
    const hipError_t he = wait_stream(&e->wait, st, what);
    if (he != hipSuccess) return fail(e, S2M_ERR_HIP, what, he);
    return S2M_OK;
}

// one line for a watchdog or an error message: where the caller is, what the streams and the side thread are doing, every
// pinned hand-back word against the number the host expects.  Reads only; racy by design (diagnostic).
std::string debug_state(const s2m_engine *e)
{
    char b[1024];
    auto mail = [](const Mailbox &m, char *out, size_t cap, const char *name) {
        if (m.h) std::snprintf(out, cap, " %s %u/%u", name, (unsigned)__atomic_load_n(m.h, __ATOMIC_ACQUIRE), (unsigned)m.seq);
        else out[0] = 0;
    };
    char m0[48], m1[48], m2[48], m3[48], m4[48];
    mail(e->mail, m0, sizeof(m0), "handle");
    mail(e->map.mail, m1, sizeof(m1), "map");
    mail(e->upd.mail, m2, sizeof(m2), "update");
    mail(e->vox.mail, m3, sizeof(m3), "voxel");
    mail(e->und.mail, m4, sizeof(m4), "undistort");
    const hipError_t q0 = e->stream ? hipStreamQuery(e->stream) : hipSuccess;
    const hipError_t q1 = e->pf.stream ? hipStreamQuery(e->pf.stream) : hipSuccess;
    auto qs = [](hipError_t q) { return q == hipSuccess ? "idle" : q == hipErrorNotReady ? "busy" : hipGetErrorName(q); };
    const unsigned long long flag = e->h_block ? __atomic_load_n(reinterpret_cast<const unsigned long long *>(e->h_block + S2M_BLOCK_DOUBLES), __ATOMIC_ACQUIRE) : 0ull;
    std::snprintf(b, sizeof(b),
                  "state: in %s%s%s; main stream %s, side stream %s; side thread busy=%d (flag %d) gpu_pending=%d ready=%d prepared=%d; "
                  "block word %llu/%llu; hand-back words (seen/expected):%s%s%s%s%s; waits %lld (slow %lld), policy %d, deadline %lld ms",
                  e->where[0] ? e->where : "(no call yet)", e->step[0] ? " / " : "", e->step, qs(q0), e->pf.stream ? qs(q1) : "none",
                  (int)e->pf.busy, e->pf.busy_a.load(), (int)e->pf.gpu_pending, (int)e->pf.ready, (int)e->pf.prepared, flag, e->seq, m0, m1, m2, m3, m4,
                  (long long)e->wait.n_waits.load(), (long long)e->wait.n_slow.load(), e->wait.policy, (long long)(e->wait.timeout_us / 1000));
    std::string out = b;
    std::snprintf(b, sizeof(b), "; layout beside the frames: state %d, %lld begun, %lld swapped in, %lld dropped, %lld failed, last reason \"%s\", density %.1f"
                  "; map code allocations (all handles): %lld, %.1f MB",
                  e->relay.state.load(), (long long)e->relay.n_started, (long long)e->n_beside, (long long)e->relay.n_dropped, (long long)e->relay.n_failed,
                  e->relay.why.load(), e->relay.density, (long long)map_allocations(), (double)map_allocated_bytes() / 1048576.0);
    return out + b;
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
    if (c->wait_policy < 0 || c->wait_policy > 2 || c->wait_timeout_ms < 0 || c->wait_spin_us < 0) return S2M_ERR_ARG;
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


}  // namespace s2m_eng

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
    c->wait_policy = 0;
    c->wait_timeout_ms = 10000;
    c->wait_spin_us = 40;
    c->layout_beside = 1;
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
        case S2M_ERR_TIMEOUT: return "the device did not answer in time";
        default: return "unknown error";
    }
}

namespace {
// s2m_config -> the handle's wait control.  S2M_WAIT_POLICY / S2M_WAIT_TIMEOUT_MS override the config (A/B runs, tests);
// a zero deadline or spin time means the default
void apply_wait_config(s2m_engine *e)
{
    int policy = e->cfg.wait_policy;
    int64_t ms = e->cfg.wait_timeout_ms > 0 ? e->cfg.wait_timeout_ms : 10000;
    if (const char *g = std::getenv("S2M_WAIT_POLICY")) {
        const std::string v(g);
        policy = v == "spin" ? 0 : v == "yield" ? 1 : (v == "sleep" || v == "block") ? 2 : std::max(0, std::min(2, std::atoi(g)));
    }
    if (const char *g = std::getenv("S2M_WAIT_TIMEOUT_MS")) ms = std::max<int64_t>(1, std::atoll(g));
    e->wait.policy = policy;
    e->wait.timeout_us = ms * 1000;
    e->wait.spin_us = e->cfg.wait_spin_us > 0 ? e->cfg.wait_spin_us : 40;
}
}  // namespace

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
    apply_wait_config(e);
    if (const char *g = std::getenv("S2M_TEST_STALL")) {  // fault injection (tests): worker | mail | reduce [: hand-backs before the stall]
        const std::string v(g);
        const size_t colon = v.find(':');
        const std::string kind = v.substr(0, colon);
        e->wait.stall_kind = kind == "worker" ? kStallWorker : kind == "mail" ? kStallMail : kind == "reduce" ? kStallReduce : kStallNone;
        e->wait.stall_after.store(colon == std::string::npos ? 0 : std::atol(v.c_str() + colon + 1));
    }
    e->wait.hosttime = std::getenv("S2M_HOSTTIME") != nullptr;
    e->relay.enabled = std::getenv("S2M_NO_BESIDE") == nullptr;
    if (const char *g = std::getenv("S2M_BESIDE_AT")) {   // test hook: a layout beside the frames behind the n-th update [",regrid": with a new cell size]
        e->relay.force_at = std::atol(g);
        e->relay.force_regrid = std::strstr(g, "regrid") != nullptr;
    }
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
    e->exact_stage = std::getenv("S2M_EXACT_STAGE") != nullptr;
    e->map.no_fused_prep = std::getenv("S2M_NO_FUSED_PREP") != nullptr;
    e->upd.fuse_stage = !e->map.no_fused_prep && std::getenv("S2M_NO_FUSED_STAGE") == nullptr;
    e->vox.no_hint = std::getenv("S2M_NO_VOXEL_HINT") != nullptr;
    e->und.always_sort = std::getenv("S2M_NO_TIME_SHORTCUT") != nullptr;
    if (const char *g = std::getenv("S2M_FIRST_GAIN")) e->first_round_gain = std::max(1.5f, std::min(256.0f, (float)std::atof(g)));
    if (const char *g = std::getenv("S2M_BLIND_ROUNDS")) e->blind_rounds = std::max(0, std::min(8, std::atoi(g)));  // (A/B runs)
    bool ok = hipSetDevice(dev) == hipSuccess;
    if (ok) {
        int least = 0, greatest = 0;
        const char *g = std::getenv("S2M_MAIN_PRIO");   // (A/B: the handle's own stream at the highest priority)
        if (g && std::strcmp(g, "high") == 0 && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess)
            ok = hipStreamCreateWithPriority(&e->own_stream, hipStreamNonBlocking, greatest) == hipSuccess;
        else
            ok = hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking) == hipSuccess;
    }
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
    tl_wait = &e->wait;
    // Nothing here may block for ever either: the streams are given the handle's deadline to drain and the side thread to
    // leave.  If either does not, the device (or the runtime under that thread) is not answering: the handle's memory is left
    // where it is -- freeing it would wait on the same device -- and the caller is told.
    bool drained = !e->stream || wait_stream(&e->wait, e->stream, "the main stream (s2m_destroy)") != kWaitTimedOut;
    if (e->pf.worker.joinable()) {
        { std::lock_guard<std::mutex> lk(e->pf.mu); e->pf.quit = true; }
        e->pf.quit_a.store(1, std::memory_order_release);
        e->pf.cv.notify_all();
        const bool left = wait_until(&e->wait, [&] { return e->pf.exited.load(std::memory_order_acquire) != 0; }, "the side thread to leave (s2m_destroy)");
        if (left) e->pf.worker.join();
        else { e->pf.worker.detach(); drained = false; }
    }
    if (drained && e->pf.stream) drained = wait_stream(&e->wait, e->pf.stream, "the side stream (s2m_destroy)") != kWaitTimedOut;
    if (drained) {
        relay_shutdown(e);
        drained = e->relay.exited.load() != 0 || !e->relay.worker.joinable();
    }
    tl_wait = nullptr;
    if (!drained) return S2M_ERR_TIMEOUT;  // (the handle is leaked on purpose)
    if (e->pf.stream) (void)hipStreamDestroy(e->pf.stream);
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

int s2m_test_stall(s2m_engine *e, int32_t kind, int64_t after)
{
    if (!e || kind < 0 || kind > 4 || after < 0) return S2M_ERR_ARG;
    if (kind == 4) {   // the hand-backs of the layout worker (its own wait control: it has its own way of waiting)
        e->relay.wait.stall_kind = kStallNone;
        e->relay.wait.stall_after.store((long)after);
        e->relay.wait.stall_kind = kStallMail;
        return S2M_OK;
    }
    e->relay.wait.stall_kind = kStallNone;
    e->wait.stall_kind = kStallNone;
    e->wait.stall_after.store((long)after);
    e->wait.stall_kind = kind;
    return S2M_OK;
}

int s2m_debug_state(const s2m_engine *e, char *buf, int64_t capacity)
{
    if (!e || !buf || capacity < 1) return S2M_ERR_ARG;
    const std::string s = debug_state(e);
    std::snprintf(buf, (size_t)capacity, "%s", s.c_str());
    return S2M_OK;
}

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
    apply_wait_config(e);
    if (e->cfg.far_point_bet >= 0 && e->cfg.far_point_bet <= 2 && !e->spec_env) e->spec_mode = e->cfg.far_point_bet;
    return S2M_OK;
}

int s2m_set_stream(s2m_engine *e, void *hip_stream)
{
    if (!e) return S2M_ERR_ARG;
    S2M_ENTER(e);
    int rc = sync_stream(e, e->stream, "the stream that is being replaced");
    if (rc) return rc;
    e->stream = hip_stream ? (hipStream_t)hip_stream : e->own_stream;
    return S2M_OK;
}


int s2m_comm_unique_id(uint8_t id[S2M_COMM_ID_BYTES])
{
    if (!id) return S2M_ERR_ARG;
    std::string err;
    return comm_unique_id(id, err) ? S2M_OK : S2M_ERR_HIP;
}

int s2m_comm_init(s2m_engine *e, const uint8_t id[S2M_COMM_ID_BYTES], int32_t nranks, int32_t rank)
{
    if (!e || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(e, S2M_ERR_ARG, "s2m_comm_init: bad argument");
    S2M_ENTER(e);
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


}  // extern "C"
