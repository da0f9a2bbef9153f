// s2m_engine.cpp -- the handle: lifetime, configuration, exchange attachment; and the helpers the other translation
// units of the engine share (s2m_engine_internal.h).  There is no CPU fallback for any compute entry point.
#include "s2m_engine_internal.h"

using namespace s2m;
using namespace s2m_eng;

namespace s2m_eng {

int fail(s2m_engine *e, int code, const char *what, hipError_t he)
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
    e->exact_stage = std::getenv("S2M_EXACT_STAGE") != nullptr;
    e->map.no_fused_prep = std::getenv("S2M_NO_FUSED_PREP") != nullptr;
    e->upd.fuse_stage = !e->map.no_fused_prep && std::getenv("S2M_NO_FUSED_STAGE") == nullptr;
    e->vox.no_hint = std::getenv("S2M_NO_VOXEL_HINT") != nullptr;
    e->und.always_sort = std::getenv("S2M_NO_TIME_SHORTCUT") != nullptr;
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


}  // extern "C"
