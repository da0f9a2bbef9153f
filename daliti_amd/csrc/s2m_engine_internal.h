#pragma once
// s2m_engine_internal.h -- what the translation units of the host engine share: the handle and a few helpers.
//
// The engine is split by what the reference's node does with it:
//   s2m_engine.cpp       handle lifetime, configuration, exchange attachment (s2m_comm_*)
//   s2m_engine_map.cpp   the map: build, share, incremental maintenance, field-of-view trim, getters, change log
//   s2m_engine_scan.cpp  the scan's front half: set / down-sample / undistort, the prefetch and prepare worker
//   s2m_engine_loop.cpp  residual passes and the iterated update in its four forms (single, batch, multi, sharded)
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
#include <deque>
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
// s2m_relay.hip
int64_t snapshot_blocks(int64_t m);
void launch_snapshot(const float4 *pts, const uint32_t *pidx, int64_t m, float4 *out, uint32_t *count, int64_t cap, uint32_t *blk, hipStream_t st);
void launch_remap_ids(uint32_t *pidx, int64_t m, const float4 *snap, hipStream_t st);
size_t snapshot_sort_tmp_bytes(int64_t n);
hipError_t snapshot_sort_by_id(const float4 *snap, int64_t n, float4 *out, uint32_t *work, void *tmp, size_t tmp_bytes, hipStream_t st);
void launch_count_cells(const uint32_t *bricks_dev, int64_t bricks_bound, const uint32_t *tab, uint32_t *cells_dev, uint32_t *host_dev, uint32_t seq,
                        hipStream_t st);
void launch_deinterleave(const float *src, int64_t stride, int64_t n, float *sx, float *sy, float *sz,
                         hipStream_t st);
void launch_scan_reset(int64_t n, uint8_t *sel, uint8_t *eff, uint8_t *flags, hipStream_t st);
}

using namespace s2m;

struct s2m_engine {
    s2m_config cfg{};
    int device = 0;
    // every wait of this handle's caller and of its side thread (s2m_wait.h): policy, deadline, the fault-injection hook
    WaitCtl wait;
    bool poisoned = false;          // a wait expired: the handle's state on the device is unknown; only s2m_destroy is served
    const char *where = "";         // the entry point the caller is inside (or was last), and the step of it that is waiting:
    const char *step = "";          // what s2m_debug_state and the message of an expired wait quote
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
    bool exact_stage = false;       // S2M_EXACT_STAGE=1: map_incremental asks for the staged count after every batch (A/B and tests)
    bool no_slab = false;           // S2M_NO_SLAB=1: no in-place update of the touched bricks, every update merges (A/B and tests)
    int64_t n_inplace = 0;          // updates applied in place (counted among the merged ones too)
    ChangeLog log;                  // what the updates added / removed since the last s2m_map_get_changes
    uint64_t log_seq = 0;
    bool last_update_merged = false;
    int64_t n_beside = 0, n_beside_regrid = 0;   // layouts produced beside the frames (s2m_map_update_stats, stats[10], [11])
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
        std::atomic<int> quit_a{0}, exited{0};  // mirrors for the waits with a deadline (s2m_destroy; a stalled job's sleep)
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

    // ---- the layout beside the frames (s2m_engine_relay.cpp): a second map that a worker thread builds from a snapshot on
    // its own stream and brings up to date by running the update calls that arrived meanwhile again; swapped in between two
    // updates.  What ikd-Tree's rebuild thread does for a subtree (ikd_Tree.cpp:192-203, 229-367).
    struct Relay {
        enum State { kIdle = 0, kStarting, kSnapReady, kBuilding, kCaughtUp, kFailed };
        struct Op {
            int kind = 0;                  // 0: lists (update_add x 1 or 2), 1: boxes (update_delete)
            int64_t off_a = 0, na = 0, off_b = 0, nb = 0;   // the lists' places in the arena
            bool ds_a = false;
            float fs = 0.0f;
            bool has_vox = false;
            VoxBox vox;
            std::vector<float> boxes;
            int64_t arena_end = 0;         // the arena is free up to here once the op is done
        };
        WaitCtl wait;                      // how the worker waits (it yields its core between polls; the handle's deadline)
        bool late_tried = false;           // the rehearsal of a map that was built too small for one has been made (or tried)
        int64_t sized_live = 0;            // the live count the other map's buffers were last sized for
        int64_t allocs_seen = 0;           // map_allocations() when the other map's buffers were last matched to the live map's
        std::thread worker;
        std::mutex mu;
        std::condition_variable cv;
        std::deque<Op> ops;                // under mu
        std::atomic<int> state{kIdle}, busy{0}, quit{0}, exited{0}, cancel{0};
        hipStream_t stream = nullptr;
        hipEvent_t ev_main = nullptr;      // recorded on the main stream behind what the worker is about to read
        hipEvent_t ev_side = nullptr;      // recorded on the layout stream behind what it has written
        hipEvent_t ev_snap = nullptr;      // ... behind the snapshot (which reads the live map: the next update of it waits)
        bool snap_fence = false;
        MapBuffers map;                    // the other map: being built, or the previous live one waiting for the next turn
        UpdateBuffers upd;
        Grid grid{};
        MapStats stats;
        float built_cell = 0.0f;
        float4 *snap = nullptr;            // the snapshot: live points with their ids
        int64_t snap_cap = 0, snap_bound = 0;
        uint32_t *snap_count = nullptr;    // device word
        float4 *snap2 = nullptr;           // ... in ascending id order (what the build reads)
        uint32_t *snap_work = nullptr;     // 4 words per point: the id sort's keys and values
        void *snap_tmp = nullptr;
        size_t snap_tmp_bytes = 0;
        uint32_t *snap_blk = nullptr;      // per block of positions: live points in front of it (scratch of the snapshot)
        int64_t snap_blk_cap = 0, extent_bound = 0;
        int64_t id_snap = 0;               // next_id of the live map at the snapshot
        bool regrid = false;               // the new layout chooses its cell size from the density
        float4 *arena = nullptr;           // the update calls' point lists on their way to the worker (a ring)
        int64_t arena_cap = 0, arena_head = 0, arena_tail = 0;
        // triggers
        int64_t commits = 0, since_layout = 0, force_at = -1;
        int64_t min_gap = 32;              // updates between two layouts: doubled by every layout that failed or was dropped, 32 again after a swap
        int64_t n_started = 0, n_dropped = 0, n_failed = 0;   // layouts begun / given up / failed (diagnostic: s2m_debug_state)
        bool force_regrid = false, enabled = true;
        uint32_t *d_cells = nullptr, *h_cells = nullptr, *h_cells_dev = nullptr;
        uint32_t cells_seq = 0;
        bool cells_posted = false;
        int64_t cells_live = 0;            // live points when the count was posted
        double density = 0.0;              // points per occupied cell, last count (0: not known)
        std::atomic<const char *> why{""};  // what triggered the last layout / why it failed (string literals only: read by s2m_debug_state from any thread)
        std::atomic<int> timed_out{0};     // ... a wait of the worker expired
    } relay;

    EskfWork work;
    Comm comm;  // attached RCCL communicator (multi-GPU form), handle == nullptr when single GPU
    ShmExchange shm;               // or: host shared-memory exchange between the processes of one node (s2m_comm_init_shm)
    std::vector<double> shm_blocks;  // the ranks' blocks of one exchange, padded to a power of two for the tree sum
    // s2m_map_get_changes: the reports that have landed and wait for the follower's arrays (kept across a call that could not
    // take them)
    struct BoxEvent { float box[6]; int64_t after_added, after_removed; };
    std::vector<float4> chg_added, chg_removed;
    std::vector<BoxEvent> chg_boxes;
    bool chg_overflow = false;
    int32_t queue[S2M_FEAT_QUEUE + 1] = {0};
    int32_t queue_len = 0;
};

// ---- helpers shared by the engine's translation units (defined in s2m_engine.cpp) -----------------------------------
namespace s2m_eng {
// he == kWaitTimedOut (a wait of s2m_wait.h expired): the code becomes S2M_ERR_TIMEOUT, the message names the wait and the
// handle's state (s2m_debug_state), and the handle serves nothing but s2m_destroy from then on
int fail(s2m_engine *e, int code, const char *what, hipError_t he = hipSuccess);
int refuse_poisoned(s2m_engine *e);
// everything enqueued on a stream of the handle has finished: hipStreamSynchronize under the handle's policy and deadline
int sync_stream(s2m_engine *e, hipStream_t st, const char *what);
std::string debug_state(const s2m_engine *e);
// one of the handle's two maps with what belongs to it (s2m_engine_map.cpp)
struct MapSide {
    s2m::MapBuffers *map;
    s2m::UpdateBuffers *upd;
    s2m::Grid *grid;
    s2m::MapStats *stats;
    float *built_cell;
    hipStream_t st;
    bool live;
};
MapSide live_side(s2m_engine *e);
void bind_update(s2m_engine *e, MapSide s);
int commit_update(s2m_engine *e, MapSide s, const float *boxes, int nb);
// the layout beside the frames (s2m_engine_relay.cpp)
int relay_after_commit(s2m_engine *e, bool inplace, bool kept_grid);   // end of every update of the live map: triggers, snapshot
int relay_record_lists(s2m_engine *e, const float4 *la, int64_t na, bool ds_a, float fs, const s2m::VoxBox *vox, const float4 *lb, int64_t nb);
int relay_record_boxes(s2m_engine *e, const float *boxes, int nb);
int relay_poll(s2m_engine *e);      // when a new scan arrives / an update begins: swap when the other map has caught up
int relay_fence(s2m_engine *e);     // before an update writes the live map: the snapshot that reads it is over
int relay_rehearse(s2m_engine *e, const float *cloud_dev, int64_t stride, int64_t m);   // behind s2m_map_build: the other map's buffers, now
int relay_cancel(s2m_engine *e);    // the live map is being replaced: whatever is in flight is dropped (waits for the worker to let go)
void relay_shutdown(s2m_engine *e); // s2m_destroy
s2m::Gates gates_of(const s2m_config &c);
s2m::Pose pose_of(const double s[S2M_STATE_DOUBLES]);
int check_config(const s2m_config *c);
// stage a host or device AoS cloud; returns a device pointer usable until the next stage call
int stage_cloud(s2m_engine *e, const float *xyz, int64_t stride, int64_t count, int on_device, const float **dev);
// rank[position] = caller index of every sorted position (s2m_engine_map.cpp)
int caller_index_table(s2m_engine *e, const uint32_t **rank);
}  // namespace s2m_eng

#define S2M_HIP(e, call)                                                          \
    do {                                                                          \
        hipError_t he_ = (call);                                                  \
        if (he_ != hipSuccess) return s2m_eng::fail((e), S2M_ERR_HIP, #call, he_); \
    } while (0)

// On entry to a compute entry point: a handle that has given up refuses, the calling thread's waits take this handle's
// policy and deadline, the device is selected
#define S2M_ENTER(e)                                                    \
    do {                                                                \
        if ((e)->poisoned) return s2m_eng::refuse_poisoned(e);          \
        s2m::tl_wait = &(e)->wait;                                      \
        (e)->where = __func__;                                          \
        (e)->step = "";                                                 \
        S2M_HIP(e, hipSetDevice((e)->device));                          \
    } while (0)
// ... inside a helper that an entry point calls (the entry point's name stays in `where`)
#define S2M_INNER(e)                                                    \
    do {                                                                \
        if ((e)->poisoned) return s2m_eng::refuse_poisoned(e);          \
        s2m::tl_wait = &(e)->wait;                                      \
        S2M_HIP(e, hipSetDevice((e)->device));                          \
    } while (0)

namespace s2m_eng {
template <class T>
int grow(s2m_engine *e, T **p, int64_t count)
{
    if (*p) S2M_HIP(e, hipFree(*p));
    *p = nullptr;
    S2M_HIP(e, hipMalloc((void **)p, (size_t)std::max<int64_t>(count, 1) * sizeof(T)));
    return S2M_OK;
}
}  // namespace s2m_eng
