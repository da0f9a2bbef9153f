// s2m_engine_relay.cpp -- a new layout of the whole map produced BESIDE the frames.
//
// An in-place update (s2m_mapedit.hip) costs what the scan changes, but the layout it works on wears out: the tail of the point
// array fills with the bricks that opened or outgrew their place, the spare rows of the brick table run out, and the cell size
// that was right for the seed is wrong for what the voxel rule leaves behind.  Until round 5 the cure ran inside an update -- a
// merge (~2 ms at 5 M points) or a re-grid (another ~2 ms) in the frame that hit the limit.  ikd-Tree moves every large rebuild
// to a second thread (ikd_Tree.cpp:192-203, 229-367: flatten, build beside the node's loop, apply the operations logged
// meanwhile, swap the subtree in); this is the same for the whole map:
//   trigger   at the end of an update, EARLY: tail or spare rows three quarters used, or the points per occupied cell (counted
//             every 64 updates by one kernel over the brick tables) a factor two off what the cell size was chosen for
//   snapshot  three small launches over the live map on the main stream, behind an update: every point with its id, in position
//             order (inside a cell that is id order: the new layout keeps the documented order of ties)
//   build     the worker thread of this file, on its own stream: a complete build from the snapshot (a new cell size when the
//             density asked for it), the ids put back
//   replay    the update CALLS that arrive meanwhile (map_incremental's two lists, map_add's list, delete_boxes' boxes) are kept
//             (a device ring) and run again on the new map by the worker.  The voxel rule is a function of the point set, ties
//             go by id or batch order (s2m_mapupd.hip): the same calls leave the same set with the same ids
//   swap      at the start of the next update once the worker has nothing left: the two maps change places (pointers), the
//             main stream waits for the layout stream's last event (long past).  No frame waits for a kernel of the layout.
// A follower of the map (s2m_map_get_changes) notices nothing: same points, same ids.  Neighbour indices of earlier passes are
// invalid after any update anyway.  If the live map has to re-lay itself after all (the limit came before the worker was done),
// or is rebuilt, the layout in flight is dropped.
#include "s2m_engine_internal.h"

using namespace s2m;
using namespace s2m_eng;

#define S2M_TRY(x)                       \
    do {                                 \
        hipError_t e_ = (x);             \
        if (e_ != hipSuccess) return e_; \
    } while (0)

namespace {
using Relay = s2m_engine::Relay;

int64_t spare_rows(const MapBuffers &b)
{
    return std::min(std::min(b.tab_cap / kBrickStride, b.bmove_cap), std::min(std::min(b.bkey_cap, b.bmark_cap), b.bend_cap));
}

MapSide other_side(s2m_engine *e)
{
    Relay &r = e->relay;
    return MapSide{&r.map, &r.upd, &r.grid, &r.stats, &r.built_cell, r.stream, false};
}

// ---- the worker ---------------------------------------------------------------------------------------------------------
hipError_t worker_prepare(s2m_engine *e)
{
    Relay &r = e->relay;
    if (!r.stream) {
        // The layout's kernels (a radix sort of millions of keys) should not take the chip from a frame's: the layout stream has
        // the lowest priority the device offers.  Measured and not kept as the default: a stream that owns only 32 or 64 of the 256
        // compute units (hipExtStreamCreateWithCUMask, S2M_BESIDE_CUS=n) -- the frames beside the build were no faster (0.59 -
        // 0.66 ms either way: they wait for the snapshot and for the update calls' copies, not for compute units) and the build,
        // eight times longer, slowed two hundred frames by 10 % instead of sixty (NOTEBOOK round 6).
        int cus = 0, want = 0;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, e->device) == hipSuccess) cus = prop.multiProcessorCount;
        if (const char *g = std::getenv("S2M_BESIDE_CUS")) want = std::atoi(g);
        if (want > 0 && want < cus) {
            std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0u);
            for (int i = 0; i < want; ++i) mask[(size_t)i / 32] |= 1u << (i % 32);
            if (hipExtStreamCreateWithCUMask(&r.stream, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
                (void)hipGetLastError();
                r.stream = nullptr;
            }
        }
        if (!r.stream) {
            int least = 0, greatest = 0;
            if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = greatest = 0;
            if (const char *g = std::getenv("S2M_BESIDE_PRIO")) least = std::strcmp(g, "normal") == 0 ? 0 : std::strcmp(g, "high") == 0 ? greatest : least;   // (A/B)
            if (hipStreamCreateWithPriority(&r.stream, hipStreamNonBlocking, least) != hipSuccess) {
                (void)hipGetLastError();
                S2M_TRY(hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking));
            }
        }
    }
    if (!r.ev_side) S2M_TRY(hipEventCreateWithFlags(&r.ev_side, hipEventDisableTiming));
    if (!r.ev_main) S2M_TRY(hipEventCreateWithFlags(&r.ev_main, hipEventDisableTiming));
    if (!r.ev_snap) S2M_TRY(hipEventCreateWithFlags(&r.ev_snap, hipEventDisableTiming));
    if (!r.snap_count) S2M_TRY(hipMalloc((void **)&r.snap_count, sizeof(uint32_t)));
    if (r.snap_cap < r.snap_bound) {
        if (r.snap) S2M_TRY(hipFree(r.snap));
        r.snap = nullptr;
        r.snap_cap = 0;
        const int64_t want = r.snap_bound + r.snap_bound / 4 + 65536;
        S2M_TRY(hipMalloc((void **)&r.snap, (size_t)want * sizeof(float4)));
        if (r.snap2) S2M_TRY(hipFree(r.snap2));
        r.snap2 = nullptr;
        S2M_TRY(hipMalloc((void **)&r.snap2, (size_t)want * sizeof(float4)));
        if (r.snap_work) S2M_TRY(hipFree(r.snap_work));
        r.snap_work = nullptr;
        S2M_TRY(hipMalloc((void **)&r.snap_work, (size_t)want * 4 * sizeof(uint32_t)));
        if (r.snap_tmp) S2M_TRY(hipFree(r.snap_tmp));
        r.snap_tmp = nullptr;
        r.snap_tmp_bytes = snapshot_sort_tmp_bytes(want);
        S2M_TRY(hipMalloc(&r.snap_tmp, std::max<size_t>(r.snap_tmp_bytes, 256)));
        r.snap_cap = want;
    }
    if (!r.d_cells) {   // the density probe's words (a device word, a pinned pair): here, not beside a frame (a pinned allocation is milliseconds)
        S2M_TRY(hipMalloc((void **)&r.d_cells, sizeof(uint32_t)));
        S2M_TRY(hipMemset(r.d_cells, 0, sizeof(uint32_t)));
        S2M_TRY(hipHostMalloc((void **)&r.h_cells, 16 * sizeof(uint32_t), hipHostMallocMapped));
        S2M_TRY(hipHostGetDevicePointer((void **)&r.h_cells_dev, r.h_cells, 0));
        r.h_cells[0] = 0u;
    }
    const int64_t blocks = snapshot_blocks(r.extent_bound) + 1;
    if (r.snap_blk_cap < blocks) {
        if (r.snap_blk) S2M_TRY(hipFree(r.snap_blk));
        r.snap_blk = nullptr;
        r.snap_blk_cap = 0;
        S2M_TRY(hipMalloc((void **)&r.snap_blk, (size_t)(blocks + blocks / 4) * sizeof(uint32_t)));
        r.snap_blk_cap = blocks + blocks / 4;
    }
    // room for the lists of a few hundred updates (two lists of at most a scan each); a queue that outgrows it gives the layout up
    const int64_t want = std::max<int64_t>((int64_t)1 << 22, 64 * std::max<int64_t>(e->n_cap, 4096));
    if (r.arena_cap < want) {
        if (r.arena) S2M_TRY(hipFree(r.arena));
        r.arena = nullptr;
        r.arena_cap = 0;
        S2M_TRY(hipMalloc((void **)&r.arena, (size_t)want * sizeof(float4)));
        r.arena_cap = want;
    }
    return hipSuccess;
}

// a complete build from the snapshot, on the layout stream; the points get the ids they had
int worker_build(s2m_engine *e, const float origin[3], float cell_live)
{
    Relay &r = e->relay;
    if (hipStreamWaitEvent(r.stream, r.ev_main, 0) != hipSuccess) return S2M_ERR_HIP;
    uint32_t n = 0;
    const uint32_t *src[1] = {r.snap_count};
    hipError_t he = mail_fetch(r.map.mail, src, 1, &n, r.stream);
    if (he != hipSuccess) return he == kWaitTimedOut ? S2M_ERR_TIMEOUT : S2M_ERR_HIP;
    if ((int64_t)n > r.snap_cap || n == 0) return S2M_ERR_CAPACITY;
    const float cell = r.regrid ? 0.0f : cell_live;
    bool too_large = false;
    // in ascending id order: whatever cells the new grid has, the points of a cell then stand in the documented order
    he = snapshot_sort_by_id(r.snap, (int64_t)n, r.snap2, r.snap_work, r.snap_tmp, r.snap_tmp_bytes, r.stream);
    if (he != hipSuccess) return S2M_ERR_HIP;
    std::swap(r.snap, r.snap2);
    he = build_map(reinterpret_cast<const float *>(r.snap), 4, (int64_t)n, cell, r.map, r.grid, r.stats, too_large, r.stream,
                   (!r.regrid && cell > 0.0f) ? origin : nullptr);
    if (he != hipSuccess) return he == kWaitTimedOut ? S2M_ERR_TIMEOUT : S2M_ERR_HIP;
    if (too_large) return S2M_ERR_CAPACITY;
    // a layout asked for by the tail or the table rows, on a map whose density has left what its cell size was chosen for (the count of
    // this very build says so): the cell size is chosen again here, not by a rebuild inside the next update
    if (!r.regrid && e->cfg.cell_size <= 0.0f && r.stats.occupied_cells > 0) {
        const double mean = (double)n / (double)r.stats.occupied_cells;
        if (mean < 5.5 || mean > 22.0) {
            r.regrid = true;
            he = build_map(reinterpret_cast<const float *>(r.snap), 4, (int64_t)n, 0.0f, r.map, r.grid, r.stats, too_large, r.stream, nullptr);
            if (he != hipSuccess) return he == kWaitTimedOut ? S2M_ERR_TIMEOUT : S2M_ERR_HIP;
            if (too_large) return S2M_ERR_CAPACITY;
        }
    }
    launch_remap_ids(r.map.pidx, r.grid.m, r.snap, r.stream);
    r.map.next_id = r.id_snap;
    r.map.ids_dense = false;
    r.built_cell = r.grid.c;
    return hipEventRecord(r.ev_side, r.stream) == hipSuccess ? S2M_OK : S2M_ERR_HIP;
}

// one update call of the live map, again, on the other one
int worker_replay(s2m_engine *e, const Relay::Op &op)
{
    Relay &r = e->relay;
    const MapSide s = other_side(e);
    if (hipStreamWaitEvent(r.stream, r.ev_main, 0) != hipSuccess) return S2M_ERR_HIP;   // (the lists were copied on the main stream)
    bind_update(e, s);
    hipError_t he = update_begin(r.upd, r.grid, r.stream);
    int rc = S2M_OK;
    if (he == hipSuccess && op.kind == 0) {
        he = update_add(r.upd, r.grid, r.arena + op.off_a, op.na, op.ds_a, op.fs, nullptr, r.stream, op.has_vox ? &op.vox : nullptr, true);
        if (he == hipSuccess && op.nb > 0) he = update_add(r.upd, r.grid, r.arena + op.off_b, op.nb, false, 0.0f, nullptr, r.stream, nullptr, true);
        if (he == hipSuccess) rc = commit_update(e, s, nullptr, 0);
    } else if (he == hipSuccess) {
        int64_t del = 0;
        he = update_delete(r.upd, r.grid, op.boxes.data(), (int)(op.boxes.size() / 6), &del, r.stream);
        if (he == hipSuccess && del > 0) rc = commit_update(e, s, op.boxes.data(), (int)(op.boxes.size() / 6));
    }
    if (he != hipSuccess) return he == kWaitTimedOut ? S2M_ERR_TIMEOUT : S2M_ERR_HIP;
    if (rc) return rc;
    return hipEventRecord(r.ev_side, r.stream) == hipSuccess ? S2M_OK : S2M_ERR_HIP;
}

void relay_worker(s2m_engine *e)
{
    Relay &r = e->relay;
    (void)hipSetDevice(e->device);
    // the worker's own way of waiting: a short spin, then it yields the core between polls -- the build's kernels take hundreds of
    // microseconds, nobody waits for them, and the frames' thread is the one that should have a core.  (Naps are too coarse: a
    // replayed update is four hand-backs, and with a nap behind each the worker falls behind the frames and never catches up --
    // measured, NOTEBOOK round 6.)
    r.wait.policy = kWaitYield;
    r.wait.spin_us = 5;
    r.wait.timeout_us = e->wait.timeout_us;
    if (const char *g = std::getenv("S2M_BESIDE_WAIT")) {   // (A/B)
        if (std::strcmp(g, "spin") == 0) { r.wait.policy = e->wait.policy; r.wait.spin_us = e->wait.spin_us; }
        else if (std::strcmp(g, "sleep") == 0) r.wait.policy = kWaitSleep;
    }
    tl_wait = &r.wait;
    struct Exited {
        std::atomic<int> &f;
        ~Exited() { f.store(1, std::memory_order_release); }
    } exited{r.exited};
    bool built = false;
    std::unique_lock<std::mutex> lk(r.mu);
    for (;;) {
        r.cv.wait(lk, [&] {   // (idle: no deadline -- the thread waits to be given work)
            const int st = r.state.load();
            return r.quit.load() != 0 || r.cancel.load() != 0 || st == Relay::kStarting || (st == Relay::kBuilding && (!built || !r.ops.empty()));
        });
        if (r.quit.load() != 0) return;
        if (r.cancel.load() != 0) {
            r.ops.clear();
            r.arena_head = r.arena_tail = 0;
            built = false;
            r.state.store(Relay::kIdle);
            r.cancel.store(0);
            r.cv.notify_all();
            continue;
        }
        const int st = r.state.load();
        r.busy.store(1);
        if (st == Relay::kStarting) {
            lk.unlock();
            const hipError_t he = worker_prepare(e);
            lk.lock();
            built = false;
            r.state.store(he == hipSuccess ? Relay::kSnapReady : Relay::kFailed);
        } else if (!built) {
            const float origin[3] = {r.grid.ox, r.grid.oy, r.grid.oz};   // (left here by the snapshot: the live map's origin and cell)
            const float cell_live = r.grid.c;
            lk.unlock();
            const int rc = worker_build(e, origin, cell_live);
            lk.lock();
            built = rc == S2M_OK;
            if (rc) { r.timed_out.store(rc == S2M_ERR_TIMEOUT); r.why.store(rc == S2M_ERR_TIMEOUT ? "a wait of the layout worker expired" : "the build beside the frames failed"); r.state.store(Relay::kFailed); }
            else if (r.ops.empty() && r.cancel.load() == 0) r.state.store(Relay::kCaughtUp);
        } else {
            Relay::Op op = std::move(r.ops.front());
            r.ops.pop_front();
            lk.unlock();
            const int rc = worker_replay(e, op);
            lk.lock();
            r.arena_tail = op.arena_end;
            if (rc) { r.timed_out.store(rc == S2M_ERR_TIMEOUT); r.why.store(rc == S2M_ERR_TIMEOUT ? "a wait of the layout worker expired" : "an update could not be applied to the map beside the frames"); r.state.store(Relay::kFailed); }
            else if (r.ops.empty() && r.cancel.load() == 0) r.state.store(Relay::kCaughtUp);
        }
        r.busy.store(0);
        r.cv.notify_all();
    }
}

// a contiguous stretch of the ring for `need` points; -1: no room
int64_t arena_take(Relay &r, int64_t need)
{
    // (under r.mu) head: next free, tail: oldest still in use; the stretch must not wrap
    if (need > r.arena_cap) return -1;
    int64_t at = r.arena_head;
    const bool empty = r.ops.empty() && r.busy.load() == 0;
    if (empty) { r.arena_head = r.arena_tail = 0; at = 0; }
    if (r.arena_head >= r.arena_tail) {            // free: [head, cap) and [0, tail)
        if (at + need <= r.arena_cap) { r.arena_head = at + need; return at; }
        if (need < r.arena_tail) { r.arena_head = need; return 0; }
        return -1;
    }
    if (at + need < r.arena_tail) { r.arena_head = at + need; return at; }   // free: [head, tail)
    return -1;
}

// the layout in flight is given up without waiting: the worker lets go at its next step
void relay_drop(s2m_engine *e)
{
    Relay &r = e->relay;
    std::lock_guard<std::mutex> lk(r.mu);
    if (r.state.load() == Relay::kIdle) return;
    ++r.n_dropped;
    r.min_gap = std::min<int64_t>(2 * r.min_gap, 4096);   // (a layout that keeps being overtaken is not begun again at once)
    r.since_layout = 0;
    r.cancel.store(1);
    r.cv.notify_all();
}
}  // namespace

namespace s2m_eng {

int relay_after_commit(s2m_engine *e, bool inplace, bool kept_grid)
{
    Relay &r = e->relay;
    if (!r.enabled || e->no_merge || e->no_slab || e->cfg.layout_beside == 0) return S2M_OK;
    ++r.commits;
    ++r.since_layout;
    // the third update after a build (a warm-up frame: the live map has allocated what a scan's batches need by now): the other
    // map's update buffers get the same capacities, here and not beside the frame of its first real update
    if (r.commits == 3 && r.state.load() == Relay::kIdle && r.map.pts && r.stream) {
        S2M_HIP(e, update_reserve_like(r.upd, e->upd, e->stream));
        // (and the ring of recorded update calls is sized by the scan's capacity, which did not exist at the build: the worker
        // would free and allocate it where the first layout starts -- a hipFree waits for the whole device, 0.85 ms in the trace)
        S2M_HIP(e, worker_prepare(e));
        int rc = sync_stream(e, e->stream, "the other map's update buffers");
        if (rc) return rc;
        r.allocs_seen = map_allocations();
    }
    const int st = r.state.load();
    if (st == Relay::kFailed) return S2M_OK;   // (relay_poll deals with it)
    if (st != Relay::kIdle && (!inplace || !kept_grid)) {   // the live map has just laid itself out again (or was rebuilt: new ids)
        relay_drop(e);
        return S2M_OK;
    }
    if (st == Relay::kSnapReady && r.cancel.load() == 0) {
        // the snapshot, behind this update: every live point with its id; the worker builds from it
        if (e->grid.live > r.snap_cap || snapshot_blocks(e->grid.m) > r.snap_blk_cap) { relay_drop(e); return S2M_OK; }
        // ... on the layout stream, behind this update (event) -- the registration of the next scan only READS the live map and runs
        // beside it; the next update of the live map, the first thing that writes it, waits for the snapshot's event (relay_fence:
        // a quarter of a millisecond later, long done)
        if (!r.ev_main) S2M_HIP(e, hipEventCreateWithFlags(&r.ev_main, hipEventDisableTiming));
        if (!r.ev_snap) S2M_HIP(e, hipEventCreateWithFlags(&r.ev_snap, hipEventDisableTiming));
        S2M_HIP(e, hipEventRecord(r.ev_main, e->stream));
        S2M_HIP(e, hipStreamWaitEvent(r.stream, r.ev_main, 0));
        launch_snapshot(e->grid.pts, e->grid.pidx, e->grid.m, r.snap, r.snap_count, r.snap_cap, r.snap_blk, r.stream);
        S2M_HIP(e, hipEventRecord(r.ev_snap, r.stream));
        r.snap_fence = true;
        std::lock_guard<std::mutex> lk(r.mu);
        r.id_snap = e->map.next_id;
        r.grid.ox = e->grid.ox; r.grid.oy = e->grid.oy; r.grid.oz = e->grid.oz;   // (what the build keeps unless it re-grids)
        r.grid.c = e->cfg.cell_size > 0.0f ? e->cfg.cell_size : e->built_cell;
        r.state.store(Relay::kBuilding);
        r.cv.notify_all();
        return S2M_OK;
    }
    if (st != Relay::kIdle) return S2M_OK;
    // ---- idle.  Has the live map (or its update buffers) grown in this call?  Then the frame has stalled for its allocations
    // anyway, and the other map's buffers follow NOW: a map that has outgrown them would otherwise have them allocated by the
    // worker beside the frames of its next layout -- a dozen allocations, 3 - 6 ms of stalled frames (NOTEBOOK round 6).
    if (!r.map.pts && !r.late_tried && e->grid.live >= 4096) {
        // a map that was BUILT too small for a rehearsal (a node's first scan, one point of a test) and has grown by updates: the
        // rehearsal now, from the live map's own points
        r.late_tried = true;
        r.snap_bound = 2 * e->grid.live + 65536;
        r.extent_bound = std::max<int64_t>(2 * e->grid.m, e->map.pts_cap) + 65536;
        S2M_HIP(e, worker_prepare(e));
        launch_snapshot(e->grid.pts, e->grid.pidx, e->grid.m, r.snap, r.snap_count, r.snap_cap, r.snap_blk, e->stream);
        uint32_t n = 0;
        S2M_HIP(e, hipMemcpyAsync(&n, r.snap_count, sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
        int rc = sync_stream(e, e->stream, "the live map's points for the rehearsal");
        if (rc) return rc;
        const int64_t commits = r.commits, since = r.since_layout;
        rc = relay_rehearse(e, reinterpret_cast<const float *>(r.snap), 4, (int64_t)n);
        r.late_tried = true;
        r.commits = commits;
        r.since_layout = since;
        if (rc) return rc;
    }
    if (map_allocations() != r.allocs_seen && r.stream && r.map.pts) {
        // (only for a map that has GROWN by a quarter beyond what the other map was last sized for: after a layout the two maps
        // have changed places and each holds arrays the other never needed -- matching those would be a dozen allocations in a
        // frame for nothing)
        if (e->grid.live > r.sized_live + r.sized_live / 4) {
            S2M_HIP(e, map_reserve_like(r.map, e->map, 2 * e->grid.live + 65536));
            S2M_HIP(e, update_reserve_like(r.upd, e->upd, e->stream));
            if (e->grid.live + e->grid.live / 4 + 65536 > r.snap_cap || snapshot_blocks(e->grid.m + e->grid.m / 4) + 1 > r.snap_blk_cap) {
                r.snap_bound = 2 * e->grid.live + 65536;
                r.extent_bound = std::max<int64_t>(2 * e->grid.m, e->map.pts_cap) + 65536;
                S2M_HIP(e, worker_prepare(e));
            }
            int rc = sync_stream(e, e->stream, "the other map's buffers following the live map's");
            if (rc) return rc;
            r.sized_live = e->grid.live;
        }
        r.allocs_seen = map_allocations();
    }
    // ---- how worn is the layout?
    if (r.cells_posted && r.h_cells && __atomic_load_n(r.h_cells, __ATOMIC_ACQUIRE) == r.cells_seq) {   // the count posted a while ago has landed
        const uint32_t cells = r.h_cells[1];
        r.density = cells > 0 ? (double)r.cells_live / (double)cells : 0.0;
        r.cells_posted = false;
    }
    if (!r.cells_posted && (r.commits & 63) == 32 && e->stats.bricks > 0 && e->cfg.cell_size <= 0.0f) {
        if (!r.d_cells) {
            S2M_HIP(e, hipMalloc((void **)&r.d_cells, sizeof(uint32_t)));
            S2M_HIP(e, hipMemsetAsync(r.d_cells, 0, sizeof(uint32_t), e->stream));
            S2M_HIP(e, hipHostMalloc((void **)&r.h_cells, 16 * sizeof(uint32_t), hipHostMallocMapped));
            S2M_HIP(e, hipHostGetDevicePointer((void **)&r.h_cells_dev, r.h_cells, 0));
            r.h_cells[0] = 0u;
        }
        if (++r.cells_seq == 0u) r.cells_seq = 1u;
        launch_count_cells(e->map.counters + kBricksWord, e->stats.bricks, e->grid.tab, r.d_cells, r.h_cells_dev, r.cells_seq, e->stream);
        r.cells_live = e->grid.live;
        r.cells_posted = true;
    }
    const char *why = nullptr;
    bool regrid = false;
    if (r.force_at >= 0 && r.commits == r.force_at) { why = "forced (test hook)"; regrid = r.force_regrid; }
    else if (r.since_layout >= r.min_gap) {
        const int64_t tail = e->grid.m - e->map.main_ext, rows = spare_rows(e->map);
        if (e->map.main_ext > 0 && tail > 0 && e->map.tail_used * 4 >= tail * 3) why = "the tail of the point array is three quarters used";
        else if (rows > 0 && e->stats.bricks * 4 >= rows * 3) why = "the brick table's spare rows are three quarters used";
        else if (e->cfg.cell_size <= 0.0f && r.density > 0.0 && (r.density < 5.5 || r.density > 22.0)) { why = "the points per occupied cell have drifted"; regrid = true; }
    }
    if (!why || e->grid.live <= 0) return S2M_OK;
    {
        std::lock_guard<std::mutex> lk(r.mu);
        r.why.store(why);
        r.timed_out.store(0);
        ++r.n_started;
        r.regrid = regrid;
        r.snap_bound = e->grid.live + e->grid.live / 8 + 65536;   // (the snapshot is taken a frame or two from now)
        r.extent_bound = e->grid.m + e->grid.m / 8 + 65536;
        r.density = 0.0;
        r.state.store(Relay::kStarting);
        if (!r.worker.joinable()) r.worker = std::thread(relay_worker, e);
        r.cv.notify_all();
    }
    return S2M_OK;
}

// Behind s2m_map_build: everything a layout beside the frames will need is allocated NOW -- the other map is built once from the
// same cloud and given two dummy updates (the first one of a dense build merges and lays the room and the tail out, the second
// one goes through the in-place path), so that its buffers exist at the sizes the live map's will have.  A device allocation
// stalls every queue of the process for a fraction of a millisecond (the driver remaps them); a dozen of them beside a frame
// were that frame's 6 and 14 ms (round 6, first version).  Costs the build about as much again, once.
int relay_rehearse(s2m_engine *e, const float *cloud_dev, int64_t stride, int64_t m)
{
    Relay &r = e->relay;
    r.late_tried = false;
    if (!r.enabled || e->cfg.layout_beside == 0 || e->no_merge || e->no_slab || m < 4096 || r.state.load() != Relay::kIdle) return S2M_OK;
    static const bool off = std::getenv("S2M_NO_REHEARSAL") != nullptr;   // (A/B runs)
    r.commits = 0;   // (updates are counted from the build)
    r.since_layout = 0;
    if (off) return S2M_OK;
    r.snap_bound = m + m / 8 + 65536;
    r.extent_bound = 4 * m + ((int64_t)1 << 23);
    hipError_t he = worker_prepare(e);
    if (he != hipSuccess) return fail(e, S2M_ERR_HIP, "the buffers of the layout beside the frames", he);
    if (!r.ev_main) S2M_HIP(e, hipEventCreateWithFlags(&r.ev_main, hipEventDisableTiming));
    // (on the main stream: the whole chip, and the cloud is staged there)
    const hipStream_t st = e->stream;
    const float origin[3] = {e->grid.ox, e->grid.oy, e->grid.oz};
    bool too_large = false;
    he = build_map(cloud_dev, stride, m, e->grid.c, r.map, r.grid, r.stats, too_large, st, origin);
    if (he != hipSuccess) return fail(e, S2M_ERR_HIP, "the rehearsal of the layout beside the frames", he);
    if (too_large) return S2M_OK;
    r.built_cell = r.grid.c;
    MapSide s = other_side(e);
    s.st = st;
    for (int round = 0; round < 2; ++round) {
        S2M_HIP(e, hipMemcpy2DAsync(r.arena, sizeof(float4), cloud_dev, (size_t)stride * sizeof(float), 3 * sizeof(float), 1, hipMemcpyDeviceToDevice, st));
        bind_update(e, s);
        he = update_begin(r.upd, r.grid, st);
        if (he == hipSuccess) he = update_add(r.upd, r.grid, r.arena, 1, false, 0.0f, nullptr, st, nullptr, true);
        if (he != hipSuccess) return fail(e, S2M_ERR_HIP, "the rehearsal of the layout beside the frames", he);
        if (commit_update(e, s, nullptr, 0) != S2M_OK) break;   // (not fatal: the first real layout allocates what is missing)
    }
    // (and the code object of s2m_relay.hip is loaded by its first launch -- milliseconds, seen as frame 23 of every drive, where
    // the first density count fell: here instead)
    launch_count_cells(r.map.counters + kBricksWord, r.stats.bricks, r.grid.tab, r.d_cells, r.h_cells_dev, 0u, st);
    // (and a stream's hardware queue is made by its first submission, with the same stall as an allocation: one word on the layout stream)
    S2M_HIP(e, hipMemsetAsync(r.snap_count, 0, sizeof(uint32_t), r.stream));
    int rc = sync_stream(e, r.stream, "the layout stream's first submission");
    if (rc) return rc;
    rc = sync_stream(e, st, "the rehearsal of the layout beside the frames");
    r.allocs_seen = map_allocations();
    r.sized_live = m;
    return rc;
}

int relay_record_lists(s2m_engine *e, const float4 *la, int64_t na, bool ds_a, float fs, const VoxBox *vox, const float4 *lb, int64_t nb)
{
    Relay &r = e->relay;
    const int st = r.state.load();
    if ((st != Relay::kBuilding && st != Relay::kCaughtUp) || r.cancel.load() != 0) return S2M_OK;
    if (na + nb <= 0) return S2M_OK;
    int64_t off = -1;
    {
        std::lock_guard<std::mutex> lk(r.mu);
        off = arena_take(r, na + nb);
    }
    if (off < 0) { relay_drop(e); return S2M_OK; }   // the worker has fallen too far behind: give this layout up
    if (na > 0) S2M_HIP(e, hipMemcpyAsync(r.arena + off, la, (size_t)na * sizeof(float4), hipMemcpyDeviceToDevice, e->stream));
    if (nb > 0) S2M_HIP(e, hipMemcpyAsync(r.arena + off + na, lb, (size_t)nb * sizeof(float4), hipMemcpyDeviceToDevice, e->stream));
    S2M_HIP(e, hipEventRecord(r.ev_main, e->stream));
    Relay::Op op;
    op.kind = 0;
    op.off_a = off; op.na = na; op.off_b = off + na; op.nb = nb;
    op.ds_a = ds_a; op.fs = fs;
    op.has_vox = vox != nullptr;
    if (vox) op.vox = *vox;
    op.arena_end = off + na + nb;
    std::lock_guard<std::mutex> lk(r.mu);
    if (r.cancel.load() != 0) return S2M_OK;
    r.ops.push_back(std::move(op));
    if (r.state.load() == Relay::kCaughtUp) r.state.store(Relay::kBuilding);
    r.cv.notify_all();
    return S2M_OK;
}

int relay_record_boxes(s2m_engine *e, const float *boxes, int nb)
{
    Relay &r = e->relay;
    const int st = r.state.load();
    if ((st != Relay::kBuilding && st != Relay::kCaughtUp) || r.cancel.load() != 0 || nb <= 0) return S2M_OK;
    S2M_HIP(e, hipEventRecord(r.ev_main, e->stream));
    Relay::Op op;
    op.kind = 1;
    op.boxes.assign(boxes, boxes + 6 * (size_t)nb);
    std::lock_guard<std::mutex> lk(r.mu);
    if (r.cancel.load() != 0) return S2M_OK;
    op.arena_end = r.arena_head;
    r.ops.push_back(std::move(op));
    if (r.state.load() == Relay::kCaughtUp) r.state.store(Relay::kBuilding);
    r.cv.notify_all();
    return S2M_OK;
}

// before anything writes the live map: a snapshot that is still being read must be over
int relay_fence(s2m_engine *e)
{
    Relay &r = e->relay;
    if (!r.snap_fence) return S2M_OK;
    r.snap_fence = false;
    S2M_HIP(e, hipStreamWaitEvent(e->stream, r.ev_snap, 0));
    return S2M_OK;
}

int relay_poll(s2m_engine *e)
{
    Relay &r = e->relay;
    const int st = r.state.load();
    if (st == Relay::kIdle || (st != Relay::kCaughtUp && st != Relay::kFailed)) return S2M_OK;
    // Nearest_Points of the current scan are positions in the live map, and map_incremental classifies the scan from them: the
    // maps change places only while no such list is in use -- behind an update, or when a new scan arrives
    if (e->nn_valid) return S2M_OK;
    std::unique_lock<std::mutex> lk(r.mu);
    if (r.state.load() == Relay::kFailed) {
        const bool timed_out = r.timed_out.load() != 0;
        ++r.n_failed;
        r.min_gap = std::min<int64_t>(2 * r.min_gap, 4096);
        r.ops.clear();
        r.arena_head = r.arena_tail = 0;
        r.state.store(Relay::kIdle);
        r.since_layout = 0;   // (not again at once)
        lk.unlock();
        if (timed_out) {
            const char *none = nullptr;
            if (e->wait.expired.compare_exchange_strong(none, r.wait.expired.load())) e->wait.waited_us.store(r.wait.waited_us.load());
            return fail(e, S2M_ERR_HIP, "the layout beside the frames", kWaitTimedOut);
        }
        return S2M_OK;
    }
    if (r.state.load() != Relay::kCaughtUp || !r.ops.empty() || r.busy.load() != 0 || r.cancel.load() != 0) return S2M_OK;
    // ---- the swap: the other map holds the live map's points, ids and all, in a fresh layout
    S2M_HIP(e, hipSetDevice(e->device));
    S2M_HIP(e, hipStreamWaitEvent(e->stream, r.ev_side, 0));
    static const bool check = std::getenv("S2M_BESIDE_CHECK") != nullptr;   // (debug: are the two maps the same set of (id, point)?)
    if (check) {
        auto fetch = [&](const Grid &g, std::vector<float4> &out) -> hipError_t {
            float4 *d = nullptr; uint32_t *cnt = nullptr, *blk = nullptr;
            S2M_TRY(hipMalloc((void **)&d, (size_t)std::max<int64_t>(g.live + 1024, 1) * sizeof(float4)));
            S2M_TRY(hipMalloc((void **)&cnt, sizeof(uint32_t)));
            S2M_TRY(hipMalloc((void **)&blk, (size_t)(snapshot_blocks(g.m) + 1) * sizeof(uint32_t)));
            launch_snapshot(g.pts, g.pidx, g.m, d, cnt, g.live + 1024, blk, e->stream);
            uint32_t n = 0;
            S2M_TRY(hipMemcpyAsync(&n, cnt, sizeof(n), hipMemcpyDeviceToHost, e->stream));
            S2M_TRY(hipStreamSynchronize(e->stream));
            out.resize(std::min<int64_t>(n, g.live + 1024));
            S2M_TRY(hipMemcpy(out.data(), d, out.size() * sizeof(float4), hipMemcpyDeviceToHost));
            (void)hipFree(d); (void)hipFree(cnt); (void)hipFree(blk);
            std::sort(out.begin(), out.end(), [](const float4 &a, const float4 &b) { uint32_t x, y; std::memcpy(&x, &a.w, 4); std::memcpy(&y, &b.w, 4); return x < y; });
            return hipSuccess;
        };
        std::vector<float4> a, b;
        S2M_HIP(e, hipStreamSynchronize(r.stream));
        S2M_HIP(e, fetch(e->grid, a));
        S2M_HIP(e, fetch(r.grid, b));
        size_t diff = 0, shown = 0;
        for (size_t i = 0, j = 0; i < a.size() || j < b.size();) {
            uint32_t x = 0xffffffffu, y = 0xffffffffu;
            if (i < a.size()) std::memcpy(&x, &a[i].w, 4);
            if (j < b.size()) std::memcpy(&y, &b[j].w, 4);
            if (x == y) {
                if (std::memcmp(&a[i], &b[j], 12) != 0) { ++diff; if (shown++ < 6) std::fprintf(stderr, "[beside check] id %u: live (%g %g %g) other (%g %g %g)\n", x, a[i].x, a[i].y, a[i].z, b[j].x, b[j].y, b[j].z); }
                ++i; ++j;
            } else if (x < y) { ++diff; if (shown++ < 6) std::fprintf(stderr, "[beside check] id %u (%g %g %g) only in the live map\n", x, a[i].x, a[i].y, a[i].z); ++i; }
            else { ++diff; if (shown++ < 6) std::fprintf(stderr, "[beside check] id %u (%g %g %g) only in the other map\n", y, b[j].x, b[j].y, b[j].z); ++j; }
        }
        std::fprintf(stderr, "[beside check] swap at update %lld: live %zu points (live count %lld, next id %lld), other %zu (%lld, next id %lld): %zu differences; snapshot at id %lld\n",
                     (long long)r.commits, a.size(), (long long)e->grid.live, (long long)e->map.next_id, b.size(), (long long)r.grid.live, (long long)r.map.next_id, diff,
                     (long long)r.id_snap);
    }
    {
        std::lock_guard<std::mutex> sk(e->stats_mu);
        std::swap(e->map, r.map);
        std::swap(e->upd, r.upd);
        std::swap(e->grid, r.grid);
        std::swap(e->stats, r.stats);
        std::swap(e->built_cell, r.built_cell);
    }
    e->map.no_fused_prep = r.map.no_fused_prep;
    e->upd.fuse_stage = r.upd.fuse_stage;
    e->nn_valid = false;
    r.sized_live = std::max<int64_t>(r.sized_live, e->grid.live);   // (the other map is now the one that held these points)
    ++e->n_beside;
    if (r.regrid) ++e->n_beside_regrid;
    r.since_layout = 0;
    r.min_gap = 32;
    r.cells_posted = false;
    r.density = 0.0;
    r.arena_head = r.arena_tail = 0;
    r.state.store(Relay::kIdle);
    return S2M_OK;
}

int relay_cancel(s2m_engine *e)
{
    Relay &r = e->relay;
    if (r.state.load() == Relay::kIdle) return S2M_OK;
    {
        std::lock_guard<std::mutex> lk(r.mu);
        if (r.state.load() == Relay::kFailed) { r.ops.clear(); r.state.store(Relay::kIdle); return S2M_OK; }
        r.cancel.store(1);
        r.cv.notify_all();
    }
    e->step = "the layout beside the frames to let go";
    if (!wait_until(&e->wait, [&] { return r.state.load() == Relay::kIdle; }, "the worker of the layout beside the frames to let go of it"))
        return fail(e, S2M_ERR_HIP, "relay_cancel", kWaitTimedOut);
    return S2M_OK;
}

void relay_shutdown(s2m_engine *e)
{
    Relay &r = e->relay;
    bool gone = true;
    if (r.worker.joinable()) {
        {
            std::lock_guard<std::mutex> lk(r.mu);
            r.quit.store(1);
            r.cv.notify_all();
        }
        gone = wait_until(&e->wait, [&] { return r.exited.load(std::memory_order_acquire) != 0; }, "the layout worker to leave (s2m_destroy)");
        if (gone) r.worker.join();
        else r.worker.detach();
    }
    if (!gone) return;   // (its memory is left where it is: the thread may still be inside the runtime)
    if (r.stream) { (void)wait_stream(&e->wait, r.stream, "the layout stream (s2m_destroy)"); (void)hipStreamDestroy(r.stream); r.stream = nullptr; }
    if (r.ev_main) (void)hipEventDestroy(r.ev_main);
    if (r.ev_side) (void)hipEventDestroy(r.ev_side);
    if (r.ev_snap) (void)hipEventDestroy(r.ev_snap);
    free_map(r.map);
    free_update(r.upd);
    if (r.snap) (void)hipFree(r.snap);
    if (r.snap_count) (void)hipFree(r.snap_count);
    if (r.snap_blk) (void)hipFree(r.snap_blk);
    if (r.snap2) (void)hipFree(r.snap2);
    if (r.snap_work) (void)hipFree(r.snap_work);
    if (r.snap_tmp) (void)hipFree(r.snap_tmp);
    if (r.arena) (void)hipFree(r.arena);
    if (r.d_cells) (void)hipFree(r.d_cells);
    if (r.h_cells) (void)hipHostFree(r.h_cells);
}

}  // namespace s2m_eng
