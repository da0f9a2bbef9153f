// s2m_engine_map.cpp -- the map behind the handle: build (ikdtree.Build, laserMapping.cpp:784-790), sharing between handles,
// incremental maintenance (map_incremental :582-630, Add_Points / Delete_Point_Boxes ikd_Tree.cpp:477-573, 631-658), the
// field-of-view trim (:313-369), the getters (flatten :1170-1175) and the change log a follower of the map reads.
#include "s2m_engine_internal.h"

using namespace s2m;
using namespace s2m_eng;

namespace {
int complete_lists(s2m_engine *e, int k, int blind, int64_t *n_completed);  // (defined with s2m_complete_neighbors)
}

extern "C" {

int s2m_map_build(s2m_engine *e, const float *xyz, int64_t stride, int64_t m, int on_device)
{
    if (!e || m < 0 || stride < 3 || (m > 0 && !xyz)) return fail(e, S2M_ERR_ARG, "s2m_map_build: bad argument");
    if (m >= ((int64_t)1 << 31)) return fail(e, S2M_ERR_CAPACITY, "map too large (>= 2^31 points)");
    S2M_ENTER(e);
    const float *dev = nullptr;
    int rc = stage_cloud(e, xyz, stride, m, on_device, &dev);
    if (rc) return rc;
    rc = relay_cancel(e);   // (a layout beside the frames belongs to the map that is being replaced)
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
    if (const char *g = std::getenv("S2M_TEST_NEXT_ID")) {   // test hook: the ids of later points begin here (a node reaches 2^32 after a day)
        const long long v = std::atoll(g);
        if (v > (long long)e->map.next_id && v < ((long long)1 << 32)) { e->map.next_id = v; e->map.ids_dense = false; }
    }
    return relay_rehearse(e, dev, stride, m);   // (what a layout beside the frames will need is allocated here, not beside a frame)
}

int s2m_map_share(s2m_engine *e, const s2m_engine *owner)
{
    if (!e || !owner || e == owner) return fail(e, S2M_ERR_ARG, "s2m_map_share: bad argument");
    if (!owner->map_ready) return fail(e, S2M_ERR_STATE, "s2m_map_share: the owner has no map");
    if (owner->device != e->device) return fail(e, S2M_ERR_ARG, "s2m_map_share: handles on different devices");
    S2M_ENTER(e);
    int rc = relay_cancel(e);
    if (rc) return rc;
    rc = sync_stream(e, e->stream, "the borrower's stream (s2m_map_share)");
    if (rc) return rc;
    rc = sync_stream(e, owner->stream, "the owner's stream (s2m_map_share)");  // the owner's build has finished
    if (rc) return rc;
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

}  // extern "C"
namespace s2m_eng {
// what the update kernels need to know about the map's layout (s2m_kernels.h, UpdateBuffers)
void bind_update(s2m_engine *e, MapSide s)
{
    s.upd->bmark = s.map->bmark;
    s.upd->layout_gen = s.map->layout_gen;
    s.upd->reserve_hint = std::max(s.upd->reserve_hint, e->n_cap);
}

MapSide live_side(s2m_engine *e) { return MapSide{&e->map, &e->upd, &e->grid, &e->stats, &e->built_cell, e->stream, true}; }

// the map after an update: merged into the sorted arrays when possible (s2m_mapedit.hip: in place, else merge_update), else rebuilt
// from upd.list (survivors in index order, then the staged points) -- the same caller order either way.
// boxes / nb: the update is a box delete (s2m_map_delete_boxes): a follower of the map gets the boxes, not the points.
// s: the live map -- or the one that is being laid out beside the frames (s2m_engine_relay.cpp), which receives the same update
// calls a little later: no follower, no counters, its own stream.
int commit_update(s2m_engine *e, MapSide s, const float *boxes, int nb)
{
    MapBuffers &map = *s.map;
    UpdateBuffers &upd = *s.upd;
    Grid &grid = *s.grid;
    MapStats &stats = *s.stats;
    const hipStream_t st = s.st;
    bool merged = false, inplace = false;
    // (the map beside the frames belongs to the worker thread: its failures go back as codes, the handle's message and state are
    // the caller's thread's)
    auto bad = [&](int code, const char *what, hipError_t he = hipSuccess) {
        if (s.live) return fail(e, code, what, he);
        return he == kWaitTimedOut ? (int)S2M_ERR_TIMEOUT : code;
    };
    if (s.live) e->map_ready = false;
    hipError_t he;
    if (s.live) {
        std::lock_guard<std::mutex> lk(e->stats_mu);
        he = resolve_stats(map, stats);  // counts of the previous build / merge
    } else {
        he = resolve_stats(map, stats);
    }
    if (he != hipSuccess) return bad(S2M_ERR_HIP, "resolve_stats", he);
    // The cell size is kept across updates (a stable grid) unless the density has drifted by more than 2x from
    // the ~11 points per occupied cell it was chosen for -- e.g. a map seeded from a handful of points and then
    // grown, or a dense seed thinned by the voxel rule: then it is chosen again from the density.  A merged update
    // does not wait for its own counts, so the drift it causes is seen when the next update begins.
    // Judged from the counts of the last build or merge and THAT layout's point count (in-place updates change the number
    // of points and of occupied cells alike, and only a layout counts the cells).
    auto drifted = [&]() {
        if (e->cfg.cell_size > 0.0f || stats.occupied_cells <= 0 || stats.layout_points <= 0) return false;
        const double mean = (double)stats.layout_points / (double)stats.occupied_cells;
        return mean < 5.5 || mean > 22.0;
    };
    const bool drift_before = drifted();
    const int64_t id0 = map.next_id;  // the first id this update hands out
    if (s.live && e->log.on && e->log.token != 0 && grid.m > 0) {   // somebody follows the map
        if (nb == 0) {   // the points about to disappear one by one, before anything moves
            launch_log_removed(e->log, map.counters + kBricksWord, stats.bricks, map.bmark, grid.tab, upd.alive_s, grid.pidx,
                               grid.pts, st);
        } else if (nb <= kLogBoxesPer && (int)e->log.boxes_per_log.size() < kLogMarks) {   // the boxes themselves, in their place in the sequence
            e->log.boxes_log.insert(e->log.boxes_log.end(), boxes, boxes + 6 * (size_t)nb);
            e->log.boxes_per_log.push_back(nb);
            launch_log_mark(e->log, st);
        } else {
            e->log.token = 0;  // more box deletes than a report holds: whoever follows the map fetches it
        }
    }
    if (!e->no_merge && !e->no_slab && !drift_before) {  // in place when every touched brick fits where it stands
        bool counted = false;
        // a scan's batches that update_add left where they were are staged by the in-place update's preparation -- when that
        // can run; otherwise here, by their own kernels
        if (upd.pend.on && !slab_fuses(map, grid, stats, upd.stage_n)) {
            he = update_materialize(upd, st);
            if (he != hipSuccess) return bad(S2M_ERR_HIP, "update_materialize", he);
        }
        he = slab_update(map, grid, stats, upd.alive_s, upd.stage, upd.stage_n, upd.counters + kUpdSlabWord, merged, st,
                         update_stage_word(upd), &counted, &upd.pend, upd.stage, upd.counters + kUpdStageWord + (upd.stage_ops & 1));
        if (he != hipSuccess) return bad(S2M_ERR_HIP, "slab_update", he);
        if (counted) upd.stage_deferred = false;  // (stage_n is the count now)
        inplace = merged;
        if (merged && s.live) ++e->n_inplace;
    }
    // (a count that stayed on the device and did not come back with the in-place update's hand-back)
    he = update_stage_count(upd, st);
    if (he != hipSuccess) return bad(S2M_ERR_HIP, "update_stage_count", he);
    if (!e->no_merge && !drift_before && !merged) {
        he = merge_update(map, grid, stats, upd.alive_s, upd.stage, upd.stage_n, merged, st, !e->no_slab);
        if (he != hipSuccess) return bad(S2M_ERR_HIP, "merge_update", he);
    }
    if (s.live) {
        e->last_update_merged = merged;
        if (merged) ++e->n_merged; else ++e->n_rebuilt;
        if (e->log.on && e->log.token != 0) {
            if (merged) launch_log_added(e->log, upd.stage, upd.stage_n, (uint32_t)id0, st);
            else e->log.token = 0;  // a rebuild numbers the points anew: whoever follows the map starts over
        }
    } else if (!merged) {
        // the map beside the frames must keep the ids the live one hands out: a rebuild would number its points anew.  Give it up;
        // the next trigger starts another one
        return bad(S2M_ERR_STATE, "the layout beside the frames could not take an update without a rebuild");
    }
    if (!merged) {
        int64_t m_new = 0;
        bool too_large = false;
        he = update_finish(upd, grid, &m_new, st);
        if (he != hipSuccess) return bad(S2M_ERR_HIP, "update_finish", he);
        if (m_new >= ((int64_t)1 << 31)) return bad(S2M_ERR_CAPACITY, "map too large (>= 2^31 points)");
        const float cell = e->cfg.cell_size > 0.0f ? e->cfg.cell_size : *s.built_cell;
        // the cells stay where they are (same origin) unless the map was empty or has wandered beyond the representable range
        const float origin[3] = {grid.ox, grid.oy, grid.oz};
        const bool keep = grid.m > 0 && cell == grid.c;
        if (grid.m > 0 && m_new > 0 && map.bbox) {
            // the map in hand is given up by the build: first make sure the build will not refuse the new box (a coordinate of
            // hundreds of kilometres in one scan must not cost a node its map -- ikd-Tree has no such limit, ikd_Tree.cpp:477-573)
            float lo[3], hi[3];
            he = cloud_bbox(reinterpret_cast<const float *>(upd.list), 4, m_new, map.bbox, map.mail, lo, hi, st);
            if (he != hipSuccess) return bad(S2M_ERR_HIP, "cloud_bbox", he);
            if (!(keep && map_build_would_fit(lo, hi, cell, origin)) && !map_build_would_fit(lo, hi, cell, nullptr)) {
                // what the abandoned update left in the update state: its removal marks (the next update starts from the ids again)
                // and the bricks it had flagged
                upd.alive_gen = ~0ull;
                if (map.bmark && map.bmark_cap > 0) S2M_HIP(e, hipMemsetAsync(map.bmark, 0, (size_t)map.bmark_cap, st));
                if (s.live) {
                    e->map_ready = true;                    // the map is as it was: nothing of it has been written
                    if (e->log.on) e->log.token = 0;        // (the removals already logged did not happen: a follower starts over)
                }
                return bad(S2M_ERR_CAPACITY, "a point of this update lies beyond what the map can represent at its cell size: the update "
                                             "was not applied, the map is as it was");
            }
        }
        he = build_map(reinterpret_cast<const float *>(upd.list), 4, m_new, cell, map, grid, stats, too_large,
                       st, keep ? origin : nullptr);
        if (he == hipSuccess && too_large && keep)   // beyond the range of the old origin: a new one
            he = build_map(reinterpret_cast<const float *>(upd.list), 4, m_new, cell, map, grid, stats, too_large, st);
        if (he != hipSuccess) return bad(S2M_ERR_HIP, "build_map", he);
        if (too_large) return bad(S2M_ERR_CAPACITY, "map bounding box too large for the cell size");
        if (drifted()) {
            he = build_map(reinterpret_cast<const float *>(upd.list), 4, m_new, 0.0f, map, grid, stats, too_large,
                           st);
            if (he != hipSuccess) return bad(S2M_ERR_HIP, "build_map", he);
            if (too_large) return bad(S2M_ERR_CAPACITY, "map bounding box too large for the cell size");
            *s.built_cell = grid.c;
            ++e->n_regrid;
        }
    }
    if (*s.built_cell <= 0.0f) *s.built_cell = grid.c;
    if (s.live) {
        e->map_ready = true;
        e->nn_valid = false;  // neighbour indices referred to the old point list
        return relay_after_commit(e, inplace, merged);   // (the layout beside the frames: triggers, and what a re-lay of the live map means for one in flight)
    }
    return S2M_OK;
}
}  // namespace s2m_eng
extern "C" {

int s2m_map_add(s2m_engine *e, const float *xyz, int64_t stride, int64_t n, int downsample_on, float downsample_size,
                int on_device, int64_t *n_added)
{
    if (!e || n < 0 || stride < 3 || (n > 0 && !xyz)) return fail(e, S2M_ERR_ARG, "s2m_map_add: bad argument");
    if (downsample_on && !(downsample_size > 0.0f)) return fail(e, S2M_ERR_ARG, "s2m_map_add: downsample size must be > 0");
    if (!e->map_ready) return fail(e, S2M_ERR_STATE, "no map: call s2m_map_build first");
    if (e->map_borrowed) return fail(e, S2M_ERR_STATE, "the map belongs to another handle (s2m_map_share)");
    S2M_ENTER(e);
    const float *dev = nullptr;
    int rc = stage_cloud(e, xyz, stride, n, on_device, &dev);
    if (rc) return rc;
    rc = relay_poll(e);   // (a layout that was produced beside the frames takes the live map's place here, between two updates)
    if (rc) return rc;
    rc = relay_fence(e);
    if (rc) return rc;
    float4 *np = nullptr;
    S2M_HIP(e, xyz_to_float4(e->upd, dev, stride, n, &np, e->stream));
    rc = relay_record_lists(e, np, n, downsample_on != 0, downsample_size, nullptr, nullptr, 0);
    if (rc) return rc;
    bind_update(e, live_side(e));
    S2M_HIP(e, update_begin(e->upd, e->grid, e->stream));
    int64_t added = 0;
    S2M_HIP(e, update_add(e->upd, e->grid, np, n, downsample_on != 0, downsample_size, &added, e->stream));
    if (n_added) *n_added = added;
    return commit_update(e, live_side(e), nullptr, 0);
}

int s2m_map_delete_boxes(s2m_engine *e, const float *boxes, int64_t n, int64_t *n_deleted)
{
    if (!e || n < 0 || (n > 0 && !boxes) || n > 4096) return fail(e, S2M_ERR_ARG, "s2m_map_delete_boxes: bad argument");
    if (!e->map_ready) return fail(e, S2M_ERR_STATE, "no map: call s2m_map_build first");
    if (e->map_borrowed) return fail(e, S2M_ERR_STATE, "the map belongs to another handle (s2m_map_share)");
    if (n_deleted) *n_deleted = 0;
    if (!delete_touches_map(e->grid, boxes, (int)n)) return S2M_OK;  // (the slab ahead of the sensor after a cube move)
    S2M_ENTER(e);
    int rc = relay_poll(e);
    if (rc) return rc;
    rc = relay_fence(e);
    if (rc) return rc;
    bind_update(e, live_side(e));
    S2M_HIP(e, update_begin(e->upd, e->grid, e->stream));
    int64_t del = 0;
    S2M_HIP(e, update_delete(e->upd, e->grid, boxes, (int)n, &del, e->stream));
    if (n_deleted) *n_deleted = del;
    if (del == 0) return S2M_OK;  // nothing changed: keep the grid and the neighbour indices
    rc = relay_record_boxes(e, boxes, (int)n);
    if (rc) return rc;
    return commit_update(e, live_side(e), boxes, (int)n);
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
    S2M_ENTER(e);
    {
        int rc = relay_poll(e);   // (a layout that was produced beside the frames takes the live map's place here, between two updates)
        if (rc) return rc;
        rc = relay_fence(e);
        if (rc) return rc;
    }
    const Pose pose = pose_of(state);
    float4 *la = nullptr, *lb = nullptr;
    int64_t na = 0, nb = 0;
    // Nearest_Points[i] of the reference is never short (unbounded search): finish the lists that ended at the gate -- when
    // the last rematch pass reported any (block[159]; a scan inside the mapped area has none: no launch, no round trip)
    // ONE launch of the far-point kernel with the radius of the whole map (its rounds grow their band -- and the bricks they
    // look at -- until the nearest neighbour is proven), nobody asked whether anything was open before or is open after: the
    // count of lists it had to leave open comes back with the classification's own hand-back, and only then (a scan point
    // further from every map point than the map is wide) the classic loop of rounds and questions runs and the scan is
    // classified again.
    const bool complete = e->nn_valid && ekf_inited != 0 && e->short_lists != 0 && !e->nn_complete && !e->nn_nearest && e->n > 0 && e->grid.m > 0;
    // how much of such a list must be proven: its nearest entry is all that :603 reads, and the other entries -- beyond the gate
    // radius from the point -- cannot pass :612-616 as long as the voxel's half diagonal stays inside that radius:
    // sqrt(3) fs <= sqrt(gate) (fs <= 1.29 m with the reference's gate; feat.yaml ships 0.5).  A larger leaf gets all five.
    const int need_k = std::sqrt(3.0) * filter_size_map <= std::sqrt((double)e->cfg.knn_d2_gate) ? 1 : kK;
    const uint32_t *open_word = nullptr;
    if (complete) {
        int rc = complete_lists(e, need_k, -1, nullptr);
        if (rc) return rc;
        open_word = e->d_hard + 3 * e->n_cap + 3;
    }
    VoxBox vox;
    bind_update(e, live_side(e));
    uint32_t left_open = 0;
    S2M_HIP(e, incr_classify(e->upd, pose, e->d_scan, e->d_scan + e->n_cap, e->d_scan + 2 * e->n_cap, (int)e->n,
                             e->d_nn_idx, e->grid, e->nn_valid && ekf_inited != 0, filter_size_map, &la, &na, &lb, &nb, e->stream,
                             &vox, true, open_word, &left_open));   // (update_begin runs inside, while the counts travel to the host)
    if (complete && left_open != 0) {
        int rc = complete_lists(e, need_k, 0, nullptr);
        if (rc) return rc;
        S2M_HIP(e, incr_classify(e->upd, pose, e->d_scan, e->d_scan + e->n_cap, e->d_scan + 2 * e->n_cap, (int)e->n,
                                 e->d_nn_idx, e->grid, true, filter_size_map, &la, &na, &lb, &nb, e->stream, &vox, true));
    }
    if (complete) { e->nn_nearest = true; e->nn_complete = need_k == kK; }
    if (n_to_add) *n_to_add = na;
    if (n_no_downsample) *n_no_downsample = nb;
    {
        static const bool hosttime = std::getenv("S2M_HOSTTIME") != nullptr;  // (diagnostic: sizes of the batches)
        if (hosttime) {
            static int64_t calls = 0, sa = 0, sb = 0, ma = 0, mb = 0;
            sa += na; sb += nb; ma = std::max(ma, na); mb = std::max(mb, nb);
            if (++calls % 200 == 0) std::fprintf(stderr, "[batches] %ld scans: to add mean %ld max %ld, no downsample mean %ld max %ld\n", (long)calls, (long)(sa / calls), (long)ma, (long)(sb / calls), (long)mb);
        }
    }
    // (nobody asks for the number of points each call adds: the count stays on the device until the commit's own hand-back)
    const bool defer = !e->exact_stage;
    {   // the same two lists for the map that is being laid out beside the frames, if one is
        int rc = relay_record_lists(e, la, na, true, (float)filter_size_map, &vox, lb, nb);
        if (rc) return rc;
    }
    S2M_HIP(e, update_add(e->upd, e->grid, la, na, true, (float)filter_size_map, nullptr, e->stream, &vox, defer));   // :627
    S2M_HIP(e, update_add(e->upd, e->grid, lb, nb, false, 0.0f, nullptr, e->stream, nullptr, defer));           // :628
    return commit_update(e, live_side(e), nullptr, 0);
}

}  // extern "C"
namespace s2m_eng {
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
}  // namespace s2m_eng
extern "C" {

int s2m_map_get_points(s2m_engine *e, float *xyz, int64_t capacity, int64_t *m)
{
    if (!e || !m) return fail(e, S2M_ERR_ARG, "null argument");
    if (!e->map_ready) return fail(e, S2M_ERR_STATE, "no map");
    *m = e->grid.live;
    if (!xyz) return S2M_OK;
    if (capacity < e->grid.live) return fail(e, S2M_ERR_CAPACITY, "point buffer too small");
    if (e->grid.live == 0) return S2M_OK;
    S2M_ENTER(e);
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
    rc = sync_stream(e, e->stream, "the map in caller order");  // (the copy into pageable memory would wait inside the runtime, without a deadline)
    if (rc) return rc;
    S2M_HIP(e, hipMemcpyAsync(xyz, e->d_stage, (size_t)floats * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    return sync_stream(e, e->stream, "the map's points on their way to the caller");
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


int s2m_map_get_ids(s2m_engine *e, uint32_t *ids, int64_t capacity, int64_t *m)
{
    if (!e || !m) return fail(e, S2M_ERR_ARG, "null argument");
    if (!e->map_ready) return fail(e, S2M_ERR_STATE, "no map");
    *m = e->grid.live;
    if (!ids || e->grid.live == 0) return S2M_OK;
    if (capacity < e->grid.live) return fail(e, S2M_ERR_CAPACITY, "id buffer too small");
    S2M_ENTER(e);
    if (e->grid.live > e->stage_cap) {
        int rc = grow(e, &e->d_stage, e->grid.live);
        if (rc) return rc;
        e->stage_cap = e->grid.live;
    }
    const uint32_t *rank = nullptr;
    int rc = caller_index_table(e, &rank);
    if (rc) return rc;
    launch_ids_by_rank(e->grid.pidx, rank, e->grid.m, reinterpret_cast<uint32_t *>(e->d_stage), e->stream);
    rc = sync_stream(e, e->stream, "the map's ids in caller order");
    if (rc) return rc;
    S2M_HIP(e, hipMemcpyAsync(ids, e->d_stage, (size_t)e->grid.live * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    return sync_stream(e, e->stream, "the map's ids on their way to the caller");
}

namespace {
// the report that has landed in pinned memory joins what is waiting for the follower (e->chg_*): entries, and the box deletes
// with their place in the sequence
void stash_report(s2m_engine *e)
{
    ChangeLog &L = e->log;
    const uint32_t *h = L.h_head + 1;
    if (h[2] != 0u) { e->chg_overflow = true; return; }
    const size_t a0 = e->chg_added.size(), r0 = e->chg_removed.size();
    e->chg_added.insert(e->chg_added.end(), L.h_added, L.h_added + h[0]);
    e->chg_removed.insert(e->chg_removed.end(), L.h_removed, L.h_removed + h[1]);
    size_t at = 0;
    for (size_t k = 0; k < L.boxes_per_posted.size() && k < (size_t)h[3]; ++k)
        for (int b = 0; b < L.boxes_per_posted[k]; ++b, ++at) {
            s2m_engine::BoxEvent ev;
            std::memcpy(ev.box, L.boxes_posted.data() + 6 * at, sizeof(ev.box));
            ev.after_added = (int64_t)(a0 + h[4 + 2 * k]);
            ev.after_removed = (int64_t)(r0 + h[5 + 2 * k]);
            e->chg_boxes.push_back(ev);
        }
    L.boxes_posted.clear();
    L.boxes_per_posted.clear();
}
void post_report(s2m_engine *e)
{
    ChangeLog &L = e->log;
    L.boxes_posted.swap(L.boxes_log);
    L.boxes_per_posted.swap(L.boxes_per_log);
    L.boxes_log.clear();
    L.boxes_per_log.clear();
    changelog_post(L, e->stream);
}
}  // namespace

int s2m_map_get_changes(s2m_engine *e, uint64_t *token, s2m_map_changes *c)
{
    if (!e || !token || !c) return fail(e, S2M_ERR_ARG, "null argument");
    if (!e->map_ready) return fail(e, S2M_ERR_STATE, "no map");
    if (e->map_borrowed) return fail(e, S2M_ERR_STATE, "the map belongs to another handle (s2m_map_share)");
    S2M_ENTER(e);
    c->n_added = 0; c->n_removed = 0; c->n_boxes = 0; c->resync = 0;
    ChangeLog &L = e->log;
    auto fresh = [&]() {
        L.on = true;
        L.token = (++e->log_seq << 8) | 1u;
        *token = L.token;
    };
    auto start_over = [&]() {  // the follower fetches the map as it is now; the log starts from here
        launch_log_reset(L, e->stream);
        L.boxes_log.clear(); L.boxes_per_log.clear(); L.boxes_posted.clear(); L.boxes_per_posted.clear();
        L.posted = false;   // (a report still on its way is stale: its sequence number is never waited for)
        e->chg_added.clear(); e->chg_removed.clear(); e->chg_boxes.clear();
        e->chg_overflow = false;
        fresh();
        c->resync = 1;
    };
    if (!L.on || L.token == 0 || *token != L.token) {
        // the caller does not hold the state the log starts from (first call, a rebuild in between, another follower's token)
        // (room for what a few frames change -- a field-of-view trim is logged as its boxes, not as its points; beyond the room
        // the follower fetches the map.  S2M_LOG_CAP: test hook, a capacity small enough to overflow)
        int64_t cap = std::max<int64_t>((int64_t)1 << 18, 4 * e->n_cap);
        if (const char *g = std::getenv("S2M_LOG_CAP")) cap = std::max<int64_t>(16, std::atoll(g));
        S2M_HIP(e, changelog_ensure(L, cap, e->stream));
        start_over();
        return S2M_OK;
    }
    // 1. what has been posted for this follower (lag: by the previous call, a frame ago -- it has landed) and, without lag,
    //    everything up to now
    e->step = "the change log";
    if (L.posted) {
        S2M_HIP(e, changelog_collect(L, e->stream));
        stash_report(e);
    }
    if (c->lag == 0) {
        post_report(e);
        S2M_HIP(e, changelog_collect(L, e->stream));
        stash_report(e);
    }
    if (e->chg_overflow) {  // more changes than the log holds: start over
        start_over();
        return S2M_OK;
    }
    // 2. hand it over
    const int64_t na = (int64_t)e->chg_added.size(), nr = (int64_t)e->chg_removed.size(), nx = (int64_t)e->chg_boxes.size();
    c->n_added = na; c->n_removed = nr; c->n_boxes = nx;
    if ((na > 0 && (!c->added_xyz || !c->added_ids || c->capacity_added < na)) || (nr > 0 && (!c->removed_ids || c->capacity_removed < nr)) ||
        (nx > 0 && (!c->boxes || !c->box_after_added || !c->box_after_removed || c->capacity_boxes < nx)))
        return fail(e, S2M_ERR_CAPACITY, "s2m_map_get_changes: buffers too small (the changes are kept)");
    for (int64_t i = 0; i < na; ++i) {
        const float4 &p = e->chg_added[(size_t)i];
        c->added_xyz[3 * i] = p.x; c->added_xyz[3 * i + 1] = p.y; c->added_xyz[3 * i + 2] = p.z;
        std::memcpy(&c->added_ids[i], &p.w, sizeof(uint32_t));
    }
    for (int64_t i = 0; i < nr; ++i) {
        const float4 &p = e->chg_removed[(size_t)i];
        if (c->removed_xyz) { c->removed_xyz[3 * i] = p.x; c->removed_xyz[3 * i + 1] = p.y; c->removed_xyz[3 * i + 2] = p.z; }
        std::memcpy(&c->removed_ids[i], &p.w, sizeof(uint32_t));
    }
    for (int64_t i = 0; i < nx; ++i) {
        std::memcpy(c->boxes + 6 * i, e->chg_boxes[(size_t)i].box, 6 * sizeof(float));
        c->box_after_added[i] = e->chg_boxes[(size_t)i].after_added;
        c->box_after_removed[i] = e->chg_boxes[(size_t)i].after_removed;
    }
    e->chg_added.clear(); e->chg_removed.clear(); e->chg_boxes.clear();
    // 3. with lag: what has happened since leaves for the host now and is handed over by the next call -- nobody waits
    if (c->lag != 0) post_report(e);
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
    S2M_ENTER(e);
    const uint32_t *rank = nullptr;
    int rc = caller_index_table(e, &rank);
    if (rc) return rc;
    rc = sync_stream(e, e->stream, "the map's order");
    if (rc) return rc;
    S2M_HIP(e, hipMemcpyAsync(order, rank, (size_t)e->grid.m * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    return sync_stream(e, e->stream, "the map's order on its way to the caller");
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

int s2m_map_update_stats(const s2m_engine *e, int64_t stats[12])
{
    if (!e || !stats) return S2M_ERR_ARG;
    stats[0] = e->n_merged;
    stats[1] = e->n_rebuilt;
    stats[2] = e->n_regrid;
    stats[3] = map_allocations();
    stats[4] = e->map.n_relaid;
    stats[5] = e->map.n_big_slab;
    for (int k = 0; k < 4; ++k) stats[6 + k] = e->map.slab_fail[k];
    stats[10] = e->n_beside;
    stats[11] = e->n_beside_regrid;
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
    if (blind < 0) {
        // the whole map's radius in one launch: the box of the bricks in use seen from anywhere within a sensor's reach of it
        const double r = 4.0 * half_diag + 1000.0;
        m.open_count = m.hard_count + 3;   // (zeroed by the collecting kernel, read by the caller's next hand-back)
        launch_collect_short(m, e->stream);
        m.band0 = 2.0f * std::sqrt(m.gates.knn_d2_gate);
        m.gates.knn_d2_gate = (float)std::min(r * r, 1.0e37);
        launch_match_hard_only(m, e->stream);
        launch_far_reset(m.hard_count, e->d_qheads, e->stream, /*keep_open=*/true);
        S2M_HIP(e, hipGetLastError());
        return S2M_OK;
    }
    launch_far_reset(m.hard_count, e->d_qheads, e->stream);  // (also the word a blind launch may have left)
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
            m.band0 = 2.0f * std::sqrt(m.gates.knn_d2_gate);  // nothing closer than the radius searched so far: start at twice that
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
    S2M_ENTER(e);
    if (e->n == 0 || e->grid.m == 0 || e->nn_complete) return S2M_OK;
    if (e->short_lists == 0) { e->nn_complete = true; return S2M_OK; }  // the last rematch pass counted them: none
    int rc = complete_lists(e, kK, 0, n_completed);
    if (rc) return rc;
    e->nn_complete = true;
    e->nn_nearest = true;
    return S2M_OK;
}


}  // extern "C"
