// s2m_kernels.h -- host-side launchers of the HIP kernels (one per .hip translation unit).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "s2m_device.h"
#include "s2m_loop.h"
#include "s2m_wait.h"

namespace s2m {

// ---- a few device words handed to the host: one kernel that writes them into pinned memory + one stream sync.
// (Every hipMemcpyAsync of 4 bytes is a kernel launch plus ~20 us of turn-around on this stack; the update path
// used to issue eleven of them.)
struct Mailbox {
    uint32_t *h = nullptr;    // pinned host words
    uint32_t *dev = nullptr;  // the same memory as the device sees it
    uint32_t seq = 0;         // word 0 of the buffer carries the sequence number of the last fetch
};
constexpr int kMailSlots = 16;
void free_mailbox(Mailbox &mb);
// out[i] = *src[i] for i < k (k <= kMailSlots); returns when the kernel -- and with it everything queued on the
// stream before it -- has finished (the host polls the pinned sequence word instead of synchronising the stream), or
// kWaitTimedOut when the handle's deadline passes first
hipError_t mail_fetch(Mailbox &mb, const uint32_t *const *src, int k, uint32_t *out, hipStream_t st);
// the same wait without a payload: returns when everything queued on the stream has finished
hipError_t mail_post(Mailbox &mb, const uint32_t *const *src, int k, hipStream_t st);   // the two halves of mail_fetch
hipError_t mail_collect(Mailbox &mb, int k, uint32_t *out, hipStream_t st);
hipError_t mail_wait(Mailbox &mb, hipStream_t st);

// ---- s2m_map.hip : map build (KD_TREE::Build, ikd-Tree/ikd_Tree.cpp:408-423) -------------------
// bounding box of an AoS cloud (s2m_map.hip): scratch holds the per-workgroup partial boxes + the result
constexpr int kBboxBlocks = 1024;
constexpr int kBboxScratchFloats = (kBboxBlocks + 1) * 6;
// the voxel grid's numbers (pcl::VoxelGrid: min_b, divb_mul) as the device derives them from a cloud's box
struct VoxelDimsDev {
    int min_b[3];
    uint32_t too_fine;  // the voxel index would overflow int32 (PCL refuses such a leaf)
    int64_t mul1, mul2;
    uint32_t bits, pad;  // bits of the largest voxel index
};
// the box only, left on the device at scratch + kBboxBlocks * 6 (and, with dims, the voxel grid's numbers): no hand-back
hipError_t cloud_bbox_launch(const float *xyz, int64_t stride, int64_t n, float *scratch, float inv_leaf, VoxelDimsDev *dims, hipStream_t st);
hipError_t cloud_bbox(const float *xyz, int64_t stride, int64_t n, float *scratch, Mailbox &mail, float lo[3], float hi[3],
                      hipStream_t st);

constexpr int kSentinelPoints = 32;  // pts[m .. m+32): padding targets of the search kernels' point batches (8 slots x up to 4 lanes
                                     // that share a run point by point)
struct MapBuffers {
    // owned by the engine, (re)allocated by build_map
    float4 *pts = nullptr;
    uint32_t *pidx = nullptr;                   // point id of every sorted position: the caller's index after a build, ascending in
                                                // caller order ever after (removals leave gaps; new points take next_id, next_id + 1, ...)
    int64_t next_id = 0;
    bool ids_dense = true;                      // the ids ARE the caller indices (no point was removed since the last build)
    float4 *pts2 = nullptr;
    uint32_t *pidx2 = nullptr;                  // the other halves of the double buffers a merge update writes into
    uint4 *top = nullptr, *top2 = nullptr;      // the toroidal top array and the spare a re-lay writes into
    uint32_t *tab = nullptr;
    int64_t pts_cap = 0, pidx_cap = 0, pts2_cap = 0, pidx2_cap = 0, top_cap = 0, top2_cap = 0, tab_cap = 0;
    // per-point arrays of scratch_cap + 1 elements.  keys_alt holds the SORTED keys of the current map (it stays
    // valid between updates: the merge update reads it); keys is the unsorted input of a build or the output of a
    // merge; vals / vals_alt are the build sort's payload
    uint64_t *keys = nullptr, *keys_alt = nullptr;
    uint32_t *vals = nullptr, *vals_alt = nullptr;
    uint32_t *work_a = nullptr, *work_b = nullptr, *work_c = nullptr;
    uint32_t *bstart = nullptr;  // per occupied brick: first position in pts
    int64_t bstart_cap = 0;
    uint64_t *bkey = nullptr;    // per occupied brick: its key (brick_key, s2m_device.h: the high part of its points' keys)
    uint8_t *bmark = nullptr;    // per occupied brick: bit 0 = a point of it was removed, bit 1 = a new point goes into it, bit 2 = opened (this update)
    uint32_t *bend = nullptr;    // per occupied brick: end of the stretch of positions it owns (its points, then room)
    uint32_t *bmove = nullptr;   // per touched brick of an in-place update: first position of its new stretch in the tail, ~0 = it stays
    void *bplan = nullptr;       // per brick: what the in-place update's plan found for it (s2m_mapedit.hip, BrickPlan: 32 bytes)
    uint32_t *blist = nullptr;   // the ids of the bricks an in-place update touches, compact
    int64_t bkey_cap = 0, bmark_cap = 0, bend_cap = 0, bmove_cap = 0, bplan_cap = 0, blist_cap = 0;
    int64_t main_ext = 0;        // positions [0, main_ext) are ordered by key (the layout of the last build or merge); [main_ext, Grid::m)
                                 // is the TAIL: stretches handed to bricks that in-place updates opened or moved, in the order they asked
    int64_t tail_used = 0;       // positions of the tail handed out so far (host copy of the device cursor, counters[kTailWord] - main_ext)
    int64_t n_moved = 0;         // bricks that in-place updates have put into the tail (diagnostic)
    int64_t slab_fail[4] = {0, 0, 0, 0};  // in-place updates given up because of: a point beyond the representable cells, no spare
                                 // table rows, a brick too large to stage, the tail exhausted (diagnostic)
    int prep_lds = 0;            // slab_prepare_kernel's dynamic LDS: 0 not asked yet, 1 granted, -1 refused (the separate kernels run)
    bool no_fused_prep = false;  // (set by the engine from S2M_NO_FUSED_PREP: the in-place update's separate kernels instead of slab_prepare_kernel)
    uint2 *run = nullptr;        // per top slot: where the brick's new points stand among the update's sorted ones (slab_head_kernel)
    int64_t run_cap = 0;
    uint32_t *grow = nullptr;    // per top slot: points the brick has gained (net) by in-place updates since the room was laid out
    int64_t grow_cap = 0;
    int64_t added_since_layout = 0;  // host bound of the sum of `grow`
    uint64_t layout_gen = 0;     // counts builds and merges: a new dense layout of pts (in-place updates keep the layout)
    uint64_t *mk = nullptr;      // merge update: sorted keys of the new points
    uint32_t *mv = nullptr;      // merge update: their stage positions, then their lower bounds among the old keys
    int64_t mk_cap = 0, mv_cap = 0;
    unsigned long long *dword = nullptr;  // merge update: bit mask of the removed points per 64 caller indices
    int64_t dword_cap = 0;
    Mailbox mail;
    uint32_t *h_stats_dev = nullptr;
    uint32_t *h_stats = nullptr;  // pinned: occupied bricks, then the occupied-cell counters, copied behind a merge
    hipEvent_t stats_event = nullptr;
    bool stats_pending = false;
    void *sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    int64_t scratch_cap = 0;
    float *bbox = nullptr;       // kBboxScratchFloats floats on device (cloud_bbox scratch)
    uint32_t *counters = nullptr; // small device counters (64 words, then the occupied-cell shards)
    int64_t n_relaid = 0;         // times the top array was re-laid for a box that had outgrown or left its window
    int big_slab = 0;             // the crowded-brick form of the in-place rewrite: 0 = not asked yet, 1 = granted, -1 = refused by the device
    int64_t n_big_slab = 0;       // bricks that went through it (diagnostic)
};
constexpr int kBricksWord = 32;   // MapBuffers::counters[kBricksWord]: the number of bricks (ids in use), kept on the device
constexpr int kBoxWords = 34;     // ... [kBoxWords .. +6): a box of bricks on its way to the host (lo xyz, hi xyz)
constexpr int kTailWord = 40;     // ... [kTailWord]: the next free position of the tail

struct MapStats {
    int64_t bricks = 0, top_entries = 0, occupied_cells = 0;
    int64_t layout_points = 0;  // points of the layout the cell count belongs to (the last build or merge)
};

// xyz_dev: device pointer, stride in floats.  cell <= 0 selects the cell size from the density.
// Returns hipSuccess or the failing HIP error; *too_large set when the grid would not fit.
// keep_origin: the origin of the map this one replaces (a rebuild inside an update keeps the cells where they are).
hipError_t build_map(const float *xyz_dev, int64_t stride, int64_t m, float cell, MapBuffers &buf, Grid &grid,
                     MapStats &stats, bool &too_large, hipStream_t st, const float *keep_origin = nullptr);
void free_map(MapBuffers &buf);
hipError_t map_reserve_like(MapBuffers &dst, const MapBuffers &src, int64_t build_points);   // dst's arrays at least as large as src's, and room to BUILD build_points points (dst not in use)
bool map_build_would_fit(const float lo[3], const float hi[3], float cell, const float *keep_origin);   // build_map's verdict on a box, nothing touched
int64_t map_allocated_bytes();   // ... and their bytes
int64_t map_allocations();  // device (re)allocations by the map build / merge / update code so far (all handles; diagnostic)
void note_allocation(const char *what = "", size_t bytes = 0);
// bricks / occupied_cells of the last build or merge (a merge does not wait for them: they arrive behind it)
hipError_t resolve_stats(MapBuffers &buf, MapStats &stats);
// The map after an update without a new sort (s2m_mapedit.hip, "merge update"): alive_s[sorted position] for the m old points
// (m + 1 readable bytes), n_new staged points in their order.  merged = false (and nothing changed) when the update cannot be
// merged -- a new point beyond the representable cell range, no room in the scratch arrays, empty map -- and the caller falls
// back to update_finish + build_map.
hipError_t merge_update(MapBuffers &buf, Grid &g, MapStats &stats, const uint8_t *alive_s,
                        const float4 *stage, int64_t n_new, bool &merged, hipStream_t st, bool with_slack = false);
// The same update in place when every touched brick still fits where it stands (s2m_mapedit.hip, slab_update): done = false and
// nothing touched otherwise.  flags: five zeroed words of the update's counters.
// Batches of update_add(defer) that have not been written to the staging list: a scan's batches are staged by the in-place
// update's own preparation kernel (slab_prepare_kernel) -- or, when that cannot run, by update_materialize.
struct StagePending {
    const float4 *la = nullptr;            // the voxel rule's batch ...
    const uint32_t *flag = nullptr;        // ... and which of it won its voxel (add_resolve_kernel)
    int na = 0;
    const uint64_t *vkey = nullptr;        // its winner table, still to be emptied
    unsigned long long *vtab = nullptr;
    const float4 *lb = nullptr;            // the batch that is added whole, behind the winners
    int nb = 0;
    bool on = false;
};
// can the in-place update prepare a batch of at most n_bound points in one workgroup (and stage pending batches itself)?
bool slab_fuses(MapBuffers &buf, const Grid &g, const MapStats &stats, int64_t n_bound);
// n_dev (optional): the device's word with the number of staged points when the host only knows the bound n_new (update_add
// with defer): the points from *n_dev on are ignored, n_new comes back as that number and counted = true once the update's
// hand-back has been read (also when the update could not be done in place).
hipError_t slab_update(MapBuffers &buf, Grid &g, MapStats &stats, uint8_t *alive_s, const float4 *stage, int64_t &n_new,
                       uint32_t *flags, bool &done, hipStream_t st, const uint32_t *n_dev = nullptr, bool *counted = nullptr,
                       StagePending *pend = nullptr, float4 *stage_out = nullptr, uint32_t *count_out = nullptr);

// ---- s2m_mapupd.hip : incremental map maintenance (map_incremental / Add_Points / Delete_Point_Boxes) ----
struct UpdateBuffers {
    uint8_t *alive_s = nullptr;    // per sorted position (position in Grid::pts): 1 = holds a point; 0 = removed by this update, or a
                                   // hole an in-place update left (m + 1 bytes)
    int64_t alive_s_cap = 0;
    uint64_t alive_gen = ~0ull, layout_gen = 0;  // alive_s was made for layout alive_gen; the map's layout now (set by the engine)
    uint8_t *bmark = nullptr;      // the map's per-brick marks (MapBuffers::bmark, set by the engine): bit 0 = a point of the brick was removed
    uint32_t *counters = nullptr;  // kUpdWords words, zeroed by every update_begin: [1] voxels rewritten (tmp_counter), [2] points deleted
                                   // by boxes, [kUpdStageWord .. +2) staged points (deferred count), [14] live points (order_by_id), [kUpdSlabWord .. +11) the in-place update's words
                                   // (s2m_mapedit.hip), [kUpdVoxWord .. +6) the box of the voxels of map_incremental's PointToAdd
    float4 *stage = nullptr;       // points to append, in order
    int64_t stage_cap = 0, stage_n = 0;
    // update_add(defer): the count stays on the device (counters[kUpdStageWord + (stage_ops & 1)]) and stage_n is an upper
    // bound of it until the update's commit reads it back with the hand-back it waits for anyway (update_stage_count)
    bool stage_deferred = false;
    int stage_ops = 0;
    StagePending pend;           // deferred batches not yet written to `stage` (update_materialize / slab_prepare_kernel)
    bool fuse_stage = true;      // (the engine: false under S2M_NO_FUSED_PREP)
    uint32_t deleted_reported = 0;  // box deletes already reported to the caller within this update
    // per-batch scratch
    uint64_t *key = nullptr, *key2 = nullptr;
    uint32_t *val = nullptr, *val2 = nullptr, *cnt = nullptr, *best_idx = nullptr, *best_pos = nullptr, *add_flag = nullptr, *pos = nullptr;
    float *dnew = nullptr, *best_d = nullptr;
    int64_t batch_cap = 0;
    // the map ordered by point id (rebuild path; caller-order getters): unsorted / sorted (id, position) pairs
    uint32_t *flag32 = nullptr, *pos_old = nullptr, *ord_key = nullptr, *ord_val = nullptr;
    int64_t ord_cap = 0;
    float4 *list = nullptr;        // the new point list handed to build_map
    int64_t list_cap = 0;
    void *tmp = nullptr;
    size_t tmp_bytes = 0;
    float *boxes = nullptr;
    int64_t boxes_cap = 0;
    float4 *cvt = nullptr;         // conversion / classification scratch
    int64_t cvt_cap = 0;
    int vox_reach[6] = {0, 0, 0, 0, 0, 0};  // voxels below / above the sensor's that the last scans' PointToAdd lists reached (incr_classify)
    int64_t reserve_hint = 0;            // points of the largest scan the handle has seen: staging arrays are sized for it at once
    unsigned long long *vtab = nullptr;  // direct-address table over a batch's voxel box (winner per voxel), all ~0 between batches
    int64_t vtab_cap = 0;                // slots
    Mailbox mail;
};
constexpr int kUpdWords = 64, kUpdStageWord = 8, kUpdSlabWord = 16, kUpdVoxWord = 32;
// [kUpdListWord .. +2) lengths of map_incremental's two lists (+2, +3 stay zero), [kUpdBatchWord] winners of update_add's batch
// (+1 stays zero): written by the one-workgroup compactions
constexpr int kUpdListWord = 40, kUpdBatchWord = 44;
// box of voxels (edge = the down-sampling size) that holds every point of a batch: the batch's winner per voxel comes from a
// direct-address table over it (or the sort that groups the batch by voxel uses a linear index of `bits` bits instead of the
// 63-bit packed key); bits == 0: not available (use the packed key).  map_incremental measures it from the scan's own
// PointToAdd list (incr_classify), so it follows the sensor and not the map.
constexpr int kVoxTableBits = 25;  // a voxel box of up to 2^25 voxels gets a winner table (256 MB at most; 45 MB at C3)
struct VoxBox {
    int lo[3] = {0, 0, 0}, d[3] = {0, 0, 0};
    int bits = 0;
};
void free_update(UpdateBuffers &u);
hipError_t update_reserve_like(UpdateBuffers &dst, const UpdateBuffers &src, hipStream_t st);  // dst's batch / staging capacities >= src's
hipError_t update_begin(UpdateBuffers &u, const Grid &g, hipStream_t st);
hipError_t update_add(UpdateBuffers &u, const Grid &g, const float4 *np, int64_t n, bool downsample, float ds,
                      int64_t *n_added, hipStream_t st, const VoxBox *vox = nullptr, bool defer = false);
// the staged count of a deferred update: its device word (nullptr when the host's stage_n is exact) / read back now
const uint32_t *update_stage_word(const UpdateBuffers &u);
hipError_t update_materialize(UpdateBuffers &u, hipStream_t st);  // pending batches into the staging list (no-op without any)
hipError_t update_stage_count(UpdateBuffers &u, hipStream_t st);
// false when no box of the call reaches the bricks in use (host arithmetic: such a call launches nothing and waits for nothing)
bool delete_touches_map(const Grid &g, const float *boxes_host, int nb);
hipError_t update_delete(UpdateBuffers &u, const Grid &g, const float *boxes_host, int nb, int64_t *n_deleted,
                         hipStream_t st);
hipError_t update_finish(UpdateBuffers &u, const Grid &g, int64_t *m_out, hipStream_t st);
// (the neighbour lists need only be exact up to the gate and in their first entry: s2m_mapupd.hip, incr_classify_kernel)
// extra / extra_out: one more device word brought back with the counts (the completion's "lists left open")
hipError_t incr_classify(UpdateBuffers &u, const Pose &pose, const float *sx, const float *sy, const float *sz, int n,
                         const int32_t *nn_idx, const Grid &g, bool have_nn, double fs, float4 **to_add, int64_t *n_add,
                         float4 **no_down, int64_t *n_no_down, hipStream_t st, VoxBox *vox = nullptr, bool begin_update = false,
                         const uint32_t *extra = nullptr, uint32_t *extra_out = nullptr);
hipError_t xyz_to_float4(UpdateBuffers &u, const float *xyz_dev, int64_t stride, int64_t n, float4 **out, hipStream_t st);
// the map in CALLER order as packed xyz (ikdtree.flatten's counterpart): xyz[3 * pidx[j]] = pts[j]
void launch_map_to_xyz(const float4 *pts, const uint32_t *rank, int64_t m, float *xyz, hipStream_t st);
// rank[position] = dense caller index of every live position (cold path: a sort of the ids), 0xffffffff for a removed one
hipError_t caller_ranks(UpdateBuffers &u, const Grid &g, const uint8_t *alive_s, const uint32_t **rank, int64_t *live, hipStream_t st);
// ---- the change log (s2m_map_get_changes): what the updates since the last report added and removed, by point id ----------
constexpr int kLogMarks = 8;                      // box deletes a log can hold between two reports (each one: up to kLogBoxesPer boxes)
constexpr int kLogBoxesPer = 8;
constexpr int kLogWords = 4 + 2 * kLogMarks;      // [0] added, [1] removed, [2] overflow, [3] marks, then {added, removed} at every mark
struct ChangeLog {
    float4 *added = nullptr;      // {x, y, z, bitcast(id)} of every point added
    float4 *removed = nullptr;    // ... and of every point removed one by one (the voxel rule; not the box deletes)
    uint32_t *counts = nullptr;   // device: kLogWords words
    // the report on its way to the follower, in pinned host memory the flush kernels write: [0] sequence word, then `counts`
    uint32_t *h_head = nullptr, *h_head_dev = nullptr;
    float4 *h_added = nullptr, *h_added_dev = nullptr, *h_removed = nullptr, *h_removed_dev = nullptr;
    uint32_t flush_seq = 0;
    bool posted = false;          // a flush is on its way (or has landed) and has not been handed to the follower
    bool landed = false;          // ... it has been waited for: h_head / h_added / h_removed hold it
    std::vector<float> boxes_log, boxes_posted;      // the box deletes since the last post / of the posted report: 6 floats per box ...
    std::vector<int32_t> boxes_per_log, boxes_per_posted;  // ... and how many boxes each marked delete had
    int64_t cap = 0;
    bool on = false;              // somebody follows the map (the first s2m_map_get_changes switches it on)
    uint64_t token = 0;           // the map state of the last report; 0: none yet
};
void free_changelog(ChangeLog &c);
hipError_t changelog_ensure(ChangeLog &c, int64_t cap, hipStream_t st);
// the points this update removes one by one (alive_s == 0 where a point was), found through the bricks the update marked as
// touched by a removal: cost proportional to the change.  Runs before the map is rewritten.
void launch_log_removed(ChangeLog &c, const uint32_t *bricks_dev, int64_t bricks_bound, const uint8_t *bmark, const uint32_t *tab,
                        const uint8_t *alive_s, const uint32_t *pidx, const float4 *pts, hipStream_t st);
void launch_log_added(ChangeLog &c, const float4 *stage, int64_t n, uint32_t first_id, hipStream_t st);
void launch_log_reset(ChangeLog &c, hipStream_t st);
void launch_log_mark(ChangeLog &c, hipStream_t st);  // a box delete takes its place in the sequence
void changelog_post(ChangeLog &c, hipStream_t st);
hipError_t changelog_collect(ChangeLog &c, hipStream_t st);
// ids in caller order: ids[rank[j]] = pidx[j]
void launch_ids_by_rank(const uint32_t *pidx, const uint32_t *rank, int64_t m, uint32_t *ids, hipStream_t st);
// caller indices of a neighbour list: out[i] = nn[i] >= 0 ? pidx[nn[i]] : -1
void launch_positions_to_indices(const int32_t *nn, const uint32_t *pidx, int64_t count, int32_t *out, hipStream_t st);

// ---- s2m_voxel.hip : scan voxel down-sampling (pcl::VoxelGrid, laserMapping.cpp:775-776) ------------
struct VoxelBuffers {
    uint32_t *key = nullptr, *key2 = nullptr;  // voxel index (below 2^31: PCL's own limit)
    uint32_t *val = nullptr, *val2 = nullptr, *head = nullptr, *pos = nullptr;
    float *box = nullptr;  // cloud_bbox scratch
    VoxelDimsDev *dims = nullptr;  // the grid's numbers on the device (the form without the box's hand-back)
    bool no_hint = false;          // (the engine: true under S2M_NO_VOXEL_HINT -- the box is fetched first, as before)
    int kbits_hint = 0;            // bits the last cloud's voxel indices had: the next one is sorted on as many before its box is known
    int64_t n_respeculated = 0;    // (diagnostic: clouds whose indices had more bits than the hint -- done again the classic way)
    void *tmp = nullptr;
    size_t tmp_bytes = 0;
    int64_t cap = 0;
    Mailbox mail;
};
void free_voxel(VoxelBuffers &v);
hipError_t voxel_downsample(VoxelBuffers &v, const float *xyz, int64_t stride, int64_t n, float leaf, float *ox,
                            float *oy, float *oz, int64_t *n_out, bool *too_fine, hipStream_t st);

// ---- s2m_undistort.hip : per-point motion compensation (IMU_Processing.hpp:333-370) ----------------
struct UndistBuffers {
    uint32_t *key = nullptr, *key2 = nullptr, *val = nullptr, *val2 = nullptr;
    void *tmp = nullptr;
    size_t tmp_bytes = 0;
    double *poses = nullptr;
    int pose_cap = 0;
    float *out = nullptr;  // n x 3 packed, device
    int64_t cap = 0;
    uint32_t *perm = nullptr;  // sorted position -> input index, on request
    int64_t perm_cap = 0;
    // records that arrive in time order (every spinning or scanning LiDAR driver delivers them so) need no sort: the key kernel
    // counts the places where the order is broken, one hand-back says whether any
    bool always_sort = false;      // (the engine: true under S2M_NO_TIME_SHORTCUT)
    uint32_t *unsorted = nullptr;  // device counter, never reset: compared with what the host saw last
    uint32_t unsorted_seen = 0;
    int64_t n_sorted_input = 0, n_unsorted_input = 0;  // (diagnostic)
    Mailbox mail;
};
void free_undist(UndistBuffers &u);
hipError_t undistort(UndistBuffers &u, const float *pts, int64_t stride, int64_t n, int off_a, int off_b,
                     const double *poses_host, int K, const Pose &end, bool sort_by_time, uint32_t *perm_dev,
                     hipStream_t st, bool order_ready = false);
hipError_t undistort_order(UndistBuffers &u, const float *pts, int64_t stride, int64_t n, int off_a, int off_b, hipStream_t st);

// ---- s2m_match.hip : exact 5-NN ------------------------------------------------------------------
// match_hard hands its list out dynamically: every resident wave takes the point of its own index first, then pulls
// further ones from one of kQueueShards heads (separate cache lines: one head would serialise at ~90 tickets per
// microsecond).  The heads are zeroed by every reduce launch, like the list length.
constexpr int kQueueShards = 64;
constexpr int kQueueStride = 32;  // words (128 B)
constexpr int kQueueWords = kQueueShards * kQueueStride;
// What the first-shell kernel hands to match_hard for a point it could not resolve -- everything that kernel needs to
// start, in one 32-byte record (it used to follow an index into three more arrays: one dependent load level per point)
struct HardRec {
    float wx, wy, wz;   // world-frame query
    uint32_t qi;        // scan point
    float d5;           // squared distance of the first shell's 5th neighbour (valid when found == 5)
    uint32_t found;     // neighbours the first shell held (0 = empty shell, 5 = radius known)
    uint32_t slot;      // scan of a batched launch (0 otherwise)
    uint32_t pad;
};
static_assert(sizeof(HardRec) == 32, "HardRec is two 16-byte loads");
struct MatchArgs {
    Grid grid;
    Pose pose;
    Gates gates;
    const float *sx, *sy, *sz;
    int n;
    int32_t *nn_idx;     // n x 5, sorted position of the neighbour in Grid::pts, -1 = missing
    float *nn_d2;        // n x 5 ascending, INFINITY = missing
    HardRec *hard_rec;   // records of scratch: the points the first-shell kernel could not resolve, without / with a radius
    int64_t hard_off1 = 0;  // where the second list (radius known) starts inside hard_rec (n for one scan; the sum of all scans' n in a batch)
    uint32_t slot = 0;   // which scan of a batched launch this is (travels in the record so that match_hard finds the outputs)
    uint32_t *hard_count; // the two lengths (device counters, reset by every reduce launch)
    uint32_t *qheads = nullptr;  // kQueueShards dequeue heads of match_hard's work queue, kQueueStride words apart
    LoopLaunch loop;             // device-resident loop (s2m_loop.h): state == nullptr for a host-stepped pass
    int32_t far_waves = 0;       // > 0: waves of the far-point launch (few far points expected); 0: as many as stay resident
    uint32_t *open_count = nullptr;  // launch_match_hard_only: += 1 for every list that is still unsettled at the launch's radius
    float band0 = 0.0f;          // launch_match_hard_only: first band of the search in metres (0: the kernel's own)
    int32_t short_k = kK;        // launch_collect_short / launch_match_hard_only: a list is open while its short_k-th entry is not proven (5: the whole list; 1: the nearest)
};
// first shell (s2m_match.hip), then -- unless `group` carries bit 0x40000 -- the far-point kernel (s2m_match_far.hip)
void launch_match(const MatchArgs &a, int group, hipStream_t st);
// the far-point kernel on its own, in its per-iteration form (the redo of a pass whose bet "no far points" was lost)
void launch_match_far_points(const MatchArgs &a, int group, hipStream_t st);
void launch_far_points(const MatchArgs &a, bool wide, hipStream_t st);
// completion of the lists that ended short at the gate (s2m_complete_neighbors): the scan points with fewer than five
// neighbours appended to far-point list 0 (a.pose = the pose of the rematch pass that produced the lists), and the
// far-point kernel on its own with the brick neighbourhood clipped to the grid (a.gates.knn_d2_gate = the radius^2 of
// this round)
void launch_collect_short(const MatchArgs &a, hipStream_t st);
void launch_match_hard_only(const MatchArgs &a, hipStream_t st);
// the far-point list's counters and queue heads back to zero (one launch; two memsets are four)
void launch_far_reset(uint32_t *hard_count, uint32_t *qheads, hipStream_t st, bool keep_open = false);  // keep_open: word 3 (lists left open) stays

// ---- s2m_reduce.hip : [plane fit +] residual + Jacobian + normal block ----------------------------
constexpr int kRedBlock = 512;   // 8 waves per workgroup: 128 partial rows at 65k points for the in-kernel final sum
constexpr int kRowsBlock = 256;
constexpr int kTicketGroups = 16;  // group counters of the two-level arrival count, one cache line apart
constexpr int kTicketStride = 32;  // words (128 B)
constexpr int kTicketWords = kTicketStride * (1 + kTicketGroups);
constexpr int kRedTerms = 96;    // 78 (upper triangle of 12x12) + 12 + total_res + count, padded
struct ReduceArgs {
    Pose pose;
    Gates gates;
    const float *sx, *sy, *sz;
    int n;
    int fit;             // rematch pass: fit the plane from the fresh neighbours first
    const int32_t *nn_idx;
    const float *nn_d2;
    const float4 *pts;   // the sorted map points {x, y, position, z}: nn_idx holds sorted positions
    float4 *plane;
    uint8_t *flags;
    uint8_t *sel;
    uint8_t *eff;
    float *pd2;
    double *partials;  // blocks x kRedTerms
    double *block;     // S2M_BLOCK_DOUBLES output (device)
    uint32_t *ticket;  // arrival counters of the in-kernel final sum (kTicketWords words, zero before the first launch)
    uint32_t *hard_count;            // reset to 0 for the next rematch pass; its sum is published as block[158]
    uint32_t *qheads = nullptr;      // match_hard's dequeue heads, reset likewise
    int spec = 0;                    // the far-point kernel was NOT launched for this pass (the host bet on an empty list):
                                     // if the list is not empty the block is void and the counters are left for the redo
    double *host_block;              // optional: pinned host copy of block, device-visible pointer
    unsigned long long *host_flag;   // optional: set to seq (system scope) after host_block is written
    unsigned long long seq;
    LoopLaunch loop;                 // device-resident loop: pose and pass kind come from loop.state, the last workgroup
                                     // runs the Kalman update and the judgement (s2m_loop.h)
};
// ---- one grid for K scans (s2m_iterated_update_batch, BASELINE configs[4] on one device) -----------------------
// Every pass of the K scans that are ready is ONE launch per kernel: blockIdx.y selects the scan, whose pose, flags
// and buffers sit in a table that travels in the kernel arguments (K <= 8 per launch: 2.9 KB of the 4 KB a launch may
// carry); the far points of all scans share one list and one set of queue heads, so the resident waves of match_hard
// balance over 8x the points.  The per-point code and the partial-sum shapes are those of the single-scan kernels:
// results are bit-identical to s2m_iterated_update.
constexpr int kBatchMax = 8;
struct ScanDesc {
    Pose pose;
    const float *sx, *sy, *sz;
    int32_t *nn_idx;
    float *nn_d2;
    float4 *plane;
    uint8_t *flags, *sel, *eff;
    float *pd2;
    double *partials, *block;
    uint32_t *ticket;
    double *host_block;
    unsigned long long *host_flag;
    unsigned long long seq;
    int32_t n;
    int32_t rematch;   // this pass runs the search (and the plane fit) for this scan
    int32_t active;    // 0: the scan has finished (or is not part of this launch)
    int32_t pad;
    LoopLaunch loop;   // device-resident loop: pose / rematch / active come from loop.state instead
};
struct BatchArgs {
    Grid grid;
    Gates gates;
    int32_t k;
    int32_t n_max;               // largest n of the table (grid size of the per-point kernels)
    HardRec *hard_rec;           // unified far-point lists of all scans: [no radius | radius known], hard_off1 apart
    int64_t hard_off1;
    uint32_t *hard_count;        // this launch's counters / queue heads ...
    uint32_t *qheads;
    uint32_t *hard_count_next;   // ... and the other set, which the reduce kernels zero for the next launch
    uint32_t *qheads_next;
    ScanDesc d[kBatchMax];
};
static_assert(sizeof(BatchArgs) <= 4096, "the table must fit the kernel-argument segment");
// the kernels of one pass for the active scans of the table: search (for the scans with rematch set), then the reduce
// kernels (FIT for the rematching scans, plain for the others: the host issues whichever the table needs)
void launch_match_batch(const BatchArgs &b, hipStream_t st);
void launch_far_points_batch(const BatchArgs &b, bool wide, hipStream_t st);
void launch_reduce_batch(const BatchArgs &b, bool any_fit, bool any_plain, hipStream_t st);
void launch_reduce_batch_loop(const BatchArgs &b, hipStream_t st);  // every scan of the table runs the device-resident loop

int reduce_blocks(int n);
int rows_blocks(int n);
void launch_publish(const double *block, double *host_block, unsigned long long *host_flag, unsigned long long seq,
                    hipStream_t st);
void launch_reduce(const ReduceArgs &a, hipStream_t st);

// dense rows of the last pass in index order (laserMapping.cpp:942-979)
struct RowsArgs {
    Pose pose;
    Gates gates;
    const float *sx, *sy, *sz;
    int n;
    const float4 *plane;
    const float *pd2;
    const uint8_t *eff;
    uint32_t *block_off;  // blocks + 1
    double *h_x;          // m x 12
    double *h;            // m
    int32_t *scan_index;  // m
};
void launch_rows(const RowsArgs &a, hipStream_t st);

}  // namespace s2m
