/*
 * daliti_s2m.h -- C ABI of the MI355X scan-to-map registration engine for DaLiTI's eskf_lio.
 *
 * The reference has no plugin/FFI interface for this path: the residual loop, Jacobian build
 * and Kalman update are inline in main() (eskf_lio/src/laserMapping.cpp:820-1102) over globals.
 * This header defines the seam a patched laserMapping.cpp (or any FFI) binds; every entry point
 * cites the reference lines it replaces.  INTEGRATION.md shows the node-side patch.
 *
 * Conventions
 *   - plain C, opaque handle, int return codes (S2M_OK == 0, negative = error), never throws;
 *   - the caller owns every host buffer; the handle owns device memory and one HIP stream;
 *   - one handle is not re-entrant (the reference caller is single-threaded,
 *     laserMapping.cpp:726-731); use one handle per GPU / per thread;
 *   - matrices are row-major doubles; the state is 36 doubles in StatesGroup member order
 *     (eskf_lio/include/common_lib.h:219-227): rot_end[9] pos_end[3] R_L_I[9] T_L_I[3]
 *     vel_end[3] bias_g[3] bias_a[3] gravity[3]; the error state is the reference's 24-vector
 *     [dtheta dpos dtheta_LI dT_LI dv dbg dba dg] (common_lib.h:146-157);
 *   - point clouds are float xyz with a caller-given stride in floats (3 for packed xyz,
 *     12 for pcl::PointXYZINormal, eskf_lio/include/my_utility.h:57);
 *   - there is no CPU fallback: every compute entry point needs a gfx950 device and returns
 *     S2M_ERR_NO_DEVICE / S2M_ERR_HIP otherwise.
 */
#ifndef DALITI_S2M_H
#define DALITI_S2M_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: s2m_map_incremental gained `ekf_inited`, s2m_get_timing_stats writes 6 doubles, s2m_set_timing(n > 2) samples
 * (all round 2, which forgot to bump it); new in round 3: s2m_iterated_update_multi, s2m_complete_neighbors,
 * s2m_map_get_order, s2m_map_update_stats.  3 (round 4): s2m_config gained `device_loop`; new: s2m_bet_stats,
 * s2m_scan_prefetch_raw, s2m_scan_prepare_raw, s2m_map_inplace_updates; block[159] carries the count of neighbour lists
 * short of the gate; s2m_map_get_order may report positions that hold no point (0xffffffff).  A caller built against an
 * older version must be recompiled.  5 (round 6): s2m_config gained `wait_policy`, `wait_timeout_ms`, `wait_spin_us`, `layout_beside`; new code
 * S2M_ERR_TIMEOUT; new: s2m_debug_state; s2m_map_get_changes reports the changes of the update BEFORE the last one when
 * asked to (lag); s2m_map_update_stats writes 12 counters. */
#define S2M_ABI_VERSION 5
#define S2M_K 5            /* NUM_MATCH_POINTS, laserMapping.cpp:77 */
#define S2M_DIM 24         /* DIM_OF_STATES, common_lib.h:23 */
#define S2M_STATE_DOUBLES 36
#define S2M_BLOCK_DOUBLES 160 /* HtH[144] Htz[12] effct total_res far_points short_lists */
#define S2M_FEAT_QUEUE 10  /* QUEUE_SIZE, laserMapping.cpp:192 */

enum {
    S2M_OK = 0,
    S2M_ERR_ARG = -1,        /* null / negative / inconsistent argument            */
    S2M_ERR_NO_DEVICE = -2,  /* no HIP device, or the device is not gfx950         */
    S2M_ERR_HIP = -3,        /* a HIP runtime call failed (see s2m_last_error)     */
    S2M_ERR_STATE = -4,      /* call order: no map / no scan / no pass yet         */
    S2M_ERR_CAPACITY = -5,   /* caller buffer too small, or grid too large         */
    S2M_ERR_NUMERIC = -6,    /* singular matrix in the Kalman update               */
    S2M_ERR_TIMEOUT = -7     /* a wait for the device (or for the handle's side thread) passed s2m_config.wait_timeout_ms:
                              * s2m_last_error names the wait and what the handle was doing.  What the device holds for this
                              * handle is unknown from then on: every later compute call on it returns this code at once and
                              * only s2m_destroy is served (the map can be rebuilt on a new handle from the node's mirror) */
};

typedef struct s2m_engine s2m_engine;

/* Gates and constants of the path (defaults = reference values) plus engine knobs. */
typedef struct {
    float  plane_thr;        /* 0.1f   esti_plane threshold        laserMapping.cpp:863   */
    float  knn_d2_gate;      /* 5.0f   d2[4] gate                  laserMapping.cpp:853   */
    double s_gate;           /* 0.9                                laserMapping.cpp:870   */
    double res_gate;         /* 2.0                                laserMapping.cpp:889   */
    double laser_point_cov;  /* 0.0015 LASER_POINT_COV             laserMapping.cpp:76    */
    double conv_rot_deg;     /* 0.01                               laserMapping.cpp:1040  */
    double conv_pos_cm;      /* 0.015                              laserMapping.cpp:1040  */
    int32_t extrinsic_est_en;/* mapping/extrinsic_est_en           laserMapping.cpp:660   */
    int32_t max_iter;        /* mapping/max_iteration              laserMapping.cpp:656   */
    int32_t feat_threshold;  /* dynamic_effect_featurepoints_threshold, laserMapping.cpp:97 */
    float  cell_size;        /* voxel edge of the GPU map in metres; <= 0 = choose from density */
    int32_t device;          /* HIP device ordinal; < 0 = current device                   */
    int32_t far_point_bet;   /* 1 (default): a rematch pass whose predecessor in the same position (first pass of a
                              * scan / later pass) left no point to the far-point kernel runs without that kernel;
                              * the reduce kernel reports whether the bet held and a lost bet is repaired (results
                              * are identical either way).  0: never bet.  2: always bet (tests)             */
    int32_t device_loop;     /* 0 (default): the host-stepped loop -- after every pass the block lands in pinned host memory,
                              * the host runs the 24x24 update and launches the next pass.  1: s2m_iterated_update and
                              * s2m_iterated_update_batch keep the state on the device between the passes of a scan: the
                              * last workgroup of every pass applies the Kalman update (matrix-inversion-lemma form, one
                              * nc x nc solve), the convergence test and the rematch / exit judgement
                              * (laserMapping.cpp:899-918, 1012-1101), the kernels of the following iterations are already
                              * enqueued, the host reads one record at the end and updates the covariance.  Both forms
                              * evaluate the same update (poses agree to ~1e-12); measured on MI355X the device form is
                              * the slower one (C3 0.161 vs 0.145 ms/step, 22.8 k vs 25.9 k scans/s at 24 in flight:
                              * NOTEBOOK.md, round 4), hence opt-in.  Forms that sum blocks over ranks or handles
                              * (communicator, shared-memory exchange, _multi, _sharded) are always host-stepped. */
    int32_t wait_policy;     /* how the calling thread waits for the device (several times per frame: the block of every pass,
                              * the hand-backs of the map update).  0 (default) spin: poll with `pause` -- lowest latency, one
                              * core busy for the duration of the call.  1 yield: poll for wait_spin_us, then sched_yield()
                              * between polls.  2 sleep: poll for wait_spin_us, then nanosleep (50 us; 200 us after 5 ms)
                              * between polls -- the core is free while the device works.  The reference's loop never blocks
                              * on its map (laserMapping.cpp:726-731); a node whose CPU is busy wants 1 or 2.          */
    int32_t wait_timeout_ms; /* deadline of every such wait (default 10000).  A wait that passes it returns
                              * S2M_ERR_TIMEOUT; no entry point blocks for ever, whatever the device does.            */
    int32_t wait_spin_us;    /* policies 1 and 2: how long a wait polls before it gives the core away (default 40)   */
    int32_t layout_beside;   /* 1 (default): when the layout an in-place map update works on wears out -- the tail of the point
                              * array or the brick table's spare rows three quarters used, the points per occupied cell a factor
                              * two off what the cell size was chosen for -- a new layout of the whole map is built BESIDE the
                              * frames: snapshot, build on a worker thread and stream of the handle, the update calls that arrive
                              * meanwhile run again on the new map, swap between two updates (ikd-Tree's rebuild thread,
                              * ikd_Tree.cpp:192-203, 229-367).  0: the layout is renewed inside the update that hits the limit
                              * (a merge, a re-grid: milliseconds in that frame).  Same map, same ids either way.            */
} s2m_config;

int s2m_abi_version(void);
int s2m_config_default(s2m_config *cfg);
const char *s2m_strerror(int code);

/* Lifetime.  Replaces the globals of laserMapping.cpp:79-195 (ikdtree :164, Nearest_Points
 * :578, point_selected_surf :812, H_T_H/G :696). */
int s2m_create(const s2m_config *cfg, s2m_engine **out);
int s2m_destroy(s2m_engine *e);
const char *s2m_last_error(const s2m_engine *e);
/* What the handle is doing, for a watchdog that sees no progress (design; callable from ANOTHER thread while the owner is
 * inside an entry point -- it only reads): the entry point and step the owner is in, whether each of the handle's streams is
 * idle, the side thread's job flags, every pinned hand-back word against the sequence number the host expects, the waits
 * counted so far.  Writes a NUL-terminated line into buf (truncated to capacity). */
int s2m_debug_state(const s2m_engine *e, char *buf, int64_t capacity);
/* Fault injection for tests of the deadline path on a healthy device (design): from the `after`-th one on, the hand-backs of
 * `kind` are withheld -- 1: the side thread never finishes its job (s2m_scan_prefetch_raw / s2m_scan_prepare_raw), 2: the
 * kernel that hands device words to the host is not launched (map updates, voxel grid, scan hand-over), 3: the reduce kernel
 * does not publish its block (every pass), 4: the hand-backs of the worker that lays the map out beside the frames (the frames
 * go on; the layout's own deadline reports it).  0 disarms.  The environment variable S2M_TEST_STALL=worker|mail|reduce[:after]
 * arms the same hook when a handle is created (for callers that are not tests' own code, e.g. tools/replay_node). */
int s2m_test_stall(s2m_engine *e, int32_t kind, int64_t after);
/* Change gates between scans (e.g. feat_threshold, laserMapping.cpp:427-430). cell_size/device
 * are fixed at creation. */
int s2m_set_config(s2m_engine *e, const s2m_config *cfg);
/* Run the handle's work on a caller stream (hipStream_t passed as void*; NULL = own stream). */
int s2m_set_stream(s2m_engine *e, void *hip_stream);

/* Map seed: ikdtree.Build(feats_down_world->points), laserMapping.cpp:784-790
 * (KD_TREE::Build, ikd-Tree/ikd_Tree.cpp:408-423).  on_device != 0: xyz is a device pointer. */
int s2m_map_build(s2m_engine *e, const float *xyz, int64_t stride_floats, int64_t m, int on_device);
/* Several handles, one map (design, not reference: the reference has one scan in flight against one
 * global ikdtree, laserMapping.cpp:164): `e` searches the map built in `owner` instead of holding a copy,
 * so K scans can be registered concurrently -- K handles, K streams, K host threads -- against one
 * HBM-resident map.  The borrower never modifies the map (S2M_ERR_STATE from the update entry points);
 * `owner` must stay alive, on the same device, and must not rebuild or update its map while a borrower
 * has a pass in flight; after an update of the owner's map the borrower calls s2m_map_share again to see it (an
 * update writes the other half of the owner's double buffers).  s2m_map_build on `e` ends the loan. */
int s2m_map_share(s2m_engine *e, const s2m_engine *owner);
int s2m_map_size(const s2m_engine *e, int64_t *m);             /* ikdtree.validnum(), :794 */
/* info[0..7]: cell size, origin xyz, bricks, top-level entries, occupied cells, mean pts/cell */
int s2m_map_info(const s2m_engine *e, double info[8]);

/* ---- incremental map maintenance (each call ends with a new GPU map -- the survivors in index order followed
 * by the added points, merged into the sorted arrays when the new points fit the current grid, rebuilt
 * otherwise; neighbour indices of earlier passes become invalid) ------------------------------------
 * ikdtree.Add_Points(points, downsample_on) (ikd-Tree/ikd_Tree.cpp:477-573): with downsample_on the
 * voxel [min, max) of edge downsample_size around each new point keeps only the point closest to
 * its centre (ties: a new point beats an old one, the later of two new ones wins).
 * *n_added = voxels rewritten (downsample_on) or n.
 * A coordinate the grid cannot address at its cell size (beyond +-2^20 cells of the origin AND of any origin that would still
 * hold the rest of the map: thousands of kilometres -- corrupt input, ikd-Tree would take it) makes the update fail with
 * S2M_ERR_CAPACITY; the map is then exactly as it was before the call and the handle goes on working (the same holds for
 * s2m_map_incremental).  Non-finite coordinates are stored in a clamped cell, where they are nobody's neighbour, or refused
 * the same way. */
int s2m_map_add(s2m_engine *e, const float *xyz, int64_t stride_floats, int64_t n, int downsample_on,
                float downsample_size, int on_device, int64_t *n_added);
/* ikdtree.Delete_Point_Boxes(cub_needrm) (ikd_Tree.cpp:631-658; lasermap_fov_segment,
 * laserMapping.cpp:313-369).  boxes: HOST array n x 6 = {min xyz, max xyz}; a point is removed
 * when min <= p < max on every axis. */
int s2m_map_delete_boxes(s2m_engine *e, const float *boxes, int64_t n, int64_t *n_deleted);
/* lasermap_fov_segment() (laserMapping.cpp:304-369): keeps the local-map cube of edge cube_len
 * (mapping/cube_side_length) around the LiDAR; when the LiDAR comes within MOV_THRESHOLD * DET_RANGE
 * = 1.5 * 300 m of a face the cube is shifted by mov_dist (:345) and the slabs that fall out are
 * removed with Delete_Point_Boxes (:368).  pos_lid = pos_end + rot_end * T_L_I (:753).  The first call
 * only initialises the cube (:320-328).  Outputs (optional): local_map[6] = the cube after the call
 * (min xyz, max xyz), *n_boxes = slabs removed (0..3), *n_deleted = kdtree_delete_counter. */
int s2m_fov_segment(s2m_engine *e, const double pos_lid[3], double cube_len, float local_map[6],
                    int32_t *n_boxes, int64_t *n_deleted);
/* forget the cube (Localmap_Initialized = false), e.g. when a rosbag loops back */
int s2m_fov_reset(s2m_engine *e);

/* map_incremental() (laserMapping.cpp:582-630) for the current scan: world points from `state`,
 * the add / no-need-downsample decision from the Nearest_Points of the last rematch pass, then
 * Add_Points(PointToAdd, true) and Add_Points(PointNoNeedDownsample, false) (:627-628).
 * filter_size_map = mapping/filter_size_map (feat.yaml: 0.5), also the ikd-Tree downsample size
 * (:784).  ekf_inited = the reference's flg_EKF_inited at the call (:593): with a zero value no point is
 * classified and all go to PointToAdd (:623).  In the shipped node the flag is 1 whenever map_incremental
 * runs: it is set every frame (:762, INIT_TIME == 0), only the EKF_stop branch clears it (:1062), and on
 * such a scan the node does not call map_incremental at all (`if (!EKF_stop_flg)`, :1165) -- a faithful
 * caller does the same (s2m_iter_log.ekf_stop) and passes 1 otherwise.
 * Outputs: the sizes of the two lists (add_point_size = their sum, :629). */
int s2m_map_incremental(s2m_engine *e, const double state[S2M_STATE_DOUBLES], double filter_size_map,
                        int32_t ekf_inited, int64_t *n_to_add, int64_t *n_no_downsample);
/* ikdtree.flatten(Root_Node, PCL_Storage) (laserMapping.cpp:1170-1175): the current map points,
 * packed xyz, in the caller's index order (the order the indices of s2m_get_neighbors refer to: the array given to
 * s2m_map_build; after updates the survivors in index order followed by the added points). */
int s2m_map_get_points(s2m_engine *e, float *xyz, int64_t capacity_points, int64_t *m);
/* The map's point ids in the order of s2m_map_get_points (ascending): the index into the array handed to s2m_map_build for
 * the points of a build, consecutive numbers for the points added since; an id never changes while the map is only updated
 * (a rebuild -- s2m_map_update_stats, stats[1] -- numbers the points anew). */
int s2m_map_get_ids(s2m_engine *e, uint32_t *ids, int64_t capacity_points, int64_t *m);
/* Map publishing proportional to the change (design; the reference flattens and publishes the whole map every frame,
 * laserMapping.cpp:1170-1175, 1229-1235 -- O(M) per frame): what the updates since the previous call did to the map, in
 * sequence.  Three kinds of entry: points ADDED (xyz + id), points REMOVED one by one (the voxel rule of Add_Points: xyz + id,
 * so that a follower can find them by place and then by id), and BOX DELETES (Delete_Point_Boxes -- the field-of-view trim
 * removes millions of points: the follower gets the boxes, not the points, and applies min <= p < max per axis, float
 * compares, to what it holds).  Order of application: box k applies after added[0 .. box_after_added[k]) and
 * removed[0 .. box_after_removed[k]) and before the entries behind them; within a stretch between two boxes additions first,
 * then removals (a point that came and went is in both lists).
 * The first call (*token = 0) and every call whose token is not the one the previous call returned -- a rebuild in between,
 * more changes than the log holds (2^18 entries or four scans' worth, 8 box deletes) -- answer resync = 1 with no changes:
 * fetch s2m_map_get_points + s2m_map_get_ids and go on from the token returned.  Otherwise the arrays receive n_added /
 * n_removed / n_boxes entries (S2M_ERR_CAPACITY if any is too small: the counts are set, the changes kept for the next call).
 * lag = 0: everything up to now; the call waits for the device (one hand-back).  lag = 1: everything up to the PREVIOUS call
 * -- that report left for pinned host memory a frame ago and has landed, so the call waits for nothing, and what has happened
 * since leaves now: the follower is one call behind the map and never makes the frame wait.  One follower per handle.
 * Cost: the removed points are collected from the bricks an update touches, the added points are the update's staging list
 * -- nothing map-sized runs, nothing map-sized crosses PCIe. */
typedef struct {
    float    *added_xyz;         /* in: capacity_added x 3                                   */
    uint32_t *added_ids;         /* in: capacity_added                                       */
    int64_t   capacity_added;
    float    *removed_xyz;       /* in: capacity_removed x 3, or NULL (ids only)             */
    uint32_t *removed_ids;       /* in: capacity_removed                                     */
    int64_t   capacity_removed;
    float    *boxes;             /* in: capacity_boxes x 6 = {min xyz, max xyz}              */
    int64_t  *box_after_added;   /* in: capacity_boxes                                       */
    int64_t  *box_after_removed; /* in: capacity_boxes                                       */
    int64_t   capacity_boxes;
    int64_t   n_added, n_removed, n_boxes;   /* out */
    int32_t   resync;            /* out */
    int32_t   lag;               /* in: 0 or 1 */
} s2m_map_changes;
int s2m_map_get_changes(s2m_engine *e, uint64_t *token, s2m_map_changes *changes);
/* How the last map_add / map_delete_boxes / fov_segment / map_incremental produced the new map (design, not
 * reference): *merged = 1 when the update was merged into the sorted arrays of the current grid, 0 when the grid
 * was rebuilt (a density drift, an empty map; see s2m_map_update_stats). */
int s2m_map_last_update(const s2m_engine *e, int32_t *merged);
/* Running counts for latency diagnosis (design, not reference; the reference hides the same events behind ikd-Tree's
 * rebuild thread, ikd_Tree.cpp:192-203, 229-367): stats[0] = updates merged into the grid, stats[1] = updates that
 * rebuilt it (an empty map, ids or buffers exhausted, a point beyond the representable range of cells, a density drift --
 * never growth as such), stats[2] = rebuilds that also chose a new cell size (density drift), stats[3] = device buffer
 * (re)allocations made by map builds and updates so far (process-wide), stats[4] = times the top-level array was re-laid
 * because the box of bricks in use had outgrown or left its window (a few thousand entries; no point moves), stats[5] =
 * bricks rewritten in place through the large staging form (more than 2 048 points), stats[6..9] = updates that could not
 * stay in place (and were merged) because of: a point beyond the representable cell range, no spare table rows for the
 * bricks to open, a brick too large to stage (more than 6 144 points), the tail of the point array exhausted; stats[10] =
 * layouts of the whole map that were produced BESIDE the frames (on the handle's side stream, from a snapshot, the updates
 * that arrived meanwhile applied again, swapped in between two frames: ikd-Tree's rebuild thread, ikd_Tree.cpp:192-203,
 * 229-367) instead of inside an update, stats[11] = those of them that also chose a new cell size. */
int s2m_map_update_stats(const s2m_engine *e, int64_t stats[12]);
/* How many of the merged updates (s2m_map_update_stats, stats[0]) were applied IN PLACE: only the bricks the update touched
 * were rewritten where they stand -- possible when each of them still fits the stretch of the point array it owns and no new
 * point opens a brick; cost proportional to the update, not to the map (ikd-Tree inserts per point in O(log M),
 * ikd_Tree.cpp:477-573).  Every other update re-lays the whole map out (merge) or rebuilds the grid.  (Design, not reference.) */
int s2m_map_inplace_updates(const s2m_engine *e, int64_t *n);
/* The engine's internal point order (design, not reference): order[j] = the caller's index of the point at sorted
 * position j.  Positions are ordered by (brick of 8x8x8 cells, cell within the brick, caller index).  The cell of a
 * coordinate v is floor(((double)v - (double)origin) * (double)(1.0f / cell_size)) -- float operands, double arithmetic --
 * with the origin and cell size of s2m_map_info, a SIGNED integer per axis: the origin is fixed by the first build of a map
 * and kept by every update, and the map grows around it in any direction.  A brick's coordinates are (cell >> 3) per axis
 * (arithmetic shift), bricks are ordered by (z, y, x) lexicographically on these signed coordinates, the cells inside a brick
 * by (cell & 7) in (z, y, x) likewise -- an order that does not depend on the box the map occupies, the same after a full
 * build and after a merged update.  An in-place update keeps this order INSIDE every brick; a brick that the update opens,
 * or that outgrows the stretch of positions it owns, is handed a stretch behind the key-ordered part of the array (the
 * price of an update whose cost follows the scan and not the map), so between two merges a brick is one contiguous run of
 * positions whose place among the other bricks' runs is history, not key order.  *m = the extent of the position
 * range; after in-place updates it may exceed s2m_map_size, and order[j] = 0xffffffff marks a position that holds no point
 * (a hole at the end of a rewritten brick).  Candidates tied at exactly the same float squared distance are ranked by
 * this position (tests hand it to the oracle as the tie order). */
int s2m_map_get_order(s2m_engine *e, uint32_t *order, int64_t capacity_points, int64_t *m);
/* bricks[6] = the box of bricks the map may occupy at the moment, in brick coordinates (lo xyz, hi xyz, inclusive; hi < lo:
 * empty map).  Informative: it is conservative (it grows with the map and is tightened only when it outgrows the window of
 * the top-level array), and the order of s2m_map_get_order does not depend on it. */
int s2m_map_grid(const s2m_engine *e, int32_t bricks[6]);

/* The down-sampled body-frame scan feats_down (laserMapping.cpp:775-778).  Resets the per-scan
 * state: point_selected_surf := true (:812), Nearest_Points cleared (:810).  Coordinates must be finite -- the
 * reference's clouds are is_dense (its preprocessing drops invalid returns); an infinite coordinate makes the voxel
 * grid of the *_downsampled / *_from_raw entry points report S2M_ERR_CAPACITY. */
int s2m_scan_set(s2m_engine *e, const float *xyz, int64_t stride_floats, int64_t n, int on_device);

/* downSizeFilterSurf.filter(*feats_down) followed by the scan hand-over (laserMapping.cpp:775-778):
 * pcl::VoxelGrid with leaf = mapping/filter_size_surf (one centroid per occupied voxel, ascending
 * voxel index), result kept on the device as the current scan (same resets as s2m_scan_set).
 * *n_out = feats_down_size.  S2M_ERR_CAPACITY when the leaf is too small for the cloud's extent
 * (PCL's int32 voxel-index overflow check). */
int s2m_scan_set_downsampled(s2m_engine *e, const float *xyz, int64_t stride_floats, int64_t n, float leaf,
                             int on_device, int64_t *n_out);
/* The current scan (feats_down), packed xyz, for publishers and callers that keep a host copy. */
int s2m_scan_get(s2m_engine *e, float *xyz, int64_t capacity_points, int64_t *n);

/* One entry of IMUpose (eskf_lio::Pose6D as filled by set_pose6d, common_lib.h:248-265;
 * IMU_Processing.hpp:224, 310): the IMU state at one IMU sample, offset_time relative to the scan start. */
typedef struct {
    double offset_time;
    double acc[3], gyr[3], vel[3], pos[3];
    double rot[9]; /* row-major */
} s2m_imu_pose;

/* Motion compensation of a raw scan: the backward-propagation loop of ImuProcess::UndistortPcl
 * (IMU_Processing.hpp:333-370) plus the time sort in front of it (:215-216).  The sequential IMU
 * forward/covariance propagation (:226-308) stays with the caller and provides `poses` (ascending
 * offset_time) and state_end (rot_end, pos_end, R_L_I, T_L_I after :319-323).
 * Each point record is `stride_floats` floats starting with x, y, z; its offset time is
 * rec[time_off_a] * rec[time_off_b] (normal_x * normal_z: 4 and 6 in pcl::PointXYZINormal) or
 * rec[time_off_a] alone when time_off_b < 0.  sort_by_time != 0 returns the points ordered by time like
 * the reference (ties keep input order).  out_xyz: n x 3 packed floats (host, or device when
 * on_device applies to input and output alike); perm (optional, host, n) receives the input index of
 * each output point. */
int s2m_undistort(s2m_engine *e, const float *points, int64_t stride_floats, int64_t n, int32_t time_off_a,
                  int32_t time_off_b, const s2m_imu_pose *poses, int32_t n_poses,
                  const double state_end[S2M_STATE_DOUBLES], int sort_by_time, int on_device, float *out_xyz,
                  uint32_t *perm);
/* The NEXT sweep's records on their way while the current one is registered (the reference's node receives the next
 * message while it works, laserMapping.cpp:726-731): starts the host-to-device copy of `points` (host memory, n records
 * of stride_floats floats) on a side stream, driven by a worker thread of the handle, and returns at once.  A following
 * s2m_scan_set_from_raw with the SAME pointer, stride and n uses that copy instead of copying itself (any other call
 * ignores it).  With time_off_a >= 0 (the time fields of s2m_undistort) the time order of the records
 * (IMU_Processing.hpp:215-216: it needs no pose) is computed behind the copy and reused by the call that consumes the
 * records; time_off_a < 0: the copy only.  The caller keeps the buffer alive and unchanged until then.  A copy that is
 * not consumed is dropped by the next s2m_scan_set* call of any kind (a node that recycles its host buffers could
 * otherwise present new records under the address of a sweep that was announced and then skipped); points == NULL
 * cancels explicitly.  (Design, not reference.) */
int s2m_scan_prefetch_raw(s2m_engine *e, const float *points, int64_t stride_floats, int64_t n, int32_t time_off_a,
                          int32_t time_off_b);
/* The whole front half of the NEXT frame while the map update of the current one runs: s2m_scan_set_from_raw's work (copy,
 * undistortion, voxel grid) on the handle's side stream and worker thread, into spare scan arrays; returns at once.  A
 * following s2m_scan_set_from_raw with the SAME arguments (pointer, layout, poses and state_end by value, leaf; host
 * input) swaps the prepared scan in instead of computing it; any other call ignores it and computes as usual -- the result
 * is the same scan either way.  Meant for a node that replays or catches up: the reference's loop has the next message's
 * IMU poses as soon as the current update has produced the state they are propagated from (IMU_Processing.hpp:246-330),
 * i.e. before map_incremental (laserMapping.cpp:1134) -- call it there.  The caller keeps `points` alive and unchanged
 * until the matching s2m_scan_set_from_raw returns.  A sweep larger than the current scan's arrays is not prepared (the
 * synchronous call handles it).  Records that s2m_scan_prefetch_raw has already brought over (same buffer, not yet
 * consumed) are not copied again: prefetch when the sweep arrives, prepare when its poses exist.  (Design, not reference.) */
int s2m_scan_prepare_raw(s2m_engine *e, const float *points, int64_t stride_floats, int64_t n, int32_t time_off_a,
                         int32_t time_off_b, const s2m_imu_pose *poses, int32_t n_poses,
                         const double state_end[S2M_STATE_DOUBLES], float leaf);
/* The node's front half of a frame kept on the device: undistort (time-sorted) -> VoxelGrid(leaf)
 * -> current scan (IMU_Processing.hpp:333-370, laserMapping.cpp:775-778).  leaf <= 0 skips the
 * down-sampling.  *n_out = feats_down_size. */
int s2m_scan_set_from_raw(s2m_engine *e, const float *points, int64_t stride_floats, int64_t n, int32_t time_off_a,
                          int32_t time_off_b, const s2m_imu_pose *poses, int32_t n_poses,
                          const double state_end[S2M_STATE_DOUBLES], float leaf, int on_device, int64_t *n_out);

/* Output of one residual/Jacobian pass. */
typedef struct {
    double  HtH[144];        /* Hsub^T * Hsub, row-major 12x12        laserMapping.cpp:1015 */
    double  Htz[12];         /* Hsub^T * meas_vec                                           */
    int32_t effct_feat_num;  /*                                       laserMapping.cpp:885-895 */
    int32_t rematch;         /* echo                                                          */
    double  total_residual;  /*                                       laserMapping.cpp:884-893 */
} s2m_pass_out;

/* One pass of laserMapping.cpp:829-979 fused with the Hsub^T*Hsub / Hsub^T*meas_vec
 * contraction of :1015-1032.  rematch = (iterCount == 0 || rematch_en) (:847). */
int s2m_residual_pass(s2m_engine *e, const double state[S2M_STATE_DOUBLES], int rematch,
                      s2m_pass_out *out);
/* Same pass, result left on the device for a collective: d_block is a DEVICE pointer to
 * S2M_BLOCK_DOUBLES doubles laid out HtH[144] Htz[12] effct total_res far_points 0 (effct as a double; far_points =
 * scan points this pass handed to the far-point kernel, a diagnostic).
 * No host synchronisation; ordered on the handle's stream. */
int s2m_residual_pass_device(s2m_engine *e, const double state[S2M_STATE_DOUBLES], int rematch,
                             double *d_block);

/* Dense rows of the last pass: Hsub (m x 12, column order [rot pos rot_LI trans_LI]) and
 * meas_vec (m), in scan index order, plus the scan index of each row (laserCloudOri order).
 * laserMapping.cpp:942-979.  Any pointer may be NULL.  *m_out = effct_feat_num. */
int s2m_get_rows(s2m_engine *e, double *h_x, double *h, int32_t *scan_index, int64_t capacity,
                 int64_t *m_out);
/* Per-point state after the last pass (any pointer may be NULL):
 * selected[n] point_selected_surf (:812,:857-873); effective[n] (:889); plane[n*4] = (n,d) of
 * esti_plane (coeffSel_tmpt xyz + fit offset); pd2[n] (coeffSel_tmpt intensity, :877). */
int s2m_get_point_state(s2m_engine *e, uint8_t *selected, uint8_t *effective, float *plane,
                        float *pd2);
/* Nearest_Points of the last rematch pass (:845-850): idx[n*5] indexes the array given to
 * s2m_map_build (-1 = missing), d2[n*5] ascending (INFINITY = missing); exact ties in d2 are ordered by the engine's
 * sorted position (s2m_map_get_order).  The per-iteration search stops at the d2 <= 5 gate (:853): a list is the exact
 * answer of ikdtree.Nearest_Search for every neighbour within the gate radius and ends there -- a scan point whose
 * surroundings are emptier than that has fewer than five entries until s2m_complete_neighbors has run. */
int s2m_get_neighbors(s2m_engine *e, int32_t *idx, float *d2);
/* Completes the lists that ended short at the gate to the unbounded result of ikdtree.Nearest_Search (max_dist =
 * INFINITY, ikd-Tree/ikd_Tree.cpp:425): five nearest map points however far away (fewer only when the map holds
 * fewer).  map_incremental reads points_near[0] of exactly such points when the sensor enters new territory
 * (laserMapping.cpp:593-607); s2m_map_incremental therefore calls this first.  Cold path: a scan inside the mapped
 * area has nothing to complete.  *n_completed (optional) = lists that were short. */
int s2m_complete_neighbors(s2m_engine *e, int64_t *n_completed);

/* Kalman update of laserMapping.cpp:1012-1046 from the normal block:
 * K_1 = (H_T_H + (P/R)^-1)^-1; solution = K z + vec - K H vec[0:12]; x [+]= solution.
 * x is updated in place; *converged = flg_EKF_converged (:1040). */
int s2m_eskf_update(s2m_engine *e, double x[S2M_STATE_DOUBLES],
                    const double x_prop[S2M_STATE_DOUBLES], const double P[S2M_DIM * S2M_DIM],
                    const double HtH[144], const double Htz[12], double solution[S2M_DIM],
                    int32_t *converged);
/* P <- (I - K H (+) 0) P with K, H of the last s2m_eskf_update (laserMapping.cpp:1084-1085). */
int s2m_cov_update(s2m_engine *e, double P[S2M_DIM * S2M_DIM]);

/* Per-iteration log of s2m_iterated_update (what Log/mat_out.txt records, :936-937). */
typedef struct {
    int32_t iters;           /* iterations executed                                          */
    int32_t rematch_passes;  /* passes that ran the kNN                                      */
    int32_t converged;       /* flg_EKF_converged at exit                                    */
    int32_t ekf_stop;        /* EKF_stop_flg at exit (:907-918); state left unchanged if set */
    int32_t effct[64];       /* effct_feat_num per iteration                                 */
    int32_t rematch[64];
    int32_t conv[64];
    double  total_residual[64];
    double  solution[64][S2M_DIM];
} s2m_iter_log;

/* The whole iterated update of one scan, laserMapping.cpp:820-1102: passes, degeneracy queue
 * (:899-918, kept in the handle across scans), update, rematch judgement (:1070-1076), exit and
 * covariance update (:1079-1101).  x, P updated in place.  max_iter <= 64. */
int s2m_iterated_update(s2m_engine *e, double x[S2M_STATE_DOUBLES],
                        const double x_prop[S2M_STATE_DOUBLES], double P[S2M_DIM * S2M_DIM],
                        s2m_iter_log *log);

/* The same update for k scans at once on ONE GPU, driven by one host thread (BASELINE configs[4] --
 * batched odometry -- on a single device): handles[i] holds scan i (its own s2m_scan_set; the map may be
 * shared through s2m_map_share), x / x_prop are k x 36 doubles, P is k x 576, logs k records (optional).
 * Every scan runs exactly the loop of s2m_iterated_update (results are identical); the host goes round the
 * handles and serves whichever pass has finished, so the scans fill each other's latency gaps.  Handles
 * must be distinct, on one device, without a communicator. */
int s2m_iterated_update_batch(s2m_engine *const *handles, int32_t k, double *x, const double *x_prop,
                              double *P, s2m_iter_log *logs);
/* ONE scan split over n handles, driven by one host thread, without a collective library (design, not reference;
 * SURVEY.md 8e "single-process peer-copy gather"): handles[i] holds shard i of the scan (contiguous ranges, handle
 * order = index order) and the map -- on n devices of one node, or several shards per device.  Every pass is launched
 * on all handles, each handle's block lands in its own pinned host page, the host sums the n blocks in handle order
 * and runs ONE fp64 update (degeneracy queue and Kalman work area of handles[0]).  x, P as in s2m_iterated_update.
 * The sums are binary trees -- over the point index inside the reduce kernel, over the handle index on the host -- so
 * when the shards are aligned power-of-two pieces of the scan (n a power of two, shard sizes 64 * 2^k: shard_range of a
 * 65,536-point scan over 2, 4 or 8 handles) the result is BIT-IDENTICAL to s2m_iterated_update on the whole scan,
 * whatever n; otherwise it agrees to summation order (~1e-13 in the pose) and is deterministic for a given n. */
int s2m_iterated_update_multi(s2m_engine *const *handles, int32_t n, double x[S2M_STATE_DOUBLES],
                              const double x_prop[S2M_STATE_DOUBLES], double P[S2M_DIM * S2M_DIM], s2m_iter_log *log);
/* Multi-GPU form: this handle holds a contiguous shard of the scan's points and the whole map.
 * After every pass the shard's block (S2M_BLOCK_DOUBLES doubles, layout as in
 * s2m_residual_pass_device) is in d_block, a DEVICE buffer the caller owns; reduce(user) must
 * sum it in place across all ranks, ordered after prior work on the handle's stream (e.g. an
 * RCCL all-reduce enqueued on that stream) and return 0.  Every rank then runs the identical
 * fp64 update, so states stay bit-identical without a broadcast.  reduce == NULL degenerates to
 * s2m_iterated_update.  (The reference has no counterpart: it is single-threaded, :827-828.) */
typedef int (*s2m_allreduce_fn)(void *user);
int s2m_iterated_update_sharded(s2m_engine *e, double x[S2M_STATE_DOUBLES],
                                const double x_prop[S2M_STATE_DOUBLES], double P[S2M_DIM * S2M_DIM],
                                s2m_iter_log *log, double *d_block, s2m_allreduce_fn reduce,
                                void *user);
/* Built-in collective for the multi-GPU form: an RCCL communicator owned by the handle.  Rank 0 calls
 * s2m_comm_unique_id and ships the 128 bytes to the other ranks by any means (bench.py uses a
 * torch.distributed broadcast); every rank then calls s2m_comm_init.  While a communicator is attached
 * s2m_iterated_update treats the handle's scan as this rank's shard: after every pass the block is
 * summed with ncclAllReduce(sum, ncclDouble, 160) on the handle's stream, straight from the C++ loop.
 * RCCL is loaded at run time; without it these calls return S2M_ERR_HIP and nothing else is affected. */
#define S2M_COMM_ID_BYTES 128
int s2m_comm_unique_id(uint8_t id[S2M_COMM_ID_BYTES]);
int s2m_comm_init(s2m_engine *e, const uint8_t id[S2M_COMM_ID_BYTES], int32_t nranks, int32_t rank);
/* The same for the processes of ONE node without a collective library: the exchange goes through a POSIX shared-
 * memory segment `name` ("/something", unique per job, the same string on every rank).  Every rank's reduce kernel
 * publishes its block into that rank's pinned page exactly as in the single-GPU loop; the rank's host thread copies the
 * 1.3 KB into its slot of the segment and reads everybody's -- no collective launch, no publish kernel, no GPU
 * peer access.  The blocks are summed pairwise over the rank index on every rank alike, so aligned power-of-two shards
 * give the unsplit scan's result bit for bit (as s2m_iterated_update_multi does inside one process).  All ranks must
 * have called this before the first of them enters s2m_iterated_update. */
int s2m_comm_init_shm(s2m_engine *e, const char *name, int32_t nranks, int32_t rank);
int s2m_comm_destroy(s2m_engine *e);

/* Degeneracy queue access (effct_feat_numQueue, laserMapping.cpp:193). */
int s2m_feat_queue_get(const s2m_engine *e, int32_t q[S2M_FEAT_QUEUE], int32_t *len);
int s2m_feat_queue_set(s2m_engine *e, const int32_t *q, int32_t len);

/* h_share_model-shaped adapter (the IKFoM / FAST-LIO2 callback convention named by the north
 * star; DaLiTI itself has no such callback): fills the dense rows for the given state so an
 * esekfom-style update_iterated_dyn_share loop can consume them.  Column order is the
 * reference's [rot pos rot_LI trans_LI] (laserMapping.cpp:971), NOT FAST-LIO2's [pos rot ...].
 * The caller allocates h_x (capacity x 12) and h (capacity). */
typedef struct {
    int32_t valid;      /* out: 0 when effct_feat_num < 1 (FAST-LIO2 semantics)       */
    int32_t converge;   /* in: caller's flag; != 0 requests a rematch like :847       */
    double *h_x;        /* out: rows x 12                                             */
    double *h;          /* out: rows; h = -pd2 like meas_vec (:977)                   */
    int64_t capacity;   /* in                                                         */
    int64_t rows;       /* out: effct_feat_num                                        */
    double  total_residual; /* out */
} s2m_dyn_share;
int s2m_h_share_model(s2m_engine *e, const double state[S2M_STATE_DOUBLES], int first_iteration,
                      s2m_dyn_share *ekfom_data);

/* Timing of the last timed pass in milliseconds, measured with HIP events on the handle's stream:
 * ms[0] = the search kernels (exact 5-NN; 0 on a reuse pass), ms[1] = the reduce kernel ([gate + plane
 * fit,] residual, Jacobian row, normal block), ms[2] = whole pass.  s2m_set_timing(e, 1 or 2) times
 * every pass (three event records each); s2m_set_timing(e, n > 2) times every n-th pass only (sampling:
 * choose n coprime to the passes per scan so that every kind of pass is visited); 0 switches it off. */
int s2m_set_timing(s2m_engine *e, int enabled);
int s2m_get_timing(const s2m_engine *e, double ms[3]);
/* The bet of s2m_config.far_point_bet, counted (design, not reference): stats[0] = rematch passes that ran without the
 * far-point kernel and were right, stats[1] = passes whose bet was lost (far-point kernel + reduce kernel run again).
 * The rule: a pass bets only when the last pass in the same position (first pass of a scan / a later one) reported no
 * far point at all; a lost bet therefore stops the betting in that position until a full pass reports zero again. */
int s2m_bet_stats(const s2m_engine *e, int64_t stats[2]);

/* Accumulated over the timed passes since the last s2m_set_timing call: stats[0] = sum of search-kernel
 * ms over rematch passes, stats[1] = their count, stats[2] = sum of the reduce kernel's ms on rematch
 * passes (gate + plane fit included), stats[3] = count, stats[4] = sum of the reduce kernel's ms on reuse
 * passes, stats[5] = count. */
int s2m_get_timing_stats(const s2m_engine *e, double stats[6]);

#ifdef __cplusplus
}
#endif
#endif
