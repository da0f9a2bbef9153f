"""ctypes binding of include/daliti_s2m.h (one Python method per C entry point)."""
import collections
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("S2M_LIB") or os.path.join(_HERE, "_lib", "libdaliti_s2m.so")  # S2M_LIB: A/B builds of the same ABI

K = 5
DIM = 24
STATE_DOUBLES = 36
BLOCK_DOUBLES = 160
FEAT_QUEUE = 10
MAX_LOG = 64

# every symbol include/daliti_s2m.h declares
ABI_SYMBOLS = [
    "s2m_abi_version", "s2m_config_default", "s2m_strerror", "s2m_create", "s2m_destroy",
    "s2m_last_error", "s2m_debug_state", "s2m_test_stall", "s2m_set_config", "s2m_set_stream", "s2m_map_build", "s2m_map_size",
    "s2m_map_info", "s2m_map_last_update", "s2m_map_share", "s2m_map_add", "s2m_map_delete_boxes", "s2m_map_incremental", "s2m_map_get_points",
    "s2m_fov_segment", "s2m_fov_reset",
    "s2m_scan_set", "s2m_scan_set_downsampled", "s2m_scan_get", "s2m_undistort", "s2m_scan_set_from_raw", "s2m_scan_prefetch_raw", "s2m_scan_prepare_raw", "s2m_residual_pass", "s2m_residual_pass_device", "s2m_get_rows",
    "s2m_get_point_state", "s2m_get_neighbors", "s2m_eskf_update", "s2m_cov_update",
    "s2m_iterated_update", "s2m_iterated_update_batch", "s2m_iterated_update_multi", "s2m_iterated_update_sharded",
    "s2m_complete_neighbors", "s2m_map_get_order", "s2m_map_get_ids", "s2m_map_get_changes", "s2m_map_grid", "s2m_map_update_stats", "s2m_map_inplace_updates", "s2m_comm_unique_id", "s2m_comm_init", "s2m_comm_init_shm", "s2m_comm_destroy", "s2m_feat_queue_get", "s2m_feat_queue_set", "s2m_h_share_model",
    "s2m_set_timing", "s2m_get_timing", "s2m_get_timing_stats", "s2m_bet_stats",
]


class S2MError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("s2m error %d: %s" % (code, msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [("plane_thr", C.c_float), ("knn_d2_gate", C.c_float), ("s_gate", C.c_double),
                ("res_gate", C.c_double), ("laser_point_cov", C.c_double),
                ("conv_rot_deg", C.c_double), ("conv_pos_cm", C.c_double),
                ("extrinsic_est_en", C.c_int32), ("max_iter", C.c_int32),
                ("feat_threshold", C.c_int32), ("cell_size", C.c_float), ("device", C.c_int32),
                ("far_point_bet", C.c_int32), ("device_loop", C.c_int32), ("wait_policy", C.c_int32),
                ("wait_timeout_ms", C.c_int32), ("wait_spin_us", C.c_int32), ("layout_beside", C.c_int32)]


class PassOut(C.Structure):
    _fields_ = [("HtH", C.c_double * 144), ("Htz", C.c_double * 12), ("effct_feat_num", C.c_int32),
                ("rematch", C.c_int32), ("total_residual", C.c_double)]


class IterLog(C.Structure):
    _fields_ = [("iters", C.c_int32), ("rematch_passes", C.c_int32), ("converged", C.c_int32),
                ("ekf_stop", C.c_int32), ("effct", C.c_int32 * MAX_LOG), ("rematch", C.c_int32 * MAX_LOG),
                ("conv", C.c_int32 * MAX_LOG), ("total_residual", C.c_double * MAX_LOG),
                ("solution", (C.c_double * DIM) * MAX_LOG)]


class DynShare(C.Structure):
    _fields_ = [("valid", C.c_int32), ("converge", C.c_int32), ("h_x", C.c_void_p), ("h", C.c_void_p),
                ("capacity", C.c_int64), ("rows", C.c_int64), ("total_residual", C.c_double)]


class MapChangesC(C.Structure):
    _fields_ = [("added_xyz", C.c_void_p), ("added_ids", C.c_void_p), ("capacity_added", C.c_int64),
                ("removed_xyz", C.c_void_p), ("removed_ids", C.c_void_p), ("capacity_removed", C.c_int64),
                ("boxes", C.c_void_p), ("box_after_added", C.c_void_p), ("box_after_removed", C.c_void_p), ("capacity_boxes", C.c_int64),
                ("n_added", C.c_int64), ("n_removed", C.c_int64), ("n_boxes", C.c_int64), ("resync", C.c_int32), ("lag", C.c_int32)]


MapChanges = collections.namedtuple("MapChanges", "token resync add_xyz add_ids rem_xyz rem_ids boxes box_after_added box_after_removed")


def apply_map_changes(ids, xyz, ch):
    """What a follower does with a MapChanges report (include/daliti_s2m_mirror.hpp in numpy, for tests): returns the new
    (ids, xyz), unordered."""
    a0 = r0 = 0
    nb = len(ch.boxes)
    for k in range(nb + 1):
        a1 = int(ch.box_after_added[k]) if k < nb else len(ch.add_ids)
        r1 = int(ch.box_after_removed[k]) if k < nb else len(ch.rem_ids)
        ids = np.concatenate([ids, ch.add_ids[a0:a1]])
        xyz = np.concatenate([xyz, ch.add_xyz[a0:a1]])
        gone = np.isin(ids, ch.rem_ids[r0:r1])
        assert gone.sum() == r1 - r0, "a removed id is not in the mirror"
        ids, xyz = ids[~gone], xyz[~gone]
        if k < nb:
            lo, hi = ch.boxes[k][:3], ch.boxes[k][3:]
            inside = ((xyz >= lo) & (xyz < hi)).all(axis=1)
            ids, xyz = ids[~inside], xyz[~inside]
        a0, r0 = a1, r1
    return ids, xyz


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)


def library_path():
    return _LIB


def build_library(force=False):
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", src_dir, "-s", "clean"])
    subprocess.check_call(["make", "-C", src_dir, "-s", "-j4"])
    return _LIB


_lib = None


def load_library():
    """Load the C-ABI library; fails loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        raise S2MError(-2, "HIP extension %s is missing: run __graft_entry__.build() "
                           "(there is no CPU fallback)" % _LIB)
    lib = C.CDLL(_LIB)
    lib.s2m_strerror.restype = C.c_char_p
    lib.s2m_last_error.restype = C.c_char_p
    lib.s2m_last_error.argtypes = [C.c_void_p]
    for name in ABI_SYMBOLS:
        if os.environ.get("S2M_LIB") and not hasattr(lib, name):
            continue        # (an A/B build of an OLDER ABI, scripts/ab.sh: the legs it is run on use what it has)
        getattr(lib, name)  # AttributeError if the ABI is incomplete
    _lib = lib
    return lib


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def default_config(**kw):
    cfg = Config()
    load_library().s2m_config_default(C.byref(cfg))
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


class Engine:
    """One handle = one GPU's scan-to-map engine (not re-entrant, like the reference caller)."""

    def __init__(self, cfg=None, **kw):
        self.lib = load_library()
        self.cfg = cfg if cfg is not None else default_config(**kw)
        h = C.c_void_p()
        rc = self.lib.s2m_create(C.byref(self.cfg), C.byref(h))
        if rc != 0:
            raise S2MError(rc, self.lib.s2m_strerror(rc).decode())
        self.h = h
        self.n = 0

    def close(self):
        """s2m_destroy; returns its code (0, or S2M_ERR_TIMEOUT when the device did not drain within the deadline)."""
        rc = 0
        if getattr(self, "h", None):
            rc = self.lib.s2m_destroy(self.h)
            self.h = None
        return rc

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise S2MError(rc, "%s (%s)" % (self.lib.s2m_strerror(rc).decode(),
                                           self.lib.s2m_last_error(self.h).decode()))

    # -- configuration ---------------------------------------------------------------------
    def set_config(self, **kw):
        for k, v in kw.items():
            setattr(self.cfg, k, v)
        self._ck(self.lib.s2m_set_config(self.h, C.byref(self.cfg)))

    def debug_state(self):
        """One line on what the handle is doing (s2m_debug_state): for watchdogs and error reports."""
        buf = C.create_string_buffer(2048)
        self._ck(self.lib.s2m_debug_state(self.h, buf, C.c_int64(len(buf))))
        return buf.value.decode()

    def test_stall(self, kind, after=0):
        """Fault injection (tests): withhold the hand-backs of `kind` ('worker', 'mail', 'reduce', 'layout'; None disarms) from the
        `after`-th one on."""
        k = {None: 0, "worker": 1, "mail": 2, "reduce": 3, "layout": 4}[kind]
        self._ck(self.lib.s2m_test_stall(self.h, C.c_int32(k), C.c_int64(after)))

    def set_stream(self, hip_stream):
        self._ck(self.lib.s2m_set_stream(self.h, C.c_void_p(hip_stream)))

    def set_timing(self, on=True):
        """1/2: HIP-event time every pass (search kernels and reduce kernel); n > 2: every n-th pass."""
        self._ck(self.lib.s2m_set_timing(self.h, C.c_int(int(on))))

    def timing(self):
        ms = (C.c_double * 3)()
        self._ck(self.lib.s2m_get_timing(self.h, ms))
        return list(ms)

    def bet_stats(self):
        """(passes that bet on an empty far-point list and won, passes whose bet was lost and redone)"""
        st = (C.c_int64 * 2)()
        self._ck(self.lib.s2m_bet_stats(self.h, st))
        return int(st[0]), int(st[1])

    def timing_stats(self):
        st = (C.c_double * 6)()
        self._ck(self.lib.s2m_get_timing_stats(self.h, st))
        return dict(match_ms=st[0], match_launches=int(st[1]), fit_ms=st[2], fit_launches=int(st[3]),
                    reduce_ms=st[4], reduce_launches=int(st[5]))

    # -- map / scan ------------------------------------------------------------------------
    def map_build(self, xyz):
        xyz = np.ascontiguousarray(xyz, np.float32)
        assert xyz.ndim == 2 and xyz.shape[1] >= 3
        self._ck(self.lib.s2m_map_build(self.h, _p(xyz), C.c_int64(xyz.shape[1]), C.c_int64(xyz.shape[0]), 0))

    def map_build_device(self, dev_ptr, stride, m):
        self._ck(self.lib.s2m_map_build(self.h, C.c_void_p(dev_ptr), C.c_int64(stride), C.c_int64(m), 1))

    def map_share(self, owner):
        """Search `owner`'s map instead of holding a copy (several scans in flight against one map)."""
        self._ck(self.lib.s2m_map_share(self.h, owner.h))
        self._map_owner = owner  # keep the owner alive as long as this handle borrows its map

    def map_size(self):
        m = C.c_int64()
        self._ck(self.lib.s2m_map_size(self.h, C.byref(m)))
        return m.value

    def map_info(self):
        info = (C.c_double * 8)()
        self._ck(self.lib.s2m_map_info(self.h, info))
        return dict(cell=info[0], origin=(info[1], info[2], info[3]), bricks=int(info[4]),
                    top_entries=int(info[5]), occupied_cells=int(info[6]), mean_per_cell=info[7])

    def map_last_update_merged(self):
        """True when the last map update kept the grid (applied in place, or the map re-laid by a merge -- no re-sort), False after a
        rebuild."""
        m = C.c_int32()
        self._ck(self.lib.s2m_map_last_update(self.h, C.byref(m)))
        return bool(m.value)

    def map_update_stats(self):
        """Running counts of how the map updates were produced, in ONE vocabulary: in_place (only the touched bricks rewritten),
        relaid (the whole map laid out again in key order by a merge), rebuilt (re-sorted), regridded (rebuilds that also chose a
        new cell size); device buffer (re)allocations (process-wide); top_relaid (times the top-level array was re-laid: no point
        moves); big_bricks; why updates could not stay in place."""
        st = (C.c_int64 * 12)()
        self._ck(self.lib.s2m_map_update_stats(self.h, st))
        inplace = self.map_inplace_updates()
        return dict(in_place=inplace, relaid=st[0] - inplace, rebuilt=st[1], regridded=st[2], allocations=st[3], top_relaid=st[4], big_bricks=st[5],
                    not_in_place=dict(unrepresentable=st[6], no_table_rows=st[7], brick_too_large=st[8], tail_exhausted=st[9]),
                    relaid_beside=st[10], regridded_beside=st[11])

    def map_inplace_updates(self):
        """Updates applied in place (only the touched bricks rewritten): s2m_map_inplace_updates."""
        n = C.c_int64()
        self._ck(self.lib.s2m_map_inplace_updates(self.h, C.byref(n)))
        return n.value

    def map_ids(self):
        """Point ids in the order of map_points() (ascending)."""
        m = C.c_int64()
        self._ck(self.lib.s2m_map_get_ids(self.h, None, C.c_int64(0), C.byref(m)))
        out = np.zeros(max(m.value, 1), np.uint32)
        self._ck(self.lib.s2m_map_get_ids(self.h, _p(out), C.c_int64(len(out)), C.byref(m)))
        return out[:m.value]

    def map_changes(self, token, capacity=1 << 20, lag=0):
        """s2m_map_get_changes since the call that returned `token`: MapChanges(token, resync, add_xyz, add_ids, rem_xyz, rem_ids,
        boxes (n x 6), box_after_added, box_after_removed) -- box k applies after add[:box_after_added[k]] and
        rem[:box_after_removed[k]]; within a stretch additions first, then removals.  lag=1: the report of the previous call."""
        tok = C.c_uint64(int(token))
        xyz = np.zeros((capacity, 3), np.float32)
        ids = np.zeros(capacity, np.uint32)
        rxyz = np.zeros((capacity, 3), np.float32)
        rem = np.zeros(capacity, np.uint32)
        boxes = np.zeros((64, 6), np.float32)
        ba = np.zeros(64, np.int64)
        br = np.zeros(64, np.int64)
        c = MapChangesC(xyz.ctypes.data, ids.ctypes.data, capacity, rxyz.ctypes.data, rem.ctypes.data, capacity, boxes.ctypes.data,
                        ba.ctypes.data, br.ctypes.data, 64, 0, 0, 0, 0, int(lag))
        self._ck(self.lib.s2m_map_get_changes(self.h, C.byref(tok), C.byref(c)))
        return MapChanges(tok.value, bool(c.resync), xyz[:c.n_added], ids[:c.n_added], rxyz[:c.n_removed], rem[:c.n_removed],
                          boxes[:c.n_boxes], ba[:c.n_boxes], br[:c.n_boxes])

    def map_order(self):
        """order[j] = caller index of the point at sorted position j (the engine's tie order); 0xffffffff where an in-place
        update left a hole."""
        m = C.c_int64()
        self._ck(self.lib.s2m_map_get_order(self.h, None, C.c_int64(0), C.byref(m)))
        out = np.zeros(max(m.value, 1), np.uint32)
        self._ck(self.lib.s2m_map_get_order(self.h, _p(out), C.c_int64(len(out)), C.byref(m)))
        return out[:m.value]

    def map_grid(self):
        """(lo xyz, hi xyz) of the box of bricks the map may occupy (conservative; hi < lo: empty)."""
        c = (C.c_int32 * 6)()
        self._ck(self.lib.s2m_map_grid(self.h, c))
        return tuple(c[:3]), tuple(c[3:])

    def map_rank(self):
        """rank[i] = sorted position of caller index i: what the oracle takes as the tie order of equal distances."""
        order = self.map_order()
        order = order[order != 0xffffffff]          # holes do not take part in the order
        rank = np.empty(len(order), np.uint32)
        rank[order] = np.arange(len(order), dtype=np.uint32)
        return rank

    def complete_neighbors(self):
        """Nearest_Points beyond the gate (unbounded ikd-Tree search); returns the number of lists that were short."""
        c = C.c_int64()
        self._ck(self.lib.s2m_complete_neighbors(self.h, C.byref(c)))
        return c.value

    def map_add(self, xyz, downsample_on, downsample_size=0.5):
        """ikdtree.Add_Points(points, downsample_on); returns voxels rewritten (or n)."""
        xyz = np.ascontiguousarray(xyz, np.float32)
        n_added = C.c_int64()
        self._ck(self.lib.s2m_map_add(self.h, _p(xyz), C.c_int64(xyz.shape[1] if xyz.ndim == 2 else 3),
                                      C.c_int64(xyz.shape[0]), C.c_int(int(downsample_on)), C.c_float(downsample_size),
                                      0, C.byref(n_added)))
        return n_added.value

    def map_delete_boxes(self, boxes):
        boxes = np.ascontiguousarray(boxes, np.float32).reshape(-1, 6)
        n_del = C.c_int64()
        self._ck(self.lib.s2m_map_delete_boxes(self.h, _p(boxes), C.c_int64(len(boxes)), C.byref(n_del)))
        return n_del.value

    def fov_segment(self, pos_lid, cube_len):
        """lasermap_fov_segment(); returns (local_map[6], n_boxes, n_deleted)."""
        pos = np.ascontiguousarray(pos_lid, np.float64)
        lm = np.zeros(6, np.float32)
        nb, nd = C.c_int32(), C.c_int64()
        self._ck(self.lib.s2m_fov_segment(self.h, _p(pos), C.c_double(cube_len), _p(lm), C.byref(nb), C.byref(nd)))
        return lm, nb.value, nd.value

    def fov_reset(self):
        self._ck(self.lib.s2m_fov_reset(self.h))

    def map_incremental(self, state, filter_size_map=0.5, ekf_inited=True):
        state = np.ascontiguousarray(state, np.float64)
        na, nb = C.c_int64(), C.c_int64()
        self._ck(self.lib.s2m_map_incremental(self.h, _p(state), C.c_double(filter_size_map),
                                              C.c_int32(int(bool(ekf_inited))), C.byref(na), C.byref(nb)))
        return na.value, nb.value

    def map_points(self):
        m = C.c_int64()
        self._ck(self.lib.s2m_map_get_points(self.h, None, C.c_int64(0), C.byref(m)))
        out = np.zeros((max(m.value, 1), 3), np.float32)
        self._ck(self.lib.s2m_map_get_points(self.h, _p(out), C.c_int64(len(out)), C.byref(m)))
        return out[:m.value]

    def scan_set(self, xyz):
        xyz = np.ascontiguousarray(xyz, np.float32)
        assert xyz.ndim == 2 and xyz.shape[1] >= 3
        self._ck(self.lib.s2m_scan_set(self.h, _p(xyz), C.c_int64(xyz.shape[1]), C.c_int64(xyz.shape[0]), 0))
        self.n = xyz.shape[0]

    def scan_set_downsampled(self, xyz, leaf=0.5):
        """pcl::VoxelGrid(leaf) of the cloud becomes the current scan; returns feats_down_size."""
        xyz = np.ascontiguousarray(xyz, np.float32)
        assert xyz.ndim == 2 and xyz.shape[1] >= 3
        m = C.c_int64()
        self._ck(self.lib.s2m_scan_set_downsampled(self.h, _p(xyz), C.c_int64(xyz.shape[1]), C.c_int64(xyz.shape[0]),
                                                   C.c_float(leaf), 0, C.byref(m)))
        self.n = m.value
        return m.value

    def undistort(self, records, time_off_a, time_off_b, poses, state_end, sort_by_time=True):
        """records (n, stride) float32 starting with x, y, z; poses (K, 22) float64 IMUpose rows
        {offset_time, acc3, gyr3, vel3, pos3, rot9}.  Returns (xyz, perm)."""
        records = np.ascontiguousarray(records, np.float32)
        poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 22)
        state_end = np.ascontiguousarray(state_end, np.float64)
        n = records.shape[0]
        out = np.zeros((max(n, 1), 3), np.float32)
        perm = np.zeros(max(n, 1), np.uint32)
        self._ck(self.lib.s2m_undistort(self.h, _p(records), C.c_int64(records.shape[1]), C.c_int64(n),
                                        C.c_int32(time_off_a), C.c_int32(time_off_b), _p(poses), C.c_int32(len(poses)),
                                        _p(state_end), C.c_int(int(sort_by_time)), 0, _p(out), _p(perm)))
        return out[:n], perm[:n]

    def scan_set_from_raw(self, records, time_off_a, time_off_b, poses, state_end, leaf=0.5):
        records = np.ascontiguousarray(records, np.float32)
        poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 22)
        state_end = np.ascontiguousarray(state_end, np.float64)
        m = C.c_int64()
        self._ck(self.lib.s2m_scan_set_from_raw(self.h, _p(records), C.c_int64(records.shape[1]),
                                                C.c_int64(records.shape[0]), C.c_int32(time_off_a),
                                                C.c_int32(time_off_b), _p(poses), C.c_int32(len(poses)),
                                                _p(state_end), C.c_float(leaf), 0, C.byref(m)))
        self.n = m.value
        return m.value

    def scan_prefetch_raw(self, records, time_off_a=-1, time_off_b=-1):
        """Start the host-to-device copy of the NEXT sweep's records (a C-contiguous float32 array the caller keeps alive and
        hands to scan_set_from_raw unchanged); with the time field offsets also their time order."""
        if records is None:     # cancel: the sweep that was announced is not coming
            self._ck(self.lib.s2m_scan_prefetch_raw(self.h, None, C.c_int64(3), C.c_int64(0), C.c_int32(-1), C.c_int32(-1)))
            return
        assert records.dtype == np.float32 and records.flags["C_CONTIGUOUS"]
        self._ck(self.lib.s2m_scan_prefetch_raw(self.h, _p(records), C.c_int64(records.shape[1]), C.c_int64(records.shape[0]),
                                                C.c_int32(time_off_a), C.c_int32(time_off_b)))

    def scan_prepare_raw(self, records, time_off_a, time_off_b, poses, state_end, leaf=0.5):
        """The next frame's scan_set_from_raw on the handle's side stream (s2m_scan_prepare_raw); the same call of
        scan_set_from_raw afterwards picks the prepared scan up.  `records` must stay alive and unchanged until then."""
        assert records.dtype == np.float32 and records.flags["C_CONTIGUOUS"]
        poses = np.ascontiguousarray(poses, np.float64)
        state_end = np.ascontiguousarray(state_end, np.float64)
        self._ck(self.lib.s2m_scan_prepare_raw(self.h, _p(records), C.c_int64(records.shape[1]), C.c_int64(records.shape[0]),
                                               C.c_int32(time_off_a), C.c_int32(time_off_b), _p(poses), C.c_int32(len(poses)),
                                               _p(state_end), C.c_float(leaf)))

    def scan_get(self):
        n = C.c_int64()
        self._ck(self.lib.s2m_scan_get(self.h, None, C.c_int64(0), C.byref(n)))
        out = np.zeros((max(n.value, 1), 3), np.float32)
        self._ck(self.lib.s2m_scan_get(self.h, _p(out), C.c_int64(len(out)), C.byref(n)))
        return out[:n.value]

    def scan_set_device(self, dev_ptr, stride, n):
        self._ck(self.lib.s2m_scan_set(self.h, C.c_void_p(dev_ptr), C.c_int64(stride), C.c_int64(n), 1))
        self.n = n

    # -- passes ----------------------------------------------------------------------------
    def residual_pass(self, state, rematch):
        state = np.ascontiguousarray(state, np.float64)
        out = PassOut()
        self._ck(self.lib.s2m_residual_pass(self.h, _p(state), C.c_int(int(rematch)), C.byref(out)))
        return dict(HtH=np.array(out.HtH).reshape(12, 12), Htz=np.array(out.Htz),
                    effct=out.effct_feat_num, total_res=out.total_residual)

    def residual_pass_device(self, state, rematch, d_block_ptr):
        state = np.ascontiguousarray(state, np.float64)
        self._ck(self.lib.s2m_residual_pass_device(self.h, _p(state), C.c_int(int(rematch)), C.c_void_p(d_block_ptr)))

    def get_rows(self):
        m = C.c_int64()
        n = max(self.n, 1)
        hx = np.zeros((n, 12))
        h = np.zeros(n)
        idx = np.zeros(n, np.int32)
        self._ck(self.lib.s2m_get_rows(self.h, _p(hx), _p(h), _p(idx), C.c_int64(n), C.byref(m)))
        return hx[:m.value].copy(), h[:m.value].copy(), idx[:m.value].copy()

    def get_point_state(self):
        n = max(self.n, 1)
        sel = np.zeros(n, np.uint8)
        eff = np.zeros(n, np.uint8)
        plane = np.zeros((n, 4), np.float32)
        pd2 = np.zeros(n, np.float32)
        self._ck(self.lib.s2m_get_point_state(self.h, _p(sel), _p(eff), _p(plane), _p(pd2)))
        return dict(selected=sel[:self.n], eff=eff[:self.n], plane=plane[:self.n], pd2=pd2[:self.n])

    def get_neighbors(self):
        n = max(self.n, 1)
        idx = np.zeros((n, K), np.int32)
        d2 = np.zeros((n, K), np.float32)
        self._ck(self.lib.s2m_get_neighbors(self.h, _p(idx), _p(d2)))
        return idx[:self.n], d2[:self.n]

    # -- filter ----------------------------------------------------------------------------
    def eskf_update(self, x, x_prop, P, HtH, Htz):
        x = np.array(x, np.float64)
        x_prop = np.ascontiguousarray(x_prop, np.float64)
        P = np.ascontiguousarray(P, np.float64)
        HtH = np.ascontiguousarray(HtH, np.float64)
        Htz = np.ascontiguousarray(Htz, np.float64)
        sol = np.zeros(DIM)
        conv = C.c_int32()
        self._ck(self.lib.s2m_eskf_update(self.h, _p(x), _p(x_prop), _p(P), _p(HtH), _p(Htz), _p(sol), C.byref(conv)))
        return x, sol, bool(conv.value)

    def cov_update(self, P):
        P = np.array(P, np.float64)
        self._ck(self.lib.s2m_cov_update(self.h, _p(P)))
        return P

    def iterated_update(self, x, x_prop, P):
        x = np.array(x, np.float64)
        x_prop = np.ascontiguousarray(x_prop, np.float64)
        P = np.array(P, np.float64)
        log = IterLog()
        self._ck(self.lib.s2m_iterated_update(self.h, _p(x), _p(x_prop), _p(P), C.byref(log)))
        it = log.iters
        return dict(x=x, P=P, iters=it, rematch_passes=log.rematch_passes, converged=bool(log.converged),
                    ekf_stop=bool(log.ekf_stop), effct=np.array(log.effct[:it]), rematch=np.array(log.rematch[:it]),
                    conv=np.array(log.conv[:it]), total_res=np.array(log.total_residual[:it]),
                    solution=np.array([list(log.solution[i]) for i in range(it)]).reshape(it, DIM))

    def iterated_update_raw(self, x, x_prop, P, log):
        """Lean form for hot loops: x (36,), x_prop (36,), P (24,24) are float64 C-contiguous numpy
        arrays updated in place, log an IterLog instance; no conversions."""
        self._ck(self.lib.s2m_iterated_update(self.h, C.c_void_p(x.ctypes.data), C.c_void_p(x_prop.ctypes.data),
                                              C.c_void_p(P.ctypes.data), C.byref(log)))

    def iterated_update_bound(self, x, x_prop, P, log):
        """The lean form with its arguments bound once: returns a zero-argument callable that runs
        s2m_iterated_update on these buffers (which must stay alive and in place)."""
        fn, h = self.lib.s2m_iterated_update, self.h
        ax, ap, aP, al = C.c_void_p(x.ctypes.data), C.c_void_p(x_prop.ctypes.data), C.c_void_p(P.ctypes.data), C.byref(log)
        ck = self._ck

        def call():
            rc = fn(h, ax, ap, aP, al)
            if rc:
                ck(rc)
        return call

    @staticmethod
    def iterated_update_batch(engines, x, x_prop, P, logs=None):
        """k scans in flight on one GPU from this thread.  x (k, 36), x_prop (k, 36), P (k, 24, 24) float64
        C-contiguous arrays, updated in place; logs: optional ctypes array (IterLog * k).  Returns logs."""
        k = len(engines)
        assert x.shape == (k, STATE_DOUBLES) and x_prop.shape == (k, STATE_DOUBLES) and P.shape == (k, DIM, DIM)
        assert x.flags.c_contiguous and x_prop.flags.c_contiguous and P.flags.c_contiguous
        hs = (C.c_void_p * k)(*[e.h for e in engines])
        if logs is None:
            logs = (IterLog * k)()
        rc = engines[0].lib.s2m_iterated_update_batch(hs, C.c_int32(k), C.c_void_p(x.ctypes.data),
                                                      C.c_void_p(x_prop.ctypes.data), C.c_void_p(P.ctypes.data), logs)
        if rc != 0:
            msgs = [e.lib.s2m_last_error(e.h).decode() for e in engines]
            raise S2MError(rc, "%s (%s)" % (engines[0].lib.s2m_strerror(rc).decode(), "; ".join(m for m in msgs if m)))
        return logs

    @staticmethod
    def iterated_update_multi(engines, x, x_prop, P, logs=None):
        """ONE scan whose shards sit in `engines` (handle order = index order), summed on the host.  x (36,),
        x_prop (36,), P (24, 24) float64 C-contiguous arrays updated in place; returns the IterLog."""
        k = len(engines)
        assert x.shape == (STATE_DOUBLES,) and x_prop.shape == (STATE_DOUBLES,) and P.shape == (DIM, DIM)
        assert x.flags.c_contiguous and x_prop.flags.c_contiguous and P.flags.c_contiguous
        hs = (C.c_void_p * k)(*[e.h for e in engines])
        if logs is None:
            logs = (IterLog * 1)()
        rc = engines[0].lib.s2m_iterated_update_multi(hs, C.c_int32(k), C.c_void_p(x.ctypes.data),
                                                      C.c_void_p(x_prop.ctypes.data), C.c_void_p(P.ctypes.data), logs)
        if rc != 0:
            msgs = [e.lib.s2m_last_error(e.h).decode() for e in engines]
            raise S2MError(rc, "%s (%s)" % (engines[0].lib.s2m_strerror(rc).decode(), "; ".join(m for m in msgs if m)))
        return logs[0]

    def iterated_update_sharded(self, x, x_prop, P, d_block_ptr, reduce_cb):
        """reduce_cb() must sum the device block across ranks on this handle's stream."""
        x = np.array(x, np.float64)
        x_prop = np.ascontiguousarray(x_prop, np.float64)
        P = np.array(P, np.float64)
        log = IterLog()

        def _cb(_user):
            try:
                reduce_cb()
                return 0
            except Exception:  # never let an exception cross the C frame
                import traceback
                traceback.print_exc()
                return 1
        cb = ALLREDUCE_FN(_cb)
        self._ck(self.lib.s2m_iterated_update_sharded(self.h, _p(x), _p(x_prop), _p(P), C.byref(log),
                                                      C.c_void_p(d_block_ptr), cb, None))
        it = log.iters
        return dict(x=x, P=P, iters=it, rematch_passes=log.rematch_passes, converged=bool(log.converged),
                    ekf_stop=bool(log.ekf_stop), effct=np.array(log.effct[:it]), rematch=np.array(log.rematch[:it]),
                    conv=np.array(log.conv[:it]), total_res=np.array(log.total_residual[:it]),
                    solution=np.array([list(log.solution[i]) for i in range(it)]).reshape(it, DIM))

    @staticmethod
    def comm_unique_id():
        """128-byte RCCL unique id (call on rank 0, ship to the other ranks)."""
        buf = (C.c_uint8 * 128)()
        rc = load_library().s2m_comm_unique_id(buf)
        if rc != 0:
            raise S2MError(rc, "s2m_comm_unique_id failed (is librccl available?)")
        return bytes(buf)

    def comm_init_shm(self, name, nranks, rank):
        """Host shared-memory exchange between the processes of one node (s2m_comm_init_shm)."""
        self._ck(self.lib.s2m_comm_init_shm(self.h, C.c_char_p(name.encode()), C.c_int32(nranks), C.c_int32(rank)))

    def comm_init(self, uid, nranks, rank):
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        self._ck(self.lib.s2m_comm_init(self.h, buf, C.c_int32(nranks), C.c_int32(rank)))

    def comm_destroy(self):
        self._ck(self.lib.s2m_comm_destroy(self.h))

    def feat_queue(self):
        q = (C.c_int32 * FEAT_QUEUE)()
        n = C.c_int32()
        self._ck(self.lib.s2m_feat_queue_get(self.h, q, C.byref(n)))
        return list(q[:n.value])

    def set_feat_queue(self, q):
        if len(q) == 0:
            self._ck(self.lib.s2m_feat_queue_set(self.h, None, C.c_int32(0)))
            return
        q = list(q)
        arr = (C.c_int32 * max(len(q), 1))(*q)
        self._ck(self.lib.s2m_feat_queue_set(self.h, arr, C.c_int32(len(q))))

    def h_share_model(self, state, first_iteration, converge=False):
        state = np.ascontiguousarray(state, np.float64)
        n = max(self.n, 1)
        hx = np.zeros((n, 12))
        h = np.zeros(n)
        d = DynShare()
        d.converge = int(converge)
        d.h_x = hx.ctypes.data
        d.h = h.ctypes.data
        d.capacity = n
        self._ck(self.lib.s2m_h_share_model(self.h, _p(state), C.c_int(int(first_iteration)), C.byref(d)))
        return dict(valid=bool(d.valid), rows=d.rows, h_x=hx[:d.rows].copy(), h=h[:d.rows].copy(),
                    total_res=d.total_residual)
