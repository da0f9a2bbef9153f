"""The drive-through workload of tools/world.h bound with ctypes: a synthetic hall with pillars, raw LiDAR sweeps from a
sensor that moves while it sweeps, the IMU poses and predicted states that go with them, a seed map of the hall's first
section -- and the C++ frame loop of tools/bench_loop.cpp that drives the engine along such a trajectory.  Workload
plumbing for bench.py and the tests; no counterpart in the reference (its only validation is a rosbag drive,
README.md:38-55)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def helper_library():
    """tools/bench_loop.cpp (+ tools/world.h) as a shared object next to the product library; normally built by
    __graft_entry__.build(), rebuilt here when it is missing or older than its sources: plain g++ against the C ABI."""
    from .engine import library_path
    lib_dir = os.path.join(ROOT, "daliti_amd", "_lib")
    path = os.path.join(lib_dir, "libs2m_benchloop.so")
    srcs = [os.path.join(ROOT, "tools", "bench_loop.cpp"), os.path.join(ROOT, "tools", "world.h")]
    if not os.path.exists(path) or any(os.path.exists(s) and os.path.getmtime(path) < os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread", "-I", os.path.join(ROOT, "include"), srcs[0],
                               "-L", lib_dir, "-ldaliti_s2m", "-Wl,-rpath," + lib_dir, "-o", path])
    C.CDLL(library_path(), mode=C.RTLD_GLOBAL)
    return C.CDLL(path)


class WorldParams(C.Structure):
    _fields_ = [("x0", C.c_double), ("len", C.c_double), ("width", C.c_double), ("H", C.c_double), ("pitch", C.c_double),
                ("lane", C.c_double), ("range", C.c_double), ("step", C.c_double), ("wobble", C.c_double),
                ("wobble_period", C.c_double), ("sensor_z", C.c_double), ("sigma", C.c_double), ("err_pos", C.c_double),
                ("err_rot", C.c_double), ("seed", C.c_uint64)]


N_POSES = 20


class World:
    """A hall `width` wide and `length` long that starts `width / 2` behind the sensor's first position (0, 0, sensor_z);
    the sensor advances `step` metres per frame along +x."""

    def __init__(self, width, length, step, H=10.0, pitch=24.0, lane=4.0, max_range=150.0, wobble=1.0, wobble_period=200.0,
                 sensor_z=1.5, sigma=0.01, err_pos=0.02, err_rot=np.deg2rad(0.1), seed=7):
        self.p = WorldParams(-0.5 * width, length, width, H, pitch, lane, max_range, step, wobble, wobble_period, sensor_z, sigma,
                             err_pos, err_rot, seed)
        self.lib = helper_library()
        self.lib.s2m_world_sweeps.restype = C.c_int
        self.lib.s2m_world_seed.restype = C.c_int

    def seed_map(self, m, span=None):
        """m points on the surfaces of the hall's first `span` metres (default: its width -- a square section)."""
        xyz = np.empty((m, 3), np.float32)
        rc = self.lib.s2m_world_seed(C.byref(self.p), C.c_double(self.p.width if span is None else span), C.c_int64(m),
                                     C.c_void_p(xyz.ctypes.data))
        assert rc == 0, rc
        return xyz

    def sweeps(self, f0, count, beams, az, threads=8):
        """`count` raw sweeps from frame f0 on: dict(rec = count x (beams * az) x 12 float32 (the first n[k] records of
        sweep k are valid), n, poses = count x 20 x 22 float64, x_prop / x_true = count x 36)."""
        rays = beams * az
        # (the hall ends `length` metres behind its near wall, half a width behind the start: a sensor that has left it sees
        # nothing the generator knows how to range)
        last_x = (f0 + count) * float(self.p.step)
        if last_x > float(self.p.len) - 0.5 * float(self.p.width) - 1.0:
            raise ValueError("the drive leaves the hall after %.0f m: %d frames of %.2f m from frame %d do not fit" % (
                float(self.p.len) - 0.5 * float(self.p.width) - 1.0, count, float(self.p.step), f0))
        rec = np.zeros((count, rays, 12), np.float32)
        n = np.zeros(count, np.int64)
        poses = np.zeros((count, N_POSES, 22))
        x_prop = np.zeros((count, 36))
        x_true = np.zeros((count, 36))
        rc = self.lib.s2m_world_sweeps(C.byref(self.p), C.c_int32(f0), C.c_int32(count), C.c_int32(beams), C.c_int32(az),
                                       C.c_void_p(rec.ctypes.data), C.c_int64(rays * 12), C.c_void_p(n.ctypes.data),
                                       C.c_void_p(poses.ctypes.data), C.c_int32(N_POSES), C.c_void_p(x_prop.ctypes.data),
                                       C.c_void_p(x_true.ctypes.data), C.c_int32(threads))
        assert rc == 0, rc
        return dict(rec=rec, n=n, poses=poses, x_prop=x_prop, x_true=x_true)


def run_frames(eng, sw, P0, frames, warm, leaf=0.5, filter_size_map=0.5, cube_len=1000.0, prefetch=2, publish=False):
    """tools/bench_loop.cpp, s2m_bench_frames_moving: the sweeps `sw` (World.sweeps, warm + frames of them) through the
    engine `eng`, one frame after the other (publish: a host mirror of the map is brought up to date after every frame, as a node
    that publishes /Laser_map would -- 1 / True: to the map as it is, 2: one call behind, nobody waits for the device).  Returns per-frame arrays (warm-up frames included): ms, how (0 = map
    rebuilt, 1 = re-laid by a merge, 2 = in place), deleted, n_scan, x (updated states), iters, effct."""
    from .engine import IterLog
    lib = helper_library()
    fn = lib.s2m_bench_frames_moving
    fn.restype = C.c_int
    total = warm + frames
    assert len(sw["n"]) >= total
    rec = sw["rec"]
    x = np.zeros((total, 36))
    us = np.zeros(total)
    how = np.zeros(total, np.int32)
    deleted = np.zeros(total, np.int64)
    n_scan = np.zeros(total, np.int64)
    logs = (IterLog * total)()
    allocs = np.zeros(total, np.int32)
    publish_us = np.zeros(total)
    mirror_stats = np.zeros(4, np.int64)
    stage_us = np.zeros((total, 6))
    fetch_us = np.zeros(total)
    P0 = np.ascontiguousarray(P0, np.float64)
    rc = fn(eng.h, C.c_int32(frames), C.c_int32(warm), C.c_void_p(rec.ctypes.data), C.c_int64(rec.shape[1] * 12),
            C.c_void_p(sw["n"].ctypes.data), C.c_int32(4), C.c_int32(6), C.c_void_p(sw["poses"].ctypes.data), C.c_int32(N_POSES),
            C.c_void_p(sw["x_prop"].ctypes.data), C.c_void_p(P0.ctypes.data), C.c_float(leaf), C.c_double(filter_size_map),
            C.c_double(cube_len), C.c_int32(prefetch), C.c_void_p(x.ctypes.data), C.c_void_p(us.ctypes.data),
            C.c_void_p(how.ctypes.data), C.c_void_p(deleted.ctypes.data), C.c_void_p(n_scan.ctypes.data), logs,
            C.c_void_p(allocs.ctypes.data), C.c_int32(int(publish)), C.c_void_p(publish_us.ctypes.data),
            C.c_void_p(mirror_stats.ctypes.data), C.c_void_p(stage_us.ctypes.data), C.c_void_p(fetch_us.ctypes.data))
    if rc != 0:
        raise RuntimeError("s2m_bench_frames_moving failed: %d (%s)" % (rc, eng.lib.s2m_last_error(eng.h).decode()))
    return dict(ms=us * 1e-3, how=how, deleted=deleted, n_scan=n_scan, x=x, allocs=allocs, publish_ms=publish_us * 1e-3, fetch_ms=fetch_us * 1e-3,
                mirror_points=int(mirror_stats[0]), map_points=int(mirror_stats[1]), mirror_resyncs=int(mirror_stats[2]), mirror_missed=int(mirror_stats[3]),
                stage_ms=np.diff(np.concatenate([np.zeros((total, 1)), stage_us], axis=1), axis=1) * 1e-3, iters=np.array([l.iters for l in logs]), rematch_passes=np.array([l.rematch_passes for l in logs]),
                effct=[np.array(l.effct[:l.iters]) for l in logs])
