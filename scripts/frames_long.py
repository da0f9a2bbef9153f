#!/usr/bin/env python3
"""Does the frame time hold over a long run?  (GPU box)  N frames of the bench's frame leg (same sweep, C3 map) through
s2m_bench_frames in the pipelined form; prints the median per block of 100 frames, how the map updates were produced and
the map size.  usage: frames_long.py [frames]"""
import sys, ctypes as C
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import numpy as np
import torch
import bench
from daliti_amd import Engine, synth

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
c = synth.CONFIGS["C3"]
m = synth.make_map(c["M"], c["L"], seed=1)
scan = synth.make_scan(c["beams"], c["az"], c["L"], seed=2)
_xt, x_prop, P0 = synth.filter_inputs(synth.SENSOR_POS)
e = Engine(max_iter=5, feat_threshold=100)
e.map_build(m)
n = len(scan)
rec = np.zeros((n, 12), np.float32); rec[:, :3] = scan
rec[:, 4] = np.linspace(0.0, 1.0, n, dtype=np.float32); rec[:, 6] = 0.1
K = 20
poses = np.zeros((K, 22)); poses[:, 0] = np.linspace(0.0, 0.102, K); poses[:, 13:22] = np.eye(3).ravel()
end = np.zeros(36); end[0:9] = np.eye(3).ravel(); end[12:21] = np.eye(3).ravel()
fn = bench.bench_helper().s2m_bench_frames
fn.restype = C.c_int
x_out = np.zeros(36); frame_us = np.zeros(frames); pose_us = np.zeros(frames); merged = np.zeros(frames, np.int32)
xp = np.ascontiguousarray(x_prop, np.float64); P0c = np.ascontiguousarray(P0, np.float64)
rc = fn(e.h, C.c_int32(frames), C.c_void_p(rec.ctypes.data), C.c_int64(12), C.c_int64(n), C.c_int32(4), C.c_int32(6),
        C.c_void_p(poses.ctypes.data), C.c_int32(K), C.c_void_p(end.ctypes.data), C.c_float(0.5), C.c_void_p(xp.ctypes.data),
        C.c_void_p(P0c.ctypes.data), C.c_double(0.5), C.c_double(1000.0), C.c_int32(2), C.c_void_p(x_out.ctypes.data),
        C.c_void_p(frame_us.ctypes.data), C.c_void_p(merged.ctypes.data), C.c_void_p(pose_us.ctypes.data))
assert rc == 0, e.lib.s2m_last_error(e.h)
torch.cuda.synchronize()
per = frame_us * 1e-3
for b in range(0, frames, 100):
    blk = per[b:b + 100]
    print("frames %4d-%4d: median %.3f ms, max %.3f" % (b, b + len(blk) - 1, np.median(blk), blk.max()))
st = e.map_update_stats()
print("updates: %s, in place %d; map %d points; frames over 0.5 ms: %d of %d" % (st, e.map_inplace_updates(), e.map_size(), int((per > 0.5).sum()), frames))
