"""Per-frame wall time of the whole on-device chain at C3 sizes (warm buffers):
raw 65 536-point sweep -> undistort + voxel grid (s2m_scan_set_from_raw) -> iterated update -> map_incremental
-> field-of-view trim.  Every stage ends with a device sync; the sum is what one LiDAR frame costs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daliti_amd import Engine, synth

c = synth.CONFIGS[os.environ.get("CONFIG", "C3")]
m = synth.make_map(c["M"], c["L"]); s = synth.make_scan(c["beams"], c["az"], c["L"])
xt, xp, P = synth.filter_inputs()
n = len(s)
rec = np.zeros((n, 12), np.float32)          # PointXYZINormal records: normal_x = time ratio, normal_z = span
rec[:, :3] = s
rec[:, 4] = np.linspace(0.0, 1.0, n, dtype=np.float32)
rec[:, 6] = 0.1
K = 20
poses = np.zeros((K, 22)); poses[:, 0] = np.linspace(0.0, 0.102, K)
poses[:, 13:22] = np.eye(3).ravel()          # sensor at rest: undistortion is the identity up to rounding
end = np.zeros(36); end[0:9] = np.eye(3).ravel(); end[12:21] = np.eye(3).ravel()
leaf = float(os.environ.get("LEAF", "0.5"))
e = Engine(max_iter=5, feat_threshold=100)
e.map_build(m)
def sync(): torch.cuda.synchronize()
rows = []
for frame in range(6):
    t = [time.perf_counter()]
    nd = e.scan_set_from_raw(rec, 4, 6, poses, end, leaf); sync(); t.append(time.perf_counter())
    r = e.iterated_update(xp, xp, P); sync(); t.append(time.perf_counter())
    na, nb = e.map_incremental(r["x"], 0.5); sync(); t.append(time.perf_counter())
    lm, nbx, ndel = e.fov_segment(r["x"][9:12], 1000.0); sync(); t.append(time.perf_counter())
    d = np.diff(t) * 1e3
    rows.append(d)
    print("frame %d: raw->scan %.3f ms (%d -> %d pts) | update %.3f ms (%d iters) | map_incremental %.3f ms (+%d) | fov %.3f ms | total %.3f ms"
          % (frame, d[0], n, nd, d[1], r["iters"], d[2], na + nb, d[3], d.sum()))
rows = np.array(rows[2:])
print("warm mean: raw->scan %.3f | update %.3f | map_incremental %.3f | fov %.3f | total %.3f ms -> %.0f frames/s"
      % (*rows.mean(0), rows.sum(1).mean(), 1e3 / rows.sum(1).mean()))
# the same frames back to back, synchronised only at the end (no per-stage sync: the asynchronous tail of one stage --
# e.g. the table build of a merged map update -- overlaps the host side of the next)
N = 20
sync(); t0 = time.perf_counter()
for frame in range(N):
    e.scan_set_from_raw(rec, 4, 6, poses, end, leaf)
    r = e.iterated_update(xp, xp, P)
    e.map_incremental(r["x"], 0.5)
    e.fov_segment(r["x"][9:12], 1000.0)
sync(); dt = (time.perf_counter() - t0) / N * 1e3
print("back to back, no per-stage sync: %.3f ms per frame -> %.0f frames/s" % (dt, 1e3 / dt))
print(e.debug_state())
print(e.map_update_stats())
