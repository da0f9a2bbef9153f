"""Why does the iterated update of a voxel-downsampled scan (34.7 k points) take longer per point than the bench's raw
65 k-point scan?  Times the same update hot (repeated) and right after a map merge (the per-frame regime), and with the
scan in its voxel-grid order vs shuffled vs sorted along the map's bricks."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daliti_amd import Engine, synth

c = synth.CONFIGS["C3"]
m = synth.make_map(c["M"], c["L"]); s = synth.make_scan(c["beams"], c["az"], c["L"])
xt, xp, P = synth.filter_inputs()
e = Engine(max_iter=5, feat_threshold=100)
e.map_build(m)
def sync(): torch.cuda.synchronize()
def timed_update(reps=20):
    ts = []
    for _ in range(reps):
        sync(); t0 = time.perf_counter(); r = e.iterated_update(xp, xp, P); sync(); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts)), r
e.scan_set(s); t_raw, _ = timed_update()
e.set_timing(1); e.iterated_update(xp, xp, P); st = e.timing_stats(); e.set_timing(0)
print("raw 65536-pt scan, hot: %.3f ms  (search %.1f us, fit %.1f us per rematch pass)" % (t_raw, 1e3 * st["match_ms"] / max(st["match_launches"], 1), 1e3 * st["fit_ms"] / max(st["fit_launches"], 1)))
n2 = e.scan_set_downsampled(s, 0.5)
down = e.scan_get().copy()
for name, order in (("voxel-grid order", np.arange(n2)), ("shuffled", np.random.RandomState(0).permutation(n2)),
                    ("sorted by 4 m blocks of the world position", None)):
    if order is None:
        w = down @ np.eye(3)  # body ~ world up to the small pose offset
        key = (np.floor(w[:, 2] / 4.0).astype(np.int64) * 4096 + np.floor(w[:, 1] / 4.0).astype(np.int64) + 2048) * 4096 + np.floor(w[:, 0] / 4.0).astype(np.int64) + 2048
        order = np.argsort(key, kind="stable")
    e.scan_set(down[order])
    t_hot, _ = timed_update()
    e.set_timing(1); e.iterated_update(xp, xp, P); st = e.timing_stats(); e.set_timing(0)
    print("%d-pt down-sampled scan, %s, hot: %.3f ms  (search %.1f us, fit %.1f us per rematch pass)" % (
        n2, name, t_hot, 1e3 * st["match_ms"] / max(st["match_launches"], 1), 1e3 * st["fit_ms"] / max(st["fit_launches"], 1)))
# the per-frame regime: every update follows a merge of the previous scan into the map
e.scan_set(down)
ts = []
for k in range(6):
    sync(); t0 = time.perf_counter(); r = e.iterated_update(xp, xp, P); sync(); ts.append(time.perf_counter() - t0)
    e.map_incremental(r["x"], 0.5); sync()
    e.scan_set(down)
print("%d-pt down-sampled scan, each update after a map merge: %.3f ms" % (n2, 1e3 * float(np.median(ts[1:]))))
