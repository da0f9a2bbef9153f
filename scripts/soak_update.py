"""Soak: several hundred frames of register + map_incremental (+ FOV trim) with the sensor wandering through the C2 room;
watches device memory, how many updates merged, and checks the final map against a brute-force 5-NN on a sample."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daliti_amd import Engine, synth

frames = int(os.environ.get("FRAMES", "300"))
c = synth.CONFIGS["C2"]
m = synth.make_map(c["M"], c["L"])
e = Engine(max_iter=5, feat_threshold=100)
e.map_build(m)
free0 = torch.cuda.mem_get_info()[0]
merged = 0
x = None
t0 = time.perf_counter()
for k in range(frames):
    pos = synth.SENSOR_POS + np.array([25.0 * np.sin(k / 150.0), 20.0 * np.sin(k / 230.0), 0.0])   # stays inside the room
    s = synth.make_scan(32, 512, c["L"], seed=100 + k, sensor_pos=pos)
    xt, xp, P = synth.filter_inputs(pos, dtheta=synth.DTHETA0 * 0.2, dpos=synth.DPOS0 * 0.2)
    e.scan_set_downsampled(s, 0.5)
    r = e.iterated_update(xp, xp, P)
    assert np.abs(r["x"][9:12] - xt[9:12]).max() < 0.02, (k, r["x"][9:12], xt[9:12])
    e.map_incremental(r["x"], 0.5)
    merged += e.map_last_update_merged()
    e.fov_segment(r["x"][9:12], 1000.0)
    if k % 100 == 0:
        print("frame %d: map %d points, free memory change %+.1f MB" % (k, e.map_size(), (torch.cuda.mem_get_info()[0] - free0) / 1e6), flush=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%d frames in %.2f s (incl. host-side scan synthesis); %d updates merged; map %d points; free memory change %+.1f MB"
      % (frames, dt, merged, e.map_size(), (torch.cuda.mem_get_info()[0] - free0) / 1e6))
# exactness of the search on the grown map: brute force (float64) on a sample of the last scan
pts = e.map_points().astype(np.float64)
sample = s[:256:8].astype(np.float64)
e.scan_set(s[:256:8].copy()); e.residual_pass(r["x"], True)
idx, d2 = e.get_neighbors()
xs = r["x"]
R, t, RLI, TLI = xs[0:9].reshape(3, 3), xs[9:12], xs[12:21].reshape(3, 3), xs[21:24]
w = ((sample @ RLI.T + TLI) @ R.T + t).astype(np.float32).astype(np.float64)
worst = 0.0
for q in range(len(sample)):
    dd = ((pts - w[q]) ** 2).sum(1)
    ref = np.sort(dd)[:5]
    if ref[4] > 5.0: continue                      # beyond the gate the engine only reports "not five within it"
    worst = max(worst, float(np.abs(d2[q] - ref).max() / ref[4]))
    assert set(idx[q]) == set(np.argsort(dd, kind="stable")[:5]) or np.abs(d2[q] - ref).max() < 1e-6 * ref[4], q
print("brute-force check of %d sample queries against the grown map: worst relative d2 difference %.1e" % (len(sample), worst))
