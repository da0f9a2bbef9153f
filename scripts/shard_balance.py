"""dev probe: how evenly do the N shards of the C3 scan load their GPUs?  Every shard is registered ALONE on this one GPU
(what its rank would do on its own device, minus the wait for the others): ms per iterated update per shard, for contiguous
index ranges (shard_range: whole beams per rank) and for azimuth sectors (every beam's azimuths [r, r + 1) * az / N)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from daliti_amd import Engine, synth
from daliti_amd.engine import IterLog
from daliti_amd.sharding import shard_range

cfg = synth.CONFIGS["C3"]
c = synth.make_config("C3")
scan = c["scan"]
beams, az = cfg["beams"], cfg["az"]
owner = Engine(max_iter=5)
owner.map_build(c["map"])
for N in (2, 4, 8):
    for scheme in ("range", "sector"):
        times, far = [], []
        for r in range(N):
            if scheme == "range":
                lo, hi = shard_range(len(scan), r, N)
                sh = scan[lo:hi]
            else:
                w = az // N
                sh = scan.reshape(beams, az, 3)[:, r * w:(r + 1) * w].reshape(-1, 3)
            e = Engine(max_iter=5)
            e.map_share(owner)
            e.scan_set(np.ascontiguousarray(sh))
            x, P, log = np.zeros(36), np.zeros((24, 24)), IterLog()
            xp = np.ascontiguousarray(c["x_prop"], np.float64)
            call = e.iterated_update_bound(x, xp, P, log)
            def step(k):
                e.set_feat_queue(()); x[:] = xp; P[:] = c["P"]; P[0, 0] += (k & 1) * 1e-15; call()
            for k in range(10): step(k)
            t0 = time.perf_counter()
            for k in range(100): step(k)
            times.append((time.perf_counter() - t0) * 10.0)   # ms per step
            e.close()
        print("N=%d %-6s per-shard ms/step: %s   max %.4f  mean %.4f" % (N, scheme, " ".join("%.4f" % t for t in times), max(times), np.mean(times)), flush=True)
owner.close()
