"""Dev experiment: K engine handles on one GPU, each on its own stream and host thread (C5-style batching)."""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daliti_amd import Engine, synth
from daliti_amd.engine import IterLog

c = synth.CONFIGS["C3"]
m = synth.make_map(c["M"], c["L"])
d_map = torch.from_numpy(m).cuda()
steps = int(os.environ.get("STEPS", "200"))
for K in [int(v) for v in os.environ.get("KS", "1,2,4,8").split(",")]:
    engs, bufs = [], []
    for k in range(K):
        pos = synth.SENSOR_POS + np.array([(k - (K - 1) / 2.0) * 2.0, 0.0, 0.0])
        s = synth.make_scan(c["beams"], c["az"], c["L"], seed=2 + k, sensor_pos=pos)
        _, xp, P0 = synth.filter_inputs(pos)
        e = Engine(max_iter=5, feat_threshold=100)
        if k > 0 and os.environ.get("SHARE", "1") != "0":
            e.map_share(engs[0])      # one HBM-resident map for all handles
        else:
            e.map_build_device(d_map.data_ptr(), 3, len(m))
        e.scan_set(s)
        engs.append(e)
        bufs.append((np.zeros(36), np.ascontiguousarray(xp, np.float64), np.zeros((24, 24)), P0, IterLog()))
    torch.cuda.synchronize()

    def work(k, n):
        e = engs[k]; xb, xpb, Pb, P0, lg = bufs[k]
        for _ in range(n):
            e.set_feat_queue(())
            xb[:] = xpb; Pb[:] = P0
            e.iterated_update_raw(xb, xpb, Pb, lg)

    for k in range(K): work(k, 5)
    ths = [threading.Thread(target=work, args=(k, steps)) for k in range(K)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    print("K=%d (%s map): %.1f scans/s total, %.3f ms per scan per handle, pos %s" % (K, "own" if os.environ.get("SHARE", "1") == "0" else "shared", K * steps / dt, dt / steps * 1e3, bufs[0][0][9:12]))
    for e in engs: e.close()
