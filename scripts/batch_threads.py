"""Dev experiment: T host threads, each driving K scans through s2m_iterated_update_batch, all against one shared map
(is the batched entry host-bound or GPU-bound at its plateau?)."""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daliti_amd import Engine, synth
from daliti_amd.engine import IterLog

c = synth.CONFIGS["C3"]
m = synth.make_map(c["M"], c["L"])
d_map = torch.from_numpy(m).cuda()
steps = int(os.environ.get("STEPS", "60"))
owner = Engine(max_iter=5, feat_threshold=100)
owner.map_build_device(d_map.data_ptr(), 3, len(m))
scans = [synth.replica_scan("C5", k % 8) for k in range(8)]
for T, K in [tuple(int(v) for v in tk.split("x")) for tk in os.environ.get("TK", "1x8,1x16,2x8,4x4,4x8").split(",")]:
    groups = []
    for t in range(T):
        engs, xps, Ps = [], [], []
        for k in range(K):
            sc, pos = scans[(t * K + k) % 8]
            _, xp, P0 = synth.filter_inputs(pos)
            e = Engine(max_iter=5, feat_threshold=100)
            e.map_share(owner)
            e.scan_set(sc)
            engs.append(e); xps.append(xp); Ps.append(P0)
        xp = np.ascontiguousarray(np.stack(xps)); P0 = np.ascontiguousarray(np.stack(Ps))
        groups.append((engs, np.zeros_like(xp), xp, np.zeros_like(P0), P0, (IterLog * K)()))
    torch.cuda.synchronize()

    def work(g, n):
        engs, x, xp, P, P0, logs = g
        for _ in range(n):
            for e in engs:
                e.set_feat_queue(())
            x[:] = xp; P[:] = P0
            Engine.iterated_update_batch(engs, x, xp, P, logs)

    for g in groups: work(g, 3)
    ths = [threading.Thread(target=work, args=(g, steps)) for g in groups]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    n = T * K * steps
    print("T=%d threads x K=%d scans: %.0f scans/s, %.3g evals/s" % (T, K, n / dt, n * 65536 * 5 / dt), flush=True)
    for g in groups:
        for e in g[0]: e.close()
