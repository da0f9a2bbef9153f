"""Dev tool: cell-size sweep of the whole iterated update and of its rematch pass with the
row-run first-shell kernel (match_rows): ms/step of s2m_iterated_update, the search kernels and reduce<FIT> per
rematch pass (HIP events), and -- for C3 -- scans/s of the fused 8-scan batch.
usage: python scripts/sweep_cells.py C3|C4|R1 [cells...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daliti_amd import Engine, synth
from daliti_amd.engine import IterLog

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
cells = [float(v) for v in sys.argv[2:]] or [0.0, 0.25, 0.3, 0.35, 0.4, 0.45, 0.5, 0.55, 0.6, 0.7]
c = synth.CONFIGS[cfg]
m = synth.make_map(c["M"], c["L"])
s = synth.make_scan(c["beams"], c["az"], c["L"])
xt, xp, P = synth.filter_inputs()


def build(e):
    if cfg == "R1":
        e.map_build(m[:1])
        for lo in range(0, len(m), 1 << 20):
            e.map_add(m[lo:lo + (1 << 20)], True, 0.5)
        e.scan_set_downsampled(s, 0.5)
    else:
        e.map_build(m)
        e.scan_set(s)


if cfg == "R1":      # the downsampled map once; every cell size then rebuilds the grid over the same points
    e0 = Engine(max_iter=5)
    build(e0)
    m_r1, s_r1 = e0.map_points(), e0.scan_get()
    print("R1: map %d pts, scan %d pts, auto cell %.3f" % (len(m_r1), len(s_r1), e0.map_info()["cell"]), flush=True)
    e0.close()
for cell in cells:
    e = Engine(cell_size=cell, max_iter=5)
    if cfg == "R1":
        e.map_build(m_r1); e.scan_set(s_r1)
    else:
        build(e)
    info = e.map_info()
    x = xp.copy(); PP = P.copy(); log = IterLog()
    xb, Pb = np.zeros(36), np.zeros((24, 24))
    call = e.iterated_update_bound(xb, np.ascontiguousarray(xp), Pb, log)

    def step(k):
        e.set_feat_queue(())
        xb[:] = xp; Pb[:] = P; Pb[0, 0] += (k & 1) * 1e-15
        call()
    for k in range(10):
        step(k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(100):
        step(k)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
    e.set_timing(1)
    for k in range(20):
        step(k)
    st = e.timing_stats(); e.set_timing(0)
    line = "%s cell %.3f pts/cell %5.2f bricks %6d  %.4f ms/step  search %.1f us  reduce<FIT> %.1f us  reduce %.1f us  iters %d" % (
        cfg, info["cell"], info["mean_per_cell"], info["bricks"], 1e3 * dt, 1e3 * st["match_ms"] / max(st["match_launches"], 1),
        1e3 * st["fit_ms"] / max(st["fit_launches"], 1), 1e3 * st["reduce_ms"] / max(st["reduce_launches"], 1), log.iters)
    if cfg == "C3":   # the throughput regime: 8 scans in flight through the fused batch
        engs, keep = [], []
        K = 8
        xs, Ps = [], []
        for k in range(K):
            sc, pos = synth.replica_scan("C5", k)
            h = Engine(max_iter=5); h.map_share(e); h.scan_set(sc); engs.append(h)
            _t, xpk, Pk = synth.filter_inputs(pos); xs.append(xpk); Ps.append(Pk)
        X = np.ascontiguousarray(np.stack(xs)); XP = X.copy(); PB = np.ascontiguousarray(np.stack(Ps)); logs = (IterLog * K)()

        def bstep():
            for h in engs:
                h.set_feat_queue(())
            X[:] = XP; PB[:] = np.stack(Ps)
            Engine.iterated_update_batch(engs, X, XP, PB, logs)
        for _ in range(5):
            bstep()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40):
            bstep()
        torch.cuda.synchronize(); bt = (time.perf_counter() - t0) / 40
        line += "  batch8 %.0f scans/s" % (K / bt)
        for h in engs:
            h.close()
    print(line, flush=True)
    e.close()
