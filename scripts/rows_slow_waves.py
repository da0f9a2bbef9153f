"""dev probe: what makes the slowest waves of match_rows slow?  Per-point cycle counts of the first-shell kernel
(S2M_DEBUG_MATCH) against range, brick straddling, and the number of map points in the 3x3x3 cells."""
import os, sys, ctypes as C
import numpy as np
os.environ["S2M_DEBUG_MATCH"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from daliti_amd import Engine, synth
from scipy.spatial import cKDTree
c = synth.make_config("C3")
e = Engine()
e.map_build(c["map"]); e.scan_set(c["scan"])
info = e.map_info(); cell = info["cell"]; org = np.array(info["origin"])
import oracle
for name, x in (("predicted", c["x_prop"]), ("converged~true", c["x_true"])):
    for _ in range(3): e.residual_pass(x, True)
    d = np.zeros((e.n, 4), np.uint32)
    assert e.lib.s2m_debug_match(e.h, C.c_void_p(d.ctypes.data)) == 0
    hard = (d[:, 3] & 0xff) > 1
    cyc = d[:, 0].astype(np.float64)
    # the hard part is added to word 0 for far points: use only the resolved points for the first-shell kernel's own time
    ok = ~hard
    w = oracle.body_to_world(x, c["scan"])
    rng = np.linalg.norm(c["scan"].astype(np.float64), axis=1)
    cc = np.floor((w - org) / cell).astype(np.int64)
    strad = ((cc[:, 0] & 7) == 0) | ((cc[:, 0] & 7) == 7)
    tree = cKDTree(c["map"])
    sub = np.random.RandomState(0).choice(e.n, 8000, replace=False)
    cnt = np.array([len(v) for v in tree.query_ball_point(w[sub], 1.5 * cell, p=np.inf)])   # points in the 3x3x3-cell cube (approx.)
    print("== %s pose: first-shell cycles (100 MHz ticks) pct50 %.0f pct90 %.0f pct99 %.0f max %.0f" % (name, *np.percentile(cyc[ok], [50, 90, 99, 100])))
    wv = cyc[: e.n // 32 * 32].reshape(-1, 32)
    wmax = wv.max(1)
    print("   per wave (32 points): max-of-wave pct50 %.0f pct90 %.0f pct99 %.0f max %.0f" % tuple(np.percentile(wmax, [50, 90, 99, 100])))
    slow = np.argsort(wmax)[-20:]
    beams = slow * 32 // 1024
    print("   slowest 20 waves: beams", sorted(set(beams.tolist())), " mean range %.1f m" % rng[: e.n // 32 * 32].reshape(-1, 32)[slow].mean())
    for lo, hi in ((0, 5), (5, 15), (15, 40), (40, 200)):
        m = ok & (rng >= lo) & (rng < hi)
        if m.any(): print("   range %3d-%3d m: n %6d  mean cycles %.0f  pct99 %.0f" % (lo, hi, m.sum(), cyc[m].mean(), np.percentile(cyc[m], 99)))
    print("   straddling a brick boundary in x: mean %.0f vs %.0f" % (cyc[ok & strad].mean(), cyc[ok & ~strad].mean()))
    r = np.corrcoef(cnt[ok[sub]], cyc[sub][ok[sub]])[0, 1]
    print("   correlation of cycles with the number of map points in the cube: %.2f; cube count pct50 %d pct99 %d; cycles of the densest 1 %% %.0f" % (
        r, *np.percentile(cnt, [50, 99]).astype(int), cyc[sub][cnt >= np.percentile(cnt, 99)].mean()))
e.close()
