"""dev probe: how the far points of the first pass are distributed over the waves of match_rows (32 consecutive scan
points per wave at G = 2), and how many of them a second shell (radius 2 cells) would resolve"""
import os, sys, ctypes as C
import numpy as np
os.environ["S2M_DEBUG_MATCH"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from daliti_amd import Engine, synth
cfg = os.environ.get("CONFIG", "C3")
c = synth.CONFIGS[cfg]
m = synth.make_map(c["M"], c["L"]); s = synth.make_scan(c["beams"], c["az"], c["L"])
xt, xp, P = synth.filter_inputs()
e = Engine()
e.map_build(m); e.scan_set(s)
info = e.map_info(); cell = info["cell"]
for name, x in (("predicted pose (pass 1)", xp), ("true pose (~ last pass)", xt)):
    e.residual_pass(x, True)
    d = np.zeros((e.n, 4), np.uint32)
    assert e.lib.s2m_debug_match(e.h, C.c_void_p(d.ctypes.data)) == 0
    hard = (d[:, 3] & 0xff) > 1
    idx, d2 = e.get_neighbors()
    n = e.n
    w = hard[: n // 32 * 32].reshape(-1, 32)
    per = w.sum(1)
    print("%s %s: hard %d of %d (%.3f); waves with any hard point %.3f; hard points per such wave: mean %.1f; waves all-hard %.3f" % (
        cfg, name, hard.sum(), n, hard.mean(), (per > 0).mean(), per[per > 0].mean(), (per == 32).mean()))
    hist = np.bincount(per, minlength=33)
    print("   waves by number of hard points: 0:%d 1-4:%d 5-16:%d 17-31:%d 32:%d" % (hist[0], hist[1:5].sum(), hist[5:17].sum(), hist[17:32].sum(), hist[32]))
    # what a radius-2 shell would settle: d5 inside (2 + fmin) cells -- fmin unknown here, use the lower bound 2 cells
    d5 = d2[:, 4]
    r2_ok = hard & np.isfinite(d5) & (d5 <= (2.0 * cell) ** 2)
    r3_ok = hard & np.isfinite(d5) & (d5 <= (3.0 * cell) ** 2)
    print("   of the hard points: final d5 within 2 cells %.3f, within 3 cells %.3f, beyond the gate or missing %.3f" % (
        r2_ok.sum() / max(hard.sum(), 1), r3_ok.sum() / max(hard.sum(), 1), (hard & ~(d5 <= 5.0)).sum() / max(hard.sum(), 1)))
e.close()
