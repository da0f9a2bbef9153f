"""Dev tool: time the match / reduce kernels over group width and cell size (HIP events)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from daliti_amd import Engine, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
c = synth.CONFIGS[cfg]
m = synth.make_map(c["M"], c["L"])
s = synth.make_scan(c["beams"], c["az"], c["L"])
xt, xp, P = synth.filter_inputs()
groups = [int(g) for g in os.environ.get("GROUPS", "1,2,4,8").split(",")]
cells = [float(g) for g in os.environ.get("CELLS", "0.2,0.25,0.3,0.35,0.5").split(",")]
hgs = [int(g) for g in os.environ.get("HARD", "32").split(",")]
for g, hg in [(g, hg) for g in groups for hg in hgs]:
    for cell in cells:
        os.environ["S2M_MATCH_GROUP"] = str(g)
        os.environ["S2M_HARD_GROUP"] = str(hg)
        e = Engine(cell_size=cell, max_iter=5)
        e.map_build(m)
        e.scan_set(s)
        info = e.map_info()
        for _ in range(3):
            e.residual_pass(xp, True)
        e.set_timing(2)
        for _ in range(20):
            e.residual_pass(xp, True)
        st = e.timing_stats()
        out = e.residual_pass(xp, True)
        e.set_timing(2)
        for _ in range(20):
            e.residual_pass(xt, True)
        st2 = e.timing_stats()
        print("hard %2d " % hg + "group %2d cell %.3f pts/cell %.2f bricks %6d  match(prop) %.1f us  match(true pose) %.1f us  reduce<FIT> %.1f us  effct %d" % (
            g, info["cell"], info["mean_per_cell"], info["bricks"], 1e3 * st["match_ms"] / st["match_launches"],
            1e3 * st2["match_ms"] / st2["match_launches"],
            1e3 * st["fit_ms"] / st["fit_launches"], out["effct"]), flush=True)
        e.close()
