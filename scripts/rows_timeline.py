"""Dev experiment (build with -DS2M_EXP_ROWS_TIMELINE, run with S2M_DEBUG_MATCH=1): where match_rows' microseconds go."""
import os, sys, ctypes as C
import numpy as np
os.environ["S2M_DEBUG_MATCH"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from daliti_amd import Engine, synth
c = synth.CONFIGS["C3"]
m = synth.make_map(c["M"], c["L"]); s = synth.make_scan(c["beams"], c["az"], c["L"])
xt, xp, P = synth.filter_inputs()
e = Engine()
e.map_build(m); e.scan_set(s)
for name, x in (("first pass (x_prop)", xp), ("later pass (true pose)", xt)):
    for _ in range(3): e.residual_pass(x, True)
    d = np.zeros((e.n, 4), np.uint32)
    assert e.lib.s2m_debug_match(e.h, C.c_void_p(d.ctypes.data)) == 0
    easy = d[:, 1] == 1          # resolved by match_rows (hard points overwrite the words)
    w2, w3 = d[easy, 2], d[easy, 3]
    t = np.stack([w2 & 0xffff, w2 >> 16, w3 & 0xffff, w3 >> 16, d[easy, 0] & 0xffff], 1).astype(float) / 100.0
    names = ["query ready", "brick ids", "home row done (tau)", "trimmed rows' words", "end"]
    print(name)
    for k, n in enumerate(names):
        print("   %-22s at p50 %.2f  p90 %.2f  max %.2f us (stage p50 %.2f)" % (n, *np.percentile(t[:, k], [50, 90, 100]), np.median(t[:, k] - (t[:, k - 1] if k else 0))))
