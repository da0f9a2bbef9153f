"""Dev tool: reduce-kernel time vs scan size (HIP events around the launch, reuse passes)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from daliti_amd import Engine, synth
c = synth.CONFIGS["C3"]
m = synth.make_map(c["M"], c["L"]); s = synth.make_scan(c["beams"] * 4, c["az"], c["L"])
_, xp, P = synth.filter_inputs()
e = Engine(max_iter=5, feat_threshold=100)
e.map_build(m)
for n in (64, 512, 4096, 16384, 65536, 131072, 262144):
    e.scan_set(s[:n])
    e.residual_pass(xp, True)
    e.set_timing(2)
    ts = []
    for _ in range(30):
        e.residual_pass(xp, False)
        ts.append(e.timing()[1])
    tf = []
    for _ in range(10):
        e.residual_pass(xp, True)
        tf.append(e.timing()[1])
    e.set_timing(0)
    print("n %7d  blocks %4d  reduce %.2f us (min %.2f)   reduce<FIT> %.2f us" % (n, (n + 511) // 512, np.median(ts) * 1e3, np.min(ts) * 1e3, np.median(tf) * 1e3))
