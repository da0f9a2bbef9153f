"""dev probe: how many scan points go to the far-point kernel in a rematch pass at the CONVERGED pose"""
import os, sys, ctypes as C
import numpy as np
os.environ["S2M_DEBUG_MATCH"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from daliti_amd import Engine, synth
for cfg in ("C1", "C2", "C3", "C4"):
    c = synth.make_config(cfg)
    e = Engine(max_iter=5)
    e.map_build(c["map"]); e.scan_set(c["scan"])
    r = e.iterated_update(c["x_prop"], c["x_prop"], c["P"])
    # the pose of the LAST rematch pass of the update = the state before the last solution was applied; the final state is
    # within a millimetre of it
    for name, x in (("predicted", c["x_prop"]), ("converged", r["x"])):
        e.residual_pass(x, True)
        d = np.zeros((e.n, 4), np.uint32)
        assert e.lib.s2m_debug_match(e.h, C.c_void_p(d.ctypes.data)) == 0
        hard = (d[:, 3] & 0xff) > 1
        print(cfg, name, "far points:", int(hard.sum()), "of", e.n, flush=True)
    e.close()
