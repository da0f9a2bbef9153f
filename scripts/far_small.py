"""dev probe: far points of the small test scene's halves at a tilted predicted pose (is the bet lost there?)"""
import os, sys, ctypes as C
import numpy as np
os.environ["S2M_DEBUG_MATCH"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from daliti_amd import Engine, synth
sc = synth.make_small()
for deg in (0.0, 2.0, 4.0, 6.0):
    a = np.deg2rad(deg)
    Ry = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    xp = sc["x_prop"].copy()
    xp[:9] = (xp[:9].reshape(3, 3) @ Ry).ravel()
    for lo, hi in ((0, 1024), (1024, 2048)):
        e = Engine(max_iter=5)
        e.map_build(sc["map"]); e.scan_set(sc["scan"][lo:hi])
        e.residual_pass(xp, True)
        d = np.zeros((e.n, 4), np.uint32)
        assert e.lib.s2m_debug_match(e.h, C.c_void_p(d.ctypes.data)) == 0
        far = int(((d[:, 3] & 0xff) > 1).sum())
        r = e.iterated_update(xp, xp, sc["P"])
        print(deg, lo, hi, "far points:", far, "iters", r["iters"], "effct", list(r["effct"]), flush=True)
        e.close()
