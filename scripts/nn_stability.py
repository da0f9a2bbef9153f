#!/usr/bin/env python3
"""How much of the neighbour search of a later rematch pass is already known from the pass before it?
GPU box: python3 scripts/nn_stability.py [C3|R1|C1|C2] [imu]
Prints, for the two rematch poses of the iterated update of the config (x_prop and the converged pose): the share of
points whose 5-NN SET is unchanged, whose ORDERED list is unchanged, and how tight the bound "farthest old neighbour
seen from the new query" is against the true 5th distance (the warm-start lever, NOTEBOOK.md round 4)."""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from daliti_amd import Engine, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfgd = synth.CONFIGS[name]
map_xyz = synth.make_map(cfgd["M"], cfgd["L"], seed=1)
scan = synth.make_scan(cfgd["beams"], cfgd["az"], cfgd["L"], seed=2)
# third argument "imu": an IMU-sized prediction error (2 cm, 0.1 deg) instead of BASELINE's 5 cm / 1 deg
if len(sys.argv) > 2 and sys.argv[2] == "imu":
    _xt, x_prop, P0 = synth.filter_inputs(synth.SENSOR_POS, dtheta=np.deg2rad(0.1) * np.array([0.6, -0.5, 0.62]), dpos=0.02 * np.array([0.66, -0.53, 0.53]))
else:
    _xt, x_prop, P0 = synth.filter_inputs(synth.SENSOR_POS)
e = Engine(max_iter=5, feat_threshold=100)
if name == "R1":
    import bench
    bench.build_reference_density_map(e, map_xyz)
    e.scan_set_downsampled(scan, 0.5)
else:
    e.map_build(map_xyz)
    e.scan_set(scan)
x = np.array(x_prop, copy=True); P = np.array(P0, copy=True)
res = e.iterated_update(x, x_prop, P)
xc = res["x"]
print("iters", res["iters"], "rematch", res["rematch"], "effct", res["effct"])
e.residual_pass(np.asarray(x_prop), True)
i0, d0 = e.get_neighbors()
e.residual_pass(np.asarray(xc), True)
i1, d1 = e.get_neighbors()
n = len(i0)
same_list = (i0 == i1).all(axis=1)
same_set = (np.sort(i0, axis=1) == np.sort(i1, axis=1)).all(axis=1)
print("%s: %d points; ordered list unchanged %.1f %%, set unchanged %.1f %%" % (name, n, 100 * same_list.mean(), 100 * same_set.mean()))
for w in (64, 256, 1024):
    k = n // w
    print("  groups of %4d consecutive points with every list unchanged: %.1f %%; mean changed per group %.1f" %
          (w, 100 * same_list[:k * w].reshape(k, w).all(axis=1).mean(), (~same_list[:k * w]).reshape(k, w).sum(axis=1).mean()))
# the bound: farthest OLD neighbour from the NEW query vs the true new 5th distance
pts = e.map_points()
st = synth  # noqa
from daliti_amd.engine import Engine as _E  # noqa
import oracle  # checker-side helper only (this is a lab script)
w1 = oracle.body_to_world(np.asarray(xc), np.asarray(e.scan_get(), np.float32))
if True:
    ok = (i0 >= 0).all(axis=1)
    old = pts[i0[ok]]
    dd = ((old - w1[ok][:, None, :]) ** 2).sum(axis=2).max(axis=1)
    chk = ((pts[i1[ok]][:, 4, :] - w1[ok]) ** 2).sum(axis=1)
    print("  sanity: max |recomputed d5^2 - reported| = %.3g" % np.abs(chk - d1[ok][:, 4]).max())
    r = np.sqrt(dd / d1[ok][:, 4])
    print("  bound / true 5th distance: median %.3f, p90 %.3f, p99 %.3f" % (np.median(r), np.quantile(r, 0.9), np.quantile(r, 0.99)))
