#!/usr/bin/env python3
"""Would a spatially blocked order of a voxel-grid scan make the search kernels faster?  (GPU box)  The voxel grid emits
points by voxel index (x fastest): a wave's 32 queries lie along lines.  Time the iterated update of the same points in
that order and sorted by 2 m / 4 m blocks (Morton order inside) -- the result differs in summation order only."""
import sys, time
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import numpy as np
import torch
from daliti_amd import Engine, synth

c = synth.CONFIGS["C3"]
m = synth.make_map(c["M"], c["L"], seed=1)
scan = synth.make_scan(c["beams"], c["az"], c["L"], seed=2)
_xt, x_prop, P0 = synth.filter_inputs(synth.SENSOR_POS)
e = Engine(max_iter=5, feat_threshold=100)
e.map_build(m)
e.scan_set_downsampled(scan, 0.5)
base = e.scan_get().copy()

def morton(q):
    q = q.astype(np.uint64)
    out = np.zeros(len(q), np.uint64)
    for b in range(10):
        for a in range(3):
            out |= ((q[:, a] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
    return out

def timed(pts, label):
    e.scan_set(pts)
    ts = []
    for k in range(30):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = e.iterated_update(x_prop, x_prop, P0)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("%-34s %d points: iterated update median %.3f ms (min %.3f), iters %d effct %s" % (label, len(pts), np.median(ts[5:]), min(ts[5:]), r["iters"], r["effct"][:2]))

timed(base, "voxel-grid order")
for blk in (1.0, 2.0, 4.0):
    cell = np.floor((base - base.min(axis=0)) / blk).astype(np.int64)
    order = np.argsort(morton(cell), kind="stable")
    timed(np.ascontiguousarray(base[order]), "blocks of %.0f m, Morton order" % blk)
rs = np.random.RandomState(0)
timed(np.ascontiguousarray(base[rs.permutation(len(base))]), "random order")
timed(scan, "raw beam order (65 k)")
