#!/usr/bin/env python3
"""map_incremental at C4's map size (20 M points) beside C3's (5 M): what the map-proportional part of the merge update
costs.  GPU box: python3 scripts/map_incr_20m.py"""
import sys, time
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import numpy as np
import torch
from daliti_amd import Engine, synth

for name in ("C3", "C4"):
    c = synth.CONFIGS[name]
    m = synth.make_map(c["M"], c["L"], seed=1)
    scan = synth.make_scan(64, 1024, c["L"], seed=2)
    _xt, x_prop, P0 = synth.filter_inputs(synth.SENSOR_POS)
    e = Engine(max_iter=5, feat_threshold=100)
    e.map_build(m)
    e.scan_set_downsampled(scan, 0.5)
    ts = []
    for f in range(12):
        r = e.iterated_update(x_prop, x_prop, P0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.map_incremental(r["x"], 0.5)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print("%s: map %d points, map_incremental median %.3f ms (min %.3f) over frames 2..11; merged %s, %d of 12 updates in place" % (
        name, e.map_size(), float(np.median(ts[2:])), min(ts[2:]), e.map_last_update_merged(), e.map_inplace_updates()))
    e.close()
