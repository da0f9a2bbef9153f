import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from daliti_amd import Engine, synth
from daliti_amd.world import World
w = World(215.0, 1290.0, 1.0)
sw = w.sweeps(0, 320, 64, 1024, threads=16)
seed = w.seed_map(5000000)
_, _, P0 = synth.filter_inputs()
e = Engine(max_iter=5)
e.map_build(seed)
for f in range(320):
    n = int(sw["n"][f])
    rec, poses, xp = sw["rec"][f][:n], sw["poses"][f], sw["x_prop"][f]
    nd = e.scan_set_from_raw(rec, 4, 6, poses, xp, 0.5)
    if f in (5, 100, 200, 300):
        import torch
        d = torch.zeros(160, dtype=torch.float64, device="cuda")
        e.residual_pass_device(xp, True, d.data_ptr())
        torch.cuda.synchronize()
        blk = d.cpu().numpy()
        print("frame", f, "scan", nd, "effct", int(blk[156]), "far points", int(blk[158]), "short lists", int(blk[159]))
    got = e.iterated_update(xp, xp, P0)
    e.map_incremental(got["x"], 0.5)
    e.fov_segment(got["x"][9:12], 1000.0)
