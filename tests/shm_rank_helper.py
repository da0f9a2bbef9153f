"""One rank of the shared-memory exchange test (tests/test_sharding.py): registers its shard of the small scene with the
far-point bet given on the command line, writes the resulting state.  usage: shm_rank_helper.py NAME NRANKS RANK BET OUT
The scene is C2 (65,536-point scan vs 1 M-point map): its first pass has thousands of far points, so a rank that always
bets loses there."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from daliti_amd import Engine, synth  # noqa: E402
from daliti_amd.sharding import shard_range  # noqa: E402

name, nranks, rank, bet, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
sc = synth.make_config("C2")
# device_loop=0: the unsplit reference run (NRANKS = 1) is host-stepped like the ranks of an exchange always are, so that
# "aligned shards reproduce the unsplit scan bit for bit" compares like with like
e = Engine(max_iter=5, device=0, far_point_bet=bet, device_loop=0)
e.map_build(sc["map"])
lo, hi = shard_range(len(sc["scan"]), rank, nranks)
e.scan_set(sc["scan"][lo:hi])
if nranks > 1:
    e.comm_init_shm(name, nranks, rank)
res = []
for _ in range(int(os.environ.get("S2M_HELPER_SCANS", "3"))):   # scans in a row: the bet's history moves between them
    e.set_feat_queue(())
    r = e.iterated_update(sc["x_prop"], sc["x_prop"], sc["P"])
    eff = np.zeros(8)
    eff[:r["iters"]] = r["effct"]   # the per-iteration effective counts are the JOB's (summed over the ranks)
    res.append(np.r_[r["x"], r["P"].ravel(), r["iters"], eff, r["total_res"].sum()])
np.save(out, np.array(res))
e.close()
