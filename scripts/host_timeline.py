"""dev probe: S2M_HOST_TIMELINE=1 -- launch / wait / solve microseconds per ESKF iteration of one C3 update"""
import os, sys
os.environ["S2M_HOST_TIMELINE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from daliti_amd import Engine, synth
from daliti_amd.engine import IterLog
c = synth.make_config("C3")
e = Engine(max_iter=5)
e.map_build(c["map"]); e.scan_set(c["scan"])
x = np.zeros(36); P = np.zeros((24, 24)); log = IterLog()
call = e.iterated_update_bound(x, np.ascontiguousarray(c["x_prop"]), P, log)
for k in range(30):
    e.set_feat_queue(()); x[:] = c["x_prop"]; P[:] = c["P"]; P[0, 0] += (k & 1) * 1e-15
    if k == 27: sys.stderr.write("---- step 27\n")
    call()
