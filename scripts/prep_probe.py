import sys, time
sys.path.insert(0, ".")
import numpy as np
from daliti_amd import Engine, synth
cfgd = synth.CONFIGS["C1"]
m = synth.make_map(cfgd["M"], cfgd["L"], seed=1)
scan = synth.make_scan(64, 1024, cfgd["L"], seed=2)
n = len(scan)
rec = np.zeros((n, 12), np.float32); rec[:, :3] = scan; rec[:, 4] = np.linspace(0, 1, n, dtype=np.float32); rec[:, 6] = 0.1
K = 20
poses = np.zeros((K, 22)); poses[:, 0] = np.linspace(0.0, 0.102, K); poses[:, 13:22] = np.eye(3).ravel()
end = np.zeros(36); end[0:9] = np.eye(3).ravel(); end[12:21] = np.eye(3).ravel()
e = Engine(); e.map_build(m)
for i in range(3):
    t0 = time.perf_counter(); e.scan_set_from_raw(rec, 4, 6, poses, end, 0.5); t1 = time.perf_counter()
    print("in place %.3f ms" % ((t1 - t0) * 1e3))
for i in range(3):
    e.scan_prepare_raw(rec, 4, 6, poses, end, 0.5); time.sleep(0.01)
    t0 = time.perf_counter(); e.scan_set_from_raw(rec, 4, 6, poses, end, 0.5); t1 = time.perf_counter()
    print("prepared %.3f ms" % ((t1 - t0) * 1e3))
