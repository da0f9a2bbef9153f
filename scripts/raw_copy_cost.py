"""dev probe: what does the host -> device copy of the raw sweep cost inside s2m_scan_set_from_raw? (host records vs records
already on the device)"""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daliti_amd import Engine, synth
c = synth.make_config("C3")
e = Engine(max_iter=5)
e.map_build(c["map"])
n = len(c["scan"])
rec = np.zeros((n, 12), np.float32); rec[:, :3] = c["scan"]; rec[:, 4] = np.linspace(0, 1, n, dtype=np.float32); rec[:, 6] = 0.1
K = 20
poses = np.zeros((K, 22)); poses[:, 0] = np.linspace(0.0, 0.102, K); poses[:, 13:22] = np.eye(3).ravel()
end = np.zeros(36); end[0:9] = np.eye(3).ravel(); end[12:21] = np.eye(3).ravel()
d_rec = torch.from_numpy(rec).cuda(); torch.cuda.synchronize()
m = C.c_int64()
def call(ptr, on_dev):
    rc = e.lib.s2m_scan_set_from_raw(e.h, C.c_void_p(ptr), C.c_int64(12), C.c_int64(n), C.c_int32(4), C.c_int32(6),
                                     C.c_void_p(poses.ctypes.data), C.c_int32(K), C.c_void_p(end.ctypes.data), C.c_float(0.5), on_dev, C.byref(m))
    assert rc == 0
for name, ptr, od in (("host records", rec.ctypes.data, 0), ("device records", d_rec.data_ptr(), 1), ("host records", rec.ctypes.data, 0)):
    for _ in range(5): call(ptr, od)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): call(ptr, od); torch.cuda.synchronize()
    print("%-16s %.3f ms per s2m_scan_set_from_raw (+sync), %d points out" % (name, (time.perf_counter() - t0) * 20, m.value), flush=True)
t0 = time.perf_counter()
for _ in range(50):
    a = np.ascontiguousarray(rec[:, [0, 1, 2, 4, 6]])
print("numpy gather of 5 of 12 floats: %.3f ms" % ((time.perf_counter() - t0) * 20))
