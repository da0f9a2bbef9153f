"""Dev tool: wall time of map build / incremental update / box delete at C3 (warm buffers)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daliti_amd import Engine, synth

cfg = os.environ.get("CONFIG", "C3")
c = synth.CONFIGS[cfg]
m = synth.make_map(c["M"], c["L"]); s = synth.make_scan(c["beams"], c["az"], c["L"])
xt, xp, P = synth.filter_inputs()
d_map = torch.from_numpy(m).cuda()
e = Engine(max_iter=5, feat_threshold=100)
def sync(): torch.cuda.synchronize()
for rep in range(4):
    sync(); t0 = time.perf_counter(); e.map_build_device(d_map.data_ptr(), 3, len(m)); sync()
    print("map_build (%d pts, rep %d): %.3f ms" % (len(m), rep, (time.perf_counter() - t0) * 1e3))
e.scan_set(s)
for rep in range(4):
    x = e.iterated_update(xp, xp, P)["x"]
    sync(); t0 = time.perf_counter(); na, nb = e.map_incremental(x, 0.5); sync()
    print("map_incremental rep %d: %.3f ms (to_add %d, no_downsample %d, map now %d)" % (rep, (time.perf_counter() - t0) * 1e3, na, nb, e.map_size()))
    e.scan_set(s)
boxes = np.array([[-5, -5, -1, 5, 5, 3]], np.float32)
sync(); t0 = time.perf_counter(); nd = e.map_delete_boxes(boxes); sync()
print("map_delete_boxes: %.3f ms (%d deleted)" % ((time.perf_counter() - t0) * 1e3, nd))
d_scan = torch.from_numpy(s).cuda()
for rep in range(3):
    sync(); t0 = time.perf_counter(); n2 = e.scan_set_downsampled(s, 0.5); sync()
    print("scan_set_downsampled (host input) rep %d: %.3f ms -> %d" % (rep, (time.perf_counter() - t0) * 1e3, n2))
