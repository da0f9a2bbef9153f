"""Dev experiment (build with -DS2M_EXP_REDUCE_TIMELINE): where the reduce kernel's microseconds go, from wall-clock
stamps of thread 0 of the last workgroup."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daliti_amd import Engine, synth
c = synth.CONFIGS["C3"]
m = synth.make_map(c["M"], c["L"]); s = synth.make_scan(c["beams"], c["az"], c["L"])
_, xp, P = synth.filter_inputs()
e = Engine(max_iter=5)
e.map_build(m); e.scan_set(s)
blk = torch.zeros(160, dtype=torch.float64, device="cuda")
names = ["start->loads", "loads->row", "row->butterfly+LDS", "->partial row stored", "->ticket known", "->rows summed", "->published/end"]
for fit in (True, False):
    acc = []
    for _ in range(30):
        e.residual_pass_device(xp, fit, blk.data_ptr())
        torch.cuda.synchronize()
        b = blk.cpu().numpy()
        lo, hi = int(b[158]), int(b[159])
        acc.append([(lo >> (12 * k)) & 0xfff for k in range(4)] + [(hi >> (12 * k)) & 0xfff for k in range(3)])
    a = np.median(np.array(acc[5:]), 0) / 100.0
    print("reduce%s: " % ("<FIT>" if fit else "     ") + "  ".join("%s %.2f" % (n, v) for n, v in zip(names, a)) + "  | sum %.2f us" % a.sum())
