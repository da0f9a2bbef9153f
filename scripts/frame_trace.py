"""Where a frame's time goes on the GPU: the tail of a rocprofv3 --kernel-trace of `bench.py` (its last leg is the
back-to-back frame loop).  usage: frame_trace.py <dir with *_kernel_trace.csv> [frames] [ms per frame]"""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.77
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t1 = max(r[1] for r in rows)
lo = t1 - int(frames * ms * 1e6 * 0.9)          # inside the back-to-back leg
rows = [r for r in rows if r[0] >= lo]
span = t1 - rows[0][0]
busy = 0; last_end = rows[0][0]
for s, e, _ in rows:
    if e > last_end:
        busy += e - max(s, last_end); last_end = e
per = collections.defaultdict(lambda: [0, 0])
for s, e, n in rows:
    k = n.split("(")[0].replace("void ", "")
    k = k.split("<")[0][-48:] if "rocprim" in k else k[:60]
    per[k][0] += 1; per[k][1] += e - s
nf = span / (ms * 1e6)
print("window %.2f ms = %.1f frames; GPU busy %.1f %%; %.1f kernel launches per frame, %.1f us of kernel time per frame" % (
    span / 1e6, nf, 100.0 * busy / span, len(rows) / nf, sum(v[1] for v in per.values()) / nf / 1e3))
for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-62s %5.1f per frame  avg %7.2f us  %7.1f us per frame" % (k, c / nf, t / c / 1e3, t / nf / 1e3))
