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
marks_ = sum(1 for r in rows if "incr_classify_kernel" in r[2])   # (one per frame of a frame loop: the honest count under a profiler)
if marks_ > 8:
    nf = float(marks_)
print("window %.2f ms = %.1f frames; GPU busy %.1f %%; %.1f kernel launches per frame, %.1f us of kernel time per frame" % (
    span / 1e6, nf, 100.0 * busy / span, len(rows) / nf, sum(v[1] for v in per.values()) / nf / 1e3))
for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-62s %5.1f per frame  avg %7.2f us  %7.1f us per frame" % (k, c / nf, t / c / 1e3, t / nf / 1e3))

# one frame as a sequence (optional 4th argument "seq"): from one incr_classify_kernel to the next, every kernel with its
# queue, start relative to the frame, duration and the idle gap since the previous kernel END on any queue
if len(sys.argv) > 4 and sys.argv[4] == "seq":
    full = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            full.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
    full.sort()
    marks = [i for i, r in enumerate(full) if "incr_classify_kernel" in r[2]]
    if len(marks) > 12:
        a, b = marks[-10], marks[-9]
        t0 = full[a][0]; last_end = t0
        print("\none frame, classify to classify: %.1f us" % ((full[b][0] - t0) / 1e3))
        for s, e, n, q in full[a:b]:
            k = n.split("(")[0].replace("void ", "")
            k = ("rocprim::" + k.split("::")[-1].split("<")[0]) if "rocprim" in k else k[:44]
            print("  q%-3s +%7.1f us  %6.1f us  gap %5.1f  %s" % (q[-3:], (s - t0) / 1e3, (e - s) / 1e3, max(0, s - last_end) / 1e3, k))
            last_end = max(last_end, e)

# a field-of-view trim as a sequence (4th argument "trim"): from the longest delete_boxes_kernel to the next classify
if len(sys.argv) > 4 and sys.argv[4] == "trim":
    full = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            full.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
    full.sort()
    dels = [i for i, r in enumerate(full) if "delete_boxes_kernel" in r[2]]
    print("\n%d delete_boxes_kernel launches" % len(dels))
    for a in sorted(dels, key=lambda i: full[i][0] - full[i][1])[:2]:
        b = next(i for i in range(a, len(full)) if "incr_classify_kernel" in full[i][2])
        t0 = full[a][0]; last_end = t0
        print("trim, delete to the next classify: %.1f us" % ((full[b][0] - t0) / 1e3))
        for s, e, n, q in full[max(a - 3, 0):b]:
            k = n.split("(")[0].replace("void ", "")
            k = ("rocprim::" + k.split("::")[-1].split("<")[0]) if "rocprim" in k else k[:44]
            if "q3" == "q" + q[-1:] and False:
                continue
            print("  q%-3s +%7.1f us  %6.1f us  gap %5.1f  %s" % (q[-3:], (s - t0) / 1e3, (e - s) / 1e3, max(0, s - last_end) / 1e3, k))
            last_end = max(last_end, e)
