"""Every kernel of every queue in a window around the n-th layout's snapshot (rocprofv3 --kernel-trace of scripts/relay_check.py):
python scripts/kernel_window.py <dir> [which snapshot] [ms before] [ms after]"""
import csv, glob, os, sys
d = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 1
before = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
after = float(sys.argv[4]) if len(sys.argv) > 4 else 2.0
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("s2m::", "").replace("(anonymous namespace)::", "").split("(")[0][:int(os.environ.get("NAMELEN", 48))]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "?")))
rows.sort()
snaps = [r for r in rows if r[2].startswith("snapshot_count_kernel")]
t0 = snaps[which][0]
qs = sorted(set(r[3] for r in rows if t0 - before * 1e6 <= r[0] <= t0 + after * 1e6))
print("queues in the window:", qs, "(the snapshot's: %s)" % snaps[which][3])
last_end = {}
for s, e, n, q in rows:
    if t0 - before * 1e6 <= s <= t0 + after * 1e6:
        gap = (s - last_end[q]) * 1e-3 if q in last_end else float("nan")
        print("%+9.3f ms  q%-3s %7.1f us  (gap %7.1f us)  %s" % ((s - t0) * 1e-6, q, (e - s) * 1e-3, gap, n))
    last_end[q] = e
