"""GPU occupancy of a run from a rocprofv3 --kernel-trace CSV: how much of the wall span has a kernel running, how
many run side by side, and each kernel's share.
usage: timeline.py <dir with *_kernel_trace.csv> [skip_fraction] [name filter, e.g. _batch]"""
import collections
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    lo = t0 + int((t1 - t0) * skip)          # skip set-up (map build, warm-up)
    flt = sys.argv[3] if len(sys.argv) > 3 else ""
    rows = [r for r in rows if r[0] >= lo and "s2m::" in r[2] and flt in r[2]]
    if flt:  # the longest stretch without a gap of more than 2 ms between such kernels (one leg of the bench)
        best, cur = [], [rows[0]]
        for r in rows[1:]:
            if r[0] - max(x[1] for x in cur[-8:]) > 2000000:
                if len(cur) > len(best):
                    best = cur
                cur = []
            cur.append(r)
        rows = cur if len(cur) > len(best) else best
    span = max(r[1] for r in rows) - rows[0][0]
    ev = []
    for s, e, _ in rows:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    busy = 0; depth = 0; last = ev[0][0]; hist = collections.Counter()
    for t, k in ev:
        hist[depth] += t - last
        if depth > 0:
            busy += t - last
        depth += k; last = t
    per = collections.defaultdict(lambda: [0, 0])
    for s, e, n in rows:
        k = n.split("(")[0].replace("void s2m::", "").replace("s2m::", "")[:40]
        per[k][0] += 1; per[k][1] += e - s
    print("span %.3f ms, some kernel running %.1f %%, kernels summed %.3f ms (mean concurrency while busy %.2f)" % (
        span / 1e6, 100.0 * busy / span, sum(v[1] for v in per.values()) / 1e6, sum(v[1] for v in per.values()) / max(busy, 1)))
    print("concurrency histogram (share of span): " + ", ".join("%d: %.1f %%" % (k, 100.0 * v / span) for k, v in sorted(hist.items())))
    for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print("%-42s calls %6d  avg %8.2f us  sum %8.3f ms" % (k, c, t / c / 1e3, t / 1e6))


if __name__ == "__main__":
    main()
