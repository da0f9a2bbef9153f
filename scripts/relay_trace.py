"""What runs on the device beside a layout (s2m_engine_relay.cpp), from a rocprofv3 --kernel-trace CSV of a drive
(ONLY=beside rocprofv3 --kernel-trace --output-format csv -d <dir> -- python scripts/relay_check.py 1100).
For every layout (a snapshot_count_kernel): the kernels of the layout's queue with their start (ms behind the snapshot) and
duration, and the frames' kernels in that window against their own medians over the whole drive.
usage: relay_trace.py <dir> [window_ms]"""
import collections
import csv
import glob
import os
import sys

import numpy as np


def short(n):
    n = n.replace("s2m::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:64]


def main():
    d = sys.argv[1]
    win = float(sys.argv[2]) if len(sys.argv) > 2 else 25.0
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")))
    rows.sort()
    snaps = [r for r in rows if r[2].startswith("snapshot_count_kernel")]
    print("%d kernels, %d snapshots" % (len(rows), len(snaps)))
    dur = collections.defaultdict(list)
    for s, e, n, q in rows:
        dur[n].append((e - s) * 1e-3)
    med = {n: float(np.median(v)) for n, v in dur.items()}
    for k, sn in enumerate(snaps[1:] if len(snaps) > 2 else snaps):    # (the first one is the rehearsal's, if it shows)
        t0, q_lay = sn[0], sn[3]
        inside = [r for r in rows if t0 - 0.5e6 <= r[0] <= t0 + win * 1e6]
        lay = [r for r in inside if r[3] == q_lay]
        oth = [r for r in inside if r[3] != q_lay]
        print("\n== layout %d: queue %s; %d kernels of the layout, %d of the frames within %.0f ms" % (k, q_lay, len(lay), len(oth), win))
        tot = 0.0
        for s, e, n, q in lay:
            tot += (e - s) * 1e-3
            if (e - s) > 20000:
                print("   +%8.3f ms  %8.1f us  %s" % ((s - t0) * 1e-6, (e - s) * 1e-3, n))
        if lay:
            print("   the layout's kernels: %.2f ms of device time between +%.3f and +%.3f ms" % (tot * 1e-3, (lay[0][0] - t0) * 1e-6, (max(r[1] for r in lay) - t0) * 1e-6))
        # the frames' kernels per ms of the window: time of kernels / their median
        agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
        for s, e, n, q in oth:
            a = agg[n]
            a[0] += 1
            a[1] += (e - s) * 1e-3
            a[2] += med[n]
        print("   the frames' kernels in the window (count, time, time at their drive medians):")
        for n, a in sorted(agg.items(), key=lambda kv: -(kv[1][1] - kv[1][2]))[:14]:
            print("     %-64s %5d  %9.1f us  %9.1f us  x%.2f" % (n, a[0], a[1], a[2], a[1] / max(a[2], 1e-9)))
        # per millisecond: slowdown of the frames' kernels, and what the layout ran
        print("   per ms behind the snapshot: frames' kernel time / median time | the layout's longest kernel starting there")
        for ms in range(0, int(win)):
            lo, hi = t0 + ms * 1e6, t0 + (ms + 1) * 1e6
            o = [r for r in oth if lo <= r[0] < hi]
            if not o:
                continue
            a = sum((e - s) * 1e-3 for s, e, n, q in o)
            b = sum(med[n] for s, e, n, q in o)
            ll = [r for r in lay if r[0] < hi and r[1] > lo]
            top = max(ll, key=lambda r: r[1] - r[0]) if ll else None
            print("     +%2d ms  x%.2f (%4d kernels)  | %s" % (ms, a / max(b, 1e-9), len(o), ("%s %.0f us" % (top[2], (top[1] - top[0]) * 1e-3)) if top else "-"))


if __name__ == "__main__":
    main()
