"""HIP API calls that took long around the start of a layout beside the frames, from a rocprofv3 --hip-trace --kernel-trace run
of scripts/relay_check.py: python scripts/api_trace_slow.py <dir> [threshold_us] [window_ms]"""
import csv, glob, os, sys, collections
d = sys.argv[1]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
win = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
api, ker = [], []
for f in glob.glob(os.path.join(d, "**", "*hip_api_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        api.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], r.get("Thread_Id", "?")))
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ker.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
api.sort(); ker.sort()
snaps = [k for k in ker if "snapshot_count_kernel" in k[2]]
print("%d api calls, %d kernels, %d snapshots" % (len(api), len(ker), len(snaps)))
threads = collections.Counter(a[3] for a in api)
print("threads:", dict(threads))
for k, sn in enumerate(snaps):
    t0 = sn[0]
    print("\n== snapshot %d; api calls longer than %.0f us within %.1f ms before / after its first kernel (ms relative to it, us, thread, call)" % (k, thr, win))
    for s, e, fn, th in api:
        if t0 - win * 1e6 <= s <= t0 + win * 1e6 and (e - s) * 1e-3 >= thr:
            print("   %+8.3f ms  %8.1f us  %s  %s" % ((s - t0) * 1e-6, (e - s) * 1e-3, th, fn))

# launch calls per thread inside the snapshot's first 1.2 ms against the 3 ms before it: is a launch slower on the HOST beside a layout?
import statistics
for k, sn in enumerate(snaps):
    t0 = sn[0]
    for name, lo, hi in (("before", t0 - 3.2e6, t0 - 0.2e6), ("beside", t0 + 0.15e6, t0 + 1.2e6)):
        per = collections.defaultdict(list)
        for s, e, fn, th in api:
            if lo <= s <= hi and fn in ("hipLaunchKernel", "hipModuleLaunchKernel", "hipExtModuleLaunchKernel", "hipLaunchKernelGGL"):
                per[th].append((e - s) * 1e-3)
        print("snapshot %d, %s: " % (k, name) + "; ".join("thread %s: %d launches, median %.1f us, mean %.1f, max %.1f" % (th, len(v), statistics.median(v), sum(v) / len(v), max(v)) for th, v in sorted(per.items())))
