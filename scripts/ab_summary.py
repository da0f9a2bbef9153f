"""Summarise A/B directories under gpurun_out/ (bench JSON lines + rocprofv3 counter CSVs) as text for profiles/.
usage: python scripts/ab_summary.py <dir> [<dir> ...]"""
import collections, csv, glob, json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for tag in sys.argv[1:]:
    d = os.path.join(root, "gpurun_out", tag)
    print("== %s" % tag)
    for f in sorted(glob.glob(os.path.join(d, "*.json"))):
        try:
            b = json.loads(open(f).read().strip().splitlines()[-1]); r = b.get("roofline", {})
            print("%-20s ms/step %.4f  scans/s %6.0f  rematch pass %.1f us (search kernels %.1f)" % (
                os.path.basename(f)[:-5], b["ms_per_step"], b["scans_per_sec"], 1e3 * (r.get("avg_launch_ms") or 0),
                1e3 * ((r.get("search_kernels_only") or {}).get("avg_ms") or 0)))
        except Exception:
            pass
    for sub in sorted(glob.glob(os.path.join(d, "C*_*"))):
        if not os.path.isdir(sub):
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(os.path.join(sub, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void s2m::", "")[:40]
                if "match" in k or "reduce" in k:
                    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in sorted(acc):
            print("%-8s %-34s %s launches=%d" % (os.path.basename(sub), k, {c: round(sum(v) / len(v)) for c, v in sorted(acc[k].items())},
                                                 len(next(iter(acc[k].values())))))
