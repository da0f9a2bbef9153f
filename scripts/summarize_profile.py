"""Summarise gpurun_out/<tag>/ (from scripts/profile_round.sh) into profiles/<tag>_*.{csv,json}."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")


def short(name):
    for k in ("match_rows_batch", "match_hard_batch", "reduce_kernel_batch<false, true>", "reduce_kernel_batch<false, false>",
              "reduce_kernel_batch<true, true>", "reduce_kernel_batch<true, false>"):
        if k in name:
            return k
    for k in ("match_rows", "match_easy", "match_hard", "reduce_kernel<false, true>", "reduce_kernel<false, false>",
              "reduce_kernel<true, true>", "reduce_kernel<true, false>"):
        if k in name:
            return k
    return None


stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, "%s_bench_kernel_stats.csv" % tag))
summary = collections.defaultdict(dict)
if stats:
    for r in csv.DictReader(open(stats[0])):
        k = short(r["Name"])
        if k:
            summary[k].update(calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]), min_ns=float(r["MinNs"]),
                              max_ns=float(r["MaxNs"]))
for sub in ("fetch", "write", "sq", "tcc"):
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            for c, x in v.items():
                summary[k][c + "_avg"] = sum(x) / len(x)
                if c in ("FETCH_SIZE", "WRITE_SIZE"):
                    summary[k][c + "_max"] = max(x)
out = dict(tag=tag, note="FETCH_SIZE / WRITE_SIZE in KiB per launch as reported by rocprofv3 (separate --pmc passes); "
           "MI355X_MICROARCH.md: on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so read bytes are up to 2x "
           "the reported figure", kernels=summary)
for name in ("bench.json", "bench_under_stats.json"):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, "%s_%s" % (tag, name)))
json.dump(out, open(os.path.join(dst, "%s_pmc.json" % tag), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True)[:3000])
