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


# TA_BUSY_avr averages the 9 x 4 x 8 = 288 addresser instances the counter is defined over; 256 of them belong to active
# CUs.  GRBM_GUI_ACTIVE is summed over the 8 XCDs.
TA_INSTANCES, CUS, XCDS = 288, 256, 8


def short(name):
    for k in ("match_rows_batch", "match_hard32_batch", "match_hard_batch", "match_hard32", "reduce_kernel_batch<false, true>", "reduce_kernel_batch<false, false>",
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
for sub in ("fetch", "write", "sq", "tcc", "ta", "tcp"):
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
# whole-run sums of the batched kernels (one grid for K scans): per-scan VALU work and the wait / active shares of
# the wave-cycles, for the issue-side reading of the throughput regime (bench.py: roofline.issue.batched)
batched = None
sq_files = glob.glob(os.path.join(src, "sq", "**", "*counter_collection.csv"), recursive=True)
bj = os.path.join(src, "bench_under_stats.json")
if sq_files and os.path.exists(bj) and os.path.getsize(bj):
    tot = collections.defaultdict(float)
    seen = False
    for r in csv.DictReader(open(sq_files[0])):
        k = short(r["Kernel_Name"])
        if k and "_batch" in k:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            seen = True
    if seen:
        try:
            b = json.loads(open(bj).read().strip().splitlines()[-1])
            scans = (b["steps"] + b["warmup"] + 40) * b["config"]["scans_per_gpu"]   # warm-up + timed + HIP-event sampling loop
            valu = tot["SQ_INSTS_VALU"] / scans
            batched = dict(scans_profiled=scans, scans_per_launch_group=b["config"]["scans_per_gpu"],
                           valu_wave_instructions_per_scan=valu,
                           valu_issue_floor_us_per_scan=valu * 2 / (1024 * 2.4e9) * 1e6,
                           wait_share_of_wave_cycles=tot["SQ_WAIT_ANY"] / tot["SQ_WAVE_CYCLES"],
                           active_share_of_wave_cycles=tot["SQ_ACTIVE_INST_ANY"] / tot["SQ_WAVE_CYCLES"],
                           scans_per_sec_unprofiled=None,
                           note="sums over every dispatch of match_rows_batch / match_hard_batch / reduce_kernel_batch of the "
                                "profiled run divided by the scans it registered; issue floor = VALU x 2 cycles / (1024 SIMDs x 2.4 GHz)")
            ta_files = glob.glob(os.path.join(src, "ta", "**", "*counter_collection.csv"), recursive=True)
            if ta_files:   # share of the batched kernels' run time in which the texture addressers are busy
                ta = collections.defaultdict(float)
                for r in csv.DictReader(open(ta_files[0])):
                    k = short(r["Kernel_Name"])
                    if k and "_batch" in k:
                        ta[r["Counter_Name"]] += float(r["Counter_Value"])
                if ta.get("GRBM_GUI_ACTIVE"):
                    batched["ta_busy_share"] = ta["TA_BUSY_avr"] * TA_INSTANCES / CUS / (ta["GRBM_GUI_ACTIVE"] / XCDS)
            ub = os.path.join(src, "bench.json")
            if os.path.exists(ub) and os.path.getsize(ub):
                u = json.loads(open(ub).read().strip().splitlines()[-1])
                batched["scans_per_sec_unprofiled"] = u["scans_per_sec"]
                batched["gpu_us_per_scan_at_that_rate"] = 1e6 / u["scans_per_sec"]
                batched["issue_utilisation"] = batched["valu_issue_floor_us_per_scan"] / batched["gpu_us_per_scan_at_that_rate"]
        except (KeyError, ValueError, IndexError):
            batched = None
out = dict(tag=tag, batched=batched, note="FETCH_SIZE / WRITE_SIZE in KiB per launch as reported by rocprofv3 (separate --pmc passes); "
           "MI355X_MICROARCH.md: on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so read bytes are up to 2x "
           "the reported figure", kernels=summary)
for name in ("bench.json", "bench_under_stats.json"):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, "%s_%s" % (tag, name)))
json.dump(out, open(os.path.join(dst, "%s_pmc.json" % tag), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True)[:3000])
