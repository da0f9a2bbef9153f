#!/bin/bash
# quick per-kernel stats (+ VALU/TA counters) of one bench configuration on the box
# usage: scripts/prof_quick.sh <tag> [bench args...]   env passes through (S2M_MATCH_GROUP, ...)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 40 --warmup 5 --no-cpu $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD > "$OUT/bench_under_stats.json" 2>/dev/null
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d "$OUT/sq" -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TA_BUSY_avr TA_TA_BUSY_sum --output-format csv -d "$OUT/ta" -- $CMD > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
out = "$OUT"
st = glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True)
rows = {}
if st:
    for r in csv.DictReader(open(st[0])):
        n = r["Name"]
        if "s2m::" in n:
            rows[n[:70]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3)
for n, v in rows.items():
    print("%-72s calls %5d avg %8.2f us min %8.2f max %8.2f" % ((n,) + v))
for sub in ("sq", "ta"):
    for f in glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "s2m::match" in r["Kernel_Name"] or "s2m::reduce" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(k, {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
