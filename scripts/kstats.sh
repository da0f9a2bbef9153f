#!/bin/bash
# per-kernel averages of one bench leg under rocprofv3 --kernel-trace --stats (GPU box).  usage: scripts/kstats.sh <tag> <bench args...>
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; shift
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu "$@" > $O/bench.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "s2m::" in r["Name"] and ("match" in r["Name"] or "reduce" in r["Name"]):
            print("%-60s calls %6s avg %8.2f us min %8.2f max %8.2f" % (r["Name"].replace("void s2m::", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
grep "^{" $O/bench.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(\"ms/step\", d[\"ms_per_step\"], \"scans/s\", d[\"scans_per_sec\"])"
rm -rf $O/stats
