#!/bin/bash
# per-kernel stats of scripts/time_map.py (map build / incremental update / box delete at C3) on the box
# usage: scripts/prof_map.sh <tag>
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $R/scripts/time_map.py > "$OUT/time_map.txt" 2>/dev/null
python3 - <<PY
import csv, glob
st = glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True)
rows = []
for r in csv.DictReader(open(st[0])):
    rows.append((float(r["TotalDurationNs"]) / 1e3, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:100]))
rows.sort(reverse=True)
with open("$OUT/kernel_stats.txt", "w") as f:
    for t, c, a, n in rows[:45]:
        line = "%10.1f us total  calls %4d  avg %8.2f us  %s" % (t, c, a, n)
        print(line); f.write(line + "\n")
PY
cat "$OUT/time_map.txt" | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
