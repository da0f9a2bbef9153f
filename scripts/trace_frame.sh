#!/bin/bash
# kernel + copy timeline of the LAST frame of scripts/frame_pipeline.py (start offsets, durations, gaps)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/trace" -- python3 $R/scripts/frame_pipeline.py > "$OUT/frames.txt" 2>/dev/null
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
for f in glob.glob("$OUT/trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
rows.sort()
idx = [i for i, r in enumerate(rows) if "undistort" in r[2] or "undist" in r[2]]
i0 = idx[-1]
# back up to the H2D copy that precedes the last undistort kernel
while i0 > 0 and rows[i0][0] - rows[i0 - 1][1] < 200000 and "incr_classify" not in rows[i0 - 1][2] and "merge" not in rows[i0 - 1][2] and "brick" not in rows[i0 - 1][2]: i0 -= 1
t0 = rows[i0][0]
prev_end = t0
with open("$OUT/timeline.txt", "w") as out:
    for s, e, n in rows[i0:]:
        if "incr_classify" in n: break
        line = "%8.1f us  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, n)
        print(line); out.write(line + "\n")
        prev_end = max(prev_end, e)
PY
