#!/bin/bash
# Which unit binds the search kernels when the chip is saturated (batched C5) and for one scan (C3)?
# Texture-addresser / L1 counters in their own --pmc passes (separate runs, kernel-trace only).
# usage (GPU box): bash scripts/r03_ta_probe.sh <tag>
set -u
TAG=${1:-r03ta}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/avail.txt" 2>&1
grep -o "^\s*Name\s*:\s*\(TA\|TCP\|TD\)_[A-Za-z0-9_]*" "$OUT/avail.txt" | sort -u > "$OUT/avail_ta_tcp.txt"
pass() {  # name, bench args, counters...
    local name=$1 args=$2; shift 2
    timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu --no-side $args > "$OUT/$name.log" 2>&1
    echo "$name rc=$?"
}
for cfg in "C3:--config C3" "C5:--config C5 --replicas 16"; do
    n=${cfg%%:*}; a=${cfg#*:}
    pass ${n}_ta1 "$a" TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE
    pass ${n}_ta2 "$a" TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum
    pass ${n}_ta3 "$a" TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum
    pass ${n}_tcp1 "$a" TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum
    pass ${n}_tcp2 "$a" TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
    pass ${n}_tcp3 "$a" TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum
    pass ${n}_td "$a" TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum
done
python3 - "$OUT" <<'PY'
import collections, csv, glob, os, sys
out = sys.argv[1]
for d in sorted(glob.glob(os.path.join(out, "C*_*"))):
    if not os.path.isdir(d):
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void s2m::", "")[:40]
            if "match" in k or "reduce" in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        print(os.path.basename(d), k, {c: round(sum(v) / len(v), 1) for c, v in acc[k].items()}, "n=%d" % len(next(iter(acc[k].values()))))
PY
