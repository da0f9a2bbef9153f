#!/bin/bash
# Round-2 experiments on the box (timing rows of DESIGN's rejected table): Morton-ordered scan, wave-uniform batch
# reject, coalesced instead of gathered neighbours in reduce<FIT>.  Ends with the stock build restored.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result"
line() { python - "$1" "$2" <<'PY'
import json,sys
d=json.load(open(sys.argv[2])); r=d["roofline"]
print("%-22s ms/step %.4f search %.1f us fit %.1f us reuse %.1f us pose_err %.5f" % (sys.argv[1], d["ms_per_step"], 1e3*r["search_kernels_only"]["avg_ms"], 1e3*r["reduce_fit_avg_ms"], 1e3*d["roofline_reuse"]["avg_launch_ms"], d["pose_error_vs_truth_m"]))
PY
}
mkdir -p gpurun_out/ab2
python bench.py --no-cpu --steps 200 > gpurun_out/ab2/stock.json 2>/dev/null; line stock gpurun_out/ab2/stock.json
python bench.py --no-cpu --steps 200 --sort-scan morton > gpurun_out/ab2/morton.json 2>/dev/null; line morton-scan gpurun_out/ab2/morton.json
for v in "reject:-DS2M_EXP_BATCH_REJECT" "fitcoal:-DS2M_EXP_FIT_COALESCED"; do
  name=${v%%:*}; flags=${v#*:}
  make -B -j8 -C daliti_amd/csrc -s CXXFLAGS="$BASE $flags" > /dev/null 2>&1 || { echo "$name build failed"; continue; }
  python bench.py --no-cpu --steps 200 > gpurun_out/ab2/$name.json 2>/dev/null; line $name gpurun_out/ab2/$name.json
  if [ $name = reject ]; then scripts/prof_quick.sh ab2_reject 2>&1 | grep -E "match_rows" | head -3; fi
done
make -B -j8 -C daliti_amd/csrc -s > /dev/null 2>&1
