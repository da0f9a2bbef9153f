#!/bin/bash
# Runs on the MI355X box (via gpurun): the gpu test suite, then the bench lines of this round.
# usage: scripts/r03_check.sh <tag> [quick]      outputs under gpurun_out/<tag>/
set -u
TAG=${1:-r03a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
timeout 1500 python3 -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest.log"
tail -5 "$OUT/pytest.log"
timeout 600 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_form.json" 2> "$OUT/bench_driver_form.err"; echo "bench20 rc=$?"
if [ "${2:-}" != "quick" ]; then
  timeout 600 python3 bench.py --config C4 --no-side > "$OUT/bench_C4.json" 2> "$OUT/bench_C4.err"; echo "C4 rc=$?"
  timeout 600 python3 bench.py --config C5 --no-cpu > "$OUT/bench_C5.json" 2> "$OUT/bench_C5.err"; echo "C5 rc=$?"
  timeout 600 python3 bench.py --collective host --shards 2 --no-cpu > "$OUT/bench_host2.json" 2> "$OUT/bench_host2.err"; echo "host2 rc=$?"
  timeout 600 python3 bench.py --collective host --shards 8 --no-cpu > "$OUT/bench_host8.json" 2> "$OUT/bench_host8.err"; echo "host8 rc=$?"
  timeout 600 python3 bench.py --py-loop --no-cpu > "$OUT/bench_pyloop.json" 2> "$OUT/bench_pyloop.err"; echo "pyloop rc=$?"
fi
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$OUT/bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d.get("roofline",{})
        print(os.path.basename(f), "ms/step %.4f"%d["ms_per_step"], "value %.3e"%d["value"], "pass_ms", r.get("avg_launch_ms"), "frac", r.get("frac"),
              "c5", (d.get("c5_batch") or {}).get("scans_per_sec"), "frame", {k:(d.get("frame_pipeline") or {}).get(k) for k in ("median_ms","p99_ms","max_ms","updates")})
    except Exception as ex:
        print(os.path.basename(f), "unparsed", ex)
PY
