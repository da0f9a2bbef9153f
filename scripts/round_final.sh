#!/bin/bash
# End-of-round measurements on the MI355X box: every BASELINE config through bench.py, the batched entry point at several
# K, the two loop forms, and the rocprofv3 profiles (stats + separate --pmc passes) of C3 / C4 / C5.
# usage: scripts/round_final.sh <tag>     outputs under gpurun_out/<tag>*/  (copy what is to be judged into profiles/)
TAG=${1:-r04z}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
O=gpurun_out/$TAG; mkdir -p $O
b() { name=$1; shift; timeout 900 python3 bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name rc=$?"; }
b C3
b C3_driver_form --gpus 1 --steps 20 --warmup 5
b C1 --config C1 --no-side
b C2 --config C2 --no-side
b C3_extrinsic --extrinsic --no-side
b C4 --config C4 --no-side
b R1 --config R1 --no-side
b C3_device_loop --device-loop --no-side
b C5_K8 --config C5 --no-cpu
b C5_K16 --config C5 --replicas 16 --no-cpu
b C5_K24 --config C5 --replicas 24 --no-cpu
b C5_K32 --config C5 --replicas 32 --no-cpu
b C5_K24_device_loop --config C5 --replicas 24 --no-cpu --device-loop
b C5b_K24 --config C5b --replicas 24 --no-cpu
b host2 --collective host --shards 2 --no-cpu
b host8 --collective host --shards 8 --no-cpu
bash scripts/profile_round.sh ${TAG}_C3 > /dev/null 2>&1; echo "prof C3 rc=$?"
BENCH_ARGS="--config C4 --no-side" bash scripts/profile_round.sh ${TAG}_C4 > /dev/null 2>&1; echo "prof C4 rc=$?"
BENCH_ARGS="--config C5" bash scripts/profile_round.sh ${TAG}_C5 > /dev/null 2>&1; echo "prof C5 rc=$?"
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d.get("roofline",{}); c=d.get("cpu_baseline",{})
        print("%-22s ms/step %.4f  evals/s %.3e  iters/s %.0f  scans/s %.0f  pass %.1f us  cpu1 %.1f ms (x%.0f)" % (
            os.path.basename(f)[6:-5], d["ms_per_step"], d["value"], d["eskf_iters_per_sec"], d["scans_per_sec"],
            1e3*(r.get("avg_launch_ms") or 0), c.get("ms_per_step",0), d.get("speedup_vs_cpu_1thread",0)))
    except Exception as ex:
        print(os.path.basename(f), "unparsed", ex)
PY
