cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03b
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03b/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03b/pytest.log
for K in 2 4 8 16 32; do
  timeout 300 python3 bench.py --config C5 --replicas $K --no-cpu --steps 100 > gpurun_out/r03b/c5_fused_$K.json 2> gpurun_out/r03b/c5_fused_$K.err; echo "fused $K rc=$?"
done
for K in 8 16; do
  S2M_BATCH_STREAMS=1 timeout 300 python3 bench.py --config C5 --replicas $K --no-cpu --steps 100 > gpurun_out/r03b/c5_streams_$K.json 2> gpurun_out/r03b/c5_streams_$K.err; echo "streams $K rc=$?"
done
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r03b/c5_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "scans/s %.0f"%d["scans_per_sec"], "ms/step %.4f"%d["ms_per_step"], "evals/s %.3e"%d["value"])
    except Exception as ex:
        print(os.path.basename(f), "unparsed", ex, open(f.replace(".json",".err")).read()[-500:])
PY
