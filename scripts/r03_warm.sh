cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03c
for i in 1 2 3; do
  timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-side > gpurun_out/r03c/d20_$i.json 2>/dev/null
  timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu > gpurun_out/r03c/d20nocpu_$i.json 2>/dev/null
done
timeout 300 python3 bench.py --steps 20 --warmup 100 --no-cpu > gpurun_out/r03c/d20w100.json 2>/dev/null
timeout 300 python3 bench.py --steps 200 --warmup 20 --no-side > gpurun_out/r03c/d200.json 2>/dev/null
timeout 300 python3 bench.py --steps 2000 --warmup 20 --no-cpu > gpurun_out/r03c/d2000.json 2>/dev/null
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r03c/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "ms/step %.4f"%d["ms_per_step"])
    except Exception as ex:
        print(os.path.basename(f), "unparsed", ex)
PY
