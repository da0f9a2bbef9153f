cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h; mkdir -p $O
run() { name=$1; K=$2; shift 2
  env "$@" timeout 300 python3 bench.py --config C5 --replicas $K --no-cpu --steps 100 > $O/$name.json 2> $O/$name.err
  python3 -c "
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', 'scans/s %.0f' % d['scans_per_sec'])
except Exception as e: print('$name', 'FAILED', e)"; }
for K in 8 16 32; do
  run g2nb2_K$K $K A=1
  run g1nb1_K$K $K S2M_BATCH_G=1 S2M_BATCH_NB=1
  run g1nb2_K$K $K S2M_BATCH_G=1 S2M_BATCH_NB=2
  run g1nb3_K$K $K S2M_BATCH_G=1 S2M_BATCH_NB=3
done
