cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e; mkdir -p $O
run() { # name, env..., K
  name=$1; K=$2; shift 2
  env "$@" timeout 300 python3 bench.py --config C5 --replicas $K --no-cpu --steps 100 > $O/$name.json 2> $O/$name.err
  python3 -c "
import json,sys
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', 'scans/s %.0f' % d['scans_per_sec'])
except Exception as e: print('$name', 'FAILED', e)"
}
for K in 8 16; do
  run base_K$K $K A=1
  run g1_K$K $K S2M_BATCH_GROUPS=1
  run g2_K$K $K S2M_BATCH_GROUPS=2
  run g4_K$K $K S2M_BATCH_GROUPS=4
  run g8_K$K $K S2M_BATCH_GROUPS=8
  run nb1_K$K $K S2M_BATCH_NB=1
  run nb3_K$K $K S2M_BATCH_NB=3
done
