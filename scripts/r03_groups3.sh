cd $GRAFT_REPO_ROOT
run() { name=$1; K=$2; shift 2
  env "$@" timeout 300 python3 bench.py --config C5 --replicas $K --no-cpu --steps 100 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', 'scans/s %.0f' % d['scans_per_sec'])"; }
for rep in 1 2; do
for K in 8 12 16; do
  run g2_K$K $K A=1
  run g3_K$K $K S2M_BATCH_GROUPS=3
done; done
