cd $GRAFT_REPO_ROOT
O=gpurun_out/r03g; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4
run() { name=$1; shift; timeout 600 "$@" > $O/$name.json 2> $O/$name.err; python3 -c "
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); r=d.get('roofline',{})
    print('%-16s ms/step %.4f  scans/s %.0f  pass %.1f us (search %.1f)' % ('$name', d['ms_per_step'], d['scans_per_sec'], 1e3*(r.get('avg_launch_ms') or 0), 1e3*((r.get('search_kernels_only') or {}).get('avg_ms') or 0)))
except Exception as e: print('$name', 'FAILED', e)"; }
for rep in 1 2; do
run c3_32_$rep python3 bench.py --no-cpu
run c3_64_$rep env S2M_HARD_LANES=64 python3 bench.py --no-cpu
done
run c4_32 python3 bench.py --config C4 --no-cpu
run c4_64 env S2M_HARD_LANES=64 python3 bench.py --config C4 --no-cpu
run c1_32 python3 bench.py --config C1 --no-cpu
run c1_64 env S2M_HARD_LANES=64 python3 bench.py --config C1 --no-cpu
run r1_32 python3 bench.py --config R1 --no-cpu
run r1_64 env S2M_HARD_LANES=64 python3 bench.py --config R1 --no-cpu
run c5k8_32 python3 bench.py --config C5 --no-cpu --steps 100
run c5k8_64 env S2M_HARD_LANES=64 python3 bench.py --config C5 --no-cpu --steps 100
run c5k16_32 python3 bench.py --config C5 --replicas 16 --no-cpu --steps 100
run c5k16_64 env S2M_HARD_LANES=64 python3 bench.py --config C5 --replicas 16 --no-cpu --steps 100
