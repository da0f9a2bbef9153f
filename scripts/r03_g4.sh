# lanes per first-shell query x batches per trip, now that the lanes of a query share runs point by point
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03m; mkdir -p $O
run() { name=$1; shift; timeout 600 "$@" > $O/$name.json 2> $O/$name.err; python3 -c "
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); r=d.get('roofline',{})
    print('%-18s ms/step %.4f  scans/s %.0f  pass %.1f us (search %.1f)' % ('$name', d['ms_per_step'], d['scans_per_sec'], 1e3*(r.get('avg_launch_ms') or 0), 1e3*((r.get('search_kernels_only') or {}).get('avg_ms') or 0)))
except Exception as e: print('$name', 'FAILED', e)"; }
for cfg in C3 C4 R1; do
for g in 2 4; do for nb in 1 2 3; do
run ${cfg}_g${g}nb${nb} env S2M_MATCH_GROUP=$g S2M_EASY_NB=$nb python3 bench.py --config $cfg --no-cpu --no-side
done; done; done
for k in 8 24; do
for g in 2 4; do for nb in 1 2; do
run c5k${k}_g${g}nb${nb} env S2M_BATCH_G=$g S2M_BATCH_NB=$nb python3 bench.py --config C5 --replicas $k --no-cpu --steps 100
done; done; done
S2M_MATCH_GROUP=4 timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
