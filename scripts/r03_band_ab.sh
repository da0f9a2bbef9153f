# first band (cells) of far points whose first shell was empty: 2.8 (default) against 2.2 / 2.5
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03u; mkdir -p $O
run() { name=$1; shift; timeout 600 "$@" > $O/$name.json 2> $O/$name.err; python3 -c "
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); r=d.get('roofline',{})
    print('%-18s ms/step %.4f  scans/s %.0f  pass %.1f us (search %.1f)' % ('$name', d['ms_per_step'], d['scans_per_sec'], 1e3*(r.get('avg_launch_ms') or 0), 1e3*((r.get('search_kernels_only') or {}).get('avg_ms') or 0)))
except Exception as e: print('$name', 'FAILED', e)"; }
for cfg in C3 C4 R1 C2; do
run ${cfg}_be28 python3 bench.py --config $cfg --no-cpu --no-side --py-loop
for v in 22 25; do
run ${cfg}_be$v env S2M_LIB=$GRAFT_REPO_ROOT/daliti_amd/_lib_ab/libdaliti_s2m_be$v.so python3 bench.py --config $cfg --no-cpu --no-side --py-loop
done; done
run c5k16_be28 python3 bench.py --config C5 --replicas 16 --no-cpu --steps 100 --py-loop
for v in 22 25; do
run c5k16_be$v env S2M_LIB=$GRAFT_REPO_ROOT/daliti_amd/_lib_ab/libdaliti_s2m_be$v.so python3 bench.py --config C5 --replicas 16 --no-cpu --steps 100 --py-loop
done
