cd $GRAFT_REPO_ROOT
O=gpurun_out/r03j; mkdir -p $O
AB=$GRAFT_REPO_ROOT/daliti_amd/_lib_ab
timeout 1700 python3 -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -5
run() { name=$1; shift; timeout 600 "$@" > $O/$name.json 2> $O/$name.err; python3 -c "
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); r=d.get('roofline',{})
    print('%-14s ms/step %.4f  scans/s %.0f  pass %.1f us (search %.1f)' % ('$name', d['ms_per_step'], d['scans_per_sec'], 1e3*(r.get('avg_launch_ms') or 0), 1e3*((r.get('search_kernels_only') or {}).get('avg_ms') or 0)))
except Exception as e: print('$name', 'FAILED', e)"; }
for cfg in C3 C4 R1 C1 C2; do
  run ${cfg}_ext3 python3 bench.py --config $cfg --no-cpu --py-loop
  run ${cfg}_ext2 env S2M_LIB=$AB/libdaliti_s2m_ext2.so python3 bench.py --config $cfg --no-cpu --py-loop
  run ${cfg}_ext0 env S2M_LIB=$AB/libdaliti_s2m_ext0.so python3 bench.py --config $cfg --no-cpu --py-loop
done
for K in 8 16; do
  run c5k${K}_ext3 python3 bench.py --config C5 --replicas $K --no-cpu --steps 100 --py-loop
  run c5k${K}_ext2 env S2M_LIB=$AB/libdaliti_s2m_ext2.so python3 bench.py --config C5 --replicas $K --no-cpu --steps 100 --py-loop
  run c5k${K}_ext0 env S2M_LIB=$AB/libdaliti_s2m_ext0.so python3 bench.py --config C5 --replicas $K --no-cpu --steps 100 --py-loop
done
