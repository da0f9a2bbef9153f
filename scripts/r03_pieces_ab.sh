cd $GRAFT_REPO_ROOT
O=gpurun_out/r03i; mkdir -p $O
AB=$GRAFT_REPO_ROOT/daliti_amd/_lib_ab/libdaliti_s2m_nopieces.so

run() { name=$1; shift; timeout 600 "$@" > $O/$name.json 2> $O/$name.err; python3 -c "
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); r=d.get('roofline',{})
    print('%-18s ms/step %.4f  scans/s %.0f  pass %.1f us (search %.1f)' % ('$name', d['ms_per_step'], d['scans_per_sec'], 1e3*(r.get('avg_launch_ms') or 0), 1e3*((r.get('search_kernels_only') or {}).get('avg_ms') or 0)))
except Exception as e: print('$name', 'FAILED', e)"; }
for rep in 1 2; do
run c3_pieces_$rep python3 bench.py --no-cpu --py-loop
run c3_nopieces_$rep env S2M_LIB=$AB python3 bench.py --no-cpu --py-loop
done
run c4_pieces python3 bench.py --config C4 --no-cpu --py-loop
run c4_nopieces env S2M_LIB=$AB python3 bench.py --config C4 --no-cpu --py-loop
run r1_pieces python3 bench.py --config R1 --no-cpu --py-loop
run r1_nopieces env S2M_LIB=$AB python3 bench.py --config R1 --no-cpu --py-loop
run c1_pieces python3 bench.py --config C1 --no-cpu --py-loop
run c1_nopieces env S2M_LIB=$AB python3 bench.py --config C1 --no-cpu --py-loop
run c5k8_pieces python3 bench.py --config C5 --no-cpu --steps 100 --py-loop
run c5k8_nopieces env S2M_LIB=$AB python3 bench.py --config C5 --no-cpu --steps 100 --py-loop
run c5k16_pieces python3 bench.py --config C5 --replicas 16 --no-cpu --steps 100 --py-loop
run c5k16_nopieces env S2M_LIB=$AB python3 bench.py --config C5 --replicas 16 --no-cpu --steps 100 --py-loop
