# extrinsic_est_en = 1 (twelve Jacobian columns): the wave-level contraction by MFMA (default) against three butterflies
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s; mkdir -p $O
AB=$GRAFT_REPO_ROOT/daliti_amd/_lib_ab/libdaliti_s2m_nomfma.so
timeout 1200 python3 -m pytest tests -m gpu -x -q -k "extrinsic or parity or pass_bit_exact or shards" 2>&1 | grep -E "passed|failed|rror" | tail -3
run() { name=$1; shift; timeout 600 "$@" > $O/$name.json 2> $O/$name.err; python3 -c "
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); r=d.get('roofline',{}); rr=d.get('roofline_reuse',{})
    print('%-18s ms/step %.4f  rematch pass %.1f us (reduce<FIT> %.1f)  reuse pass %.1f us' % ('$name', d['ms_per_step'], 1e3*(r.get('avg_launch_ms') or 0), 1e3*(r.get('reduce_fit_avg_ms') or 0), 1e3*(rr.get('avg_launch_ms') or 0)))
except Exception as e: print('$name', 'FAILED', e)"; }
for rep in 1 2 3; do
run ext_mfma_$rep python3 bench.py --extrinsic --no-cpu --no-side --py-loop
run ext_shuffle_$rep env S2M_LIB=$AB python3 bench.py --extrinsic --no-cpu --no-side --py-loop
done
run c4ext_mfma python3 bench.py --config C4 --extrinsic --no-cpu --no-side --py-loop
run c4ext_shuffle env S2M_LIB=$AB python3 bench.py --config C4 --extrinsic --no-cpu --no-side --py-loop
