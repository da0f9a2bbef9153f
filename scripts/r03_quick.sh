cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -15
timeout 300 python3 bench.py --no-side > gpurun_out/r03f_bench.json 2>/dev/null; python3 -c "
import json; d=json.loads(open('gpurun_out/r03f_bench.json').read().strip().splitlines()[-1]); print('ms/step', d['ms_per_step'], 'reuse reduce ms', d['roofline_reuse']['avg_launch_ms'], 'fit', d['roofline']['reduce_fit_avg_ms'])"
