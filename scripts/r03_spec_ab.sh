cd $GRAFT_REPO_ROOT
timeout 1700 python3 -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
for i in 1 2; do
timeout 300 python3 bench.py --no-cpu | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spec on  ms/step %.4f'%d['ms_per_step'])"
S2M_SPEC=0 timeout 300 python3 bench.py --no-cpu | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spec off ms/step %.4f'%d['ms_per_step'])"
done
for cfg in C1 C2 C4 R1; do
timeout 300 python3 bench.py --no-cpu --config $cfg | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg spec on  ms/step %.4f'%d['ms_per_step'])"
S2M_SPEC=0 timeout 300 python3 bench.py --no-cpu --config $cfg | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg spec off ms/step %.4f'%d['ms_per_step'])"
done
