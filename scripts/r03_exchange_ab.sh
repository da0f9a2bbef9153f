# what each exchange form adds to the loop with ONE rank (lower bound of its per-iteration cost; N > 1 adds rank skew):
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03r; mkdir -p $O
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29655
run() { name=$1; shift; timeout 600 "$@" > $O/$name.json 2> $O/$name.err; python3 -c "
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('%-22s ms/step %.4f  collective %s' % ('$name', d['ms_per_step'], d['config'].get('collective')))
except Exception as e: print('$name', 'FAILED', e)"; }
for rep in 1 2; do
run plain_$rep python3 bench.py --no-cpu --no-side
run shm_$rep python3 bench.py --no-cpu --no-side --force-collective --collective shm
run rccl_$rep python3 bench.py --no-cpu --no-side --force-collective --collective rccl
run torchcb_$rep python3 bench.py --no-cpu --no-side --force-collective --torch-collective
done
# two ranks on the same GPU (gloo for the rendezvous only), strong scaling of C3: shm against the torch callback
run two_shm python3 bench.py --gpus 2 --no-cpu --no-side --backend gloo --all-on-device0 --collective shm
run two_torchcb python3 bench.py --gpus 2 --no-cpu --no-side --backend gloo --all-on-device0 --torch-collective
