# Generic A/B on the GPU box: the built library against daliti_amd/_lib_ab/libdaliti_s2m_<variant>.so
# usage: bash scripts/r03_ab.sh <out-tag> <variant> [counters: 0|1]
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
V=$2
AB=$GRAFT_REPO_ROOT/daliti_amd/_lib_ab/libdaliti_s2m_$V.so
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -5
run() { name=$1; shift; timeout 600 "$@" > $O/$name.json 2> $O/$name.err; python3 -c "
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); r=d.get('roofline',{})
    print('%-18s ms/step %.4f  scans/s %.0f  pass %.1f us (search %.1f)' % ('$name', d['ms_per_step'], d['scans_per_sec'], 1e3*(r.get('avg_launch_ms') or 0), 1e3*((r.get('search_kernels_only') or {}).get('avg_ms') or 0)))
except Exception as e: print('$name', 'FAILED', e)"; }
for rep in 1 2; do
run c3_new_$rep python3 bench.py --no-cpu --py-loop
run c3_${V}_$rep env S2M_LIB=$AB python3 bench.py --no-cpu --py-loop
done
for cfg in C4 R1 C1; do
run ${cfg}_new python3 bench.py --config $cfg --no-cpu --py-loop
run ${cfg}_$V env S2M_LIB=$AB python3 bench.py --config $cfg --no-cpu --py-loop
done
for k in 8 16 24; do
run c5k${k}_new python3 bench.py --config C5 --replicas $k --no-cpu --steps 100 --py-loop
run c5k${k}_$V env S2M_LIB=$AB python3 bench.py --config C5 --replicas $k --no-cpu --steps 100 --py-loop
done
[ "${3:-1}" = "1" ] || exit 0
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
pass() { local name=$1 args=$2; shift 2
    timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$R/$O/$name" -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu --no-side $args > "$R/$O/$name.log" 2>&1; echo "$name rc=$?"; }
pass C3_ta "--config C3" TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE
pass C3_tcp "--config C3" TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum
pass C3_sq "--config C3" SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD
pass C5_ta "--config C5 --replicas 16" TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE
cd $R
python3 - $O <<'PY'
import collections, csv, glob, os, sys
out = sys.argv[1]
for d in sorted(glob.glob(os.path.join(out, "C*_*"))):
    if not os.path.isdir(d):
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void s2m::", "")[:40]
            if "match" in k or "reduce" in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        print(os.path.basename(d), k, {c: round(sum(v) / len(v), 1) for c, v in acc[k].items()}, "n=%d" % len(next(iter(acc[k].values()))))
PY
