#!/bin/bash
# Runs on the MI355X box (via gpurun): kernel-trace stats and, in separate passes as the microarch
# guide prescribes, the FETCH_SIZE / WRITE_SIZE counters of the same bench command.
# usage: scripts/profile_round.sh <tag>      outputs under gpurun_out/<tag>/
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 50 --warmup 5 --no-cpu ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD > "$OUT/bench_under_stats.json" 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d "$OUT/sq" -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d "$OUT/tcc" -- $CMD > /dev/null 2>&1
# texture addresser / L1 (two TA counters per pass at most; an unknown counter name makes rocprofv3 abort and hang: bounded)
timeout 300 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE --output-format csv -d "$OUT/ta" -- $CMD > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d "$OUT/tcp" -- $CMD > /dev/null 2>&1
python3 $R/bench.py --steps 200 --warmup 20 ${BENCH_ARGS:-} > "$OUT/bench.json" 2> "$OUT/bench.err"
ls "$OUT"
