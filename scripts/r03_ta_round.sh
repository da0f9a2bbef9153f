# TA / TCP passes of the C3 and C5 profiles only (the rest of gpurun_out/<tag>_C3|_C5 comes from profile_round.sh)
TAG=${1:-r03zz}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for cfg in "C3:" "C5:--config C5"; do
  n=${cfg%%:*}; a=${cfg#*:}
  OUT=$R/gpurun_out/${TAG}_$n; mkdir -p $OUT
  CMD="python3 $R/bench.py --steps 50 --warmup 5 --no-cpu $a"
  timeout 300 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE --output-format csv -d "$OUT/ta" -- $CMD > /dev/null 2>&1; echo "$n ta rc=$?"
  timeout 300 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d "$OUT/tcp" -- $CMD > /dev/null 2>&1; echo "$n tcp rc=$?"
done
