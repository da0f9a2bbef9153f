# rocprofv3 kernel stats of the extrinsic_est_en bench with the MFMA contraction (default build) and with the three
# butterflies (-DS2M_EXT_MFMA=0), plus the in-kernel probe.  Output: gpurun_out/r03t/summary.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03t; mkdir -p $O
AB=$R/daliti_amd/_lib_ab/libdaliti_s2m_nomfma.so
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/mfma -- python3 $R/bench.py --extrinsic --no-cpu --no-side --steps 100 --warmup 10 > /dev/null 2>&1
S2M_LIB=$AB rocprofv3 --kernel-trace --stats --output-format csv -d $O/shuffle -- python3 $R/bench.py --extrinsic --no-cpu --no-side --steps 100 --warmup 10 --py-loop > /dev/null 2>&1
cd $R
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I daliti_amd/csrc -I include scripts/hth_mfma_probe.hip -o /tmp/hth 2>/dev/null
{
echo "== in-kernel probe (scripts/hth_mfma_probe.hip): wave-level contraction of 64 points, two waves per SIMD"
/tmp/hth | tail -4
for v in mfma shuffle; do
  echo "== rocprofv3 --kernel-trace --stats, bench.py --extrinsic (C3, 12 Jacobian columns), build: $v"
  python3 - $O/$v <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "reduce_kernel" in r["Name"]:
            print("  %-60s calls %5s  avg %8.2f us  min %8.2f  max %8.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
} > $O/summary.txt
cat $O/summary.txt
