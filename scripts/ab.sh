#!/bin/bash
# One A/B runner for kernel variants (replaces the per-experiment scripts of rounds 1-3; results go to NOTEBOOK.md).
#
#   scripts/ab.sh build <name> "<extra hipcc flags>"     (build container: hipcc cross-compiles gfx950)
#       builds the library with the extra flags into daliti_amd/_lib_ab/libdaliti_s2m_<name>.so (travels with gpurun)
#   scripts/ab.sh run <tag> "<variants>" "<legs>" [reps] [env for the variant legs]   (GPU box)
#       runs every leg alternately with the in-tree library ("base") and with each variant, `reps` times (default 2),
#       and prints one line per run; JSON lines under gpurun_out/<tag>/.
#       legs: C1 C2 C3 C4 R1 (one scan in flight), K8 K16 K24 ... (C5 with that many scans in flight), B8 B24 (the same against C4's 20 M-point map), FRAME (C3 with
#       the frame_pipeline side leg); a variant named "-" means "the in-tree library again" with the given env and with the
#       extra bench.py arguments in $AB_ARGS (e.g. AB_ARGS=--device-loop: an A/B of two forms inside one library)
#   scripts/ab.sh counters <tag> "<variants>" "<leg>"    (GPU box) SQ / TA / TCP counters of one leg, base and variants
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mode=${1:?mode}; shift
leg_args() {
  case $1 in
    C1|C2|C3|C4|R1) echo "--config $1 --no-cpu --no-side" ;;
    K*) echo "--config C5 --replicas ${1#K} --no-cpu --steps 100" ;;
    B*) echo "--config C5b --replicas ${1#B} --no-cpu --steps 40" ;;
    FRAME) echo "--config C3 --cpu-steps 0 --steps 50 --moving-frames 0" ;;
    *) echo "unknown leg $1" >&2; exit 2 ;;
  esac
}
lib_of() { [ "$1" = base ] && echo "" || echo "$R/daliti_amd/_lib_ab/libdaliti_s2m_$1.so"; }
case $mode in
build)
  name=${1:?name}; flags=${2:-}
  tmp=$(mktemp -d)
  make -C daliti_amd/csrc -s -j8 OUT_DIR=$tmp CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result $flags" || exit 1
  mkdir -p daliti_amd/_lib_ab && cp $tmp/libdaliti_s2m.so daliti_amd/_lib_ab/libdaliti_s2m_$name.so && rm -rf $tmp
  echo "built daliti_amd/_lib_ab/libdaliti_s2m_$name.so"
  ;;
run)
  tag=${1:?tag}; variants=${2:-}; legs=${3:-"C3 K8 K24"}; reps=${4:-2}; venv=${5:-}
  O=gpurun_out/$tag; mkdir -p $O
  one() {  # name leg env...
    local name=$1 leg=$2 extra=$3; shift 3
    env "$@" timeout 900 python3 bench.py $(leg_args $leg) $extra > $O/$name.json 2> $O/$name.err
    python3 - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get("roofline", {}); f = d.get("frame_pipeline") or {}
    print("%-22s ms/step %.4f  scans/s %7.0f  rematch pass %.1f us%s" % (sys.argv[2], d["ms_per_step"], d["scans_per_sec"],
          1e3 * (r.get("avg_launch_ms") or 0), ("  frame median %.3f ms" % f["median_ms"]) if f.get("median_ms") else ""))
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
  }
  for rep in $(seq 1 $reps); do
    for leg in $legs; do
      one ${leg}_base_$rep $leg "" S2M_AB=base
      for v in $variants; do
        if [ "$v" = "-" ]; then one ${leg}_alt_$rep $leg "${AB_ARGS:-}" ${venv:-S2M_AB=alt}; else one ${leg}_${v}_$rep $leg "" S2M_LIB=$(lib_of $v) $venv; fi
      done
    done
  done
  ;;
counters)
  tag=${1:?tag}; variants=${2:-}; leg=${3:-K8}
  O=$R/gpurun_out/$tag; mkdir -p $O
  cd /tmp && export TMPDIR=/tmp
  for v in base $variants; do
    lib=$(lib_of $v); [ -n "$lib" ] && export S2M_LIB=$lib || unset S2M_LIB
    args="$(leg_args $leg) --steps 30 --warmup 5"
    pass() { local name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$O/${v}_$name" -- python3 $R/bench.py $args > "$O/${v}_$name.log" 2>&1; }
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${v}_stats" -- python3 $R/bench.py $args > "$O/${v}_stats.log" 2>&1
    pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
    pass ta TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE
    pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum
    pass fetch FETCH_SIZE
  done
  python3 $R/scripts/ab_counters.py $O base $variants
  ;;
*) echo "usage: scripts/ab.sh build|run|counters ..." >&2; exit 2 ;;
esac
