#!/bin/bash
# Dev tool (runs on the MI355X box): rebuild the library with extra -D flags per variant and bench each.
# usage: scripts/ab_build.sh "<name>:<flags>" ...     e.g.  "w4b8:-DS2M_HARD_WAVES=4" "w3:-DS2M_HARD_WAVES=3"
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  make -B -j8 -C daliti_amd/csrc -s CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result $flags" > /dev/null 2>&1 || { echo "$name build failed"; continue; }
  python bench.py --steps 200 --warmup 20 --no-cpu 2>/dev/null | tail -1 > gpurun_out/ab_$name.json
  python - "$name" <<'PY'
import json,sys
d=json.load(open('gpurun_out/ab_%s.json'%sys.argv[1])); print(sys.argv[1], 'ms/step %.4f'%d['ms_per_step'], 'match launch us %.1f'%(d['roofline']['avg_launch_ms']*1e3))
PY
done
