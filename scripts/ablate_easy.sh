#!/bin/bash
# Dev tool (runs on the GPU box): time match_easy with parts removed.  Builds throw-away libraries.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/daliti_amd/csrc
cp ../_lib/libdaliti_s2m.so /tmp/lib_orig.so
make -s -j8 >/dev/null 2>&1   # object files are not shipped to the box: rebuild them once
for ab in 0 1 2; do
  if [ $ab = 0 ]; then FL=""; else FL="-DS2M_ABLATE=$ab"; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $FL -c s2m_match.hip -o /tmp/match_ab.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../_lib/libdaliti_s2m.so ../_lib/s2m_map.o ../_lib/s2m_mapupd.o ../_lib/s2m_voxel.o ../_lib/s2m_undistort.o /tmp/match_ab.o ../_lib/s2m_reduce.o ../_lib/s2m_eskf.o ../_lib/s2m_engine.o
  echo "== ablate $ab"
  cd /tmp && TMPDIR=/tmp PASSES=8 CELL=0.5 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab$ab -- python3 $R/scripts/one_pass.py >/dev/null 2>&1
  grep -h "match_easy\|match_hard" /tmp/ab$ab/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-80
  cd $R/daliti_amd/csrc
done
cp /tmp/lib_orig.so ../_lib/libdaliti_s2m.so
