# soak of the shared-memory exchange: 4 processes on one GPU, 2,000 scans each, different bets; all results must be equal
cd $GRAFT_REPO_ROOT
T=$(mktemp -d)
export S2M_HELPER_SCANS=2000
for r in 0 1 2 3; do
  python3 tests/shm_rank_helper.py /s2m_soak_$$ 4 $r $(( (r * 2) % 3 )) $T/r$r.npy &
done
wait
python3 - $T <<'PY'
import sys, numpy as np
a = [np.load("%s/r%d.npy" % (sys.argv[1], r)) for r in range(4)]
ok = all(np.array_equal(a[0], x) for x in a[1:])
print("shm soak:", a[0].shape, "ranks equal:", ok, "scans identical to the first:", bool((a[0] == a[0][0]).all()))
sys.exit(0 if ok else 1)
PY
