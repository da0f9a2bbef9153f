cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 600 python3 scripts/sweep_cells.py C3 > $O/sweep_C3.txt 2>&1; echo "sweep C3 rc=$?"
timeout 900 python3 scripts/sweep_cells.py C4 > $O/sweep_C4.txt 2>&1; echo "sweep C4 rc=$?"
timeout 900 python3 scripts/sweep_cells.py R1 0.0 0.6 0.8 1.0 1.1 1.24 1.4 1.6 1.9 > $O/sweep_R1.txt 2>&1; echo "sweep R1 rc=$?"
cat $O/sweep_C3.txt $O/sweep_C4.txt $O/sweep_R1.txt
bash scripts/profile_round.sh r03d_C3 > /dev/null 2>&1; echo "prof C3 rc=$?"
BENCH_ARGS="--config C4 --no-side" bash scripts/profile_round.sh r03d_C4 > /dev/null 2>&1; echo "prof C4 rc=$?"
BENCH_ARGS="--config C5" bash scripts/profile_round.sh r03d_C5 > /dev/null 2>&1; echo "prof C5 rc=$?"
