cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r04v
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r04v/tr -- python3 $R/bench.py --config C3 --cpu-steps 1 --steps 20 > $R/gpurun_out/r04v/bench.log 2>&1
python3 $R/scripts/frame_trace.py $R/gpurun_out/r04v/tr 64 0.36 seq > $R/gpurun_out/r04v/frame_trace.txt
rm -rf $R/gpurun_out/r04v/tr
tail -130 $R/gpurun_out/r04v/frame_trace.txt
