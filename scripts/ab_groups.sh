#!/bin/bash
# A/B on the box: bench.py (C3, no CPU leg) for the row-run search over G (lanes per query) and NB (batches per
# trip), and the per-cell form; per-kernel times from the engine's HIP events
OUT=gpurun_out/${1:-ab}
mkdir -p $OUT
for g in ${GROUPS_:-1 2 4}; do for nb in ${NBS_:-1 2 3}; do
  S2M_MATCH_GROUP=$g S2M_EASY_NB=$nb python bench.py --no-cpu --steps 100 > $OUT/rows_g${g}_nb$nb.json 2> $OUT/rows_g${g}_nb$nb.err
done; done
S2M_EASY_CELLS=1 python bench.py --no-cpu --steps 100 > $OUT/cells_g4.json 2> $OUT/cells_g4.err
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]
        print(f.split("/")[-1], "ms/step %.4f" % d["ms_per_step"], "search %.1f us" % (1e3*r["search_kernels_only"]["avg_ms"]), "fit %.1f us" % (1e3*r["reduce_fit_avg_ms"]), "reuse %.1f us" % (1e3*d["roofline_reuse"]["avg_launch_ms"]), "pose_err %.5f" % d["pose_error_vs_truth_m"])
    except Exception as e: print(f, "ERR", e)
PY
