"""Prints the DESIGN.md section-6 results table from profiles/<tag>_bench_*.json (scripts/r03_final.sh)."""
import json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r03z"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def load(n):
    return json.loads(open(os.path.join(root, "profiles", "%s_bench_%s.json" % (tag, n))).read().strip().splitlines()[-1])
def k(v): return ("%.1f k" % (v / 1e3))
def sci(v):
    e = int(("%e" % v).split("e")[1]); return "%.2fe%d" % (v / 10 ** e, e)
names = {"C1": "C1 10 k vs 100 k", "C2": "C2 65 k vs 1 M", "C3": "**C3 65 k vs 5 M** (headline)", "C3_extrinsic": "C3, `extrinsic_est_en`",
         "C4": "C4 131 k vs 20 M, one GPU", "R1": "R1 reference density: 34 734-pt scan vs 803 k-pt map"}
for n in ("C1", "C2", "C3", "C3_extrinsic", "C4", "R1"):
    d = load(n); c = d["cpu_baseline"]; r = d["roofline"]; rr = d["roofline_reuse"]
    print("| %s | %.3f | %s | %s | %.1f µs (%.1f + %.1f) | %.1f µs | %.0f ms (%.0f) | %.0f× (%.0f×) |" % (
        names[n], d["ms_per_step"], sci(d["value"]), k(d["eskf_iters_per_sec"]), 1e3 * r["avg_launch_ms"], 1e3 * r["search_kernels_only"]["avg_ms"],
        1e3 * r["reduce_fit_avg_ms"], 1e3 * rr["avg_launch_ms"], c["ms_per_step"], c["all_cores_variant"]["ms_per_step"],
        d["speedup_vs_cpu_1thread"], d["value"] / c["all_cores_variant"]["value"]))
for n in ("C5_K8", "C5_K16", "C5_K32", "host2", "host8", "C3_driver_form", "pyloop"):
    d = load(n)
    print("%s: ms/step %.4f evals/s %s iters/s %s scans/s %.0f" % (n, d["ms_per_step"], sci(d["value"]), k(d["eskf_iters_per_sec"]), d["scans_per_sec"]))
d = load("C3")
print("c5_batch", {q: d["c5_batch"][q] for q in ("scans_per_sec", "value", "algorithmic_GBps", "frac")})
fp = d["frame_pipeline"]; print("frame", {q: fp[q] for q in ("ms_per_frame", "median_ms", "p99_ms", "max_ms", "max_over_median", "updates", "stages_ms")})
print("step_times", d["step_times"]); print("roofline", {q: d["roofline"][q] for q in ("frac", "achieved", "traffic", "frac_of_measured_traffic", "peak_measured_copy")})
print("issue", {q: d["roofline"]["issue"][q] for q in ("valu_wave_instructions", "valu_issue_floor_ms", "frac_of_pass", "wait_share_of_wave_cycles")})
print("batched", d["roofline"]["issue"].get("batched"))
