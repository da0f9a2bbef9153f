"""Per-kernel averages of the rocprofv3 passes scripts/ab.sh counters wrote: one line per (variant, kernel).
usage: ab_counters.py <dir> <variant>..."""
import collections
import csv
import glob
import os
import sys


def short(name):
    n = name.split("(")[0].replace("void s2m::", "").replace("s2m::", "")
    return n[:44]


def main():
    out, variants = sys.argv[1], sys.argv[2:]
    for v in variants:
        stats = {}
        for f in glob.glob(os.path.join(out, v + "_stats", "**", "*kernel_stats.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "s2m::" in r["Name"]:
                    stats[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for sub in ("sq", "ta", "tcp", "fetch"):
            for f in glob.glob(os.path.join(out, "%s_%s" % (v, sub), "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    k = short(r["Kernel_Name"])
                    if "match" in k or "reduce" in k or "fit" in k:
                        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in sorted(acc):
            c = {n: sum(x) / len(x) for n, x in acc[k].items()}
            calls, avg = stats.get(k, (0, 0.0))
            wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
            waves = c.get("SQ_WAVES", 0.0) or 1.0
            print("%-6s %-44s calls %5d avg %7.2f us | VALU/wave %6.0f wait %4.1f%% issue-stall %4.1f%% active %4.1f%% | "
                  "TA busy %8.0f  L1 req %9.0f  fetch %8.0f KiB" % (
                      v, k, calls, avg, c.get("SQ_INSTS_VALU", 0.0) / waves, 100 * c.get("SQ_WAIT_ANY", 0.0) / wc,
                      100 * c.get("SQ_WAIT_INST_ANY", 0.0) / wc, 100 * c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc,
                      c.get("TA_BUSY_avr", 0.0), c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0), c.get("FETCH_SIZE", 0.0)))


if __name__ == "__main__":
    main()
