"""The follower of the change log on the bench's long drive, with and without layouts beside the frames, lag 1 and 0:
python scripts/follower_check.py [frames]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from daliti_amd import Engine, synth
from daliti_amd.world import World, run_frames
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1100
warm = 8
L = synth.CONFIGS["C3"]["L"]
w = World(L, 6.0 * L, 1.0)
seed = w.seed_map(5_000_000)
sw = w.sweeps(0, frames + warm, 64, 1024, threads=32)
_, _, P0 = synth.filter_inputs()
cases = (("inside, lag 1", "S2M_NO_BESIDE", 2), ("beside, lag 1", None, 2), ("beside, lag 0", None, 1), ("beside, thread", None, 3), ("refdens, thread", None, 3))
if os.environ.get("ONLY"):
    cases = tuple(c for c in cases if c[0] == os.environ["ONLY"]) * int(os.environ.get("REPEAT", 1))
for name, env, publish in cases:
    if env:
        os.environ[env] = "1"
    e = Engine(max_iter=5)
    if env:
        del os.environ[env]
    if name.startswith("refdens"):   # the seed through the voxel rule (bench.py, build_reference_density_map)
        e.map_build(seed[:1])
        for lo in range(0, len(seed), 1 << 20):
            e.map_add(seed[lo:lo + (1 << 20)], True, 0.5)
    else:
        e.map_build(seed)
    r = run_frames(e, sw, P0, frames, warm, publish=int(os.environ.get("PUB", publish)))
    ms = r["ms"][warm:]
    st = e.map_update_stats()
    print("%-14s median %.3f p99 %.3f max %.3f ms; publish median %.3f (fetch %.3f) ms; beside %d; mirror %d map %d resyncs %d missed %d; trims at %s" % (
        name, np.median(ms), np.percentile(ms, 99), ms.max(), np.median(r["publish_ms"][warm:]), np.median(r["fetch_ms"][warm:]), st["relaid_beside"],
        r["mirror_points"], r["map_points"], r["mirror_resyncs"], r["mirror_missed"], [int(i) for i in np.nonzero(r["deleted"])[0]]), flush=True)
    if os.environ.get("TOP"):   # the slowest frames: frame, ms, of which the follower's part (and of that the fetch), the calls' wall times
        pm, fm = r["publish_ms"][warm:], r["fetch_ms"][warm:]
        for i in np.argsort(ms)[-int(os.environ["TOP"]):][::-1]:
            print("   frame %4d: %.3f ms, follower %.3f (fetch %.3f), calls %s, %d device allocations, update %s" % (i, ms[i], pm[i], fm[i], np.round(r["stage_ms"][warm + i], 3), r["allocs"][warm + i], ("rebuilt", "relaid", "in place")[r["how"][warm + i]]))
        print("   ", {k: v for k, v in st.items() if v})
        print("   follower's part of the frame: median %.3f p90 %.3f p99 %.3f max %.3f ms; frames above 2 x the median: %d" % (
            np.median(pm), np.percentile(pm, 90), np.percentile(pm, 99), pm.max(), int((ms > 2 * np.median(ms)).sum())))
        print("   median per 100 frames: frame", " ".join("%.3f" % np.median(ms[k:k + 100]) for k in range(0, frames, 100)))
        print("                       follower", " ".join("%.3f" % np.median(pm[k:k + 100]) for k in range(0, frames, 100)))
    e.close()
