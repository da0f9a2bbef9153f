"""Hunting the intermittent stall of the drive (NOTEBOOK round 5, last row): the C3-scale drive of bench.py's
`frame_pipeline_moving`, again and again -- a fresh handle every other drive, the three wait policies in rotation, every wait
under a short deadline so that a stall comes back as S2M_ERR_TIMEOUT with the handle's state instead of a hung process.
python scripts/hang_hunt.py [drives] [frames] [deadline_ms]
Prints one line per drive and a summary per policy: median / p99 / max frame, frames over 1 ms and over 5 ms, errors."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from daliti_amd import Engine, synth, S2MError
from daliti_amd.world import World, run_frames

drives = int(sys.argv[1]) if len(sys.argv) > 1 else 40
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 520
deadline = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
M = int(os.environ.get("M", 5_000_000))
warm = 8
L = synth.CONFIGS["C3"]["L"]
w = World(L, 6.0 * L, 1.0)
t0 = time.time()
seed = w.seed_map(M)
sw = w.sweeps(0, frames + warm, 64, 1024, threads=int(os.environ.get("THREADS", 16)))
_, _, P0 = synth.filter_inputs()
print("workload ready in %.1f s: %d sweeps, seed %d points; nproc %d, affinity %d" % (
    time.time() - t0, frames + warm, len(seed), os.cpu_count(), len(os.sched_getaffinity(0))), flush=True)
names = ["spin", "yield", "sleep"]
stats = {k: [] for k in range(3)}
errors = []
e = None
for d in range(drives):
    pol = d % 3
    if e is None or d % 2 == 0:
        if e is not None:
            rc = e.close()
            if rc:
                errors.append((d, "close", rc))
        e = Engine(max_iter=5, wait_policy=pol, wait_timeout_ms=deadline)
    else:
        e.set_config(wait_policy=pol)
    t1 = time.time()
    try:
        e.map_build(seed)
        e.fov_reset()
        r = run_frames(e, sw, P0, frames, warm)
    except (S2MError, RuntimeError) as ex:
        errors.append((d, names[pol], str(ex)))
        print("drive %d (%s): ERROR after %.1f s: %s" % (d, names[pol], time.time() - t1, ex), flush=True)
        try:
            e.close()
        except Exception:
            pass
        e = None
        continue
    ms = r["ms"][warm:]
    how = r["how"][warm:]
    stats[pol].append(ms)
    print("drive %d (%s, %.1f s): median %.3f p99 %.3f max %.3f ms; >1 ms: %d, >5 ms: %d; in place %d relaid %d rebuilt %d; %s" % (
        d, names[pol], time.time() - t1, np.median(ms), np.percentile(ms, 99), ms.max(), (ms > 1.0).sum(), (ms > 5.0).sum(),
        (how == 2).sum(), (how == 1).sum(), (how == 0).sum(),
        [(int(i), round(float(ms[i]), 2)) for i in np.argsort(ms)[-3:][::-1]]), flush=True)
    for i in np.nonzero(ms > 2.0)[0]:   # where a frame of milliseconds went: the wall times of its calls (set_from_raw, prefetch, iterated_update, prepare, map_incremental, fov)
        print("   frame %d: %.3f ms, calls %s, %d device allocations" % (i, ms[i], np.round(r["stage_ms"][warm + i], 3), r["allocs"][warm + i]), flush=True)
if e is not None:
    e.close()
for pol in range(3):
    if stats[pol]:
        a = np.concatenate(stats[pol])
        per = np.array([np.median(m) for m in stats[pol]])
        print("policy %s: %d drives, %d frames: median %.3f (drive medians %.3f..%.3f) p99 %.3f p99.9 %.3f max %.3f ms; frames >1 ms: %d, >5 ms: %d" % (
            names[pol], len(stats[pol]), len(a), np.median(a), per.min(), per.max(), np.percentile(a, 99), np.percentile(a, 99.9), a.max(),
            (a > 1.0).sum(), (a > 5.0).sum()))
print("errors: %d %s" % (len(errors), errors))
