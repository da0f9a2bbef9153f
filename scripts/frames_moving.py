"""The moving-trajectory frame leg on its own (bench.py runs the same as `frame_pipeline_moving`):
python scripts/frames_moving.py [frames] [step_m] [M] [beams] [az]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from daliti_amd import Engine, synth
from daliti_amd.world import World, run_frames

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 300
step = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
M = int(sys.argv[3]) if len(sys.argv) > 3 else 5_000_000
beams = int(sys.argv[4]) if len(sys.argv) > 4 else 64
az = int(sys.argv[5]) if len(sys.argv) > 5 else 1024
L = synth.CONFIGS["C3"]["L"]
warm = 4
w = World(L, 6.0 * L, step)
t0 = time.time()
seed = w.seed_map(M)
extra = 12
import sys, time
_t0 = time.time()
sys.stderr.write('[frames_moving] generating sweeps\n'); sys.stderr.flush()
sw = w.sweeps(0, frames + warm + extra, beams, az, threads=16)
sys.stderr.write('[frames_moving] sweeps in %.1f s\n' % (time.time() - _t0)); sys.stderr.flush()
print("gen %.1fs: seed %d pts, sweeps n min/mean/max %d/%d/%d" % (time.time() - t0, len(seed), sw["n"].min(), sw["n"].mean(), sw["n"].max()), flush=True)
e = Engine(max_iter=5)
if os.environ.get("SEED") == "r1":   # the reference's map density: the seed cloud through Add_Points(downsample 0.5 m), as config R1
    e.map_build(seed[:1])
    for lo in range(0, len(seed), 1 << 20):
        e.map_add(seed[lo:lo + (1 << 20)], True, 0.5)
    print("seed at the reference's density:", e.map_size(), "points", e.map_info())
else:
    e.map_build(seed)
_, _, P0 = synth.filter_inputs()
st0 = e.map_update_stats()
sys.stderr.write('[frames_moving] map ready at %.1f s, driving\n' % (time.time() - _t0)); sys.stderr.flush()
r = run_frames(e, sw, P0, frames, warm, cube_len=float(os.environ.get("CUBE", "901")))
st1 = e.map_update_stats()
ms = r["ms"][warm:]
how = r["how"][warm:]
err = np.linalg.norm(r["x"][:, 9:12] - sw["x_true"][:frames + warm, 9:12], axis=1)
print("frames %d step %.1f m: median %.3f p99 %.3f max %.3f ms, max/median %.2f" % (frames, step, np.median(ms), np.percentile(ms, 99), ms.max(), ms.max() / np.median(ms)))
print("how: in place %d, relaid %d, rebuilt %d; stats delta %s" % ((how == 2).sum(), (how == 1).sum(), (how == 0).sum(), {k: (st1[k] - st0[k]) if not isinstance(st1[k], dict) else st1[k] for k in st1}))
print("deleted by trim:", [(int(i), int(d)) for i, d in enumerate(r["deleted"]) if d > 0])
print("scan pts after voxel grid: mean %d; iters mean %.2f; pose err vs truth: median %.4f max %.4f m; map size %d; bets %s" % (
    r["n_scan"].mean(), r["iters"].mean(), np.median(err), err.max(), e.map_size(), e.bet_stats()))
slow = np.argsort(ms)[-8:][::-1]
print("slowest frames:", [(int(i + warm), round(float(ms[i]), 3), int(how[i]), int(r["deleted"][i + warm])) for i in slow])
print("every 25th frame ms:", " ".join("%.3f" % v for v in ms[::25]))
print("host wall per call, median ms (set_from_raw, prefetch, iterated_update, prepare, map_incremental, fov):", np.round(np.median(r["stage_ms"][warm:], axis=0), 3))
for f in range(6, 14):
    print("  frame %d: %.3f ms, stages %s, iters %d, scan %d, how %d" % (f, r["ms"][f], np.round(r["stage_ms"][f], 3), r["iters"][f], r["n_scan"][f], r["how"][f]))
it = r["iters"][warm:]
for k in sorted(set(it.tolist())):
    sel = it == k
    print("  frames with %d iterations: %d, median %.3f ms (update %.3f, map_incremental %.3f)" % (k, sel.sum(), np.median(ms[sel]),
          np.median(r["stage_ms"][warm:][sel, 2]), np.median(r["stage_ms"][warm:][sel, 4])))
print("frames with allocations:", [(int(i), int(a)) for i, a in enumerate(r["allocs"]) if a > 0])
print("map grid", e.map_grid(), e.map_info())
if os.environ.get("DUMP"):
    print("all frames ms:", " ".join("%.3f" % v for v in r["ms"]))
# a few more frames with a device sync after every stage
import torch
rows = []
if os.environ.get("NOSTAGE"):
    sys.exit(0)
for k in range(frames + warm, frames + warm + extra):
    n = int(sw["n"][k]); t = [time.perf_counter()]
    e.scan_set_from_raw(sw["rec"][k][:n], 4, 6, sw["poses"][k], sw["x_prop"][k], 0.5); torch.cuda.synchronize(); t.append(time.perf_counter())
    rr = e.iterated_update(sw["x_prop"][k], sw["x_prop"][k], P0); torch.cuda.synchronize(); t.append(time.perf_counter())
    e.map_incremental(rr["x"], 0.5); torch.cuda.synchronize(); t.append(time.perf_counter())
    e.fov_segment(rr["x"][9:12], 901.0); torch.cuda.synchronize(); t.append(time.perf_counter())
    rows.append(np.diff(t) * 1e3)
print("staged (raw_to_scan, update, map_incremental, fov) median ms:", np.round(np.median(np.array(rows[2:]), axis=0), 3), "iters", rr["iters"], "rematch", rr.get("rematch"))
for f in np.nonzero(r["deleted"])[0]:
    for g in (f - 1, f, f + 1, f + 2):
        if 0 <= g < len(r["ms"]):
            print("frame", int(g), "deleted", int(r["deleted"][g]), "ms %.3f" % r["ms"][g], "stages", np.round(r["stage_ms"][g], 3))
