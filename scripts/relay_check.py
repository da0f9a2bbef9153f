"""The layout beside the frames on the bench's long drive (s2m_engine_relay.cpp): the same 1 100 frames through a handle that
renews its layout beside the frames (default) and through one that does not (S2M_NO_BESIDE=1: the layout is renewed inside the
update that hits the limit, as in round 5) -- poses, final map (ids and points), pose error against the truth, slow frames;
then once more with a follower of the change log one call behind.   python scripts/relay_check.py [frames]"""
import os, sys, time
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
out = {}
cases = (("beside", None, 0), ("inside", "S2M_NO_BESIDE", 0), ("beside+follower", None, 2))
if os.environ.get("ONLY"):
    cases = tuple(c for c in cases if c[0] == os.environ["ONLY"])
for name, env, publish in cases:
    if env:
        os.environ[env] = "1"
    e = Engine(max_iter=5)
    if env:
        del os.environ[env]
    e.map_build(seed)
    t0 = time.time()
    r = run_frames(e, sw, P0, frames, warm, publish=publish)
    ms, how = r["ms"][warm:], r["how"][warm:]
    err = np.linalg.norm(r["x"][warm:, 9:12] - sw["x_true"][warm:warm + frames, 9:12], axis=1)
    st = e.map_update_stats()
    ids, pts = e.map_ids(), e.map_points()
    out[name] = (r["x"].copy(), ids, pts)
    print("%-16s median %.3f p99 %.3f max %.3f ms; in place %d relaid %d rebuilt %d; beside %d (regrid %d), regridded %d; map %d points, cell %.3f; "
          "pose error vs truth median %.3f max %.3f m (frame %d); slowest %s; mirror %s" % (
              name, np.median(ms), np.percentile(ms, 99), ms.max(), (how == 2).sum(), (how == 1).sum(), (how == 0).sum(), st["relaid_beside"],
              st["regridded_beside"], st["regridded"], len(ids), e.map_info()["cell"], np.median(err), err.max(), int(err.argmax()),
              [(int(i), round(float(ms[i]), 2), int(r["allocs"][warm + i])) for i in np.argsort(ms)[-6:][::-1]],
              (r["mirror_points"], r["map_points"], r["mirror_resyncs"], r["mirror_missed"]) if publish else "-"), flush=True)
    if os.environ.get("STAGES"):   # host wall time of the slow frames' calls: set_from_raw, prefetch, iterated_update, prepare, map_incremental, fov
        med = np.median(r["stage_ms"][warm:], axis=0)
        print("   median stages (ms):", np.round(med, 3))
        for i in np.argsort(ms)[-8:][::-1]:
            print("   frame %d: %.3f ms, stages %s, %d iterations, %d rematch passes, %d scan points" % (i, ms[i], np.round(r["stage_ms"][warm + i], 3), r["iters"][warm + i], r["rematch_passes"][warm + i], r["n_scan"][warm + i]))
        it, rp = r["iters"][warm:], r["rematch_passes"][warm:]
        for k in sorted(set(zip(it.tolist(), rp.tolist()))):
            sel = (it == k[0]) & (rp == k[1])
            print("   %d iterations / %d rematch passes: %d frames, median %.3f ms (iterated_update %.3f)" % (k[0], k[1], sel.sum(), np.median(ms[sel]), np.median(r["stage_ms"][warm:][sel, 2])))
    if os.environ.get("PER100"):
        print("   median per 100 frames:", " ".join("%.3f" % np.median(ms[k:k + 100]) for k in range(0, frames, 100)))
    e.close()
if len(out) < 3:
    sys.exit(0)
a, b = out["beside"], out["inside"]
dx = np.abs(a[0] - b[0]).max(axis=1)
print("poses: max |beside - inside| = %.3e (first frame that differs: %s)" % (dx.max(), int(np.argmax(dx > 0)) if (dx > 0).any() else None))
print("final maps: ids equal %s, points equal %s (%d / %d)" % (np.array_equal(a[1], b[1]), a[2].shape == b[2].shape and (a[2].view(np.uint32) == b[2].view(np.uint32)).all(), len(a[1]), len(b[1])))
c = out["beside+follower"]
print("with a follower: ids equal %s, points equal %s" % (np.array_equal(a[1], c[1]), a[2].shape == c[2].shape and (a[2].view(np.uint32) == c[2].view(np.uint32)).all()))
