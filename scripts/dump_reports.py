"""The change log's reports along the bench's drive, written to a file that scripts/mirror_replay.cpp applies to the follower's
mirror on a host without a device (what applying a report costs is host work: measure it where the profiler is).
python scripts/dump_reports.py <out file> [frames] [dense|refdens]
File: int64 m, then m x {float x, y, z; uint32 id} (the map the follower starts from); then per frame int64 na, nr, nb and
the arrays add_xyz (na x 3 float), add_ids (na uint32), rem_xyz, rem_ids, boxes (nb x 6 float), box_after_added (nb int64), box_after_removed."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from daliti_amd import Engine, synth
from daliti_amd.world import World
out = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 250
seedkind = sys.argv[3] if len(sys.argv) > 3 else "refdens"
L = synth.CONFIGS["C3"]["L"]
w = World(L, 6.0 * L, 1.0)
seed = w.seed_map(5_000_000)
sw = w.sweeps(0, frames, 64, 1024, threads=32)
_, _, P0 = synth.filter_inputs()
e = Engine(max_iter=5)
if seedkind == "refdens":
    e.map_build(seed[:1])
    for lo in range(0, len(seed), 1 << 20):
        e.map_add(seed[lo:lo + (1 << 20)], True, 0.5)
else:
    e.map_build(seed)
e.fov_reset()
ch = e.map_changes(0)
token = ch.token
ids, pts = e.map_ids(), e.map_points()
with open(out, "wb") as f:
    np.int64([len(ids)]).tofile(f)
    rec = np.zeros(len(ids), dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("id", "<u4")])
    rec["x"], rec["y"], rec["z"], rec["id"] = pts[:, 0], pts[:, 1], pts[:, 2], ids
    rec.tofile(f)
    tot = [0, 0, 0]
    for k in range(frames):
        n = int(sw["n"][k])
        r, poses, xp = sw["rec"][k][:n], sw["poses"][k], sw["x_prop"][k]
        e.scan_set_from_raw(r, 4, 6, poses, xp, 0.5)
        got = e.iterated_update(xp, xp, P0)
        e.map_incremental(got["x"], 0.5)
        e.fov_segment(got["x"][9:12], 1000.0)
        ch = e.map_changes(token)
        token = ch.token
        assert not ch.resync
        np.int64([len(ch.add_ids), len(ch.rem_ids), len(ch.boxes)]).tofile(f)
        ch.add_xyz.astype("<f4").tofile(f); ch.add_ids.astype("<u4").tofile(f)
        ch.rem_xyz.astype("<f4").tofile(f); ch.rem_ids.astype("<u4").tofile(f)
        ch.boxes.astype("<f4").tofile(f); ch.box_after_added.astype("<i8").tofile(f); ch.box_after_removed.astype("<i8").tofile(f)
        tot[0] += len(ch.add_ids); tot[1] += len(ch.rem_ids); tot[2] += len(ch.boxes)
print("%s: map %d points, %d frames, %.0f added %.0f removed per frame, %d boxes; %.1f MB" % (out, len(ids), frames, tot[0] / frames, tot[1] / frames, tot[2], os.path.getsize(out) / 1e6))
assert e.map_size() == len(ids) + tot[0] - tot[1] or tot[2] > 0
e.close()
