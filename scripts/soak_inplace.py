#!/usr/bin/env python3
"""Soak test of the in-place map update (GPU box): a few hundred random updates -- down-sampled adds near existing points,
plain adds (into existing bricks, into empty bricks, bursts), box deletes, points outside the grid -- applied to two handles,
one updating in place (default) and one merging every update (S2M_NO_SLAB=1 cannot differ per handle: the second handle runs
in a child process).  After every step: same size, same points in the same caller order, and every 10th step the same
neighbour lists for a fixed set of queries.  usage: soak_inplace.py [steps] [seed]"""
import os, sys, subprocess, pickle
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import numpy as np

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
child = len(sys.argv) > 3 and sys.argv[3] == "child"


def run():
    from daliti_amd import Engine, synth
    sc = synth.make_small(M=40000)
    rs = np.random.RandomState(seed)
    base = sc["map"]
    lo, hi = base.min(axis=0), base.max(axis=0)
    q = np.concatenate([sc["scan"], rs.uniform(lo, hi, (1500, 3)).astype(np.float32)]).astype(np.float32)
    e = Engine(cell_size=0.45)
    e.map_build(base)
    out = []
    for k in range(steps):
        kind = rs.randint(8)
        cur = e.map_points()
        if kind in (0, 1, 2):     # voxel-rule add near existing points
            pick = rs.choice(len(cur), rs.randint(50, 2500))
            e.map_add(cur[pick] + rs.normal(0, rs.choice([0.01, 0.05, 0.3]), (len(pick), 3)).astype(np.float32), True, 0.5)
        elif kind == 3:           # plain add near existing points
            pick = rs.choice(len(cur), rs.randint(1, 600))
            e.map_add(cur[pick] + rs.normal(0, 0.02, (len(pick), 3)).astype(np.float32), False, 0.5)
        elif kind == 4:           # blobs anywhere in the box (empty bricks too)
            c = rs.uniform(lo, hi, (rs.randint(1, 6), 3))
            pts = np.concatenate([ci + rs.uniform(-0.4, 0.4, (rs.randint(5, 200), 3)) for ci in c]).astype(np.float32)
            e.map_add(pts, bool(rs.randint(2)), 0.5)
        elif kind == 5:           # box delete
            c = rs.uniform(lo, hi)
            s = rs.uniform(0.2, 2.5, 3)
            e.map_delete_boxes(np.float32([np.r_[c - s, c + s]]))
        elif kind == 6:           # a burst into one cubic metre
            c = cur[rs.randint(len(cur))]
            e.map_add((c + rs.uniform(-0.5, 0.5, (rs.randint(500, 4000), 3))).astype(np.float32), bool(rs.randint(2)), 0.5)
        else:                     # now and then a point outside the grid
            if rs.randint(4) == 0:
                e.map_add((hi + rs.uniform(5, 30, (3, 3))).astype(np.float32), False, 0.5)
            else:
                e.map_delete_boxes(np.float32([np.r_[lo - 1, lo - 0.5]]))     # hits nothing
        rec = [e.map_size(), e.map_points().copy(), None]
        if k % 10 == 9:
            e.scan_set(q)
            e.residual_pass(sc["x_true"], True)
            rec[2] = tuple(a.copy() for a in e.get_neighbors())
        out.append(rec)
    stats = (e.map_inplace_updates(), e.map_update_stats())
    e.close()
    return out, stats


if child:
    res = run()
    pickle.dump(res, open(sys.argv[4], "wb"))
    sys.exit(0)
tmp = "/tmp/soak_child_%d.pkl" % os.getpid()
env = dict(os.environ, S2M_NO_SLAB="1")
p = subprocess.Popen([sys.executable, __file__, str(steps), str(seed), "child", tmp], env=env)
a, sa = run()
assert p.wait() == 0
b, sb = pickle.load(open(tmp, "rb"))
os.remove(tmp)
bad = 0
for k, (ra, rb) in enumerate(zip(a, b)):
    ok = ra[0] == rb[0] and ra[1].shape == rb[1].shape and (ra[1].view(np.uint32) == rb[1].view(np.uint32)).all()
    if ok and ra[2] is not None:
        ok = (ra[2][0] == rb[2][0]).all() and (ra[2][1].view(np.uint32) == rb[2][1].view(np.uint32)).all()
    if not ok:
        bad += 1
        print("step %d differs: sizes %d / %d" % (k, ra[0], rb[0]))
        if bad > 5:
            break
print("soak: %d steps, seed %d: %s; in place %d, relaid %d, rebuilt %d; all-merge run: %d in place" % (
    steps, seed, "IDENTICAL" if bad == 0 else "%d steps differ" % bad, sa[0], sa[1]["relaid"], sa[1]["rebuilt"], sb[0]))
sys.exit(1 if bad else 0)
