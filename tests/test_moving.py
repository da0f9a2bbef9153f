"""A drive through a world that is longer than any fixed grid (tools/world.h): raw sweeps from a moving sensor -> undistort +
voxel grid -> iterated update -> map_incremental -> field-of-view trim, sixty frames, the engine against the oracle's
sequential restatement of the same loop (laserMapping.cpp:731-1175).  The map leaves the box of its seed by twenty times
its size, the trim deletes what falls behind, and nothing is ever rebuilt: the map grows brick by brick like ikd-Tree grows
node by node (ikd_Tree.cpp:477-573).

CPU: the world generator itself (determinism, geometry, the undistortion contract)."""
import numpy as np
import pytest

from conftest import bits, ranked_tree


def _world(step=9.0):
    from daliti_amd.world import World
    return World(24.0, 700.0, step, pitch=8.0, lane=2.5, max_range=40.0, wobble=0.5, wobble_period=60.0)


def _rows(a):
    a = np.ascontiguousarray(a, np.float32).reshape(-1, 3)
    return a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]


def test_world_sweeps_are_deterministic_and_undistort_onto_the_world(oracle):
    w = _world()
    a = w.sweeps(3, 2, 16, 256, threads=1)
    b = w.sweeps(4, 1, 16, 256, threads=2)
    n = int(b["n"][0])
    assert a["n"][1] == n and (bits(a["rec"][1][:n]) == bits(b["rec"][0][:n])).all()
    assert (a["x_prop"][1] == b["x_prop"][0]).all() and (a["poses"][1] == b["poses"][0]).all()
    # returns beyond the range are dropped, the rest carry their firing time
    rec = b["rec"][0][:n]
    assert 0.5 * 16 * 256 < n <= 16 * 256 and np.linalg.norm(rec[:, :3], axis=1).max() <= 40.0 + 0.1
    assert (np.diff(rec[:, 4]) >= 0).all() and rec[0, 6] == np.float32(0.1)
    # the sweep is distorted by the motion (9 m per frame: up to 9 m); the reference's backward propagation, fed with the
    # poses that come with it, puts the points back onto the world's surfaces as seen from the TRUE end-of-sweep pose
    und, _ = oracle.undistort(rec, 4, 6, b["poses"][0], b["x_prop"][0], True)
    pw = und.astype(np.float64) + b["x_true"][0][9:12]

    def off_planes(p):
        return np.minimum.reduce([np.abs(p[:, 2]), np.abs(p[:, 2] - 10.0), np.abs(p[:, 1] - 12.0), np.abs(p[:, 1] + 12.0)])
    hit_plane = off_planes(pw) < 0.05
    raw = off_planes(rec[:, :3].astype(np.float64) + b["x_true"][0][9:12]) < 0.05
    assert hit_plane.mean() > 0.7 and hit_plane.sum() >= raw.sum()
    # the prediction error is the size asked for
    e = b["x_prop"][0][9:12] - b["x_true"][0][9:12]
    assert abs(np.linalg.norm(e) - 0.02) < 1e-12
    seed = w.seed_map(5000)
    assert seed[:, 0].min() > -12.1 and seed[:, 0].max() < 12.1 and np.abs(seed[:, 1]).max() < 12.1


@pytest.mark.gpu
@pytest.mark.parametrize("beside", [None, "12", "25,regrid"])
def test_sixty_frames_of_a_drive_match_the_oracle(oracle, monkeypatch, beside):
    """beside: a new layout of the whole map is produced BESIDE the frames behind the n-th update (S2M_BESIDE_AT: snapshot, build
    on the handle's layout thread -- with a new cell size for "regrid" --, the update calls that arrive meanwhile run again on
    it, swap between two updates: s2m_engine_relay.cpp; ikd-Tree's rebuild thread, ikd_Tree.cpp:192-203, 229-367).  Every
    stage of every frame still equals the oracle's, the follower of the change log notices nothing (same ids), and the final
    map and neighbour lists are the oracle's."""
    from daliti_amd import Engine, synth
    if beside:
        monkeypatch.setenv("S2M_BESIDE_AT", beside)
    else:
        monkeypatch.setenv("S2M_NO_BESIDE", "1")
    frames, beams, az, fs = 60, 16, 256, 0.5
    w = _world()
    seed = w.seed_map(30000)
    sw = w.sweeps(0, frames, beams, az, threads=8)
    _, _, P0 = synth.filter_inputs()
    cfg = oracle.default_cfg(max_iter=5, feat_threshold=50)
    e = Engine(max_iter=5, feat_threshold=50, cell_size=0.5)
    e.map_build(seed)
    om = oracle.Map(seed)
    fov = oracle.FovSegmenter(1000.0)
    trimmed = 0
    seed_box = (seed.min(axis=0), seed.max(axis=0))
    # a follower of the map (s2m_map_get_changes): a mirror keyed by point id, brought up to date after every frame
    from daliti_amd.engine import apply_map_changes
    ch = e.map_changes(0)
    token = ch.token
    assert ch.resync
    mirror_ids, mirror_xyz = e.map_ids(), e.map_points().copy()
    assert (mirror_ids == np.arange(len(seed))).all() and (bits(mirror_xyz) == bits(seed)).all()
    for f in range(frames):
        n = int(sw["n"][f])
        rec, poses, xp = sw["rec"][f][:n], sw["poses"][f], sw["x_prop"][f]
        # -- the engine
        nd = e.scan_set_from_raw(rec, 4, 6, poses, xp, fs)
        got = e.iterated_update(xp, xp, P0)
        na, nb = e.map_incremental(got["x"], fs)
        assert e.map_last_update_merged(), f
        deleted = e.fov_segment(got["x"][9:12], 1000.0)[2]
        # -- the oracle, stage by stage
        und, _ = oracle.undistort(rec, 4, 6, poses, xp, True)
        down = oracle.voxel_downsample(und, fs)
        assert nd == len(down), f
        tree = oracle.KdTree(om.points())
        ref = oracle.iterated_update(cfg, tree, down, xp, xp, P0)
        assert got["iters"] == ref["iters"] and (got["effct"] == ref["effct"]).all(), (f, got["effct"], ref["effct"])
        assert np.abs(got["x"] - ref["x"]).max() < 1e-9, (f, np.abs(got["x"] - ref["x"]).max())
        nn = ref["nn_idx"]
        to_add, no_down = oracle.map_incremental_lists(down, ref["x"], tree.xyz[np.maximum(nn, 0)], (nn >= 0).sum(1).astype(np.int32), fs)
        assert (na, nb) == (len(to_add), len(no_down)), (f, na, nb, len(to_add), len(no_down))
        om.add(to_add, True, fs)
        om.add(no_down, False)
        want_deleted = 0
        for box in fov.step(ref["x"][9:12]):
            want_deleted += om.delete_box(box)
        assert deleted == want_deleted, (f, deleted, want_deleted)
        trimmed += int(want_deleted > 0)
        assert e.map_size() == om.size(), f
        ch = e.map_changes(token)
        token = ch.token
        assert not ch.resync and len(np.unique(ch.rem_ids)) == len(ch.rem_ids), f
        assert len(ch.boxes) == (1 if want_deleted > 0 else 0), f          # a trim is reported as its box, not as its points
        mirror_ids, mirror_xyz = apply_map_changes(mirror_ids, mirror_xyz, ch)
        order = np.argsort(mirror_ids)
        mirror_ids, mirror_xyz = mirror_ids[order], mirror_xyz[order]
        assert len(mirror_ids) == e.map_size() and (np.diff(mirror_ids.astype(np.int64)) > 0).all(), f
    st = e.map_update_stats()
    if beside:   # the layout beside the frames happened, once, and no update waited for it
        assert st["relaid_beside"] == 1 and st["regridded_beside"] == (1 if "regrid" in beside else 0), st
    else:
        assert st["relaid_beside"] == 0, st
    # the drive left the seed's box far behind, the trim removed map points, nothing was rebuilt -- and most updates touched
    # only the bricks they changed
    lo, hi = e.map_grid()
    assert sw["x_true"][frames - 1][9] > 20 * (seed_box[1][0] - seed_box[0][0]) and trimmed >= 1
    assert st["rebuilt"] == 0 and st["regridded"] == 0 and e.map_inplace_updates() >= frames // 2, (st, e.map_inplace_updates())
    # 4 m bricks: the box of bricks grew 20 x and the window followed (a layout beside the frames sizes its own window)
    assert (hi[0] - lo[0]) * e.map_info()["cell"] * 8 > 200 and (beside or st["top_relaid"] >= 1), (lo, hi, st)
    pts = e.map_points()
    assert (bits(_rows(pts)) == bits(_rows(om.points()))).all()
    # the mirror that only ever saw the changes IS the map: same ids, same points, same (ascending id) order
    assert (mirror_ids == e.map_ids()).all() and (bits(mirror_xyz) == bits(pts)).all()
    # exact neighbours in the engine's order on the final map, for a scan at the last pose
    x = got["x"]
    e.residual_pass(x, True)
    idx, d2 = e.get_neighbors()
    oi, od, _ = ranked_tree(oracle, e, pts).knn5(oracle.body_to_world(x, e.scan_get()))
    near = od[:, 4] <= 5.0
    assert near.sum() > 100 and (bits(d2[near]) == bits(od[near])).all() and (idx[near] == oi[near]).all()
    e.close()


@pytest.mark.gpu
def test_the_drive_by_both_roads(monkeypatch):
    """The same thirty frames through a handle as it comes (the scan's batches staged and prepared by one workgroup, counts
    left on the device, the voxel grid launched before its box is known, sweeps in time order not sorted) and through one made
    with every such shortcut switched off: the same poses, the same map in the same layout."""
    from daliti_amd import Engine, synth
    frames, beams, az, fs = 30, 16, 256, 0.5
    w = _world()
    seed = w.seed_map(30000)
    sw = w.sweeps(0, frames, beams, az, threads=8)
    _, _, P0 = synth.filter_inputs()
    runs = []
    for road in (0, 1):
        if road == 1:
            for k in ("S2M_NO_FUSED_PREP", "S2M_EXACT_STAGE", "S2M_NO_FUSED_STAGE", "S2M_NO_VOXEL_HINT", "S2M_NO_TIME_SHORTCUT"):
                monkeypatch.setenv(k, "1")
        e = Engine(max_iter=5, feat_threshold=50, cell_size=0.5)
        e.map_build(seed)
        xs = []
        for f in range(frames):
            n = int(sw["n"][f])
            rec, poses, xp = sw["rec"][f][:n], sw["poses"][f], sw["x_prop"][f]
            if f + 1 < frames:
                e.scan_prefetch_raw(sw["rec"][f + 1][:int(sw["n"][f + 1])], 4, 6)
            e.scan_set_from_raw(rec, 4, 6, poses, xp, fs)
            got = e.iterated_update(xp, xp, P0)
            e.map_incremental(got["x"], fs)
            e.fov_segment(got["x"][9:12], 1000.0)
            xs.append(got["x"].copy())
        assert e.map_inplace_updates() >= frames // 2
        runs.append((np.array(xs), e.map_points().copy(), e.map_ids().copy(), e.map_rank().copy()))
        e.close()
    for a, b in zip(runs[0], runs[1]):
        assert a.shape == b.shape
        assert (a.view(np.uint64) == b.view(np.uint64)).all() if a.dtype == np.float64 else (bits(a) == bits(b)).all() if a.dtype == np.float32 else (a == b).all()


def _voxel_key(p, fs=0.5):
    k = np.floor(p.astype(np.float32) / np.float32(fs)).astype(np.int64) + (1 << 20)
    return (k[:, 0] << 42) | (k[:, 1] << 21) | k[:, 2]


def _oracle_map_update(oracle, omap, to_add, no_down, fs=0.5):
    """Add_Points(PointToAdd, true) + Add_Points(PointNoNeedDownsample, false) on a map of millions of points: the voxel rule
    looks at the old points of a new point's own voxel only (ikd_Tree.cpp:491-520; 0.5 m voxels: Box_of_Point's float
    arithmetic is exact), so the oracle's sequential restatement runs on the old points of the touched voxels and the rest
    of the map comes through untouched (as tests/test_gpu_fullsize.py::test_fullsize_map_incremental_matches_oracle)."""
    touched = np.isin(_voxel_key(omap, fs), np.unique(_voxel_key(to_add, fs))) if len(to_add) else np.zeros(len(omap), bool)
    om = oracle.Map(omap[touched] if touched.any() else omap[:0])
    om.add(to_add, True, fs)
    om.add(no_down, False)
    return np.concatenate([omap[~touched], om.points()])


@pytest.mark.gpu
def test_c3_scale_drive_matches_the_oracle_where_it_is_sampled(oracle):
    """Parity at the scale of the drive (the only validation the reference has is a drive, README.md:38-55).
    The drive of bench.py's `frame_pipeline_moving` -- 65 536 rays per sweep, a 5 M-point seed, 1 m per frame, cube_len 1000 m:
    the first field-of-view trim at frame 343 + 8 warm-up frames -- on the engine, and the oracle's restatement of
    laserMapping.cpp:731-1175 beside it:
      * chained over the first 8 frames on ITS OWN map (voxel rule, both Add_Points): poses <= 1e-9, effective counts per
        iteration, list sizes equal every frame -- and the same pose error against the truth, and after the 8 frames the
        same map as a set of 5 M points;
      * at four later frames (100, 300, the first trim, a few behind it) on the map fetched from the engine just before the
        frame (tie order = the engine's, checked against the documented order): voxel-grid count, iterations, effective
        counts, pose <= 1e-9, list sizes, points the trim deletes."""
    from daliti_amd import Engine, synth
    from daliti_amd.world import World, run_frames
    fs, cube = 0.5, 1000.0
    L = synth.CONFIGS["C3"]["L"]
    w = World(L, 6.0 * L, 1.0)
    seed = w.seed_map(synth.CONFIGS["C3"]["M"])
    total = 358
    sw = w.sweeps(0, total, 64, 1024, threads=16)
    _, _, P0 = synth.filter_inputs()
    cfg = oracle.default_cfg(max_iter=5)
    e = Engine(max_iter=5)
    e.map_build(seed)
    fov = oracle.FovSegmenter(cube)

    def engine_frame(f):
        n = int(sw["n"][f])
        nd = e.scan_set_from_raw(sw["rec"][f][:n], 4, 6, sw["poses"][f], sw["x_prop"][f], fs)
        scan = e.scan_get()
        got = e.iterated_update(sw["x_prop"][f], sw["x_prop"][f], P0)
        na, nb = e.map_incremental(got["x"], fs)
        deleted = e.fov_segment(got["x"][9:12], cube)[2]
        return nd, scan, got, na, nb, deleted

    def oracle_frame(f, tree, scan):
        n = int(sw["n"][f])
        und, _ = oracle.undistort(sw["rec"][f][:n], 4, 6, sw["poses"][f], sw["x_prop"][f], True)
        down = oracle.voxel_downsample(und, fs)
        # (the undistortion is within one float ulp of this libm, not bit-identical to it -- DESIGN 9(4) -- so the update is
        # compared on the engine's own scan; the front half by its count and to the last bit's size)
        assert len(down) == len(scan) and np.abs(down - scan).max() <= 2e-5, (f, len(down), len(scan))
        ref = oracle.iterated_update(cfg, tree, scan, sw["x_prop"][f], sw["x_prop"][f], P0)
        nn = ref["nn_idx"]
        to_add, no_down = oracle.map_incremental_lists(scan, ref["x"], tree.xyz[np.maximum(nn, 0)], (nn >= 0).sum(1).astype(np.int32), fs)
        return ref, to_add, no_down

    def same_update(f, got, ref):
        assert got["iters"] == ref["iters"] and (got["effct"] == ref["effct"]).all(), (f, got["effct"], ref["effct"])
        assert np.abs(got["x"] - ref["x"]).max() < 1e-9, (f, np.abs(got["x"] - ref["x"]).max())

    # -- the oracle chained on its own map
    omap = seed.copy()
    chain = 8
    for f in range(chain):
        nd, scan, got, na, nb, deleted = engine_frame(f)
        ref, to_add, no_down = oracle_frame(f, oracle.KdTree(omap), scan)
        same_update(f, got, ref)
        assert (na, nb) == (len(to_add), len(no_down)) and deleted == 0, (f, na, nb, len(to_add), len(no_down))
        assert fov.step(ref["x"][9:12]) == []
        err_e = np.linalg.norm(got["x"][9:12] - sw["x_true"][f][9:12])
        err_o = np.linalg.norm(ref["x"][9:12] - sw["x_true"][f][9:12])
        assert abs(err_e - err_o) < 1e-9 and err_o < 0.05, (f, err_e, err_o)     # the oracle is as far from the truth as the engine
        omap = _oracle_map_update(oracle, omap, to_add, no_down, fs)
    pts = e.map_points()
    assert len(pts) == len(omap) and (bits(_rows(pts)) == bits(_rows(omap))).all()
    del omap
    # -- the drive goes on in the C++ loop; at the sampled frames the map is fetched and the frame runs on both sides
    at = chain
    samples = (100, 300, 351, 355)
    trims = sampled_trims = 0
    for s in samples:
        if s > at:
            part = {k: v[at:] for k, v in sw.items()}
            r = run_frames(e, part, P0, s - at, 0, cube_len=cube)
            assert (r["how"] == 2).all(), (at, s, r["how"])                      # every update in place
            for x in r["x"]:
                fov.step(x[9:12])                                                # (no trim before the first sample behind it)
            trims += int((r["deleted"] > 0).sum())
        before = e.map_points()
        tree = ranked_tree(oracle, e, before)
        nd, scan, got, na, nb, deleted = engine_frame(s)
        ref, to_add, no_down = oracle_frame(s, tree, scan)
        same_update(s, got, ref)
        assert (na, nb) == (len(to_add), len(no_down)), (s, na, nb, len(to_add), len(no_down))
        want_deleted = 0
        boxes = fov.step(ref["x"][9:12])
        if boxes:
            after = _oracle_map_update(oracle, before, to_add, no_down, fs)
            gone = np.zeros(len(after), bool)
            for b in boxes:
                b = np.asarray(b, np.float32)
                gone |= ((after >= b[:3]) & (after < b[3:])).all(axis=1)
            want_deleted = int(gone.sum())
        assert deleted == want_deleted, (s, deleted, want_deleted)
        sampled_trims += int(deleted > 0)
        at = s + 1
    assert trims == 0 and sampled_trims == 1 and deleted == 0, (trims, sampled_trims)   # the first trim was one of the sampled frames
    st = e.map_update_stats()
    assert st["rebuilt"] == 0 and st["relaid"] <= 1, st                          # (the first update of a dense build lays the room out)
    e.close()
