"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Bar: neighbour sets, d2, plane, pd2, masks, effct_feat_num and the dense Jacobian rows are
bit-exact; the fp64 normal block agrees to summation-order round-off (rel 1e-12); the iterated
pose update agrees to 1e-9 (the north-star bar is 1e-4 m / 1e-4 rad).
"""
import os

import numpy as np
import pytest

from conftest import bits, ranked_tree, s_gate_candidates

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(small_scene):
    from daliti_amd import Engine
    e = Engine(max_iter=5, cell_size=0.0)
    e.map_build(small_scene["map"])
    e.scan_set(small_scene["scan"])
    yield e
    e.close()


def _oracle_pass(oracle, tree, scan, x, rematch, ps=None, ext=0):
    cfg = oracle.default_cfg(extrinsic_est_en=ext)
    ps = ps or oracle.PassState(len(scan))
    return oracle.residual_pass(cfg, tree, scan, x, rematch, ps, want_rows=True)


def test_library_is_native():
    from daliti_amd import library_path, load_library
    assert os.path.exists(library_path())
    assert load_library().s2m_abi_version() == 5


def test_knn_exact(eng, oracle, small_scene):
    x = small_scene["x_prop"]
    eng.scan_set(small_scene["scan"])
    eng.residual_pass(x, True)
    idx, d2 = eng.get_neighbors()
    qw = oracle.body_to_world(x, small_scene["scan"])
    oi, od, oc = ranked_tree(oracle, eng, small_scene["map"]).knn5(qw)
    assert (bits(d2) == bits(od)).all()
    assert (small_scene["map"][idx] == small_scene["map"][oi]).all()
    assert (idx == oi).all()


@pytest.mark.parametrize("ext", [0, 1])
def test_pass_bit_exact(oracle, small_scene, small_tree, ext):
    from daliti_amd import Engine
    e = Engine(max_iter=5, extrinsic_est_en=ext)
    e.map_build(small_scene["map"])
    e.scan_set(small_scene["scan"])
    x = small_scene["x_prop"].copy()
    if ext:  # a non-trivial extrinsic so the B, C columns are exercised
        x[12:21] = oracle.so3_exp([0.02, -0.01, 0.03]).ravel()
        x[21:24] = [0.05, -0.02, 0.1]
    ps = oracle.PassState(len(small_scene["scan"]))
    for rematch in (True, False, True):
        out = e.residual_pass(x, rematch)
        _oracle_pass(oracle, small_tree, small_scene["scan"], x, rematch, ps, ext)
        st = e.get_point_state()
        assert (st["selected"] == ps.selected).all()
        assert (st["eff"] == ps.eff).all()
        ok = ps.plane_ok.astype(bool)
        assert (bits(st["plane"][ok]) == bits(ps.plane[ok])).all()
        assert (bits(st["pd2"][ok]) == bits(ps.pd2[ok])).all()
        assert out["effct"] == ps.effct
        scale = np.abs(ps.HtH).max()
        assert np.abs(out["HtH"] - ps.HtH).max() <= 1e-12 * scale
        assert np.abs(out["Htz"] - ps.Htz).max() <= 1e-12 * max(np.abs(ps.Htz).max(), 1.0)
        assert abs(out["total_res"] - ps.total_res) <= 1e-12 * max(ps.total_res, 1.0)
        hx, h, ridx = e.get_rows()
        assert (ridx == np.nonzero(ps.eff)[0]).all()
        assert (bits(hx) == bits(ps.Hsub)).all()
        assert (bits(h) == bits(ps.meas)).all()
        # move the pose a little so the reuse pass sees a different state
        x = oracle.boxplus(x, np.r_[1e-3, -2e-3, 1e-3, 0.01, 0.01, -0.01, np.zeros(18)])
    e.close()


@pytest.mark.parametrize("ext", [0, 1])
def test_iterated_update_matches_oracle(oracle, small_scene, small_tree, ext):
    from daliti_amd import Engine
    e = Engine(max_iter=5, extrinsic_est_en=ext)
    e.map_build(small_scene["map"])
    e.scan_set(small_scene["scan"])
    xp, P = small_scene["x_prop"], small_scene["P"]
    got = e.iterated_update(xp, xp, P)
    cfg = oracle.default_cfg(extrinsic_est_en=ext, max_iter=5)
    ref = oracle.iterated_update(cfg, small_tree, small_scene["scan"], xp, xp, P)
    dense = oracle.iterated_update(cfg, small_tree, small_scene["scan"], xp, xp, P, use_dense=True)
    assert got["iters"] == ref["iters"] and got["rematch_passes"] == ref["rematch_passes"]
    assert (got["effct"] == ref["effct"]).all()
    assert (got["rematch"] == ref["rematch"]).all() and (got["conv"] == ref["conv"]).all()
    for r in (ref, dense):  # normal-equation form and the literal K-materialising form
        assert np.abs(got["x"][9:12] - r["x"][9:12]).max() < 1e-9          # metres
        dR = got["x"][:9].reshape(3, 3).T @ r["x"][:9].reshape(3, 3)
        assert np.abs(oracle.so3_log(dR)).max() < 1e-9                      # radians
        assert np.abs(got["x"] - r["x"]).max() < 1e-9
        assert np.abs(got["P"] - r["P"]).max() < 1e-12
    assert np.abs(got["solution"] - ref["solution"]).max() < 1e-9
    # and it actually registers: pose error shrinks from (5 cm, 1 deg) to the noise floor
    assert np.abs(got["x"][9:12] - small_scene["x_true"][9:12]).max() < 0.01
    e.close()


@pytest.mark.parametrize("cell", [0.08, 0.3, 1.5, 0.0])
def test_cell_size_invariance(eng, small_scene, cell):
    """The result must not depend on the grid: exercises ring growth (small cells) and crowded
    cells (large cells) against the default engine."""
    from daliti_amd import Engine
    x = small_scene["x_prop"]
    eng.scan_set(small_scene["scan"])
    eng.residual_pass(x, True)
    ref_idx, ref_d2 = eng.get_neighbors()
    ref = eng.get_point_state()
    e = Engine(cell_size=cell)
    e.map_build(small_scene["map"])
    e.scan_set(small_scene["scan"])
    e.residual_pass(x, True)
    idx, d2 = e.get_neighbors()
    st = e.get_point_state()
    assert (idx == ref_idx).all() and (bits(d2) == bits(ref_d2)).all()
    assert (bits(st["plane"]) == bits(ref["plane"])).all()
    assert (st["eff"] == ref["eff"]).all()
    e.close()


@pytest.mark.parametrize("var,val", [("S2M_EASY_NB", "2"), ("S2M_EASY_NB", "3")])
def test_first_shell_kernel_variants(eng, small_scene, var, val):
    """Both instantiations of the first-shell kernel (two point batches per trip: scans beyond 98 k points and batched
    launches; three: everything else) return the identical neighbours and block."""
    from daliti_amd import Engine
    x = small_scene["x_prop"]
    eng.scan_set(small_scene["scan"])
    ref_out = eng.residual_pass(x, True)
    ref_idx, ref_d2 = eng.get_neighbors()
    os.environ[var] = val
    try:
        e = Engine()
    finally:
        del os.environ[var]
    e.map_build(small_scene["map"])
    e.scan_set(small_scene["scan"])
    out = e.residual_pass(x, True)
    idx, d2 = e.get_neighbors()
    assert (idx == ref_idx).all() and (bits(d2) == bits(ref_d2)).all()
    assert (bits(out["HtH"]) == bits(ref_out["HtH"])).all()
    e.close()


def test_wide_address_path(eng, small_scene):
    """Maps beyond 2^28 points address the sorted array with 64-bit pointers; force that code path on a
    small map (S2M_WIDE_ADDR=1) and require the identical result."""
    from daliti_amd import Engine
    x = small_scene["x_prop"]
    eng.scan_set(small_scene["scan"])
    ref_out = eng.residual_pass(x, True)
    ref_idx, ref_d2 = eng.get_neighbors()
    os.environ["S2M_WIDE_ADDR"] = "1"
    try:
        e = Engine()
    finally:
        del os.environ["S2M_WIDE_ADDR"]
    e.map_build(small_scene["map"])
    e.scan_set(small_scene["scan"])
    out = e.residual_pass(x, True)
    idx, d2 = e.get_neighbors()
    assert (idx == ref_idx).all() and (bits(d2) == bits(ref_d2)).all()
    assert (bits(out["HtH"]) == bits(ref_out["HtH"])).all()
    e.close()


def test_map_share(eng, small_scene):
    """A second handle borrowing the first one's map returns the identical pass and may not modify the map."""
    from daliti_amd import Engine, S2MError
    x = small_scene["x_prop"]
    eng.scan_set(small_scene["scan"])
    ref_out = eng.residual_pass(x, True)
    ref_idx, ref_d2 = eng.get_neighbors()
    e = Engine()
    e.map_share(eng)
    assert e.map_size() == eng.map_size()
    e.scan_set(small_scene["scan"][::-1].copy())          # its own scan (reversed order)
    out = e.residual_pass(x, True)
    idx, d2 = e.get_neighbors()
    assert (idx[::-1] == ref_idx).all() and (bits(d2[::-1]) == bits(ref_d2)).all()
    assert out["effct"] == ref_out["effct"]
    with pytest.raises(S2MError):
        e.map_add(small_scene["scan"][:10], False)
    with pytest.raises(S2MError):
        e.map_delete_boxes(np.float32([[-1, -1, -1, 1, 1, 1]]))
    e.map_build(small_scene["map"][:1000])                # ends the loan
    assert e.map_size() == 1000 and eng.map_size() == len(small_scene["map"])
    e.close()


def test_edge_cases(oracle, small_scene):
    from daliti_amd import Engine, S2MError
    e = Engine(cell_size=0.25)
    x = small_scene["x_prop"]
    with pytest.raises(S2MError):          # no map yet
        e.residual_pass(x, True)
    # map smaller than k: nobody can be selected (laserMapping.cpp:852)
    e.map_build(small_scene["map"][:3])
    e.scan_set(small_scene["scan"][:100])
    out = e.residual_pass(x, True)
    assert out["effct"] == 0 and np.all(out["HtH"] == 0)
    idx, d2 = e.get_neighbors()
    assert (idx[:, 3:] == -1).all() and np.isinf(d2[:, 3:]).all()
    # empty scan
    e.map_build(small_scene["map"])
    e.scan_set(np.zeros((0, 3), np.float32))
    out = e.residual_pass(x, True)
    assert out["effct"] == 0 and out["total_res"] == 0
    # reuse pass before any rematch is a call-order error
    e.scan_set(small_scene["scan"][:77])
    with pytest.raises(S2MError):
        e.residual_pass(x, False)
    # ragged size + points far outside the map: the d2 gate rejects them
    far = small_scene["scan"][:77].copy()
    far[:10] += 1000.0
    e.scan_set(far)
    out = e.residual_pass(x, True)
    st = e.get_point_state()
    assert (st["selected"][:10] == 0).all() and out["effct"] > 0
    # duplicated map points: ties are broken by value, results equal the oracle's
    m = np.concatenate([small_scene["map"][:5000], small_scene["map"][:5000]])
    e.map_build(m)
    e.scan_set(small_scene["scan"][:300])
    e.residual_pass(x, True)
    idx, d2 = e.get_neighbors()
    tree = ranked_tree(oracle, e, m)
    oi, od, _ = tree.knn5(oracle.body_to_world(x, small_scene["scan"][:300]))
    assert (bits(d2) == bits(od)).all() and (idx == oi).all()
    # strided input (pcl::PointXYZINormal is 12 floats)
    wide = np.zeros((len(small_scene["map"]), 12), np.float32)
    wide[:, :3] = small_scene["map"]
    e.map_build(wide)
    assert e.map_size() == len(wide)
    e.close()


def test_h_share_model_adapter(eng, oracle, small_scene, small_tree):
    x = small_scene["x_prop"]
    eng.scan_set(small_scene["scan"])
    d = eng.h_share_model(x, first_iteration=True)
    ps = _oracle_pass(oracle, small_tree, small_scene["scan"], x, True)
    assert d["valid"] and d["rows"] == ps.effct
    assert (bits(d["h_x"]) == bits(ps.Hsub)).all() and (bits(d["h"]) == bits(ps.meas)).all()


def test_map_permutation_invariance(small_scene):
    """Property: shuffling the map's point order changes nothing but the returned indices."""
    from daliti_amd import Engine
    rs = np.random.RandomState(7)
    perm = rs.permutation(len(small_scene["map"]))
    x = small_scene["x_prop"]
    res = []
    for m in (small_scene["map"], small_scene["map"][perm]):
        e = Engine()
        e.map_build(m)
        e.scan_set(small_scene["scan"])
        out = e.residual_pass(x, True)
        idx, d2 = e.get_neighbors()
        res.append((m[idx], d2, e.get_point_state(), out))
        e.close()
    assert (res[0][0] == res[1][0]).all() and (bits(res[0][1]) == bits(res[1][1])).all()
    assert (bits(res[0][2]["plane"]) == bits(res[1][2]["plane"])).all()
    assert (bits(res[0][3]["HtH"]) == bits(res[1][3]["HtH"])).all()


def test_s_gate_float_rounding_edge(oracle):
    """`float s` of laserMapping.cpp:868: points whose double s lies in (0.9, 0.9000000059604645] are
    rejected; the kernel's masks equal the oracle's on a scene built to hit that window."""
    from daliti_amd import Engine
    patch, scan = s_gate_candidates()
    e = Engine(max_iter=5)
    e.map_build(patch)
    e.scan_set(scan)
    x = oracle.make_state()
    out = e.residual_pass(x, True)
    ps = oracle.residual_pass(oracle.default_cfg(), oracle.KdTree(patch), scan, x, True, oracle.PassState(len(scan)))
    st = e.get_point_state()
    pbn = np.sqrt((scan[:, 0].astype(np.float64) ** 2 + scan[:, 1].astype(np.float64) ** 2)
                  + scan[:, 2].astype(np.float64) ** 2)
    s_d = 1 - 0.9 * np.abs(st["pd2"].astype(np.float64)) / np.sqrt(pbn)
    want = s_d.astype(np.float32).astype(np.float64) > 0.9
    edge = (s_d > 0.9) & ~want
    assert edge.sum() >= 5
    assert (bits(st["pd2"]) == bits(ps.pd2)).all()
    assert (st["selected"].astype(bool) == want).all()
    assert (st["selected"] == ps.selected).all() and (st["eff"] == ps.eff).all()
    assert out["effct"] == ps.effct
    e.close()


@pytest.mark.parametrize("loop", [0, 1])
@pytest.mark.parametrize("K", [2, 5, 9, 17])
def test_batch_of_ragged_scans_equals_one_by_one(small_scene, K, loop, monkeypatch):
    """s2m_iterated_update_batch (one grid for K scans) on scans of different sizes -- including an empty one and one so
    small that the degeneracy queue stops its update after the first pass -- from different predicted poses: every
    scan's log, state and covariance equal what s2m_iterated_update gives for it alone, bit for bit; K = 9 and 17 spread
    over two and three launch groups; the per-handle-stream form (handles that cannot share a launch) gives the same.  loop = 1: all of it
    with the state on the device (s2m_config.device_loop)."""
    from daliti_amd import Engine, synth
    rs = np.random.RandomState(K)
    owner = Engine(max_iter=5, device_loop=loop)
    owner.map_build(small_scene["map"])
    full = small_scene["scan"]
    engs, xs, Ps, sizes = [], [], [], []
    for k in range(K):
        n = [len(full), 777, 0, 64, 1500, 1024, 2000][k % 7]
        lo = rs.randint(0, len(full) - n + 1)
        e = Engine(max_iter=5, device_loop=loop)
        e.map_share(owner)
        e.scan_set(full[lo:lo + n])
        d = np.zeros(24); d[:6] = rs.normal(0, [2e-3, 2e-3, 2e-3, 0.01, 0.01, 0.01])
        import oracle as orc
        xs.append(orc.boxplus(small_scene["x_prop"], d))
        Ps.append(small_scene["P"] * (1.0 + 0.1 * k))
        engs.append(e); sizes.append(n)
    one = []
    for e, x, P in zip(engs, xs, Ps):
        e.set_feat_queue(())
        one.append(e.iterated_update(x, x, P))
    if K >= 4:
        assert any(o["ekf_stop"] for o in one) and any(not o["ekf_stop"] for o in one)
        assert len({o["iters"] for o in one}) > 1                  # the scans do not all end in the same pass
    for mode in ("fused", "streams"):
        for e in engs:
            e.set_feat_queue(())
        X = np.ascontiguousarray(np.stack(xs)); XP = X.copy(); PB = np.ascontiguousarray(np.stack(Ps))
        if mode == "streams":
            # the env switch is read once per process: ask for the fallback by giving one handle a different gate
            engs[0].set_config(res_gate=2.0 + 1e-12)
            one[0] = None
        logs = Engine.iterated_update_batch(engs, X, XP, PB)
        for k in range(K):
            if one[k] is None:
                continue
            o = one[k]
            assert logs[k].iters == o["iters"] and logs[k].rematch_passes == o["rematch_passes"], (mode, k)
            assert bool(logs[k].ekf_stop) == o["ekf_stop"] and bool(logs[k].converged) == o["converged"]
            assert list(logs[k].effct[:logs[k].iters]) == list(o["effct"]), (mode, k)
            assert (bits(X[k]) == bits(o["x"])).all() and (bits(PB[k]) == bits(o["P"])).all(), (mode, k, sizes[k])
    for e in engs:
        e.close()
    owner.close()


def test_batch_and_multi_with_extrinsic_estimation(oracle, small_scene, small_tree):
    """extrinsic_est_en = 1 (twelve Jacobian columns: the MFMA contraction, laserMapping.cpp:968-972) through the two
    multi-scan entry points, which until round 4 only ever ran the six-column default: K = 5 ragged scans from
    different predicted poses with a non-trivial R_L_I / T_L_I through s2m_iterated_update_batch
    (reduce_kernel_batch<true, .>) == one by one bit for bit, one by one == oracle <= 1e-9; and ONE scan in two aligned
    shards through s2m_iterated_update_multi == the single handle bit for bit."""
    from daliti_amd import Engine
    from daliti_amd.sharding import shard_range
    rs = np.random.RandomState(5)
    owner = Engine(max_iter=5, extrinsic_est_en=1)
    owner.map_build(small_scene["map"])
    full = small_scene["scan"]
    base = small_scene["x_prop"].copy()
    base[12:21] = oracle.so3_exp([0.02, -0.01, 0.03]).ravel()
    base[21:24] = [0.05, -0.02, 0.1]
    # the body-frame points that give the same LiDAR-frame cloud under this extrinsic: p_b = R_LI^T (p_l - T_LI)
    R_LI = base[12:21].reshape(3, 3)
    body = ((full.astype(np.float64) - base[21:24]) @ R_LI).astype(np.float32)
    engs, xs, Ps, scans = [], [], [], []
    for k, n in enumerate([len(body), 777, 1500, 1024, 2000]):
        lo = rs.randint(0, len(body) - n + 1)
        e = Engine(max_iter=5, extrinsic_est_en=1)
        e.map_share(owner)
        e.scan_set(body[lo:lo + n])
        d = np.zeros(24); d[:6] = rs.normal(0, [2e-3, 2e-3, 2e-3, 0.01, 0.01, 0.01])
        xs.append(oracle.boxplus(base, d))
        Ps.append(small_scene["P"] * (1.0 + 0.1 * k))
        engs.append(e); scans.append(body[lo:lo + n])
    cfg = oracle.default_cfg(extrinsic_est_en=1, max_iter=5)
    one = []
    for e, x, P, sc in zip(engs, xs, Ps, scans):
        e.set_feat_queue(())
        o = e.iterated_update(x, x, P)
        ref = oracle.iterated_update(cfg, small_tree, sc, x, x, P)
        assert o["iters"] == ref["iters"] and (o["effct"] == ref["effct"]).all()
        assert np.abs(o["x"] - ref["x"]).max() < 1e-9 and np.abs(o["P"] - ref["P"]).max() < 1e-11
        assert np.abs(o["x"][21:24] - x[21:24]).max() > 0.0         # the extrinsic columns did move the extrinsic
        one.append(o)
    for e in engs:
        e.set_feat_queue(())
    X = np.ascontiguousarray(np.stack(xs)); XP = X.copy(); PB = np.ascontiguousarray(np.stack(Ps))
    logs = Engine.iterated_update_batch(engs, X, XP, PB)
    for k, o in enumerate(one):
        assert logs[k].iters == o["iters"] and list(logs[k].effct[:logs[k].iters]) == list(o["effct"]), k
        assert (bits(X[k]) == bits(o["x"])).all() and (bits(PB[k]) == bits(o["P"])).all(), k
    # one scan (the whole cloud, 2 048 points) in two aligned shards, host-summed
    shards = []
    for r in range(2):
        lo, hi = shard_range(len(body), r, 2)
        e = Engine(max_iter=5, extrinsic_est_en=1)
        e.map_share(owner)
        e.scan_set(body[lo:hi])
        shards.append(e)
    x = xs[0].copy(); P = Ps[0].copy()
    log = Engine.iterated_update_multi(shards, x, np.ascontiguousarray(xs[0]), P)
    host = Engine(max_iter=5, extrinsic_est_en=1, device_loop=0)   # the multi-handle form is host-stepped: compare like with like
    host.map_share(owner)
    host.scan_set(scans[0])
    href = host.iterated_update(xs[0], xs[0], Ps[0])
    assert log.iters == href["iters"] and list(log.effct[:log.iters]) == list(href["effct"])
    assert (bits(x) == bits(href["x"])).all() and (bits(P) == bits(href["P"])).all()
    assert np.abs(href["x"] - one[0]["x"]).max() < 1e-11 and np.abs(href["P"] - one[0]["P"]).max() < 1e-13
    host.close()
    for e in engs + shards:
        e.close()
    owner.close()


@pytest.mark.parametrize("ext", [0, 1])
def test_device_loop_equals_host_loop(oracle, small_scene, small_tree, ext):
    """The iterated update with the state on the device (the last workgroup of every pass applies the Kalman update in the
    matrix-inversion-lemma form, the convergence test and the rematch / exit judgement; the host reads one record) against
    the host-stepped loop (two 24x24 LU inverses per iteration, laserMapping.cpp:1017-1018): identical schedules and
    effective counts, states within 1e-11, covariances within 1e-13 -- with max_iter = 5 (one chunk), with the yaml's 10
    (the loop leaves early; the plan of the second scan follows the first one's schedule) and from a perturbed start
    whose schedule differs from the previous scan's (the plan does not hold: the chain stops and is resumed)."""
    from daliti_amd import Engine
    for max_iter in (5, 10, 2, 1):
        dev = Engine(max_iter=max_iter, extrinsic_est_en=ext, device_loop=1)
        host = Engine(max_iter=max_iter, extrinsic_est_en=ext, device_loop=0)
        for e in (dev, host):
            e.map_build(small_scene["map"])
            e.scan_set(small_scene["scan"])
        xp, P = small_scene["x_prop"], small_scene["P"]
        starts = [xp, xp, oracle.boxplus(xp, np.r_[-9e-3, 7e-3, -1.4e-2, -0.045, 0.035, -0.025, np.zeros(18)]), xp,
                  oracle.boxplus(xp, np.r_[2e-2, 1e-2, -2e-2, 0.1, -0.1, 0.05, np.zeros(18)])]
        scheds = set()
        for x0 in starts:      # the third start is (nearly) the true pose: it converges at once and rematches early
            for e in (dev, host):
                e.set_feat_queue(())
            a = dev.iterated_update(x0, xp, P)
            b = host.iterated_update(x0, xp, P)
            assert a["iters"] == b["iters"] and a["rematch_passes"] == b["rematch_passes"], max_iter
            assert (a["effct"] == b["effct"]).all() and (a["rematch"] == b["rematch"]).all() and (a["conv"] == b["conv"]).all()
            assert a["converged"] == b["converged"] and a["ekf_stop"] == b["ekf_stop"]
            assert np.abs(a["x"] - b["x"]).max() < 1e-11 and np.abs(a["P"] - b["P"]).max() < 1e-13
            assert np.abs(a["solution"] - b["solution"]).max() < 1e-11
            assert np.abs(a["total_res"] - b["total_res"]).max() <= 1e-12 * max(1.0, np.abs(b["total_res"]).max())
            assert list(dev.feat_queue()) == list(host.feat_queue())
            # what the handle serves after the update is the last pass's, whichever side ran the loop
            sa, sb = dev.get_point_state(), host.get_point_state()
            assert (sa["selected"] == sb["selected"]).all() and (sa["eff"] == sb["eff"]).all()
            ha, hb = dev.get_rows(), host.get_rows()
            assert (ha[2] == hb[2]).all() and np.abs(ha[0] - hb[0]).max() < 1e-9
            scheds.add(tuple(a["rematch"]))
        if max_iter >= 5:
            assert len(scheds) > 1          # the plan taken from the previous scan did not always hold
        ref = oracle.iterated_update(oracle.default_cfg(extrinsic_est_en=ext, max_iter=max_iter), small_tree,
                                     small_scene["scan"], xp, xp, P)
        for e in (dev, host):
            e.set_feat_queue(())
        a = dev.iterated_update(xp, xp, P)
        assert a["iters"] == ref["iters"] and (a["effct"] == ref["effct"]).all() and np.abs(a["x"] - ref["x"]).max() < 1e-9
        dev.close(); host.close()
    # a degenerate scan: the degeneracy queue stops the update on the device exactly as on the host
    dev = Engine(max_iter=5, device_loop=1); host = Engine(max_iter=5, device_loop=0)
    for e in (dev, host):
        e.map_build(small_scene["map"])
        e.scan_set(small_scene["scan"][:64])
    a = dev.iterated_update(xp, xp, P); b = host.iterated_update(xp, xp, P)
    assert a["ekf_stop"] and b["ekf_stop"] and a["iters"] == b["iters"] == 1
    assert (bits(a["x"]) == bits(b["x"])).all() and (bits(a["P"]) == bits(b["P"])).all()      # untouched on both sides
    assert list(dev.feat_queue()) == list(host.feat_queue())
    dev.close(); host.close()


def test_complete_neighbors_edge_cases(oracle, small_scene):
    """s2m_complete_neighbors where it cannot succeed: a map of fewer than five points (every list stays short, the call
    terminates), non-finite scan points (left alone), an empty scan; and map_incremental on such inputs."""
    from daliti_amd import Engine, synth
    x = synth.make_state()
    q = small_scene["scan"][:500].copy()
    q[7] = np.nan
    q[9, 1] = np.inf
    e = Engine(cell_size=0.5)
    e.map_build(small_scene["map"][:3])
    e.scan_set(q)
    e.residual_pass(x, True)
    e.complete_neighbors()
    idx, d2 = e.get_neighbors()
    oi, od, oc = oracle.KdTree(small_scene["map"][:3]).knn5(q)
    ok = np.isfinite(q).all(axis=1)
    assert (idx[ok][:, :3] == oi[ok][:, :3]).all() and (idx[:, 3:] == -1).all()
    assert (bits(d2[ok][:, :3]) == bits(od[ok][:, :3])).all()
    assert (idx[~ok] == -1).all()
    clean = q[ok]
    e.scan_set(clean)
    e.residual_pass(x, True)
    na, nb = e.map_incremental(x, 0.5)                     # completes what it can (three points), then classifies
    assert na + nb == len(clean)
    e.scan_set(np.zeros((0, 3), np.float32))
    e.residual_pass(x, True)
    assert e.complete_neighbors() == 0
    e.close()
