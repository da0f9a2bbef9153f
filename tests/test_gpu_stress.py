"""Randomised exactness stress of the GPU search against the oracle: cell sizes from far too small to
far too large, coordinates kilometres from the origin (float cell arithmetic loses bits there; the
termination bound carries a slop for it), clustered / planar / collinear / duplicated maps, queries
inside, at the rim of and far outside the map."""
import numpy as np
import pytest

from conftest import bits, ranked_tree

pytestmark = pytest.mark.gpu


def _check(oracle, m, q, cell, x=None):
    from daliti_amd import Engine, synth
    e = Engine(cell_size=cell)
    e.map_build(m)
    e.scan_set(q)
    x = synth.make_state() if x is None else x
    e.residual_pass(x, True)
    idx, d2 = e.get_neighbors()
    st = e.get_point_state()
    oi, od, oc = ranked_tree(oracle, e, m).knn5(oracle.body_to_world(x, q))
    near = (oc == 5) & (od[:, 4] <= 5.0)
    assert (bits(d2[near]) == bits(od[near])).all(), "d2 mismatch at cell %g" % cell
    assert (idx[near] == oi[near]).all(), "index mismatch at cell %g" % cell
    assert (st["selected"][~near] == 0).all()
    far = ~near
    assert ((d2[far, 4] > 5.0) | np.isinf(d2[far, 4])).all()
    e.close()
    return near.mean()


@pytest.mark.parametrize("seed", range(6))
def test_random_clouds_and_cells(oracle, seed):
    rs = np.random.RandomState(100 + seed)
    kind = seed % 3
    off = np.float32([0, 0, 0]) if seed < 3 else np.float32([3100.0, -2750.0, 410.0])
    if kind == 0:      # volumetric uniform
        m = rs.uniform(-8, 8, (30000, 3))
    elif kind == 1:    # a thin slab + a line + duplicates
        slab = np.c_[rs.uniform(-10, 10, (20000, 2)), rs.normal(0, 0.01, 20000)]
        line = np.c_[np.linspace(-10, 10, 3000), np.zeros(3000), np.full(3000, 2.0)]
        m = np.r_[slab, line, slab[:2000]]
    else:              # clusters of very different density
        c = rs.uniform(-10, 10, (12, 3))
        m = np.r_[tuple(c[k] + rs.normal(0, 0.05 * (1 + k), (2500, 3)) for k in range(12))]
    m = (m + off).astype(np.float32)
    q = np.r_[rs.uniform(-11, 11, (1500, 3)), m[rs.choice(len(m), 300)] + rs.normal(0, 0.02, (300, 3)),
              rs.uniform(-40, 40, (200, 3))]
    q = (q + off * (np.arange(len(q))[:, None] < 1800)).astype(np.float32)   # the last 200 stay far away
    fracs = [_check(oracle, m, q, cell) for cell in (0.07, 0.31, 1.0, 2.9, 0.0)]
    assert min(fracs) == max(fracs)          # the same points are "near" whatever the grid


def test_tiny_and_degenerate_maps(oracle):
    rs = np.random.RandomState(7)
    q = rs.uniform(-1, 1, (500, 3)).astype(np.float32)
    for m in (np.zeros((1, 3)), np.zeros((5, 3)), np.zeros((7, 3)) + [0.1, 0, 0],      # identical points
              rs.uniform(-1, 1, (6, 3)), np.c_[np.linspace(-1, 1, 50), np.zeros(50), np.zeros(50)]):
        for cell in (0.05, 0.5, 4.0):
            _check(oracle, m.astype(np.float32), q, cell)


def test_exact_ties_follow_the_documented_order(oracle):
    """A lattice map (coordinates exactly representable) and queries on lattice points, edge and cell centres: most
    queries see several candidates at exactly the same float d2, also across the 5th place.  The engine ranks them by
    sorted position = (brick, cell, caller index) -- s2m_map_get_order -- and the oracle, given that order computed
    independently from the grid parameters, must return the identical lists, whatever the cell size."""
    g = np.arange(-8, 9) * 0.25
    m = np.stack(np.meshgrid(g, g, g[:9], indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    rs = np.random.RandomState(3)
    m = m[rs.permutation(len(m))]                         # caller order unrelated to position
    q = np.r_[m[:600], m[600:1200] + np.float32(0.125), m[1200:1800] + np.float32([0.125, 0.0, 0.0])].astype(np.float32)
    tied = 0
    for cell in (0.11, 0.5, 0.8, 0.0):
        from daliti_amd import Engine, synth
        e = Engine(cell_size=cell)
        e.map_build(m)
        e.scan_set(q)
        e.residual_pass(synth.make_state(), True)
        idx, d2 = e.get_neighbors()
        oi, od, oc = ranked_tree(oracle, e, m).knn5(q)
        assert (bits(d2) == bits(od)).all() and (idx == oi).all(), "cell %g" % cell
        # the scene does what it is for: with the plain index order the lists differ
        pi, pd, _ = oracle.KdTree(m).knn5(q)
        assert (bits(pd) == bits(od)).all()
        tied += int((pi != oi).any(axis=1).sum())
        e.close()
    assert tied > 100


def test_rotated_far_pose(oracle):
    """A non-trivial body->world pose with a large translation (double transform, float result)."""
    from scipy.spatial.transform import Rotation
    from daliti_amd import synth
    rs = np.random.RandomState(9)
    x = synth.make_state(Rotation.from_rotvec([0.4, -0.9, 2.0]).as_matrix(), [1500.0, 900.0, -30.0])
    x[12:21] = Rotation.from_rotvec([0.02, 0.01, -0.03]).as_matrix().ravel()
    x[21:24] = [0.3, -0.1, 0.2]
    body = rs.uniform(-20, 20, (3000, 3)).astype(np.float32)
    world = oracle.body_to_world(x, body)
    m = (world[rs.choice(3000, 2500)] + rs.normal(0, 0.1, (2500, 3))).astype(np.float32)
    m = np.r_[m, (world.mean(0) + rs.uniform(-25, 25, (40000, 3))).astype(np.float32)]
    for cell in (0.2, 0.9, 0.0):
        _check(oracle, m, body, cell, x)


def test_non_finite_scan_points_are_harmless(oracle, small_scene):
    """NaN / Inf coordinates in the scan (the reference's clouds are is_dense, a caller's may not be): no fault, the bad
    points end up unselected without neighbours, and every other point's result is unchanged -- in the single pass and
    through the whole iterated update."""
    from daliti_amd import Engine
    scan = small_scene["scan"][:2000].copy()
    bad = np.array([3, 64, 65, 511, 1024, 1999])
    dirty = scan.copy()
    dirty[bad[0]] = np.nan
    dirty[bad[1], 0] = np.inf
    dirty[bad[2], 2] = -np.inf
    dirty[bad[3], 1] = np.nan
    dirty[bad[4]] = [np.inf, np.nan, 0.0]
    dirty[bad[5]] = 3.0e38
    x, P = small_scene["x_prop"], small_scene["P"]
    res = {}
    for name, s in (("clean", np.delete(scan, bad, axis=0)), ("dirty", dirty)):
        e = Engine(max_iter=5)
        e.map_build(small_scene["map"])
        e.scan_set(s)
        e.residual_pass(x, True)
        idx, d2 = e.get_neighbors()
        st = e.get_point_state()
        r = e.iterated_update(x, x, P)
        res[name] = (idx, d2, st["selected"], r)
        e.close()
    keep = np.ones(len(scan), bool); keep[bad] = False
    ci, cd, cs, cr = res["clean"]
    di, dd, ds, dr = res["dirty"]
    assert (ds[bad] == 0).all()
    assert (di[keep] == ci).all() and (bits(dd[keep]) == bits(cd)).all() and (ds[keep] == cs).all()
    assert dr["iters"] == cr["iters"] and (dr["effct"] == cr["effct"]).all()
    assert np.abs(dr["x"] - cr["x"]).max() < 1e-12


def test_half_wave_far_point_kernel_passes_the_stress_suite():
    """The far-point kernel exists in two forms -- a wave per point (one scan in flight) and 32 lanes per point, two
    points per wave (batched launches).  The exactness stress of this file and the small-scene parity checks run once
    more in a child process with the half-wave form forced for single scans too (test hook S2M_HARD_LANES=32)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sel = "random_clouds or tiny_and_degenerate or exact_ties or rotated_far or knn_exact or pass_bit_exact or ragged or new_territory"
    env = dict(os.environ, S2M_HARD_LANES="32")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-k", sel,
                        os.path.join(here, "test_gpu_stress.py"), os.path.join(here, "test_gpu_parity.py"),
                        os.path.join(here, "test_map_update.py")], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " passed" in r.stdout


def test_bet_on_no_far_points_is_exact_whether_won_or_lost(oracle):
    """A rematch pass whose predecessor in the same position (first pass of a scan / later pass) reported no far points
    runs WITHOUT the far-point kernel; the reduce kernel reports the list length, and a lost bet is repaired by running
    the far-point kernel and the reduce kernel again.  (1) History: the first update on a handle bets on nothing, the
    second one bets on the later pass -- same bits.  (2) In child processes: S2M_SPEC=2 (always bet, i.e. the first pass
    of C3 with its 9,981 far points LOSES every time) and S2M_SPEC=0 (never bet) give the oracle-checked results."""
    import os
    import subprocess
    import sys
    from daliti_amd import Engine, synth
    c = synth.make_config("C2")
    e = Engine(max_iter=5)
    e.map_build(c["map"])
    e.scan_set(c["scan"])
    runs = []
    for _ in range(3):
        e.set_feat_queue(())
        runs.append(e.iterated_update(c["x_prop"], c["x_prop"], c["P"]))
    for r in runs[1:]:
        assert r["iters"] == runs[0]["iters"] and (r["effct"] == runs[0]["effct"]).all()
        assert (bits(r["x"]) == bits(runs[0]["x"])).all() and (bits(r["P"]) == bits(runs[0]["P"])).all()
    # the same through the config field: never bet / always bet (every pass of this scan with far points loses)
    for bet in (0, 2):
        h = Engine(max_iter=5, far_point_bet=bet)
        h.map_share(e)
        h.scan_set(c["scan"])
        r = h.iterated_update(c["x_prop"], c["x_prop"], c["P"])
        assert r["iters"] == runs[0]["iters"] and (r["effct"] == runs[0]["effct"]).all(), bet
        assert (bits(r["x"]) == bits(runs[0]["x"])).all() and (bits(r["P"]) == bits(runs[0]["P"])).all(), bet
        h.close()
    # a different scan on the same handle whose LATER pass does have far points: far outliers that no pose can fix
    rs = np.random.RandomState(5)
    bad = c["scan"].copy()
    pick = rs.choice(len(bad), 4000, replace=False)
    bad[pick] *= np.float32(0.93)                       # pulled 7 % towards the sensor: 0.5-7 m off their surface
    e.scan_set(bad)
    e.set_feat_queue(())
    lost = e.iterated_update(c["x_prop"], c["x_prop"], c["P"])        # bets on the later pass (history 0) and loses
    e.set_feat_queue(())
    again = e.iterated_update(c["x_prop"], c["x_prop"], c["P"])       # history now says: do not bet
    assert lost["iters"] == again["iters"] and (lost["effct"] == again["effct"]).all()
    assert (bits(lost["x"]) == bits(again["x"])).all() and (bits(lost["P"]) == bits(again["P"])).all()
    # the rule, counted: a position bets only after a pass there reported NO far point; the lost bet above is the only one,
    # and the scan after it did not bet again in that position (s2m_bet_stats)
    won, n_lost = e.bet_stats()
    assert n_lost == 1 and won >= 2, (won, n_lost)
    e.set_feat_queue(())
    e.iterated_update(c["x_prop"], c["x_prop"], c["P"])
    assert e.bet_stats() == (won, n_lost)
    ro = oracle.iterated_update(oracle.default_cfg(max_iter=5, nthreads=8), oracle.KdTree(c["map"]), bad, c["x_prop"], c["x_prop"], c["P"])
    assert lost["iters"] == ro["iters"] and (lost["effct"] == ro["effct"]).all() and np.abs(lost["x"] - ro["x"]).max() < 1e-9
    e.close()
    here = os.path.dirname(os.path.abspath(__file__))
    sel = "iterated_update_matches_oracle or fullsize_registration or c1_one_iteration or multi_frame_odometry or ragged"
    for mode in ("2", "0"):
        env = dict(os.environ, S2M_SPEC=mode)
        r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-k", sel,
                            os.path.join(here, "test_gpu_parity.py"), os.path.join(here, "test_gpu_fullsize.py"),
                            os.path.join(here, "test_map_update.py")], env=env, capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, "S2M_SPEC=%s\n%s" % (mode, r.stdout[-3000:])
        assert " passed" in r.stdout
