"""Scan undistortion: the backward-propagation loop of ImuProcess::UndistortPcl."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from conftest import bits


def _scenario(n=20000, K=12, seed=0, tmin=0.0):
    """A 0.1 s sweep with the sensor turning and accelerating; records mimic PointXYZINormal
    (normal_x = time ratio at float 4, normal_z = time span at float 6)."""
    rs = np.random.RandomState(seed)
    span = 0.1
    rec = np.zeros((n, 12), np.float32)
    rec[:, :3] = rs.uniform(-30, 30, (n, 3))
    rec[:, 4] = rs.uniform(0, 1, n)           # normal_x: time ratio
    rec[:, 6] = span                          # normal_z: time span
    rec[:5, 4] = 0.0                          # points at t = 0 are left untouched (t <= IMUpose[0].offset_time)
    rec[5:10, 4] = rec[10, 4]                 # equal times: the stable order must hold
    if tmin > 0.0:                            # a sweep whose first point is later than several IMU poses
        rec[:, 4] = tmin + (1.0 - tmin) * rec[:, 4]
    times = np.linspace(0.0, span * 1.02, K)  # the last pose lies beyond the scan end, like imu_end > pcl_end
    poses = np.zeros((K, 22))
    R = np.eye(3); p = np.array([1.0, 2.0, 0.5]); v = np.array([2.0, -1.0, 0.2])
    for k in range(K):
        w = np.array([0.3, -0.2, 0.8]) + 0.05 * k
        a = np.array([0.5, 0.1, -0.3]) * (1 + 0.1 * k)
        poses[k, 0] = times[k]
        poses[k, 1:4], poses[k, 4:7], poses[k, 7:10], poses[k, 10:13] = a, w, v, p
        poses[k, 13:22] = R.ravel()
        if k + 1 < K:
            dt = times[k + 1] - times[k]
            R = R @ Rotation.from_rotvec(w * dt).as_matrix()
            p = p + v * dt + 0.5 * a * dt * dt
            v = v + a * dt
    end = np.zeros(36)
    end[0:9] = (R @ Rotation.from_rotvec([0.001, 0.002, -0.001]).as_matrix()).ravel()
    end[9:12] = p + v * 0.001
    end[12:21] = Rotation.from_rotvec([0.01, -0.02, 0.03]).as_matrix().ravel()
    end[21:24] = [0.05, -0.02, 0.1]
    return rec, poses, end


def test_oracle_undistort_against_direct_formula(oracle):
    rec, poses, end = _scenario(n=3000)
    out, perm = oracle.undistort(rec, 4, 6, poses, end, sort=True)
    t = (rec[:, 4] * rec[:, 6])
    assert (np.diff(t[perm]) >= 0).all()                         # time order
    same = t[perm][1:] == t[perm][:-1]
    assert (np.diff(perm.astype(np.int64))[same] > 0).all()      # ties keep input order
    Rend, pend = end[0:9].reshape(3, 3), end[9:12]
    RLI, TLI = end[12:21].reshape(3, 3), end[21:24]
    for s in (0, 7, 100, 1500, 2999):
        i = perm[s]
        ti = float(t[i])
        heads = [h for h in range(len(poses) - 1) if poses[h, 0] < ti]
        if not heads:
            assert (out[s] == rec[i, :3]).all()
            continue
        h = heads[-1]
        dt = ti - poses[h, 0]
        Ri = poses[h, 13:22].reshape(3, 3) @ Rotation.from_rotvec(poses[h, 4:7] * dt).as_matrix()
        Tei = poses[h, 10:13] + poses[h, 7:10] * dt + 0.5 * poses[h, 1:4] * dt * dt - pend
        ref = RLI.T @ (Rend.T @ (Ri @ (RLI @ rec[i, :3].astype(np.float64) + TLI) + Tei) - TLI)
        assert np.abs(out[s] - ref).max() < 2e-5
    # unsorted variant: same values, input order
    out2, perm2 = oracle.undistort(rec, 4, 6, poses, end, sort=False)
    assert (perm2 == np.arange(len(rec))).all() and (bits(out2[perm]) == bits(out)).all()


def _direct(rec_xyz, ti, h, poses, end):
    Rend, pend = end[0:9].reshape(3, 3), end[9:12]
    RLI, TLI = end[12:21].reshape(3, 3), end[21:24]
    dt = ti - poses[h, 0]
    Ri = poses[h, 13:22].reshape(3, 3) @ Rotation.from_rotvec(poses[h, 4:7] * dt).as_matrix()
    Tei = poses[h, 10:13] + poses[h, 7:10] * dt + 0.5 * poses[h, 1:4] * dt * dt - pend
    return RLI.T @ (Rend.T @ (Ri @ (RLI @ np.asarray(rec_xyz, np.float64) + TLI) + Tei) - TLI)


def test_oracle_first_point_is_recompensated(oracle):
    """IMU_Processing.hpp:366-367: the `break` at the first point leaves only the inner loop, so the
    earliest point of the sorted cloud is compensated once more per remaining earlier head."""
    rec, poses, end = _scenario(n=2000, tmin=0.35)
    out, perm = oracle.undistort(rec, 4, 6, poses, end, sort=True)
    t = rec[:, 4] * rec[:, 6]
    i0, t0 = perm[0], float(t[perm[0]])
    heads = [h for h in range(len(poses) - 1) if poses[h, 0] < t0]
    assert len(heads) >= 3
    p = rec[i0, :3].astype(np.float64)
    for h in reversed(heads):                       # own head first, then every earlier one
        p = _direct(p, t0, h, poses, end).astype(np.float32).astype(np.float64)
    assert np.abs(out[0] - p).max() < 2e-5
    once = _direct(rec[i0, :3], t0, heads[-1], poses, end)
    assert np.abs(out[0] - once).max() > 1e-3       # it really differs from a single application
    i1, t1 = perm[1], float(t[perm[1]])            # the second point is handled once
    h1 = [h for h in range(len(poses) - 1) if poses[h, 0] < t1][-1]
    assert np.abs(out[1] - _direct(rec[i1, :3], t1, h1, poses, end)).max() < 2e-5


@pytest.mark.gpu
def test_gpu_undistort_first_point_quirk(oracle):
    from daliti_amd import Engine
    rec, poses, end = _scenario(n=4096, tmin=0.35)
    e = Engine()
    got, perm = e.undistort(rec, 4, 6, poses, end, sort_by_time=True)
    ref, rperm = oracle.undistort(rec, 4, 6, poses, end, sort=True)
    assert (perm == rperm).all()
    assert (np.abs(got.astype(np.float64) - ref) <= 4 * np.spacing(np.abs(ref)).astype(np.float64)).all()
    e.close()


@pytest.mark.gpu
def test_gpu_undistort_matches_oracle(oracle):
    from daliti_amd import Engine, synth
    rec, poses, end = _scenario(n=65536)
    e = Engine()
    e.map_build(synth.make_map(20000))
    for sort in (True, False):
        got, perm = e.undistort(rec, 4, 6, poses, end, sort_by_time=sort)
        ref, rperm = oracle.undistort(rec, 4, 6, poses, end, sort=sort)
        assert (perm == rperm).all()
        # device sin/cos may differ from glibc in the last bit: at most one float ulp, and rarely
        diff = np.abs(got.astype(np.float64) - ref.astype(np.float64))
        assert (diff <= np.spacing(np.abs(ref)).astype(np.float64)).all()
        assert (bits(got) != bits(ref)).mean() < 1e-3
    # single time field, no second factor
    rec1 = rec.copy()
    rec1[:, 4] = rec[:, 4] * rec[:, 6]
    got1, _ = e.undistort(rec1, 4, -1, poses, end)
    ref1, _ = oracle.undistort(rec1, 4, -1, poses, end)
    assert (np.abs(got1 - ref1) <= np.spacing(np.abs(ref1))).all()
    # fused front half of a frame: undistort -> VoxelGrid -> current scan, all on the device
    m = e.scan_set_from_raw(rec, 4, 6, poses, end, leaf=0.5)
    und, _ = oracle.undistort(rec, 4, 6, poses, end, sort=True)
    ref_ds = oracle.voxel_downsample(und, 0.5)
    got_ds = e.scan_get()
    assert m == len(got_ds) and abs(m - len(ref_ds)) <= 2      # a one-ulp difference can move a point across a voxel face
    if m == len(ref_ds):
        assert np.abs(got_ds - ref_ds).max() < 1e-3
    e.close()


@pytest.mark.gpu
def test_prefetched_records_give_the_same_scan(oracle):
    """s2m_scan_prefetch_raw: the next sweep's records copied to the device by the handle's worker thread on a side stream;
    s2m_scan_set_from_raw with the same buffer uses that copy -- same scan, bit for bit, as when it copies itself; a call
    with another buffer ignores the prefetch; prefetching twice in a row, or never consuming, is harmless."""
    from daliti_amd import Engine, synth
    sc = synth.make_small()
    n = len(sc["scan"])
    rs = np.random.RandomState(3)
    rec = np.zeros((n, 12), np.float32)
    rec[:, :3] = sc["scan"]
    rec[:, 4] = rs.permutation(n).astype(np.float32) / n
    rec[:, 6] = 0.1
    other = rec.copy(); other[:, :3] += np.float32(0.25)
    K = 12
    poses = np.zeros((K, 22)); poses[:, 0] = np.linspace(0.0, 0.101, K); poses[:, 13:22] = np.eye(3).ravel()
    poses[:, 1:4] = rs.normal(0, 0.3, (K, 3)); poses[:, 4:7] = rs.normal(0, 0.2, (K, 3)); poses[:, 7:10] = rs.normal(0, 0.5, (K, 3))
    end = np.zeros(36); end[0:9] = np.eye(3).ravel(); end[12:21] = np.eye(3).ravel()
    e = Engine()
    e.map_build(sc["map"])
    e.scan_set_from_raw(rec, 4, 6, poses, end, 0.3)
    plain = e.scan_get()
    e.scan_set_from_raw(other, 4, 6, poses, end, 0.3)
    plain_other = e.scan_get()
    for _ in range(3):
        e.scan_prefetch_raw(rec)
        e.scan_set_from_raw(rec, 4, 6, poses, end, 0.3)
        assert (bits(e.scan_get()) == bits(plain)).all()
    e.scan_prefetch_raw(rec)
    e.scan_prefetch_raw(rec)                                   # a second request while (or after) the first one runs
    e.scan_set_from_raw(other, 4, 6, poses, end, 0.3)          # another buffer: the prefetch is ignored -- and dropped ...
    assert (bits(e.scan_get()) == bits(plain_other)).all()
    e.scan_set_from_raw(rec, 4, 6, poses, end, 0.3)            # ... this call copies for itself
    assert (bits(e.scan_get()) == bits(plain)).all()
    # a node that recycles its host buffers: a sweep is announced and then dropped, something else is registered, and NEW
    # records arrive at the same address with the same size -- the copy made for the dropped sweep must not be used
    recycled = rec.copy()
    e.scan_prefetch_raw(recycled, 4, 6)
    e.scan_set(sc["scan"])                                     # (any other road a scan arrives by drops the copy)
    recycled[:, :3] = other[:, :3]
    e.scan_set_from_raw(recycled, 4, 6, poses, end, 0.3)
    assert (bits(e.scan_get()) == bits(plain_other)).all()
    recycled[:, :3] = rec[:, :3]
    e.scan_prefetch_raw(recycled, 4, 6)
    e.scan_prefetch_raw(None)                                  # explicit cancel
    recycled[:, :3] = other[:, :3]
    e.scan_set_from_raw(recycled, 4, 6, poses, end, 0.3)
    assert (bits(e.scan_get()) == bits(plain_other)).all()
    e.scan_prefetch_raw(other)                                 # never consumed: close() must not hang
    e.close()


@pytest.mark.gpu
def test_prepared_frames_equal_frames_computed_in_place(oracle):
    """s2m_scan_prepare_raw: the next frame's copy + undistortion + voxel grid on the handle's side stream while the map
    update of the current frame runs on the main one.  Four frames (each with its own poses) through two handles -- one
    computes every scan inside s2m_scan_set_from_raw, the other prepares frame k + 1 between the update and
    map_incremental of frame k: scans, poses, iteration logs and maps are identical bit for bit.  A prepared scan whose
    arguments do not match the next call (other poses) is ignored; preparing and never consuming is harmless."""
    from daliti_amd import Engine, synth
    sc = synth.make_small()
    n = len(sc["scan"])
    rs = np.random.RandomState(5)
    K = 12
    frames = []
    for f in range(4):
        rec = np.zeros((n, 12), np.float32)
        rec[:, :3] = sc["scan"] + rs.normal(0, 0.02, (n, 3)).astype(np.float32)
        rec[:, 4] = rs.permutation(n).astype(np.float32) / n
        rec[:, 6] = 0.1
        poses = np.zeros((K, 22)); poses[:, 0] = np.linspace(0.0, 0.101, K); poses[:, 13:22] = np.eye(3).ravel()
        poses[:, 1:4] = rs.normal(0, 0.3, (K, 3)); poses[:, 7:10] = rs.normal(0, 0.2 + 0.1 * f, (K, 3))
        frames.append((np.ascontiguousarray(rec), poses))
    end = np.zeros(36); end[0:9] = np.eye(3).ravel(); end[12:21] = np.eye(3).ravel()
    x_prop, P0 = sc["x_prop"], sc["P"]

    def run(prepare):
        e = Engine(feat_threshold=20)
        e.map_build(sc["map"])
        out = []
        for f, (rec, poses) in enumerate(frames):
            nd = e.scan_set_from_raw(rec, 4, 6, poses, end, 0.3)
            scan = e.scan_get().copy()
            r = e.iterated_update(x_prop, x_prop, P0)
            if prepare and f + 1 < len(frames):
                e.scan_prepare_raw(frames[f + 1][0], 4, 6, frames[f + 1][1], end, 0.3)
            na, nb = e.map_incremental(r["x"], 0.5)
            out.append((nd, scan, r["x"].copy(), tuple(r["effct"]), r["iters"], na, nb, e.map_size()))
        out.append(e.map_points().copy())
        if prepare:   # a prepared scan that the next call does not match, then one that is never consumed
            e.scan_prepare_raw(frames[0][0], 4, 6, frames[0][1], end, 0.3)
            e.scan_set_from_raw(frames[0][0], 4, 6, frames[1][1], end, 0.3)
            other = e.scan_get().copy()
            e.scan_set_from_raw(frames[0][0], 4, 6, frames[1][1], end, 0.3)
            assert (bits(e.scan_get()) == bits(other)).all()
            e.scan_prepare_raw(frames[2][0], 4, 6, frames[2][1], end, 0.3)
        e.close()
        return out

    a, b = run(False), run(True)
    for fa, fb in zip(a[:-1], b[:-1]):
        assert fa[0] == fb[0] and (bits(fa[1]) == bits(fb[1])).all()
        assert (fa[2] == fb[2]).all() and fa[3:] == fb[3:]
    assert (bits(a[-1]) == bits(b[-1])).all()


@pytest.mark.gpu
def test_prefetch_then_prepare_is_the_same_scan(oracle):
    """Records brought over by s2m_scan_prefetch_raw are used by a following s2m_scan_prepare_raw of the same buffer (no
    second copy), and the prepared scan equals the one computed in place."""
    from daliti_amd import Engine, synth
    sc = synth.make_small()
    n = len(sc["scan"])
    rs = np.random.RandomState(9)
    rec = np.zeros((n, 12), np.float32)
    rec[:, :3] = sc["scan"]
    rec[:, 4] = rs.permutation(n).astype(np.float32) / n
    rec[:, 6] = 0.1
    K = 12
    poses = np.zeros((K, 22)); poses[:, 0] = np.linspace(0.0, 0.101, K); poses[:, 13:22] = np.eye(3).ravel()
    poses[:, 7:10] = rs.normal(0, 0.4, (K, 3))
    end = np.zeros(36); end[0:9] = np.eye(3).ravel(); end[12:21] = np.eye(3).ravel()
    e = Engine()
    e.map_build(sc["map"])
    e.scan_set_from_raw(rec, 4, 6, poses, end, 0.3)
    plain = e.scan_get().copy()
    for k in range(6):
        # with the time fields (k even) the side thread also sorts the records by time behind the copy, and the call that
        # consumes them -- prepare (k < 4) or scan_set_from_raw itself -- skips its own sort
        if k % 2 == 0:
            e.scan_prefetch_raw(rec, 4, 6)
        else:
            e.scan_prefetch_raw(rec)
        if k < 4:
            e.scan_prepare_raw(rec, 4, 6, poses, end, 0.3)
        assert e.scan_set_from_raw(rec, 4, 6, poses, end, 0.3) == len(plain)
        assert (bits(e.scan_get()) == bits(plain)).all()
    # an order that was computed and then overtaken by another call's sort must not be used
    other = rec.copy(); other[:, 4] = other[::-1, 4]
    e.scan_prefetch_raw(rec, 4, 6)
    e.scan_set_from_raw(other, 4, 6, poses, end, 0.3)
    assert e.scan_set_from_raw(rec, 4, 6, poses, end, 0.3) == len(plain)
    assert (bits(e.scan_get()) == bits(plain)).all()
    e.close()


@pytest.mark.gpu
def test_records_in_and_out_of_time_order_alternate(oracle):
    """Records that arrive in time order are not sorted (the key kernel counts the places where the order is broken, one
    hand-back: s2m_undistort.hip, undistort_order); records that do not are sorted by a stable sort like the reference's
    list after std::sort (IMU_Processing.hpp:216).  One handle, sweeps in order, out of order, with ties, in order again --
    by the plain road and through the prefetching worker: every scan equals the oracle's."""
    from daliti_amd import Engine, synth
    sc = synth.make_small()
    n = len(sc["scan"])
    rs = np.random.RandomState(11)
    K = 12
    poses = np.zeros((K, 22)); poses[:, 0] = np.linspace(0.0, 0.101, K); poses[:, 13:22] = np.eye(3).ravel()
    poses[:, 1:4] = rs.normal(0, 0.3, (K, 3)); poses[:, 4:7] = rs.normal(0, 0.2, (K, 3)); poses[:, 7:10] = rs.normal(0, 0.5, (K, 3))
    end = np.zeros(36); end[0:9] = np.eye(3).ravel(); end[12:21] = np.eye(3).ravel()

    def sweep(kind):
        rec = np.zeros((n, 12), np.float32)
        rec[:, :3] = sc["scan"] + rs.normal(0, 0.01, (n, 3)).astype(np.float32)
        t = np.sort(rs.uniform(0, 1, n)).astype(np.float32)
        if kind == "ties":
            t = np.floor(t * 64) / np.float32(64)              # long runs of equal stamps, still in order
        elif kind == "one_swap":
            t[[n // 2, n // 2 + 1]] = t[[n // 2 + 1, n // 2]]  # a single pair out of order (in ONE wave)
        elif kind == "shuffled":
            t = rs.permutation(t)
        rec[:, 4] = t
        rec[:, 6] = 0.1
        return rec

    e = Engine()
    e.map_build(sc["map"])
    for k, kind in enumerate(["sorted", "shuffled", "sorted", "ties", "one_swap", "one_swap", "sorted", "shuffled", "sorted"]):
        rec = sweep(kind)
        und, _ = oracle.undistort(rec, 4, 6, poses, end, True)
        want = oracle.voxel_downsample(und, 0.3)
        if k % 2:
            e.scan_prefetch_raw(rec, 4, 6)
        got_n = e.scan_set_from_raw(rec, 4, 6, poses, end, 0.3)
        assert got_n == len(want), (k, kind)
        assert (bits(e.scan_get()) == bits(want)).all(), (k, kind)
    e.close()
