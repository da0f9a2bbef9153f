"""Scan voxel down-sampling: pcl::VoxelGrid restated."""
import numpy as np
import pytest

from conftest import bits


def test_oracle_voxel_grid_properties(oracle, small_scene):
    pts = small_scene["scan"]
    leaf = 0.5
    out = oracle.voxel_downsample(pts, leaf)
    # one point per occupied voxel, the float64 centroid within float round-off, ascending voxel index
    inv = np.float32(1.0) / np.float32(leaf)
    ijk = np.floor(pts * inv).astype(np.int64)
    ijk -= np.floor(pts.min(0) * inv).astype(np.int64)
    div = ijk.max(0) + 1
    idx = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    uniq = np.unique(idx)
    assert len(out) == len(uniq)
    for k in (0, len(uniq) // 2, len(uniq) - 1):
        ref = pts[idx == uniq[k]].astype(np.float64).mean(0)
        assert np.abs(out[k] - ref).max() < 1e-5
    # a voxel with a single point returns that point exactly
    single = np.nonzero(np.bincount(np.searchsorted(uniq, idx)) == 1)[0]
    assert len(single) > 0
    k = single[0]
    assert (bits(out[k]) == bits(pts[idx == uniq[k]][0])).all()
    with pytest.raises(OverflowError):
        oracle.voxel_downsample(np.float32([[0, 0, 0], [3e4, 3e4, 3e4]]), 0.01)


@pytest.mark.gpu
@pytest.mark.parametrize("leaf", [0.5, 0.2, 2.0])
def test_gpu_voxel_grid_matches_oracle(oracle, leaf):
    from daliti_amd import Engine, S2MError, synth
    scan = synth.make_scan(64, 1024, 95.0, seed=5)      # a full 65,536-point scan
    e = Engine()
    e.map_build(synth.make_map(20000))
    m = e.scan_set_downsampled(scan, leaf)
    got = e.scan_get()
    ref = oracle.voxel_downsample(scan, leaf)
    assert m == len(ref) == len(got) < len(scan)
    assert (bits(got) == bits(ref)).all()               # same order, same bits
    # strided input (PointXYZINormal = 12 floats) and the overflow check
    wide = np.zeros((len(scan), 12), np.float32)
    wide[:, :3] = scan
    assert e.scan_set_downsampled(wide, leaf) == m
    with pytest.raises(S2MError) as ei:
        e.scan_set_downsampled(np.float32([[0, 0, 0], [3e4, 3e4, 3e4]]), 0.01)
    assert ei.value.code == -5
    # the down-sampled scan registers like any other scan
    sc = synth.make_small()
    e.map_build(sc["map"])
    n = e.scan_set_downsampled(sc["scan"], 0.3)
    r = e.iterated_update(sc["x_prop"], sc["x_prop"], sc["P"])
    tree = oracle.KdTree(sc["map"])
    ro = oracle.iterated_update(oracle.default_cfg(max_iter=5), tree, oracle.voxel_downsample(sc["scan"], 0.3),
                                sc["x_prop"], sc["x_prop"], sc["P"])
    assert n == e.n and (r["effct"] == ro["effct"]).all() and np.abs(r["x"] - ro["x"]).max() < 1e-9
    e.close()


@pytest.mark.gpu
def test_voxel_grid_with_and_without_the_box_in_hand(oracle):
    """The grid's kernels are launched before the host has seen the cloud's box when the last cloud's voxel indices had a
    known number of bits (VoxelBuffers::kbits_hint): clouds of the same extent, a cloud eight times as wide (more bits
    than the hint: sorted on too few, found out by the final hand-back and done again), a small one again, a leaf that
    overflows PCL's index in between -- every result equals the oracle's, bit for bit and in order."""
    from daliti_amd import Engine, S2MError, synth
    rs = np.random.RandomState(23)
    e = Engine()
    e.map_build(synth.make_map(20000))

    def cloud(n, half):
        return (rs.uniform(-1, 1, (n, 3)) * [half, half, 0.1 * half]).astype(np.float32)

    for k, (half, leaf) in enumerate([(20.0, 0.5), (20.0, 0.5), (21.0, 0.5), (160.0, 0.5), (160.0, 0.5), (5.0, 0.5), (5.0, 0.25), (40.0, 0.5)]):
        pts = cloud(30000 + 500 * k, half)
        ref = oracle.voxel_downsample(pts, leaf)
        m = e.scan_set_downsampled(pts, leaf)
        assert m == len(ref), (k, m, len(ref))
        assert (bits(e.scan_get()) == bits(ref)).all(), k
        if k == 4:
            with pytest.raises(S2MError) as ei:
                e.scan_set_downsampled(np.float32([[0, 0, 0], [3e4, 3e4, 3e4]]), 0.01)
            assert ei.value.code == -5
    e.close()
