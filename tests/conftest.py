import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def small_scene():
    from daliti_amd import synth
    return synth.make_small()


@pytest.fixture(scope="session")
def small_tree(oracle, small_scene):
    return oracle.KdTree(small_scene["map"])


def ranked_tree(oracle, eng, map_xyz):
    """The oracle's k-d tree over `map_xyz` with the tie order of the GPU engine `eng` (sorted position = brick, cell,
    caller index).  The order is computed here from the grid parameters alone (oracle.grid_rank) and must equal what
    the engine reports -- that pins the documented order, not only consistency between the two sides."""
    info = eng.map_info()
    doc = oracle.grid_rank(map_xyz, info["cell"], info["origin"])
    rank = eng.map_rank()
    if not np.array_equal(doc, rank):
        # In-place updates hand a brick that opens, or outgrows its place, a stretch BEHIND the key-ordered part of the array
        # (include/daliti_s2m.h, s2m_map_get_order): until the next merge the documented order holds inside every brick, and
        # every brick is one contiguous run of positions.
        assert eng.map_inplace_updates() > 0, "the engine's point order differs from the documented one"
        brick = oracle.grid_rank(map_xyz, info["cell"], info["origin"], bricks_only=True)
        along = brick[np.argsort(rank)]
        assert 1 + int((along[1:] != along[:-1]).sum()) == len(np.unique(brick)), "a brick's points are not contiguous"
        assert np.array_equal(np.lexsort((doc, brick)), np.lexsort((rank, brick))), "the order inside a brick differs from the documented one"
    return oracle.KdTree(map_xyz).set_rank(rank)


def bits(a):
    """View a float array as integers so that equality is bit-exact (and NaN == NaN)."""
    a = np.ascontiguousarray(a)
    return a.view(np.uint32 if a.dtype == np.float32 else np.uint64)


def s_gate_candidates():
    """A five-point planar patch (z = 1) and 4,096 scan points whose s = 1 - 0.9 |pd2| / sqrt(|p_body|)
    straddles 0.9 within a few float ulps: the s gate of laserMapping.cpp:868-870 at its rounding edge."""
    patch = np.array([[10.0, 0.0, 1.0], [10.2, 0.1, 1.0], [9.8, 0.1, 1.0], [10.1, -0.15, 1.0], [9.9, -0.1, 1.0]],
                     np.float32)
    k = np.arange(4096)
    bx = (np.float32(10.0) + k.astype(np.float32) * np.float32(2 ** -20)).astype(np.float32)
    by = np.full_like(bx, 0.02)
    # |pd2| = bz - 1 chosen so that 0.9 |pd2| / sqrt(norm) ~ 0.1 (s ~ 0.9); the float grid of bz scatters
    # s over ~ +-3e-7 around the gate, the window of interest is 6e-9 wide
    bz0 = np.full(len(bx), 1.35)
    for _ in range(8):  # fixed point of bz = 1 + (0.1 / 0.9) sqrt(|p|)
        bz0 = 1.0 + (0.1 / 0.9) * np.sqrt(np.sqrt(bx.astype(np.float64) ** 2 + float(by[0]) ** 2 + bz0 ** 2))
    bz = bz0.astype(np.float32)
    return patch, np.stack([bx, by, bz], 1)
