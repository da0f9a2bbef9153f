"""Golden vectors (tests/golden/s2m_small.npz, made by tests/golden/make_golden.py from the oracle).

CPU: the oracle still reproduces them bit-for-bit (guards the checker against drift).
GPU: the HIP path reproduces them without needing the oracle at run time.
"""
import os

import numpy as np
import pytest

from conftest import bits

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "s2m_small.npz")


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(GOLD))


@pytest.mark.parametrize("ext", [0, 1])
def test_oracle_reproduces_golden(oracle, gold, ext):
    k = "e%d_" % ext
    tree = oracle.KdTree(gold["map"])
    cfg = oracle.default_cfg(extrinsic_est_en=ext, max_iter=5)
    ps = oracle.residual_pass(cfg, tree, gold["scan"], gold[k + "x0"], True, oracle.PassState(len(gold["scan"])),
                              want_rows=True)
    assert (ps.nn_idx == gold[k + "nn_idx"]).all() and (bits(ps.nn_d2) == bits(gold[k + "nn_d2"])).all()
    assert (bits(ps.plane) == bits(gold[k + "plane"])).all() and (bits(ps.pd2) == bits(gold[k + "pd2"])).all()
    assert (ps.eff == gold[k + "eff"]).all() and ps.effct == int(gold[k + "effct"])
    assert (bits(ps.HtH) == bits(gold[k + "HtH"])).all() and (bits(ps.Hsub) == bits(gold[k + "Hsub"])).all()
    r = oracle.iterated_update(cfg, tree, gold["scan"], gold[k + "x0"], gold[k + "x0"], gold["P"])
    assert (r["effct"] == gold[k + "it_effct"]).all() and (r["rematch"] == gold[k + "it_rematch"]).all()
    # libm sin/cos/acos may differ in the last bit between hosts: allow round-off, not more
    assert np.abs(r["x"] - gold[k + "it_x"]).max() < 1e-12
    assert np.abs(r["P"] - gold[k + "it_P"]).max() < 1e-15


@pytest.mark.gpu
@pytest.mark.parametrize("ext", [0, 1])
def test_gpu_reproduces_golden(gold, ext):
    from daliti_amd import Engine
    k = "e%d_" % ext
    e = Engine(max_iter=5, extrinsic_est_en=ext)
    e.map_build(gold["map"])
    e.scan_set(gold["scan"])
    out = e.residual_pass(gold[k + "x0"], True)
    idx, d2 = e.get_neighbors()
    st = e.get_point_state()
    ok = gold[k + "plane_ok"].astype(bool)
    assert (idx == gold[k + "nn_idx"]).all() and (bits(d2) == bits(gold[k + "nn_d2"])).all()
    assert (bits(st["plane"][ok]) == bits(gold[k + "plane"][ok])).all()
    assert (bits(st["pd2"][ok]) == bits(gold[k + "pd2"][ok])).all()
    assert (st["selected"] == gold[k + "selected"]).all() and (st["eff"] == gold[k + "eff"]).all()
    assert out["effct"] == int(gold[k + "effct"])
    assert np.abs(out["HtH"] - gold[k + "HtH"]).max() <= 1e-12 * np.abs(gold[k + "HtH"]).max()
    hx, h, _ = e.get_rows()
    assert (bits(hx) == bits(gold[k + "Hsub"])).all() and (bits(h) == bits(gold[k + "meas"])).all()
    r = e.iterated_update(gold[k + "x0"], gold[k + "x0"], gold["P"])
    assert (r["effct"] == gold[k + "it_effct"]).all() and (r["rematch"] == gold[k + "it_rematch"]).all()
    assert (r["conv"] == gold[k + "it_conv"]).all()
    assert np.abs(r["x"] - gold[k + "it_x"]).max() < 1e-9 and np.abs(r["P"] - gold[k + "it_P"]).max() < 1e-12
    e.close()
