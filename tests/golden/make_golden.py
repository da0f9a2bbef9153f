"""Generates tests/golden/s2m_small.npz from the CPU oracle.

The reference has no tests, fixtures or runnable build in this image, so
these vectors are produced by the repo's own restatement (oracle/s2m_oracle.c) and pin it against
drift; they are not reference outputs.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from daliti_amd import synth  # noqa: E402


def main():
    sc = synth.make_small(M=20000, beams=16, az=128)
    tree = oracle.KdTree(sc["map"])
    out = dict(map=sc["map"], scan=sc["scan"], x_prop=sc["x_prop"], x_true=sc["x_true"], P=sc["P"])
    for ext in (0, 1):
        cfg = oracle.default_cfg(extrinsic_est_en=ext, max_iter=5)
        x = sc["x_prop"].copy()
        if ext:
            x[12:21] = oracle.so3_exp([0.02, -0.01, 0.03]).ravel()
            x[21:24] = [0.05, -0.02, 0.1]
        ps = oracle.residual_pass(cfg, tree, sc["scan"], x, True, oracle.PassState(len(sc["scan"])), want_rows=True)
        k = "e%d_" % ext
        out.update({k + "x0": x, k + "nn_idx": ps.nn_idx, k + "nn_d2": ps.nn_d2, k + "plane": ps.plane,
                    k + "plane_ok": ps.plane_ok, k + "pd2": ps.pd2, k + "selected": ps.selected, k + "eff": ps.eff,
                    k + "HtH": ps.HtH, k + "Htz": ps.Htz, k + "effct": np.int32(ps.effct),
                    k + "total_res": np.float64(ps.total_res), k + "Hsub": ps.Hsub, k + "meas": ps.meas})
        r = oracle.iterated_update(cfg, tree, sc["scan"], x, x, sc["P"])
        out.update({k + "it_x": r["x"], k + "it_P": r["P"], k + "it_effct": r["effct"], k + "it_rematch": r["rematch"],
                    k + "it_conv": r["conv"], k + "it_solution": r["solution"], k + "it_total_res": r["total_res"]})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "s2m_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
