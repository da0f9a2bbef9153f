"""Dev tool: randomised exactness soak of the search kernels against the oracle, beyond what the test suite runs:
more seeds, every first-shell variant (lanes per point, batches per trip, per-cell form), brick-straddling cell sizes."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from test_gpu_stress import _check

variants = [{}, {"S2M_MATCH_GROUP": "1"}, {"S2M_MATCH_GROUP": "4", "S2M_EASY_NB": "1"}, {"S2M_MATCH_GROUP": "8", "S2M_EASY_NB": "2"},
            {"S2M_EASY_CELLS": "1"}, {"S2M_WIDE_ADDR": "1"}]
n = 0
for seed in range(int(os.environ.get("SEEDS", "24"))):
    rs = np.random.RandomState(1000 + seed)
    kind = seed % 4
    off = np.float32([0, 0, 0]) if seed % 2 == 0 else np.float32(rs.uniform(-4000, 4000, 3))
    if kind == 0:
        m = rs.uniform(-8, 8, (20000, 3))
    elif kind == 1:
        m = np.c_[rs.uniform(-10, 10, (15000, 2)), rs.normal(0, 0.01, 15000)]
        m = np.r_[m, np.c_[np.full(5000, 3.0) + rs.normal(0, 0.01, 5000), rs.uniform(-10, 10, (5000, 2))]]
    elif kind == 2:
        c = rs.uniform(-10, 10, (10, 3))
        m = np.r_[tuple(c[k] + rs.normal(0, 0.03 * (1 + k), (2000, 3)) for k in range(10))]
    else:   # lattice: exact distance ties everywhere
        g = np.stack(np.meshgrid(*[np.arange(-12, 13) * 0.25] * 3, indexing="ij"), -1).reshape(-1, 3)
        m = g[rs.permutation(len(g))]
    m = (m + off).astype(np.float32)
    q = np.r_[rs.uniform(-11, 11, (1200, 3)), m[rs.choice(len(m), 300)] + rs.normal(0, 0.02, (300, 3)), rs.uniform(-30, 30, (100, 3))]
    q = (q + off).astype(np.float32)
    for cell in (rs.uniform(0.05, 0.2), rs.uniform(0.2, 0.8), rs.uniform(0.8, 3.0), 0.0):
        v = variants[n % len(variants)]
        for k, val in v.items(): os.environ[k] = val
        try:
            _check(oracle, m, q, float(cell))
        finally:
            for k in v: del os.environ[k]
        n += 1
print("soak ok: %d (map, cell, variant) combinations exact" % n)
