"""dev probe: which lists differ after s2m_complete_neighbors (new-territory scene of tests/test_map_update.py)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle
from daliti_amd import Engine, synth
from conftest import ranked_tree, bits
sc = synth.make_small()
m = sc["map"]; x = synth.make_state(); rs = np.random.RandomState(11); L = sc["L"]
near = (m[rs.choice(len(m), 1500)] + rs.normal(0, 0.05, (1500, 3))).astype(np.float32)
d = rs.normal(size=(900, 3)); d /= np.linalg.norm(d, axis=1)[:, None]
far = (d * (L * 0.75 + rs.uniform(2.3, 60.0, (900, 1)))).astype(np.float32) + np.float32([0, 0, 5.0])
rim = (m[rs.choice(len(m), 300)] + d[:300] * rs.uniform(2.0, 2.6, (300, 1))).astype(np.float32)
q = np.r_[near, far, rim].astype(np.float32)
for cell in (0.5, 0.0, 1.3):
    e = Engine(cell_size=cell); e.map_build(m); e.scan_set(q); e.residual_pass(x, True)
    tree = ranked_tree(oracle, e, m)
    oi, od, oc = tree.knn5(q)
    i0, d0 = e.get_neighbors()
    ns = e.complete_neighbors()
    idx, d2 = e.get_neighbors()
    bad = np.nonzero((idx != oi).any(1) | (bits(d2) != bits(od)).any(1))[0]
    print("cell", cell, "grid", e.map_grid(), e.map_info()["cell"], "short", ns, "beyond", int((od[:, 4] > 5).sum()), "bad", len(bad))
    for b in bad[:12]:
        print("  q", b, q[b], "oracle d", np.sqrt(od[b]), "gpu d", np.sqrt(d2[b]), "gpu idx", idx[b], "orc idx", oi[b], "before", np.sqrt(d0[b]))
    e.close()
