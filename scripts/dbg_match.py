"""Dev tool: per-query cycle / radius statistics of the match kernel (S2M_DEBUG_MATCH=1)."""
import os, sys, ctypes as C
import numpy as np
os.environ["S2M_DEBUG_MATCH"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from daliti_amd import Engine, synth
c = synth.CONFIGS[os.environ.get("CONFIG", "C3")]
m = synth.make_map(c["M"], c["L"]); s = synth.make_scan(c["beams"], c["az"], c["L"])
_, xp, P = synth.filter_inputs()
for g in [int(x) for x in os.environ.get("GROUPS", "4").split(",")]:
    for cell in [float(x) for x in os.environ.get("CELLS", "0.25,0.5").split(",")]:
        os.environ["S2M_MATCH_GROUP"] = str(g)
        e = Engine(cell_size=cell)
        e.map_build(m); e.scan_set(s)
        for _ in range(3): e.residual_pass(xp, True)
        e.set_timing(2)
        e.residual_pass(xp, True)
        ms = e.timing()[0]
        d = np.zeros((e.n, 4), np.uint32)
        assert e.lib.s2m_debug_match(e.h, C.c_void_p(d.ctypes.data)) == 0
        cyc = d[:, 0].astype(float)
        print("G %d cell %.2f match %.1f us | cycles(100MHz ticks?) pct50 %.0f pct90 %.0f pct99 %.0f max %.0f sum %.3g" % (
            g, cell, ms * 1e3, *np.percentile(cyc, [50, 90, 99, 100]), cyc.sum()))
        print("   hard fraction %.3f" % ((d[:, 3] & 0xff) > 1).mean())
        hm = (d[:, 3] & 0xff) > 1
        if hm.any():
            st = d[hm, 2].astype(np.int64); st -= st.min()
            print("   hard starts (us): pct50 %.1f pct90 %.1f max %.1f" % tuple(np.percentile(st, [50, 90, 100]) / 100.0))
        if hm.any():
            hd = (d[hm, 3] >> 8) / 100.0
            print("   hard part per point (us): mean %.1f pct50 %.1f pct90 %.1f max %.1f" % (hd.mean(), *np.percentile(hd, [50, 90, 100])))
            en = (st + (d[hm, 3] >> 8).astype(np.int64)) / 100.0
            print("   hard ends (us after first start): pct50 %.1f pct90 %.1f max %.1f" % tuple(np.percentile(en, [50, 90, 100])))
            for rr in range(2, 6):
                sl = (d[hm, 3] & 0xff) == rr
                if sl.any(): print("     rounds %d: n %d  hard part mean %.1f us" % (rr - 1, sl.sum(), hd[sl].mean()))
        for r in range(1, 12):
            sel = d[:, 1] == r
            if sel.any():
                print("   r=%2d n=%6d  mean cyc %.0f  rounds %.2f" % (r, sel.sum(), cyc[sel].mean(), (d[sel, 3] & 0xff).mean()))
        idx, d2 = e.get_neighbors()
        miss = idx[:, 4] < 0
        print("   points without five neighbours inside the gate: %d (%.3f); of the hard points: %.3f" % (miss.sum(), miss.mean(), miss[hm].mean() if hm.any() else 0))
        none = idx[:, 0] < 0
        print("   points with NO neighbour inside the gate: %d" % none.sum())
        e.close()
