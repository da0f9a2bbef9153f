import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from daliti_amd import Engine, synth, S2MError
sc = synth.make_small()
for what, vals in (("nan", [np.nan]*3), ("inf", [np.inf, -np.inf, np.inf]), ("huge", [1e30, -1e30, 3e38]), ("far", [5e5, -5e5, 2e6])):
    e = Engine(max_iter=5, wait_timeout_ms=3000)
    e.map_build(sc["map"])
    scan = sc["scan"].copy()
    scan[100] = vals; scan[777] = vals[::-1]; scan[-1] = vals
    try:
        e.scan_set(scan)
        r = e.iterated_update(sc["x_prop"], sc["x_prop"], sc["P"])
        print(what, "update ok: iters", r["iters"], "effct", list(r["effct"]), "x finite", bool(np.isfinite(r["x"]).all()))
        na, nb = e.map_incremental(r["x"], 0.5)
        print(what, "map_incremental ok:", na, nb, "map size", e.map_size(), "finite map", bool(np.isfinite(e.map_points()).all()))
        ds = e.scan_set_downsampled(scan, 0.5)
        print(what, "voxel grid ok:", ds)
    except S2MError as ex:
        print(what, "S2MError", ex.code, str(ex)[:200])
    try:
        e.map_add(np.float32([vals, [0.1, 0.2, 0.3]]), True, 0.5)
        print(what, "map_add ok; size", e.map_size())
        r = e.iterated_update(sc["x_prop"], sc["x_prop"], sc["P"])
        print(what, "update after map_add ok", list(r["effct"]))
    except S2MError as ex:
        print(what, "map_add S2MError", ex.code, str(ex)[:200])
    try:
        e.map_build(np.vstack([sc["map"][:5000], np.float32([vals])]))
        print(what, "map_build ok; size", e.map_size())
    except S2MError as ex:
        print(what, "map_build S2MError", ex.code, str(ex)[:200])
    print(what, "close", e.close(), flush=True)

# the raw path: records with non-finite coordinates / times, a non-finite state
rs = np.random.RandomState(3)
n = 4096
rec = np.zeros((n, 12), np.float32)
rec[:, :3] = sc["scan"][rs.randint(0, len(sc["scan"]), n)]
rec[:, 4] = np.sort(rs.uniform(0, 1, n)).astype(np.float32)
rec[:, 6] = 0.1
poses = np.zeros((4, 22)); poses[:, 0] = np.linspace(0.0, 0.11, 4); poses[:, 13:22] = np.eye(3).ravel()
end = np.zeros(36); end[0:9] = np.eye(3).ravel(); end[12:21] = np.eye(3).ravel()
for what in ("nan xyz", "inf xyz", "nan time", "inf time", "nan state", "nan P"):
    e = Engine(max_iter=5, wait_timeout_ms=3000)
    e.map_build(sc["map"])
    r2 = rec.copy()
    if what == "nan xyz": r2[10:20, :3] = np.nan
    if what == "inf xyz": r2[10:20, 1] = np.inf
    if what == "nan time": r2[10:20, 4] = np.nan
    if what == "inf time": r2[10:20, 4] = np.inf
    try:
        m = e.scan_set_from_raw(r2, 4, 6, poses, end, 0.5)
        xp = sc["x_prop"].copy(); P = sc["P"].copy()
        if what == "nan state": xp[9] = np.nan
        if what == "nan P": P[3, 3] = np.nan
        r = e.iterated_update(xp.copy(), xp, P)
        print(what, "ok: scan", m, "iters", r["iters"], "effct", list(r["effct"])[:2], "x finite", bool(np.isfinite(r["x"]).all()))
        if np.isfinite(r["x"]).all():
            print(what, "map_incremental", e.map_incremental(r["x"], 0.5))
    except S2MError as ex:
        print(what, "S2MError", ex.code, str(ex)[:160])
    print(what, "close", e.close(), flush=True)
