"""No entry point blocks for ever.  The reference's loop never waits on its map (laserMapping.cpp:726-731;
ikd_Tree.cpp:436-449 only fences short critical sections); this engine's caller waits for the device several times per
frame, so every such wait has a deadline (s2m_config.wait_timeout_ms) and a policy (s2m_config.wait_policy).  The tests
withhold one hand-back of each kind on a HEALTHY device (s2m_test_stall: the side thread's job, the mailbox kernel, the
reduce kernel's block) and assert that the entry point that waits for it returns S2M_ERR_TIMEOUT in time, names the wait, that
the handle then refuses work instead of hanging, can be destroyed, and that a fresh handle works."""
import time

import numpy as np
import pytest

TIMEOUT_MS = 400
BOUND_S = 2.0            # the bar: an error within 2 s


def _engine(small_scene, **kw):
    from daliti_amd import Engine
    e = Engine(max_iter=5, wait_timeout_ms=TIMEOUT_MS, **kw)
    e.map_build(small_scene["map"])
    e.scan_set(small_scene["scan"])
    return e


def _raw(small_scene, n=4096):
    """a raw sweep in the wire layout (12 floats per record, time fields at 4 and 6) with identity motion"""
    rs = np.random.RandomState(3)
    rec = np.zeros((n, 12), np.float32)
    rec[:, :3] = small_scene["scan"][rs.randint(0, len(small_scene["scan"]), n)]
    rec[:, 4] = np.sort(rs.uniform(0, 1, n)).astype(np.float32)
    rec[:, 6] = 0.1
    poses = np.zeros((4, 22))
    poses[:, 0] = np.linspace(0.0, 0.11, 4)
    poses[:, 13:22] = np.eye(3).ravel()
    end = np.zeros(36)
    end[0:9] = np.eye(3).ravel()
    end[12:21] = np.eye(3).ravel()
    return rec, poses, end


def _expect_timeout(call, e, needle):
    from daliti_amd import S2MError
    t0 = time.perf_counter()
    with pytest.raises(S2MError) as ei:
        call()
    dt = time.perf_counter() - t0
    assert ei.value.code == -7, ei.value                       # S2M_ERR_TIMEOUT
    assert dt < BOUND_S, "the entry point took %.2f s to give up" % dt
    msg = str(ei.value)
    assert needle in msg and "state: in " in msg, msg          # names the wait and carries the handle's state
    return msg


def _after(e, small_scene):
    """a handle that has given up refuses at once, can be destroyed, and the device is fine"""
    from daliti_amd import S2MError
    t0 = time.perf_counter()
    with pytest.raises(S2MError) as ei:
        e.scan_set(small_scene["scan"])
    assert ei.value.code == -7 and "given up" in str(ei.value)
    assert time.perf_counter() - t0 < 0.1
    t0 = time.perf_counter()
    rc = e.close()
    assert rc == 0 and time.perf_counter() - t0 < BOUND_S
    f = _engine(small_scene)
    r = f.iterated_update(small_scene["x_prop"], small_scene["x_prop"], small_scene["P"])
    assert r["iters"] >= 2 and r["effct"][0] > 100
    f.close()


REDUCE_CALLS = {
    "residual_pass": lambda e, sc: e.residual_pass(sc["x_prop"], True),
    "iterated_update": lambda e, sc: e.iterated_update(sc["x_prop"], sc["x_prop"], sc["P"]),
    "h_share_model": lambda e, sc: e.h_share_model(sc["x_prop"], True),
}


@pytest.mark.gpu
@pytest.mark.parametrize("policy", [0, 1, 2])
@pytest.mark.parametrize("entry", sorted(REDUCE_CALLS))
def test_block_that_never_arrives_times_out(small_scene, entry, policy):
    e = _engine(small_scene, wait_policy=policy)
    good = e.iterated_update(small_scene["x_prop"], small_scene["x_prop"], small_scene["P"])
    assert good["iters"] >= 2
    e.test_stall("reduce", 0)
    msg = _expect_timeout(lambda: REDUCE_CALLS[entry](e, small_scene), e, "the block of a pass")
    assert "policy %d" % policy in msg
    _after(e, small_scene)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [2, 8])
def test_batch_whose_block_never_arrives_times_out(small_scene, k):
    """both batched forms: k = 2 (every handle on its own stream) and k = 8 (one grid per pass, two launch groups)"""
    from daliti_amd import Engine
    owner = _engine(small_scene)
    engs = [owner]
    for _ in range(k - 1):
        b = Engine(max_iter=5, wait_timeout_ms=TIMEOUT_MS)
        b.map_share(owner)
        b.scan_set(small_scene["scan"])
        engs.append(b)
    x = np.tile(small_scene["x_prop"], (k, 1)).copy()
    xp = x.copy()
    P = np.tile(small_scene["P"], (k, 1, 1)).copy()
    Engine.iterated_update_batch(engs, x, xp, P)           # healthy first
    x[:] = xp
    engs[k - 1].test_stall("reduce", 1)                     # the second pass of the last scan is lost
    from daliti_amd import S2MError
    t0 = time.perf_counter()
    with pytest.raises(S2MError) as ei:
        Engine.iterated_update_batch(engs, x, xp, P)
    assert ei.value.code == -7 and time.perf_counter() - t0 < BOUND_S, ei.value
    assert "batch" in str(ei.value)
    for b in engs[::-1]:
        assert b.close() == 0


@pytest.mark.gpu
def test_multi_handle_update_whose_block_never_arrives_times_out(small_scene):
    from daliti_amd import Engine
    sc = small_scene
    half = len(sc["scan"]) // 2
    engs = []
    for piece in (sc["scan"][:half], sc["scan"][half:]):
        e = Engine(max_iter=5, wait_timeout_ms=TIMEOUT_MS)
        e.map_build(sc["map"])
        e.scan_set(piece)
        engs.append(e)
    x, P = sc["x_prop"].copy(), sc["P"].copy()
    Engine.iterated_update_multi(engs, x, sc["x_prop"].copy(), P)
    engs[1].test_stall("reduce", 0)
    from daliti_amd import S2MError
    t0 = time.perf_counter()
    with pytest.raises(S2MError) as ei:
        Engine.iterated_update_multi(engs, sc["x_prop"].copy(), sc["x_prop"].copy(), sc["P"].copy())
    assert ei.value.code == -7 and time.perf_counter() - t0 < BOUND_S, ei.value
    for e in engs:
        assert e.close() == 0


def _mail_calls():
    def changes(e, sc):
        tok = e.map_changes(0).token
        e.map_add(sc["scan"][:64] + np.float32(0.3), True, 0.5)
        e.test_stall("mail", 0)
        e.map_changes(tok)

    def from_raw(e, sc):
        rec, poses, end = _raw(sc)
        e.test_stall("mail", 0)
        e.scan_set_from_raw(rec, 4, 6, poses, end, 0.5)

    def incremental(e, sc):
        r = e.iterated_update(sc["x_prop"], sc["x_prop"], sc["P"])
        e.test_stall("mail", 0)
        e.map_incremental(r["x"], 0.5)

    def stalled(fn):
        def call(e, sc):
            e.test_stall("mail", 0)
            fn(e, sc)
        return call

    return {
        "scan_set": stalled(lambda e, sc: e.scan_set(sc["scan"])),
        "scan_set_downsampled": stalled(lambda e, sc: e.scan_set_downsampled(sc["scan"], 0.5)),
        "scan_set_from_raw": from_raw,
        "map_incremental": incremental,
        "map_add": stalled(lambda e, sc: e.map_add(sc["scan"][:256] + np.float32(0.2), True, 0.5)),
        "map_delete_boxes": stalled(lambda e, sc: e.map_delete_boxes([[-1, -1, -1, 1, 1, 1]])),
        "map_get_changes": changes,
    }


@pytest.mark.gpu
@pytest.mark.parametrize("entry", sorted(_mail_calls()))
def test_handback_that_never_arrives_times_out(small_scene, entry):
    e = _engine(small_scene)
    _expect_timeout(lambda: _mail_calls()[entry](e, small_scene), e, "hand-back")
    _after(e, small_scene)


@pytest.mark.gpu
@pytest.mark.parametrize("entry", ["scan_set_from_raw", "scan_set", "undistort", "scan_prepare_raw"])
def test_side_thread_that_never_comes_back_times_out(small_scene, entry):
    """the next sweep is announced (prefetch / prepare), the side thread never finishes that job: whoever needs the side
    thread idle next hears of it within the deadline -- and s2m_destroy still gets the thread to leave"""
    e = _engine(small_scene)
    rec, poses, end = _raw(small_scene)
    assert e.scan_set_from_raw(rec, 4, 6, poses, end, 0.5) > 100    # a healthy frame first (the side thread is not running yet)
    e.scan_prefetch_raw(rec, 4, 6)
    assert e.scan_set_from_raw(rec, 4, 6, poses, end, 0.5) > 100    # ... and one through the side thread
    e.test_stall("worker", 0)
    e.scan_prefetch_raw(rec, 4, 6)                                   # returns at once; the job is never finished
    call = {
        "scan_set_from_raw": lambda: e.scan_set_from_raw(rec, 4, 6, poses, end, 0.5),
        "scan_set": lambda: e.scan_set(small_scene["scan"]),
        "undistort": lambda: e.undistort(rec, 4, 6, poses, end),
        "scan_prepare_raw": lambda: e.scan_prepare_raw(rec, 4, 6, poses, end, 0.5),
    }[entry]
    msg = _expect_timeout(call, e, "side thread")
    assert "busy=1" in msg
    _after(e, small_scene)


@pytest.mark.gpu
@pytest.mark.parametrize("policy", [0, 1, 2])
def test_wait_policies_give_the_same_results(small_scene, oracle, small_tree, policy):
    """spin / yield / sleep change how the host waits, nothing else: a frame's worth of calls under each policy gives the pose,
    the counts and the map the spinning default gives"""
    sc = small_scene
    out = []
    for pol in (0, policy):
        e = _engine(sc, wait_policy=pol)
        r = e.iterated_update(sc["x_prop"], sc["x_prop"], sc["P"])
        na, nb = e.map_incremental(r["x"], 0.5)
        pts = e.map_points()
        out.append((r["x"].copy(), list(r["effct"]), na, nb, pts[np.lexsort(pts.T)].copy()))
        st = e.debug_state()
        assert "policy %d" % pol in st and "main stream idle" in st, st
        e.close()
    a, b = out
    assert np.array_equal(a[0], b[0]) and a[1] == b[1] and a[2:4] == b[2:4] and np.array_equal(a[4], b[4])


@pytest.mark.gpu
def test_debug_state_from_another_thread(small_scene):
    """a watchdog thread may ask what the handle is doing while its owner is inside an entry point"""
    import threading
    e = _engine(small_scene)
    seen, stop = [], threading.Event()

    def watch():
        while not stop.is_set():
            seen.append(e.debug_state())

    t = threading.Thread(target=watch)
    t.start()
    for _ in range(200):
        e.iterated_update(small_scene["x_prop"], small_scene["x_prop"], small_scene["P"])
    stop.set()
    t.join()
    assert any("in s2m_iterated_update" in s for s in seen)
    e.close()


@pytest.mark.gpu
def test_a_layout_whose_worker_stalls_does_not_stop_the_frames(small_scene, monkeypatch):
    """The map is laid out again BESIDE the updates by a worker thread of the handle (ikd-Tree's rebuild thread,
    ikd_Tree.cpp:192-203, 229-367).  Its first hand-back never comes: registration and map updates go on at their usual pace --
    nobody waits for the layout -- until the layout's own deadline has passed; then the next call that looks after it reports
    S2M_ERR_TIMEOUT with the worker's wait by name, the handle refuses further work and can be destroyed."""
    from daliti_amd import Engine
    sc = small_scene
    monkeypatch.setenv("S2M_BESIDE_AT", "3")
    e = Engine(max_iter=5, wait_timeout_ms=TIMEOUT_MS, cell_size=0.5)
    e.map_build(sc["map"])
    e.scan_set(sc["scan"])
    rs = np.random.RandomState(2)
    for _ in range(2):   # (the first update of a dense build lays the room and the tail out: a layout in flight would be let go for it)
        e.map_add((sc["map"][rs.choice(len(sc["map"]), 100)] + rs.normal(0, 0.2, (100, 3))).astype(np.float32), False)
    e.test_stall("layout", 0)
    t0 = time.perf_counter()
    e.map_add(sc["map"][:300] + np.float32(0.11), True, 0.5)      # the layout begins behind this update
    done, slowest = 0, 0.0
    while time.perf_counter() - t0 < 0.04:      # (a wait whose stream has gone idle without the word is given up after 0.1 s: s2m_wait.h)
        t1 = time.perf_counter()
        e.scan_set(sc["scan"])
        r = e.iterated_update(sc["x_prop"], sc["x_prop"], sc["P"])
        e.map_add((sc["map"][rs.choice(len(sc["map"]), 100)] + rs.normal(0, 0.2, (100, 3))).astype(np.float32), False)
        assert r["effct"][0] > 1000
        slowest = max(slowest, time.perf_counter() - t1)
        done += 1
    assert done >= 3 and slowest < 0.05, (done, slowest)            # (a frame of this size takes a millisecond or two)
    st = e.debug_state()
    assert "1 begun" in st.split("layout beside")[1] and "0 dropped" in st.split("layout beside")[1], st
    time.sleep(max(0.0, 1.3 * TIMEOUT_MS * 1e-3 - (time.perf_counter() - t0)))

    def look_after_it():
        e.scan_set(sc["scan"])                                    # (a new scan arrives: the moment a finished -- or failed -- layout is dealt with)
        e.iterated_update(sc["x_prop"], sc["x_prop"], sc["P"])
    msg = _expect_timeout(look_after_it, e, "the layout beside the frames")
    assert "hand-back" in msg, msg
    _after(e, small_scene)
