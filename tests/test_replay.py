"""wire formats, the ROS-free replay harness and the node-side patch.

CPU: include/daliti_s2m_wire.h (compiled as plain C) round-trips a serialised sensor_msgs/PointCloud2 byte
for byte against the Python writer; integration/laserMapping_s2m.patch applies to the reference tree (in
this container only: /root/reference does not exist on the GPU box).
GPU: tools/replay_node (g++, links only libdaliti_s2m.so) replays a multi-frame stream of serialised
/laser_cloud_surf messages; its mat_out rows, /cloud_effected clouds, odometry and final map are compared
with the oracle driven through the same frames.
"""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from conftest import bits

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import replay_format as rf  # noqa: E402

REF = "/root/reference"


def test_wire_header_roundtrip_in_plain_c(tmp_path):
    exe = str(tmp_path / "wire_roundtrip")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "wire_roundtrip.c"), "-o", exe])
    rs = np.random.RandomState(0)
    for kind, fields, width, frame in ((1, rf.FIELDS_XYZINORMAL, 12, "camera_init"), (0, rf.FIELDS_XYZI, 8, "/aft_mapped"),
                                       (1, rf.FIELDS_XYZINORMAL, 12, "ab")):
        rec = rs.uniform(-50, 50, (257, width)).astype(np.float32)
        if kind:
            rec[:, 4] = rs.uniform(0, 1, len(rec)); rec[:, 6] = 0.1
        msg = rf.serialize_pointcloud2(rec, fields, 1234.000000789, frame, seq=7)
        src, dst = tmp_path / "in.bin", tmp_path / "out.bin"
        src.write_bytes(msg)
        out = subprocess.run([exe, str(src), str(dst), str(kind)], capture_output=True, text=True, check=True).stdout
        assert "used %d n 257 step %d" % (len(msg), width * 4) in out and "sec 1234 nsec 789" in out
        assert "frame %s" % frame in out and "truncated rc -1" in out
        if kind:
            assert "stride 12 oa 4 ob 6 rc 0" in out   # normal_x / normal_z relative to x, in floats
            assert ("first %.9g %.9g %.9g t %.9g span %.9g" % tuple(rec[0, [0, 1, 2, 4, 6]])) in out
        assert dst.read_bytes() == msg                  # C writer == Python writer, byte for byte
        back, used = rf.parse_pointcloud2(msg)
        assert used == len(msg) and (bits(back["records"]) == bits(rec)).all() and back["point_step"] == width * 4


def test_wire_parser_never_reads_past_its_input(tmp_path):
    """Every truncation and a few thousand corrupted copies of a valid message through s2m_pc2_parse, under
    AddressSanitizer, in buffers of exactly the size handed over."""
    exe = str(tmp_path / "wire_fuzz")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-O1", "-g", "-fsanitize=address", "-fno-omit-frame-pointer",
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "wire_fuzz.c"), "-o", exe])
    rs = np.random.RandomState(1)
    for kind, fields, width in ((1, rf.FIELDS_XYZINORMAL, 12), (0, rf.FIELDS_XYZI, 8)):
        rec = rs.uniform(-50, 50, (33, width)).astype(np.float32)
        msg = rf.serialize_pointcloud2(rec, fields, 99.5, "camera_init", seq=3)
        src = tmp_path / ("fuzz%d.bin" % kind)
        src.write_bytes(msg)
        r = subprocess.run([exe, str(src)], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.startswith("ok:"), r.stdout + r.stderr


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree only exists in the build container")
def test_node_patch_applies_to_the_reference():
    patch = os.path.join(ROOT, "integration", "laserMapping_s2m.patch")
    assert os.path.exists(patch)
    r = subprocess.run(["patch", "--dry-run", "-p1", "-d", os.path.join(REF, "eskf_lio"), "-i", patch],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "src/laserMapping.cpp" in r.stdout and "CMakeLists.txt" in r.stdout
    assert "FAILED" not in r.stdout and "fuzz" not in r.stdout


# ---- the replay: generator, oracle-side runner, comparison -----------------------------------------------
def make_frames(n_frames=5, L=12.0, seed=20):
    """A sensor translating through a 12 m box; every sweep lasts 0.1 s at constant velocity (pure translation
    inside a sweep keeps the undistortion free of sin/cos, so GPU and oracle agree bit for bit and everything
    downstream can be compared exactly).  Raw points are the end-of-sweep ray-cast moved back along the
    motion: undistorting them returns the ray-cast."""
    from daliti_amd import synth
    rs = np.random.RandomState(seed)
    span, K = 0.1, 6
    frames, truth = [], []
    pos = np.array([0.0, 0.0, 1.5])
    vel = np.array([0.6, 0.2, 0.0])
    P = np.eye(24) * 1e-4
    P[:6, :6] = np.eye(6) * 1e-3
    for f in range(n_frames):
        p_start, p_end = pos, pos + vel * span
        scan_end = synth.make_scan(32, 256, L, seed=seed + f, sensor_pos=p_end).astype(np.float64)  # body frame at sweep end
        n = len(scan_end)
        ratio = rs.uniform(0.0, 1.0, n).astype(np.float32)
        t = (ratio * np.float32(span)).astype(np.float64)
        # inverse of IMU_Processing.hpp:358 for R = I, R_L_I = I, T_L_I = 0: P_i = P_c - T_ei,
        # T_ei = pos_head + vel * dt - pos_end = position at the point's time - pos_end
        raw = scan_end - ((p_start[None, :] + vel[None, :] * t[:, None]) - p_end[None, :])
        rec = rf.xyzinormal_records(raw.astype(np.float32), ratio, rs.randint(0, 32, n), span,
                                    intensity=rs.uniform(0, 255, n))
        msg = rf.serialize_pointcloud2(rec, rf.FIELDS_XYZINORMAL, 100.0 + f * span, "camera_init", seq=f)
        # the propagated state the node would hold after p_imu->Process: truth + a small error
        # (frame 0 seeds the map, so its state defines the world frame: exact)
        state = synth.make_state(synth.so3_exp(rs.normal(0, 0.004, 3) * (f > 0)), p_end + rs.normal(0, 0.02, 3) * (f > 0))
        # IMUpose is propagated from that same state (IMU_Processing.hpp:224-310): it ends at state.pos_end
        times = np.linspace(0.0, span * 1.02, K)
        imu = np.zeros((K, 22))
        for k in range(K):
            imu[k, 0] = times[k]
            imu[k, 7:10] = vel
            imu[k, 10:13] = state[9:12] - vel * (span - times[k])
            imu[k, 13:22] = state[0:9]
        frames.append(dict(state=state, P=P, imu=imu, msg=msg))
        truth.append(p_end)
        pos = p_end + vel * 0.0   # back-to-back sweeps
    return frames, truth


def oracle_replay(oracle, frames, fs_surf, fs_map, max_iter, feat_threshold):
    """The same loop as tools/replay_node.cpp, every stage on the CPU oracle."""
    cfg = oracle.default_cfg(max_iter=max_iter, feat_threshold=feat_threshold)
    om = None
    rows, odom, effected = [], [], []
    t0 = None
    queue = []
    for fr in frames:
        msg, _ = rf.parse_pointcloud2(fr["msg"])
        t0 = msg["stamp"] if t0 is None else t0
        und, _ = oracle.undistort(msg["records"], 4, 6, fr["imu"], fr["state"], sort=True)
        down = oracle.voxel_downsample(und, fs_surf)
        if om is None:
            om = oracle.Map(oracle.body_to_world(fr["state"], down))
            continue
        tree = oracle.KdTree(om.points())
        r = oracle.iterated_update(cfg, tree, down, fr["state"], fr["state"], fr["P"], feat_queue=queue)
        queue = list(r["feat_queue"])
        for it in range(r["iters"]):
            stop = int(r["ekf_stop"]) if it == r["iters"] - 1 else 0
            rows.append((msg["stamp"] - t0, int(r["effct"][it]), r["total_res"][it] / r["effct"][it],
                         int(r["conv"][it - 1]) if it else 0, stop, len(down)))
        # effective set of the last pass: replay it on the final neighbours with the state BEFORE the last update
        ps = oracle.PassState(len(down))
        xs = np.array(fr["state"], float)
        x_hist = [xs.copy()]
        for it in range(r["iters"] - 1):
            x_hist.append(oracle.boxplus(x_hist[-1], r["solution"][it]))
        for it in range(r["iters"]):
            oracle.residual_pass(cfg, tree, down, x_hist[it], bool(r["rematch"][it]), ps)
        assert ps.effct == r["effct"][-1]
        effected.append(oracle.body_to_world(r["x"], down[ps.eff.astype(bool)]))
        if not r["ekf_stop"]:
            nn = r["nn_idx"]
            cnt = (nn >= 0).sum(1).astype(np.int32)
            to_add, no_down = oracle.map_incremental_lists(down, r["x"], tree.xyz[np.maximum(nn, 0)], cnt, fs_map)
            om.add(to_add, True, fs_map)
            om.add(no_down, False)
        odom.append((msg["stamp"] - t0, r["x"].copy(), r["iters"], int(r["effct"][-1]), om.size()))
    return rows, odom, effected, om


@pytest.mark.gpu
def test_replay_node_matches_the_oracle(tmp_path, oracle):
    exe = str(tmp_path / "replay_node")
    lib = os.path.join(ROOT, "daliti_amd", "_lib")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "replay_node.cpp"), "-L", lib, "-ldaliti_s2m",
                           "-Wl,-rpath," + lib, "-o", exe])
    frames, truth = make_frames()
    fs, mi, thr = 0.25, 5, 50
    stream = str(tmp_path / "frames.bin")
    rf.write_stream(stream, frames, max_iter=mi, feat_threshold=thr, filter_size_surf=fs, filter_size_map=fs)
    out = tmp_path / "out"
    out.mkdir()
    r = subprocess.run([exe, stream, str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rows, odom, effected, om = oracle_replay(oracle, frames, fs, fs, mi, thr)
    # Log/mat_out.txt: [t, effct_feat_num, res_mean_last, converged, EKF_stop_flg, N, lidar_fail, edge_n] (:936-937)
    got_rows = [ln.split() for ln in (out / "mat_out.txt").read_text().splitlines()]
    assert len(got_rows) == len(rows) > 0
    for g, w in zip(got_rows, rows):
        assert len(g) == 8 and g[6:] == ["0", "0"]
        assert abs(float(g[0]) - w[0]) < 1e-5 and int(g[1]) == w[1] and [int(g[3]), int(g[4]), int(g[5])] == list(w[3:6])
        assert g[2] == "%g" % w[2] or abs(float(g[2]) - w[2]) <= 2e-6 * abs(w[2])   # iostream default = %g, 6 digits
    # odometry: pose per frame within 1e-9 of the oracle's, and it tracks the truth
    lines = [ln.split() for ln in (out / "odometry.txt").read_text().splitlines()]
    assert lines[0][1] == "seed" and len(lines) == len(frames)
    for ln, (t, x, iters, eff, msize), tr in zip(lines[1:], odom, truth[1:]):
        v = np.array([float(s) for s in ln])
        assert abs(v[0] - t) < 1e-6 and np.abs(v[1:4] - x[9:12]).max() < 1e-9 and np.abs(v[4:13] - x[:9]).max() < 1e-9
        assert (int(v[13]), int(v[14]), int(v[17])) == (iters, eff, msize)
        assert np.abs(v[1:4] - tr).max() < 0.02
    # /cloud_effected: laserCloudOri of the last pass in the world frame, PointXYZI records
    msgs = rf.parse_length_prefixed_messages((out / "cloud_effected.pc2s").read_bytes())
    assert len(msgs) == len(effected)
    for m, w in zip(msgs, effected):
        assert m["point_step"] == 32 and m["frame_id"] == "camera_init" and [f[0] for f in m["fields"]] == ["x", "y", "z", "intensity"]
        assert m["records"].shape[0] == len(w)
        assert (np.abs(m["records"][:, :3] - w) <= np.spacing(np.abs(w))).all()
    # /Laser_map after every registered frame (a host mirror fed by s2m_map_get_changes, one whole-map fetch at the start);
    # after the last frame: the same point set as the oracle's map
    maps = rf.parse_length_prefixed_messages((out / "laser_map.pc2s").read_bytes())
    assert len(maps) == len(frames) - 1 and "1 whole-map fetches" in r.stderr, r.stderr[-300:]
    assert [m["records"].shape[0] for m in maps] == [o[4] for o in odom]
    mm = maps[-1]
    assert mm["point_step"] == 48 and mm["records"].shape[0] == om.size()

    def rows_sorted(a):
        a = np.ascontiguousarray(a, np.float32).reshape(-1, 3)
        return a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]
    assert (bits(rows_sorted(mm["records"][:, :3])) == bits(rows_sorted(om.points()))).all()
