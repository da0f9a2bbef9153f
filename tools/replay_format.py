"""Byte formats of the ROS-free replay harness (tools/replay_node.cpp) and of the node's topics.

* serialised sensor_msgs/PointCloud2 (ROS 1 wire format) for pcl::PointXYZINormal (48-byte records,
  /laser_cloud_surf and /Laser_map) and pcl::PointXYZI (32-byte records, /cloud_effected): writer and
  parser, the Python mirror of include/daliti_s2m_wire.h;
* the "S2MREPL1" stream replay_node reads (header + per frame: IMUpose, propagated state, covariance,
  the serialised /laser_cloud_surf message).
Used by tests/test_replay.py and by anyone who wants to feed recorded scans to the engine without ROS.
"""
import struct

import numpy as np

FLOAT32 = 7
FIELDS_XYZINORMAL = [("x", 0), ("y", 4), ("z", 8), ("normal_x", 16), ("normal_y", 20), ("normal_z", 24),
                     ("intensity", 32), ("curvature", 36)]
FIELDS_XYZI = [("x", 0), ("y", 4), ("z", 8), ("intensity", 16)]


def xyzinormal_records(xyz, time_ratio, ring, timespan, intensity=None):
    """n x 12 float32 = pcl::PointXYZINormal as feature_extract.cpp:335-346 fills it."""
    n = len(xyz)
    rec = np.zeros((n, 12), np.float32)
    rec[:, 0:3] = xyz
    rec[:, 4] = time_ratio          # normal_x = t / timespan
    rec[:, 5] = ring                # normal_y
    rec[:, 6] = timespan            # normal_z, seconds
    if intensity is not None:
        rec[:, 8] = intensity
    return rec


def _string(s):
    b = s.encode()
    return struct.pack("<I", len(b)) + b


def serialize_pointcloud2(records, fields, stamp, frame_id, seq=0):
    """records: (n, point_step / 4) float32.  stamp in seconds (ros::Time().fromSec)."""
    records = np.ascontiguousarray(records, np.float32)
    n, step = records.shape[0], records.shape[1] * 4
    sec = int(stamp)
    nsec = int((stamp - sec) * 1e9 + 0.5)
    if nsec >= 1000000000:
        sec, nsec = sec + 1, nsec - 1000000000
    out = [struct.pack("<III", seq, sec, nsec), _string(frame_id), struct.pack("<II", 1, n),
           struct.pack("<I", len(fields))]
    for name, off in fields:
        out.append(_string(name) + struct.pack("<IBI", off, FLOAT32, 1))
    data = records.tobytes()
    out.append(struct.pack("<BII", 0, step, step * n) + struct.pack("<I", len(data)) + data + struct.pack("<B", 1))
    return b"".join(out)


def parse_pointcloud2(buf, at=0):
    """Returns (dict, next offset)."""
    def u32():
        nonlocal at
        v = struct.unpack_from("<I", buf, at)[0]
        at += 4
        return v

    def string():
        nonlocal at
        n = u32()
        s = bytes(buf[at:at + n]).decode()
        at += n
        return s
    seq, sec, nsec = u32(), u32(), u32()
    frame_id = string()
    height, width = u32(), u32()
    fields = []
    for _ in range(u32()):
        name = string()
        off = u32()
        dt = buf[at]
        at += 1
        cnt = u32()
        fields.append((name, off, dt, cnt))
    big = buf[at]
    at += 1
    step, row = u32(), u32()
    dlen = u32()
    data = np.frombuffer(bytes(buf[at:at + dlen]), np.float32).reshape(width * height, step // 4) if step else np.zeros((0, 0), np.float32)
    at += dlen
    dense = buf[at]
    at += 1
    return dict(seq=seq, stamp=sec + 1e-9 * nsec, frame_id=frame_id, height=height, width=width, fields=fields,
                is_bigendian=big, point_step=step, row_step=row, records=data, is_dense=dense), at


def parse_length_prefixed_messages(buf):
    """cloud_effected.pc2s / laser_map.pc2s: uint32 length + serialised PointCloud2, repeated."""
    out, at = [], 0
    while at < len(buf):
        n = struct.unpack_from("<I", buf, at)[0]
        msg, _ = parse_pointcloud2(buf, at + 4)
        out.append(msg)
        at += 4 + n
    return out


def write_stream(path, frames, max_iter=5, extrinsic_est_en=0, feat_threshold=100, filter_size_surf=0.5,
                 filter_size_map=0.5, cube_len=1000.0):
    """frames: list of dict(state (36,), P (24, 24), imu (K, 22), msg bytes)."""
    with open(path, "wb") as f:
        f.write(b"S2MREPL1" + struct.pack("<Iiii", len(frames), max_iter, extrinsic_est_en, feat_threshold))
        f.write(struct.pack("<ddd", filter_size_surf, filter_size_map, cube_len))
        for fr in frames:
            imu = np.ascontiguousarray(fr["imu"], np.float64).reshape(-1, 22)
            msg = fr["msg"]
            f.write(struct.pack("<II", len(imu), len(msg)))
            f.write(np.ascontiguousarray(fr["state"], np.float64).tobytes())
            f.write(np.ascontiguousarray(fr["P"], np.float64).tobytes())
            f.write(imu.tobytes())
            f.write(msg + b"\0" * ((-len(msg)) % 8))
