"""Deterministic synthetic map / scan / filter inputs of SURVEY.md section 8(d).

Scene: a closed box (floor z=0, ceiling z=H, walls x,y=+-L/2) sampled at ~50 pts/m^2 with
sigma=1 cm noise along the face normal; scan: a spinning LiDAR ray-cast analytically against the
box.  numpy's legacy ``RandomState`` (MT19937) keeps the streams stable across numpy versions.
This is workload plumbing for tests and bench.py; it has no counterpart in the reference.
"""
import numpy as np

BOX_H = 10.0
# named BASELINE.json configs: (beams, azimuths, map points, box side L)
CONFIGS = {
    "C1": dict(beams=16, az=625, M=100_000, L=22.0),
    "C2": dict(beams=64, az=1024, M=1_000_000, L=95.0),
    "C3": dict(beams=64, az=1024, M=5_000_000, L=215.0),
    "C4": dict(beams=128, az=1024, M=20_000_000, L=440.0),
    # batched odometry: 8 independent C3-sized scans (seeds 2..9, sensor offsets (k - 3.5) * 2 m) vs one map
    "C5": dict(beams=64, az=1024, M=5_000_000, L=215.0, replicas=8),
    # the same batch against C4's map: 8 scans of 65 536 points vs 20 M points (320 MB of points: beyond the 256 MB Infinity Cache,
    # where C5 sits inside it) -- the engine where HBM is the roof
    "C5b": dict(beams=64, az=1024, M=20_000_000, L=440.0, replicas=8),
    # C3's cloud at the reference's map density: map through Add_Points(downsample 0.5 m), scan through
    # VoxelGrid(0.5 m) (bench.py / tests build it with s2m_map_add and s2m_scan_set_downsampled)
    "R1": dict(beams=64, az=1024, M=5_000_000, L=215.0),
}
SENSOR_POS = np.array([0.0, 0.0, 1.5])
DTHETA0 = np.array([0.010, -0.008, 0.015])
DPOS0 = np.array([0.05, -0.04, 0.03])


def side_for_points(M, density=50.0, H=BOX_H):
    """L such that (2 L^2 + 4 L H) * density == M."""
    a = M / density
    return (-4 * H + np.sqrt(16 * H * H + 8 * a)) / 4


def make_map(M, L=None, seed=1, sigma=0.01, H=BOX_H):
    """M x 3 float32 map points on the six faces of the box, area-proportional."""
    if L is None:
        L = side_for_points(M)
    rs = np.random.RandomState(seed)
    areas = np.array([L * L, L * L, L * H, L * H, L * H, L * H])
    face = rs.choice(6, size=M, p=areas / areas.sum())
    u = rs.uniform(-0.5, 0.5, size=M)
    v = rs.uniform(0.0, 1.0, size=M)
    off = rs.normal(0.0, sigma, size=M)
    p = np.empty((M, 3))
    w = rs.uniform(-0.5, 0.5, size=M)  # second in-plane coordinate for floor / ceiling
    for f in range(6):
        m = face == f
        if f == 0:    # floor
            p[m] = np.stack([u[m] * L, w[m] * L, off[m]], 1)
        elif f == 1:  # ceiling
            p[m] = np.stack([u[m] * L, w[m] * L, H + off[m]], 1)
        elif f == 2:  # wall x = -L/2
            p[m] = np.stack([-L / 2 + off[m], u[m] * L, v[m] * H], 1)
        elif f == 3:  # wall x = +L/2
            p[m] = np.stack([L / 2 + off[m], u[m] * L, v[m] * H], 1)
        elif f == 4:  # wall y = -L/2
            p[m] = np.stack([u[m] * L, -L / 2 + off[m], v[m] * H], 1)
        else:         # wall y = +L/2
            p[m] = np.stack([u[m] * L, L / 2 + off[m], v[m] * H], 1)
    return p.astype(np.float32)


def make_scan(beams, az, L, seed=2, sigma=0.01, H=BOX_H, sensor_pos=SENSOR_POS, fov_deg=22.5):
    """(beams*az) x 3 float32 body-frame points, index = beam * az + azimuth."""
    rs = np.random.RandomState(seed)
    el = np.deg2rad(np.linspace(-fov_deg, fov_deg, beams))
    th = np.arange(az) * (2 * np.pi / az)
    E, T = np.meshgrid(el, th, indexing="ij")
    d = np.stack([np.cos(E) * np.cos(T), np.cos(E) * np.sin(T), np.sin(E)], -1).reshape(-1, 3)
    o = np.asarray(sensor_pos, float)
    lo = np.array([-L / 2, -L / 2, 0.0])
    hi = np.array([L / 2, L / 2, H])
    with np.errstate(divide="ignore", invalid="ignore"):
        t_lo = (lo - o) / d
        t_hi = (hi - o) / d
    t_exit = np.where(d > 0, t_hi, t_lo)
    t_exit = np.where(d == 0, np.inf, t_exit)
    rng = t_exit.min(axis=1) + rs.normal(0.0, sigma, size=len(d))
    # the sensor frame is the body frame (R_L_I = I, T_L_I = 0, identity attitude)
    return (d * rng[:, None]).astype(np.float32)


def so3_exp(v):
    v = np.asarray(v, float)
    n = np.linalg.norm(v)
    if n <= 1e-5:
        return np.eye(3)
    r = v / n
    K = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
    return np.eye(3) + np.sin(n) * K + (1 - np.cos(n)) * K @ K


def make_state(rot=None, pos=None):
    """36-double flat state: rot9 pos3 R_LI9 T_LI3 vel3 bg3 ba3 grav3 (row-major matrices)."""
    s = np.zeros(36)
    s[0:9] = (np.eye(3) if rot is None else np.asarray(rot, float)).ravel()
    s[9:12] = 0.0 if pos is None else pos
    s[12:21] = np.eye(3).ravel()
    return s


def filter_inputs(sensor_pos=SENSOR_POS, dtheta=DTHETA0, dpos=DPOS0):
    """(x_truth, x_prop, P0): x_prop = truth [+] (dtheta, dpos, 0...), P0 diagonal."""
    x_true = make_state(np.eye(3), sensor_pos)
    x_prop = make_state(so3_exp(dtheta), np.asarray(sensor_pos) + np.asarray(dpos))
    P = np.eye(24) * 1e-4
    P[:6, :6] = np.eye(6) * 1e-3
    return x_true, x_prop, P


def replica_offset(k):
    """Sensor offset along x of replica k of a batched-odometry config ((k - 3.5) * 2 m)."""
    return (k - 3.5) * 2.0


def replica_scan(name, k):
    """(body-frame scan, sensor position) of replica k: seed 2 + k, sensor moved by replica_offset(k)."""
    c = CONFIGS[name]
    pos = SENSOR_POS + np.array([replica_offset(k), 0.0, 0.0])
    return make_scan(c["beams"], c["az"], c["L"], seed=2 + k, sensor_pos=pos), pos


def make_config(name, scan_seed=2, sensor_dx=0.0):
    c = CONFIGS[name]
    pos = SENSOR_POS + np.array([sensor_dx, 0.0, 0.0])
    m = make_map(c["M"], c["L"], seed=1)
    s = make_scan(c["beams"], c["az"], c["L"], seed=scan_seed, sensor_pos=pos)
    x_true, x_prop, P = filter_inputs(pos)
    return dict(map=m, scan=s, x_true=x_true, x_prop=x_prop, P=P, L=c["L"])


def make_small(M=20000, beams=16, az=128, seed_map=1, seed_scan=2):
    """Small scene for oracle-sized parity tests and the golden fixture."""
    L = side_for_points(M)
    m = make_map(M, L, seed=seed_map)
    s = make_scan(beams, az, L, seed=seed_scan)
    x_true, x_prop, P = filter_inputs()
    return dict(map=m, scan=s, x_true=x_true, x_prop=x_prop, P=P, L=L)
