// world.h -- a synthetic world long enough to DRIVE through, for the moving-trajectory legs of bench.py and the tests
// (workload plumbing, no counterpart in the reference: its only validation is a rosbag drive, README.md:38-55).
//
// A hall: floor z = 0, ceiling z = H, side walls y = +-W/2, end walls x = x0 and x0 + len, and box-shaped pillars on a
// jittered lattice (pitch P) that leave a lane |y| < lane free for the sensor.  A spinning LiDAR (beams x azimuths rays,
// elevation +-fov) is ray-cast analytically from a sensor that MOVES during the sweep (constant velocity over the 0.1 s of
// a sweep, identity attitude): ray i leaves from o(t_i), t_i = i / (n - 1) * 0.1 s, and its return is stored in the sensor
// frame of THAT instant -- the raw, distorted sweep a driver would publish.  Returns beyond `range` are dropped (a real
// sensor has a maximum range: new ground comes into view as the sensor advances).  The IMU poses handed out with the sweep
// are the true trajectory shifted by the frame's prediction error, so that the reference's backward propagation
// (IMU_Processing.hpp:333-370) undoes the distortion exactly, as it does for a well-propagated state.
// Everything is a pure function of (world, frame): sweeps can be generated in any order and on any number of threads.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#include "daliti_s2m.h"

typedef struct {
    double x0, len;      // the hall spans x0 .. x0 + len
    double width, H;     // y in +-width / 2, z in 0 .. H
    double pitch, lane;  // pillar lattice pitch; no pillar centre within `lane` of y = 0
    double range;        // maximum range of a return
    double step;         // metres the sensor advances per frame (along +x)
    double wobble, wobble_period;  // lateral sine of the path: amplitude, period in metres of x
    double sensor_z;
    double sigma;        // range noise (1 sigma, metres)
    double err_pos, err_rot;  // size of the predicted pose's error: metres, radians
    uint64_t seed;
} s2m_world;

namespace s2mw {

inline uint64_t mix(uint64_t z)
{
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
inline double u01(uint64_t h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0); }
struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(mix(seed)) {}
    uint64_t next() { s = mix(s); return s; }
    double uniform() { return u01(next()); }
    double normal()
    {
        const double a = uniform(), b = uniform();
        return std::sqrt(-2.0 * std::log(a > 1e-300 ? a : 1e-300)) * std::cos(6.283185307179586 * b);
    }
};

struct Pillar {
    double lo[3], hi[3];
    bool present;
};
// the pillar of lattice cell (i, j): a pure function of the world's seed
inline Pillar pillar(const s2m_world &w, int64_t i, int64_t j)
{
    Pillar p;
    const uint64_t h0 = mix(w.seed ^ mix((uint64_t)i * 0x9e3779b1ull + 17) ^ mix((uint64_t)j * 0x85ebca77ull + 91));
    const double cx = w.x0 + ((double)i + 0.5) * w.pitch + (u01(mix(h0 + 1)) - 0.5) * 0.5 * w.pitch;
    const double cy = -0.5 * w.width + ((double)j + 0.5) * w.pitch + (u01(mix(h0 + 2)) - 0.5) * 0.5 * w.pitch;
    const double hx = 0.75 + 0.75 * u01(mix(h0 + 3)), hy = 0.75 + 0.75 * u01(mix(h0 + 4));
    const double top = u01(mix(h0 + 5)) < 0.4 ? w.H : 3.0 + (w.H - 3.5) * u01(mix(h0 + 6));
    p.lo[0] = cx - hx; p.hi[0] = cx + hx;
    p.lo[1] = cy - hy; p.hi[1] = cy + hy;
    p.lo[2] = 0.0; p.hi[2] = top;
    p.present = std::fabs(cy) - hy > w.lane && cy + hy < 0.5 * w.width - 0.5 && cy - hy > -0.5 * w.width + 0.5 &&
                cx - hx > w.x0 + 0.5 && cx + hx < w.x0 + w.len - 0.5;
    return p;
}
inline int64_t lattice_i(const s2m_world &w, double x) { return (int64_t)std::floor((x - w.x0) / w.pitch); }
inline int64_t lattice_j(const s2m_world &w, double y) { return (int64_t)std::floor((y + 0.5 * w.width) / w.pitch); }

// the sensor at the START of frame f and its velocity during the frame's sweep (0.1 s)
inline void path(const s2m_world &w, int f, double p[3], double v[3])
{
    auto at = [&](int k, double q[3]) {
        q[0] = w.step * (double)k;
        q[1] = w.wobble_period > 0.0 ? w.wobble * std::sin(6.283185307179586 * q[0] / w.wobble_period) : 0.0;
        q[2] = w.sensor_z;
    };
    double q[3];
    at(f, p);
    at(f + 1, q);
    for (int k = 0; k < 3; ++k) v[k] = (q[k] - p[k]) / 0.1;
}

// first hit of the ray o + t d (t > 0) with the world; o inside the hall and outside every pillar
inline double cast(const s2m_world &w, const Pillar *pl, int npl, const double o[3], const double d[3])
{
    const double lo[3] = {w.x0, -0.5 * w.width, 0.0}, hi[3] = {w.x0 + w.len, 0.5 * w.width, w.H};
    double t = INFINITY;
    for (int k = 0; k < 3; ++k) {  // leaving the hall
        if (d[k] > 0.0) t = std::fmin(t, (hi[k] - o[k]) / d[k]);
        else if (d[k] < 0.0) t = std::fmin(t, (lo[k] - o[k]) / d[k]);
    }
    for (int q = 0; q < npl; ++q) {  // entering a pillar
        double t0 = 0.0, t1 = t;
        bool hit = true;
        for (int k = 0; k < 3 && hit; ++k) {
            if (d[k] != 0.0) {
                double a = (pl[q].lo[k] - o[k]) / d[k], b = (pl[q].hi[k] - o[k]) / d[k];
                if (a > b) { const double s = a; a = b; b = s; }
                t0 = std::fmax(t0, a);
                t1 = std::fmin(t1, b);
                hit = t0 <= t1;
            } else {
                hit = o[k] >= pl[q].lo[k] && o[k] <= pl[q].hi[k];
            }
        }
        if (hit && t0 > 0.0) t = std::fmin(t, t0);
    }
    return t;
}

inline void so3_exp(const double r[3], double R[9])
{
    const double n = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    double K[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (n > 1e-12) { K[1] = -r[2] / n; K[2] = r[1] / n; K[3] = r[2] / n; K[5] = -r[0] / n; K[6] = -r[1] / n; K[7] = r[0] / n; }
    const double s = std::sin(n), c = 1.0 - std::cos(n);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double kk = 0.0;
            for (int q = 0; q < 3; ++q) kk += K[i * 3 + q] * K[q * 3 + j];
            R[i * 3 + j] = (i == j ? 1.0 : 0.0) + s * K[i * 3 + j] + c * kk;
        }
}

}  // namespace s2mw

// One raw sweep of frame f: records (beams * az at most, 12 floats each: x y z - time_ratio - sweep_seconds - ...,
// pcl::PointXYZINormal's layout, time = rec[4] * rec[6]), in firing order (index = azimuth-major: the sensor spins while
// all beams fire together); n_poses IMU poses across the sweep; x_prop = the predicted state at the END of the sweep
// (truth [+] the frame's prediction error: also the state_end of the undistortion); x_true = the true one.
// Returns the number of records.
inline int64_t s2m_world_sweep_impl(const s2m_world &w, int f, int beams, int az, float *rec, s2m_imu_pose *poses, int n_poses,
                                    double x_prop[S2M_STATE_DOUBLES], double x_true[S2M_STATE_DOUBLES])
{
    using namespace s2mw;
    double p0[3], v[3];
    path(w, f, p0, v);
    const double T = 0.1;
    const int64_t n_rays = (int64_t)beams * az;
    // the pillars a ray of this sweep can reach
    const int64_t i0 = lattice_i(w, p0[0] - w.range - w.pitch), i1 = lattice_i(w, p0[0] + v[0] * T + w.range + w.pitch);
    const int64_t j0 = lattice_j(w, p0[1] - w.range - w.pitch), j1 = lattice_j(w, p0[1] + w.range + w.pitch);
    Pillar *pl = new Pillar[(size_t)((i1 - i0 + 1) * (j1 - j0 + 1))];
    int npl = 0;
    for (int64_t i = i0; i <= i1; ++i)
        for (int64_t j = j0; j <= j1; ++j) {
            const Pillar p = pillar(w, i, j);
            if (p.present) pl[npl++] = p;
        }
    const double fov = 22.5 * 3.14159265358979323846 / 180.0;
    int64_t n = 0;
    for (int64_t r = 0; r < n_rays; ++r) {
        const int a = (int)(r / beams), b = (int)(r % beams);
        const double el = beams > 1 ? -fov + 2.0 * fov * (double)b / (double)(beams - 1) : 0.0;
        const double th = 6.283185307179586 * (double)a / (double)az;
        const double d[3] = {std::cos(el) * std::cos(th), std::cos(el) * std::sin(th), std::sin(el)};
        const float ratio = n_rays > 1 ? (float)((double)r / (double)(n_rays - 1)) : 0.0f;
        const double t = (double)ratio * T;
        const double o[3] = {p0[0] + v[0] * t, p0[1] + v[1] * t, p0[2] + v[2] * t};
        double range = cast(w, pl, npl, o, d);
        if (!(range <= w.range)) continue;
        Rng g(w.seed ^ mix((uint64_t)f * 1000003ull + (uint64_t)r));
        range += w.sigma * g.normal();
        float *q = rec + n * 12;
        std::memset(q, 0, 12 * sizeof(float));
        q[0] = (float)(d[0] * range); q[1] = (float)(d[1] * range); q[2] = (float)(d[2] * range);
        q[4] = ratio;
        q[6] = (float)T;
        ++n;
    }
    delete[] pl;
    // the frame's prediction error and the states
    Rng g(w.seed ^ mix(0xabcdefull + (uint64_t)f));
    double er[3], ep[3], nr = 0.0, np_ = 0.0;
    for (int k = 0; k < 3; ++k) { er[k] = g.normal(); ep[k] = g.normal(); nr += er[k] * er[k]; np_ += ep[k] * ep[k]; }
    for (int k = 0; k < 3; ++k) { er[k] *= w.err_rot / std::sqrt(nr > 0 ? nr : 1.0); ep[k] *= w.err_pos / std::sqrt(np_ > 0 ? np_ : 1.0); }
    double Re[9];
    so3_exp(er, Re);
    const double I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::memset(x_true, 0, S2M_STATE_DOUBLES * sizeof(double));
    std::memset(x_prop, 0, S2M_STATE_DOUBLES * sizeof(double));
    std::memcpy(x_true, I9, sizeof(I9));
    std::memcpy(x_true + 12, I9, sizeof(I9));
    std::memcpy(x_prop, Re, sizeof(Re));
    std::memcpy(x_prop + 12, I9, sizeof(I9));
    for (int k = 0; k < 3; ++k) {
        x_true[9 + k] = p0[k] + v[k] * T;
        x_prop[9 + k] = x_true[9 + k] + ep[k];
        x_true[24 + k] = v[k];  // (state layout: rot 0..8, pos 9..11, R_L_I 12..20, T_L_I 21..23, vel 24..26, ...)
        x_prop[24 + k] = v[k];
    }
    // IMU poses of the propagated (erroneous) trajectory: the same error all along the sweep.  The reference's loop holds
    // the pose's offset time, mean acceleration and angular velocity, velocity, position, attitude (common_lib.h, Pose6D);
    // world-frame velocity: the propagated attitude is constant, the motion a translation.
    for (int k = 0; k < n_poses; ++k) {
        s2m_imu_pose &q = poses[k];
        std::memset(&q, 0, sizeof(q));
        q.offset_time = n_poses > 1 ? 1.02 * T * (double)k / (double)(n_poses - 1) : 0.0;
        for (int c = 0; c < 3; ++c) {
            q.vel[c] = v[c];
            q.pos[c] = p0[c] + v[c] * q.offset_time + ep[c];
        }
        std::memcpy(q.rot, Re, sizeof(Re));
    }
    return n;
}

// M points on the surfaces of the hall between x0 and x0 + span (floor, ceiling, side walls, the end wall at x0, the
// pillars' sides and tops inside), area-proportional, sigma along the face normal: the seed map of a run that starts there
inline void s2m_world_seed_impl(const s2m_world &w, double span, int64_t M, float *xyz)
{
    using namespace s2mw;
    // faces: {origin, edge u, edge v, normal axis}
    struct Face { double o[3], u[3], v[3]; int axis; double area; };
    const int64_t i1 = lattice_i(w, w.x0 + span), j1 = lattice_j(w, 0.5 * w.width);
    const size_t max_faces = 5 + 5 * (size_t)((i1 + 2) * (j1 + 2));
    Face *fs = new Face[max_faces];
    int nf = 0;
    auto add = [&](double ox, double oy, double oz, double ux, double uy, double uz, double vx, double vy, double vz, int axis) {
        Face f = {{ox, oy, oz}, {ux, uy, uz}, {vx, vy, vz}, axis, 0.0};
        f.area = std::sqrt(ux * ux + uy * uy + uz * uz) * std::sqrt(vx * vx + vy * vy + vz * vz);
        if (f.area > 0.0) fs[nf++] = f;
    };
    const double y0 = -0.5 * w.width;
    add(w.x0, y0, 0.0, span, 0, 0, 0, w.width, 0, 2);
    add(w.x0, y0, w.H, span, 0, 0, 0, w.width, 0, 2);
    add(w.x0, y0, 0.0, span, 0, 0, 0, 0, w.H, 1);
    add(w.x0, -y0, 0.0, span, 0, 0, 0, 0, w.H, 1);
    add(w.x0, y0, 0.0, 0, w.width, 0, 0, 0, w.H, 0);
    for (int64_t i = 0; i <= i1; ++i)
        for (int64_t j = 0; j <= j1; ++j) {
            const Pillar p = pillar(w, i, j);
            if (!p.present || p.lo[0] >= w.x0 + span) continue;
            const double dx = p.hi[0] - p.lo[0], dy = p.hi[1] - p.lo[1], dz = p.hi[2] - p.lo[2];
            add(p.lo[0], p.lo[1], 0.0, dx, 0, 0, 0, 0, dz, 1);
            add(p.lo[0], p.hi[1], 0.0, dx, 0, 0, 0, 0, dz, 1);
            add(p.lo[0], p.lo[1], 0.0, 0, dy, 0, 0, 0, dz, 0);
            add(p.hi[0], p.lo[1], 0.0, 0, dy, 0, 0, 0, dz, 0);
            if (p.hi[2] < w.H) add(p.lo[0], p.lo[1], p.hi[2], dx, 0, 0, 0, dy, 0, 2);
        }
    double total = 0.0;
    for (int k = 0; k < nf; ++k) total += fs[k].area;
    double *cum = new double[(size_t)nf];
    double acc = 0.0;
    for (int k = 0; k < nf; ++k) { acc += fs[k].area / total; cum[k] = acc; }
    Rng g(w.seed ^ 0x5eedull);
    for (int64_t m = 0; m < M; ++m) {
        const double r = g.uniform();
        int lo = 0, hi = nf - 1;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (cum[mid] < r) lo = mid + 1; else hi = mid; }
        const Face &f = fs[lo];
        const double a = g.uniform(), b = g.uniform(), s = w.sigma * g.normal();
        double q[3];
        for (int k = 0; k < 3; ++k) q[k] = f.o[k] + a * f.u[k] + b * f.v[k];
        q[f.axis] += s;
        xyz[3 * m] = (float)q[0]; xyz[3 * m + 1] = (float)q[1]; xyz[3 * m + 2] = (float)q[2];
    }
    delete[] cum;
    delete[] fs;
}
