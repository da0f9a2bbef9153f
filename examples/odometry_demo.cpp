// odometry_demo.cpp -- the scan-to-map engine driven from C++ through the C ABI only
// (include/daliti_s2m.h), the way a patched eskf_lio/src/laserMapping.cpp would drive it:
//   seed the map from the first scan (laserMapping.cpp:780-793), then per frame
//   scan hand-over (:775-778) -> iterated ESKF update (:820-1102) -> map growth (:1165-1168).
// Synthetic data: a LiDAR moving through a closed box (floor, ceiling, four walls).
//
// build:  g++ -O2 -std=c++17 -I include examples/odometry_demo.cpp -L daliti_amd/_lib -ldaliti_s2m \
//             -Wl,-rpath,$PWD/daliti_amd/_lib -o /tmp/odometry_demo
// run:    /tmp/odometry_demo [frames]        (needs a gfx950 GPU; exits 2 when none is present)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "daliti_s2m.h"

namespace {
const double L = 40.0, H = 8.0;

// analytic ray / box intersection from `o` along `d`
double hit(const double o[3], const double d[3])
{
    const double lo[3] = {-L / 2, -L / 2, 0.0}, hi[3] = {L / 2, L / 2, H};
    double t = 1e30;
    for (int k = 0; k < 3; ++k) {
        if (d[k] > 0) t = std::fmin(t, (hi[k] - o[k]) / d[k]);
        else if (d[k] < 0) t = std::fmin(t, (lo[k] - o[k]) / d[k]);
    }
    return t;
}

std::vector<float> make_scan(const double pos[3], int beams, int az, std::mt19937 &rng)
{
    std::normal_distribution<double> noise(0.0, 0.01);
    std::vector<float> s;
    s.reserve((size_t)beams * az * 3);
    for (int b = 0; b < beams; ++b) {
        const double el = (-22.5 + 45.0 * b / (beams - 1)) * M_PI / 180.0;
        for (int a = 0; a < az; ++a) {
            const double th = 2 * M_PI * a / az;
            const double d[3] = {std::cos(el) * std::cos(th), std::cos(el) * std::sin(th), std::sin(el)};
            const double r = hit(pos, d) + noise(rng);
            for (int k = 0; k < 3; ++k) s.push_back((float)(d[k] * r));  // body frame (identity attitude)
        }
    }
    return s;
}

#define CK(call)                                                                            \
    do {                                                                                    \
        int rc_ = (call);                                                                   \
        if (rc_ != S2M_OK) {                                                                \
            std::fprintf(stderr, "%s -> %d (%s) %s\n", #call, rc_, s2m_strerror(rc_),       \
                         eng ? s2m_last_error(eng) : "");                                   \
            return rc_ == S2M_ERR_NO_DEVICE ? 2 : 1;                                        \
        }                                                                                   \
    } while (0)
}  // namespace

int main(int argc, char **argv)
{
    const int frames = argc > 1 ? std::atoi(argv[1]) : 10;
    s2m_engine *eng = nullptr;
    s2m_config cfg;
    s2m_config_default(&cfg);
    cfg.max_iter = 5;
    cfg.feat_threshold = 50;
    CK(s2m_create(&cfg, &eng));

    std::mt19937 rng(7);
    double truth[3] = {0.0, 0.0, 1.5};
    // state: rot_end pos_end R_L_I T_L_I vel bg ba gravity (36 doubles), covariance 24 x 24
    double x[S2M_STATE_DOUBLES] = {0};
    x[0] = x[4] = x[8] = 1.0;
    x[12] = x[16] = x[20] = 1.0;
    x[9] = truth[0]; x[10] = truth[1]; x[11] = truth[2];
    std::vector<double> P0(S2M_DIM * S2M_DIM, 0.0);
    for (int i = 0; i < S2M_DIM; ++i) P0[i * S2M_DIM + i] = i < 6 ? 1e-3 : 1e-4;

    // frame 0 seeds the map with its world-frame points.  A dense first scan (128 x 2048): with a single
    // sparse scan as the whole map the 5 nearest neighbours of a point lie along one scan ring, the plane
    // through them is ill-defined, and every registration (the reference's included) picks up a bias
    std::vector<float> scan = make_scan(truth, 128, 2048, rng);
    std::vector<float> seed(scan);
    for (size_t i = 0; i < seed.size(); i += 3) { seed[i] += (float)truth[0]; seed[i + 1] += (float)truth[1]; seed[i + 2] += (float)truth[2]; }
    CK(s2m_map_build(eng, seed.data(), 3, (int64_t)seed.size() / 3, 0));

    double worst = 0.0;
    double prev[3] = {x[9], x[10], x[11]};
    for (int f = 1; f <= frames; ++f) {
        truth[0] += 0.05; truth[1] += 0.02;                      // the sensor moves 5.4 cm per frame
        scan = make_scan(truth, 32, 512, rng);
        int64_t n_down = 0;
        CK(s2m_scan_set_downsampled(eng, scan.data(), 3, (int64_t)scan.size() / 3, 0.25f, 0, &n_down));
        // prediction: constant velocity from the last two estimates (stands in for the IMU propagation,
        // IMU_Processing.hpp:226-323); like there, the prediction is both the prior and the starting point
        double xp[S2M_STATE_DOUBLES];
        for (int i = 0; i < S2M_STATE_DOUBLES; ++i) xp[i] = x[i];
        for (int k = 0; k < 3; ++k) {
            xp[9 + k] = x[9 + k] + (f > 1 ? x[9 + k] - prev[k] : 0.0);
            prev[k] = x[9 + k];
            x[9 + k] = xp[9 + k];
        }
        std::vector<double> P(P0);                                 // re-inflated: stands in for process noise
        s2m_iter_log log;
        CK(s2m_iterated_update(eng, x, xp, P.data(), &log));
        int64_t n_add = 0, n_nodown = 0, m = 0;
        if (!log.ekf_stop) CK(s2m_map_incremental(eng, x, 0.25, 1, &n_add, &n_nodown));  // laserMapping.cpp:1165
        CK(s2m_map_size(eng, &m));
        const double err = std::sqrt((x[9] - truth[0]) * (x[9] - truth[0]) + (x[10] - truth[1]) * (x[10] - truth[1]) +
                                     (x[11] - truth[2]) * (x[11] - truth[2]));
        worst = std::fmax(worst, err);
        std::printf("frame %2d: scan %lld pts, iters %d (rematch %d), effective %d, pos err %.4f m, map %lld (+%lld)\n", f,
                    (long long)n_down, log.iters, log.rematch_passes, log.effct[log.iters - 1], err, (long long)m,
                    (long long)(n_add + n_nodown));
        if (log.ekf_stop) { std::fprintf(stderr, "degenerate scan\n"); return 1; }
    }
    s2m_destroy(eng);
    std::printf("worst position error %.4f m\n", worst);
    return worst < 0.03 ? 0 : 1;
}
