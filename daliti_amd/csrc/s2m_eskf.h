// s2m_eskf.h -- host side of the iterated error-state Kalman update of eskf_lio, fp64.
//
// Mirrors the reference's StatesGroup and the update lines of laserMapping.cpp:
//   StatesGroup [+]= delta, a [-] b          eskf_lio/include/common_lib.h:146-157, 173-187
//   Exp / Log                                 eskf_lio/include/so3_math.h:55-72, 76-81
//   K_1, solution, convergence test           eskf_lio/src/laserMapping.cpp:1012-1046
//   covariance update                         eskf_lio/src/laserMapping.cpp:1084-1085
// The 24x24 algebra is a few microseconds of fp64 on the host and is evaluated from the 12x12
// normal block the GPU reduces (K*z = K_1[:, :12] * H^T z, K*H = K_1[:, :12] * H^T H), so the
// m x 12 Jacobian and the 24 x m gain of the reference are never materialised.
#pragma once
#include <array>
#include <cstdint>

namespace s2m {

constexpr int kDim = 24;

// same member order as the reference's StatesGroup (common_lib.h:219-226), row-major matrices
struct State {
    double rot[9], pos[3], R_LI[9], T_LI[3], vel[3], bg[3], ba[3], grav[3];
};
static_assert(sizeof(State) == 36 * sizeof(double), "State must be 36 packed doubles");

using Vec24 = std::array<double, kDim>;
using Mat24 = std::array<double, kDim * kDim>;

void so3_exp(double v1, double v2, double v3, double R[9]);
void so3_log(const double R[9], double out[3]);
void boxplus(State &x, const Vec24 &d);
Vec24 boxminus(const State &a, const State &b);  // a [-] b

struct EskfParams {
    double laser_point_cov = 0.0015;
    double conv_rot_deg = 0.01;
    double conv_pos_cm = 0.015;
};

struct EskfWork {
    std::array<double, kDim * 12> K1c{};  // first 12 columns of K_1 = (H_T_H + (P/R)^-1)^-1, row-major 24x12
    std::array<double, 144> HtH{};        // normal block of the last update
    bool valid = false;
    // (P/R)^-1 is constant while the covariance is (it only changes at the exit of the iterated
    // update, laserMapping.cpp:1085), so it is inverted once per distinct P
    Mat24 Pkey{}, Pinv{};
    double Rkey = 0.0;
    bool pinv_valid = false;
};

// (P/R)^-1 for the given covariance into work (a no-op when work already holds it).  The covariance only changes at
// the exit of the iterated update, so the engine calls this once per scan while the first pass runs on the GPU.
bool eskf_prepare(const EskfParams &p, const Mat24 &P, EskfWork &work);
// Returns false when a matrix is singular.  x is updated in place.
bool eskf_update(const EskfParams &p, State &x, const State &x_prop, const Mat24 &P,
                 const double HtH[144], const double Htz[12], Vec24 &solution, bool &converged,
                 EskfWork &work);
// P <- (I - K_1[:, :12] * HtH (+) 0) * P
void cov_update(const EskfWork &work, Mat24 &P);

// ---- the same update in the form the device-resident loop uses (s2m_loop.h): with P' = P / R, C = P'[0:nc, 0:nc],
// K_1[:, 0:nc] = G M^-1, G = P'[:, 0:nc] C^-1 (24 x nc, row-major), M = C^-1 + H^T H.
// G and C^-1 (nc x nc) for the scan; false when C is not positive definite (the caller falls back to the LU form)
bool loop_prepare(double laser_point_cov, const double *P, int nc, double *G, double *Cinv);
// P <- P - G M^-1 (A P[0:nc, :]) with A = HtH (12 x 12 layout); false when M is not positive definite
bool loop_cov_update(const double *G, const double *Cinv, const double *HtH, int nc, double *P);

}  // namespace s2m
