// s2m_point.h -- the per-point arithmetic of one residual pass after the plane is known: residual, the two gates and
// the Jacobian row (eskf_lio/src/laserMapping.cpp:866-889, 948-978).  __host__ __device__: reduce_kernel runs it per
// lane, tests/point_check.cpp runs the same code on the CPU against the oracle.
#pragma once
#include <cmath>

#include "s2m_device.h"

namespace s2m {

// pd2 = n . p_w + d for a point whose plane passed the fit; keep = the s-gate re-selects the point (:868-873),
// eff = it also passes the residual gate and contributes a row (:889)
__host__ __device__ __forceinline__ float point_residual(const Pose &pose, const Gates &gates, float bx, float by, float bz,
                                                         const float4 &pl, bool &keep, bool &eff)
{
    float wx, wy, wz;
    body_to_world(pose, bx, by, bz, wx, wy, wz);
    const float pd2 = ((pl.x * wx + pl.y * wy) + pl.z * wz) + pl.w;                    // :866
    const double pbn = sqrt(((double)bx * (double)bx + (double)by * (double)by) + (double)bz * (double)bz);
    // "float s" (:868): the double expression is rounded to float before the compare of :870
    const float s = (float)(1 - 0.9 * fabs((double)pd2) / sqrt(pbn));
    keep = (double)s > gates.s_gate;
    eff = keep && fabs((double)pd2) <= gates.res_gate;                                  // :889
    return pd2;
}

// one Jacobian row (laserMapping.cpp:948-978): h = [A, n, B, C] or [A, n, 0, 0], z = -pd2
template <bool EXT>
__host__ __device__ __forceinline__ void jac_row(const Pose &P, float bx, float by, float bz, const float4 &pl, float pd2,
                                        double (&h)[12], double &z)
{
    const double p0 = (double)bx, p1 = (double)by, p2 = (double)bz;
    const double i0 = ((P.RLI[0] * p0 + P.RLI[1] * p1) + P.RLI[2] * p2) + P.TLI[0];
    const double i1 = ((P.RLI[3] * p0 + P.RLI[4] * p1) + P.RLI[5] * p2) + P.TLI[1];
    const double i2 = ((P.RLI[6] * p0 + P.RLI[7] * p1) + P.RLI[8] * p2) + P.TLI[2];
    const double n0 = (double)pl.x, n1 = (double)pl.y, n2 = (double)pl.z;
    // C = rot_end^T * n
    const double c0 = (P.R[0] * n0 + P.R[3] * n1) + P.R[6] * n2;
    const double c1 = (P.R[1] * n0 + P.R[4] * n1) + P.R[7] * n2;
    const double c2 = (P.R[2] * n0 + P.R[5] * n1) + P.R[8] * n2;
    // A = [p_I]x * C
    h[0] = (0.0 * c0 + -i2 * c1) + i1 * c2;
    h[1] = (i2 * c0 + 0.0 * c1) + -i0 * c2;
    h[2] = (-i1 * c0 + i0 * c1) + 0.0 * c2;
    h[3] = n0; h[4] = n1; h[5] = n2;
    if (EXT) {
        // B = ([p_b]x * R_L_I^T) * C, left to right (laserMapping.cpp:970)
        const double S[9] = {0.0, -p2, p1, p2, 0.0, -p0, -p1, p0, 0.0};
        double M[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                M[i * 3 + j] = (S[i * 3 + 0] * P.RLI[j * 3 + 0] + S[i * 3 + 1] * P.RLI[j * 3 + 1]) +
                               S[i * 3 + 2] * P.RLI[j * 3 + 2];
        h[6] = (M[0] * c0 + M[1] * c1) + M[2] * c2;
        h[7] = (M[3] * c0 + M[4] * c1) + M[5] * c2;
        h[8] = (M[6] * c0 + M[7] * c1) + M[8] * c2;
        h[9] = c0; h[10] = c1; h[11] = c2;
    } else {
#pragma unroll
        for (int i = 6; i < 12; ++i) h[i] = 0.0;
    }
    z = -(double)pd2;
}

}  // namespace s2m
