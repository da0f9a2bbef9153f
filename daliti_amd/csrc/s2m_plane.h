// s2m_plane.h -- esti_plane<float> (eskf_lio/include/common_lib.h:267-299) as a device function.
//
// Solves A x = -1 (A = the 5 neighbour coordinates) in the least-squares sense with a column-pivoted
// Householder QR in float -- the algorithm behind Eigen's colPivHouseholderQr().solve (common_lib.h:283):
// pivot on the largest remaining column norm with LAPACK-style norm down-dating, Householder
// reflectors, Eigen's rank threshold -- then n = x/|x|, d = 1/|x| and the 5-point inlier check.
// Every expression is evaluated in the order of orc_esti_plane (oracle/s2m_oracle.c) and the file is
// compiled with -ffp-contract=off, so the result is bit-identical to the oracle's.
#pragma once
#include <cfloat>
#include <cmath>

#include "s2m_device.h"

namespace s2m {

// esti_plane<float>: column-pivoted Householder QR least squares of A x = -1, same operation
// order as orc_esti_plane (oracle/s2m_oracle.c).  Returns the inlier verdict.  (__host__ too: tests/plane_check.cpp
// runs it on the CPU against the oracle.)
__host__ __device__ __forceinline__ bool fit_plane(const float (&nx)[kK], const float (&ny)[kK], const float (&nz)[kK], float thr,
                                          float4 &pl)
{
    float A[kK][3], c[kK];
    float tau[3], nu[3], nd[3];
    int trans[3];
#pragma unroll
    for (int i = 0; i < kK; ++i) {
        A[i][0] = nx[i]; A[i][1] = ny[i]; A[i][2] = nz[i];
        c[i] = -1.0f;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < kK; ++i) s = s + A[i][k] * A[i][k];
        nd[k] = __builtin_sqrtf(s);
        nu[k] = nd[k];
    }
    float nmax = nu[0];
    if (nu[1] > nmax) nmax = nu[1];
    if (nu[2] > nmax) nmax = nu[2];
    const float th = nmax * FLT_EPSILON;
    const float threshold_helper = (th * th) / (float)kK;
    const float downdate_thr = __builtin_sqrtf(FLT_EPSILON);
    int nonzero = 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int big = k;
        float bigv = nu[k];
#pragma unroll
        for (int j = k + 1; j < 3; ++j)
            if (nu[j] > bigv) { big = j; bigv = nu[j]; }
        const float big_sq = bigv * bigv;
        if (nonzero == 3 && big_sq < threshold_helper * (float)(kK - k)) nonzero = k;
        trans[k] = big;
        // column swap k <-> big with static indices (big is k, k+1 or 2)
#pragma unroll
        for (int j = k + 1; j < 3; ++j) {
            const bool sw = (big == j);
#pragma unroll
            for (int i = 0; i < kK; ++i) {
                const float a = A[i][k], b = A[i][j];
                A[i][k] = sw ? b : a;
                A[i][j] = sw ? a : b;
            }
            const float u0 = nu[k], u1 = nu[j], d0 = nd[k], d1 = nd[j];
            nu[k] = sw ? u1 : u0; nu[j] = sw ? u0 : u1;
            nd[k] = sw ? d1 : d0; nd[j] = sw ? d0 : d1;
        }
        float tail = 0.0f;
#pragma unroll
        for (int i = k + 1; i < kK; ++i) tail = tail + A[i][k] * A[i][k];
        const float c0 = A[k][k];
        float beta;
        if (tail <= FLT_MIN) {
            tau[k] = 0.0f;
            beta = c0;
#pragma unroll
            for (int i = k + 1; i < kK; ++i) A[i][k] = 0.0f;
        } else {
            beta = __builtin_sqrtf(c0 * c0 + tail);
            if (c0 >= 0.0f) beta = -beta;
            const float den = c0 - beta;
#pragma unroll
            for (int i = k + 1; i < kK; ++i) A[i][k] = A[i][k] / den;
            tau[k] = (beta - c0) / beta;
        }
        A[k][k] = beta;
        if (tau[k] != 0.0f) {
#pragma unroll
            for (int j = k + 1; j < 3; ++j) {
                float tmp = 0.0f;
#pragma unroll
                for (int i = k + 1; i < kK; ++i) tmp = tmp + A[i][k] * A[i][j];
                tmp = tmp + A[k][j];
                A[k][j] = A[k][j] - tau[k] * tmp;
#pragma unroll
                for (int i = k + 1; i < kK; ++i) A[i][j] = A[i][j] - (tau[k] * A[i][k]) * tmp;
            }
        }
#pragma unroll
        for (int j = k + 1; j < 3; ++j) {
            if (nu[j] != 0.0f) {
                float t = fabsf(A[k][j]) / nu[j];
                t = (1.0f + t) * (1.0f - t);
                if (t < 0.0f) t = 0.0f;
                const float q = nu[j] / nd[j];
                const float t2 = t * (q * q);
                if (t2 <= downdate_thr) {
                    float s = 0.0f;
#pragma unroll
                    for (int i = k + 1; i < kK; ++i) s = s + A[i][j] * A[i][j];
                    nd[j] = __builtin_sqrtf(s);
                    nu[j] = nd[j];
                } else {
                    nu[j] = nu[j] * __builtin_sqrtf(t);
                }
            }
        }
    }
    // permutation = identity with the transpositions applied on the right (trans[k] >= k)
    int p0 = 0, p1 = 1, p2 = 2;
    if (trans[0] == 1) { const int t = p0; p0 = p1; p1 = t; }
    else if (trans[0] == 2) { const int t = p0; p0 = p2; p2 = t; }
    if (trans[1] == 2) { const int t = p1; p1 = p2; p2 = t; }
    const int perm[3] = {p0, p1, p2};
    float xs[3] = {0.0f, 0.0f, 0.0f};
    if (nonzero > 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (k < nonzero && tau[k] != 0.0f) {
                float tmp = 0.0f;
#pragma unroll
                for (int i = k + 1; i < kK; ++i) tmp = tmp + A[i][k] * c[i];
                tmp = tmp + c[k];
                c[k] = c[k] - tau[k] * tmp;
#pragma unroll
                for (int i = k + 1; i < kK; ++i) c[i] = c[i] - (tau[k] * A[i][k]) * tmp;
            }
        }
#pragma unroll
        for (int i = 2; i >= 0; --i) {
            if (i < nonzero) {
                c[i] = c[i] / A[i][i];
#pragma unroll
                for (int r = 0; r < i; ++r) c[r] = c[r] - c[i] * A[r][i];
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (i < nonzero) {
                if (perm[i] == 0) xs[0] = c[i];
                else if (perm[i] == 1) xs[1] = c[i];
                else xs[2] = c[i];
            }
        }
    }
    const float n = __builtin_sqrtf((xs[0] * xs[0] + xs[1] * xs[1]) + xs[2] * xs[2]);
    pl.x = xs[0] / n;
    pl.y = xs[1] / n;
    pl.z = xs[2] / n;
    pl.w = (float)(1.0 / (double)n);
    bool ok = true;
#pragma unroll
    for (int j = 0; j < kK; ++j) {
        const float v = ((pl.x * nx[j] + pl.y * ny[j]) + pl.z * nz[j]) + pl.w;
        if (fabsf(v) > thr) ok = false;
    }
    return ok;
}

}  // namespace s2m
