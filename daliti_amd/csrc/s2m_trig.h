// s2m_trig.h -- sin, cos and acos in double for the device-resident loop (s2m_loop.h).
//
// The device library's sin / cos carry a large-argument reduction that costs ~100 vector registers; inlined into the
// last workgroup of the reduce kernel they would halve that kernel's occupancy for three calls per iteration.  These are
// the classic fdlibm forms (Sun Microsystems, "freely granted" licence: k_sin.c, k_cos.c, e_acos.c) restated for the
// arguments the loop has -- rotation increments and angles, |x| well below 1e5: Cody-Waite reduction by pi/2 in two
// pieces, the minimax kernels on [-pi/4, pi/4], the rational kernel of acos.  __host__ __device__: tests/trig_check.cpp
// compares them with the C library on the CPU (agreement within 2 ulp over the ranges used).
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define S2M_HD __host__ __device__
#else
#define S2M_HD
#endif

namespace s2m {

S2M_HD inline double trig_ksin(double x)  // |x| <= pi/4
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = x * x;
    const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return x + (z * x) * (S1 + z * r);
}
S2M_HD inline double trig_kcos(double x)  // |x| <= pi/4
{
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = x * x;
    const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + z * r);
}
// x = n * pi/2 + r, |r| <= pi/4 (+ a little); returns n mod 4
S2M_HD inline int trig_reduce(double x, double &r)
{
    const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11;
    const double fn = floor(x * invpio2 + 0.5);
    r = (x - fn * pio2_1) - fn * pio2_1t;
    return (int)((long long)fn & 3);
}
S2M_HD inline double trig_sin(double x)
{
    double r;
    const int n = trig_reduce(x, r);
    const double s = trig_ksin(r), c = trig_kcos(r);
    return n == 0 ? s : (n == 1 ? c : (n == 2 ? -s : -c));
}
S2M_HD inline double trig_cos(double x)
{
    double r;
    const int n = trig_reduce(x, r);
    const double s = trig_ksin(r), c = trig_kcos(r);
    return n == 0 ? c : (n == 1 ? -s : (n == 2 ? -c : s));
}
S2M_HD inline double trig_acos(double x)  // |x| <= 1
{
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17, pi = 3.14159265358979311600e+00;
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
                 pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                 qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
                 qS4 = 7.70381505559019352791e-02;
    auto R = [&](double z) {
        const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        return p / q;
    };
    if (x >= 1.0) return 0.0;
    if (x <= -1.0) return pi;
    if (fabs(x) < 0.5) return pio2_hi - (x - (pio2_lo - x * R(x * x)));
    if (x < 0.0) {
        const double z = (1.0 + x) * 0.5, s = sqrt(z);
        const double w = R(z) * s - pio2_lo;
        return pi - 2.0 * (s + w);
    }
    const double z = (1.0 - x) * 0.5, s = sqrt(z);
    // s split into a head with 26 significant bits and the rest, so that df * df is exact (the compensated form of e_acos.c)
    const double df = (double)(float)s;
    const double c = (z - df * df) / (s + df);
    const double w = R(z) * s + c;
    return 2.0 * (df + w);
}

}  // namespace s2m
