// Test helper: the algebra of the device-resident loop (daliti_amd/csrc/s2m_loop.h) on the CPU -- G and C^-1 from
// loop_prepare, the nc x nc system by elimination without pivoting exactly as the device wave does it, solution = vec + G w,
// the covariance through loop_cov_update -- against cases written by tests/test_host_logic.py from the oracle's literal
// two-inverse form (laserMapping.cpp:1017-1032, 1084-1085); and the loop's own sin / cos / acos (s2m_trig.h) against the C
// library.  Built with -fsanitize=address,undefined.  Per case (doubles): x[36] x_prop[36] P[576] HtH[144] Htz[12] |
// expected x[36] solution[24] converged P[576].  Only cases whose normal block is confined to the first nc columns apply.
// usage: loop_algebra_check <cases file> <n cases> <nc>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "s2m_eskf.h"
#include "s2m_trig.h"

using namespace s2m;

int main(int argc, char **argv)
{
    if (argc < 4) return 64;
    const int n = std::atoi(argv[2]), nc = std::atoi(argv[3]);
    const size_t in_d = 36 + 36 + 576 + 144 + 12, out_d = 36 + 24 + 1 + 576;
    std::vector<double> buf((in_d + out_d) * (size_t)n);
    FILE *f = std::fopen(argv[1], "rb");
    if (!f || std::fread(buf.data(), sizeof(double), buf.size(), f) != buf.size()) return 66;
    std::fclose(f);
    double worst_s = 0, worst_p = 0;
    int failed = 0;
    for (int k = 0; k < n; ++k) {
        const double *c = buf.data() + (in_d + out_d) * (size_t)k;
        State x, xp;
        std::memcpy(&x, c, sizeof(State));
        std::memcpy(&xp, c + 36, sizeof(State));
        std::vector<double> P(c + 72, c + 72 + 576);
        const double *HtH = c + 72 + 576, *Htz = HtH + 144, *es = Htz + 12 + 36, *ep = es + 24 + 1;
        double G[24 * 12], Cinv[144];
        if (!loop_prepare(0.0015, P.data(), nc, G, Cinv)) { ++failed; continue; }
        const Vec24 vec = boxminus(xp, x);
        // M = C^-1 + A, b = H^T z - A vec: forward elimination with one reciprocal per pivot, back substitution (s2m_loop.h)
        double M[144], b[12], rp[12], w[12];
        for (int i = 0; i < nc; ++i) {
            double acc = 0.0;
            for (int j = 0; j < nc; ++j) { M[i * nc + j] = Cinv[i * nc + j] + HtH[i * 12 + j]; acc += HtH[i * 12 + j] * vec[j]; }
            b[i] = Htz[i] - acc;
        }
        bool ok = true;
        for (int p = 0; p < nc; ++p) {
            ok = ok && M[p * nc + p] > 0.0;
            rp[p] = 1.0 / M[p * nc + p];
            for (int i = p + 1; i < nc; ++i) {
                const double fct = M[i * nc + p] * rp[p];
                for (int j = p + 1; j < nc; ++j) M[i * nc + j] -= fct * M[p * nc + j];
                b[i] -= fct * b[p];
            }
        }
        if (!ok) { ++failed; continue; }
        for (int p = nc - 1; p >= 0; --p) {
            w[p] = b[p] * rp[p];
            for (int i = 0; i < p; ++i) b[i] -= M[i * nc + p] * w[p];
        }
        double sscale = 1e-30;
        for (int i = 0; i < 24; ++i) sscale = std::fmax(sscale, std::fabs(es[i]));
        for (int r = 0; r < 24; ++r) {
            double acc = 0.0;
            for (int j = 0; j < nc; ++j) acc += G[r * nc + j] * w[j];
            worst_s = std::fmax(worst_s, std::fabs(vec[r] + acc - es[r]) / sscale);
        }
        if (!loop_cov_update(G, Cinv, HtH, nc, P.data())) { ++failed; continue; }
        double pscale = 1e-30;
        for (int i = 0; i < 576; ++i) pscale = std::fmax(pscale, std::fabs(ep[i]));
        for (int i = 0; i < 576; ++i) worst_p = std::fmax(worst_p, std::fabs(P[i] - ep[i]) / pscale);
    }
    // the loop's trigonometry against the C library over the ranges it is used on
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(0, 1);
    double ts = 0, tc = 0, ta = 0;
    for (int i = 0; i < 400000; ++i) {
        const double a = (i % 3 == 0) ? u(g) * 0.1 : ((i % 3 == 1) ? u(g) * 6.5 : (u(g) - 0.5) * 2000);
        ts = std::fmax(ts, std::fabs(trig_sin(a) - std::sin(a)));
        tc = std::fmax(tc, std::fabs(trig_cos(a) - std::cos(a)));
        const double cc = (i % 2) ? 1.0 - u(g) * 1e-3 : (u(g) * 2 - 1);
        ta = std::fmax(ta, std::fabs(trig_acos(cc) - std::acos(cc)));
    }
    std::printf("cases %d failed %d worst_solution_rel %.3e worst_P_rel %.3e trig_sin %.3e trig_cos %.3e trig_acos %.3e\n", n, failed,
                worst_s, worst_p, ts, tc, ta);
    return 0;
}
