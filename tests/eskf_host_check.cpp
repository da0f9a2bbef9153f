// Test helper: the product's host-side ESKF algebra (daliti_amd/csrc/s2m_eskf.cpp, pure C++, no HIP) run on the CPU
// against cases written by tests/test_host_logic.py from the oracle.  Built with -fsanitize=address,undefined.
// File layout per case (doubles): x[36] x_prop[36] P[576] HtH[144] Htz[12] | expected x[36] solution[24] converged P[576]
// usage: eskf_host_check <cases file> <n cases>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "s2m_eskf.h"

using namespace s2m;

int main(int argc, char **argv)
{
    if (argc < 3) return 64;
    const int n = std::atoi(argv[2]);
    const size_t in_d = 36 + 36 + 576 + 144 + 12, out_d = 36 + 24 + 1 + 576;
    std::vector<double> buf((in_d + out_d) * (size_t)n);
    FILE *f = std::fopen(argv[1], "rb");
    if (!f || std::fread(buf.data(), sizeof(double), buf.size(), f) != buf.size()) return 66;
    std::fclose(f);
    double worst_x = 0, worst_s = 0, worst_p = 0;
    int conv_mismatch = 0, failed = 0;
    EskfWork work;  // one work area across the cases, like one engine across scans: the (P/R)^-1 cache must follow P
    for (int k = 0; k < n; ++k) {
        const double *c = buf.data() + (in_d + out_d) * (size_t)k;
        State x, xp;
        std::memcpy(&x, c, sizeof(State));
        std::memcpy(&xp, c + 36, sizeof(State));
        Mat24 P;
        std::memcpy(P.data(), c + 72, sizeof(double) * 576);
        const double *HtH = c + 72 + 576, *Htz = HtH + 144, *ex = Htz + 12, *es = ex + 36, *ec = es + 24, *ep = ec + 1;
        EskfParams prm;
        Vec24 sol;
        bool conv = false;
        if (k % 2 == 0) (void)eskf_prepare(prm, P, work);  // the engine's order on even cases, the lazy path on odd ones
        if (!eskf_update(prm, x, xp, P, HtH, Htz, sol, conv, work)) { ++failed; continue; }
        cov_update(work, P);
        const double *xo = reinterpret_cast<const double *>(&x);
        for (int i = 0; i < 36; ++i) worst_x = std::fmax(worst_x, std::fabs(xo[i] - ex[i]));
        double sscale = 1e-30;
        for (int i = 0; i < 24; ++i) sscale = std::fmax(sscale, std::fabs(es[i]));
        for (int i = 0; i < 24; ++i) worst_s = std::fmax(worst_s, std::fabs(sol[i] - es[i]) / sscale);
        double pscale = 1e-30;
        for (int i = 0; i < 576; ++i) pscale = std::fmax(pscale, std::fabs(ep[i]));
        for (int i = 0; i < 576; ++i) worst_p = std::fmax(worst_p, std::fabs(P[i] - ep[i]) / pscale);
        conv_mismatch += (conv ? 1.0 : 0.0) != *ec;
    }
    std::printf("cases %d failed %d conv_mismatch %d worst_x %.3e worst_solution_rel %.3e worst_P_rel %.3e\n", n, failed,
                conv_mismatch, worst_x, worst_s, worst_p);
    return 0;
}
