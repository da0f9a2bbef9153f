// s2m_eskf.cpp -- see s2m_eskf.h.
#include "s2m_eskf.h"

#include <cmath>
#include <cstring>
#include <utility>

namespace s2m {
// Wider vectors for the 24-wide rows where the CPU has them (resolved once at load time).  Same operations
// in the same order per element -- the file is compiled with -ffp-contract=off, so no clone fuses a multiply
// with an add -- hence bit-identical results on every clone.
#define S2M_CPU_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))

namespace {

struct M3 {
    double a[9];
    double operator()(int r, int c) const { return a[r * 3 + c]; }
    double &operator()(int r, int c) { return a[r * 3 + c]; }
};

M3 load3(const double *p) { M3 m; std::memcpy(m.a, p, sizeof(m.a)); return m; }
void store3(const M3 &m, double *p) { std::memcpy(p, m.a, sizeof(m.a)); }

M3 mul(const M3 &A, const M3 &B)
{
    M3 C;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) C(r, c) = A(r, 0) * B(0, c) + A(r, 1) * B(1, c) + A(r, 2) * B(2, c);
    return C;
}
M3 transposed(const M3 &A)
{
    M3 T;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) T(r, c) = A(c, r);
    return T;
}

struct Lu24 {
    Mat24 lu;
    int perm[kDim];
};
// LU with partial pivoting (what Eigen's fixed-size inverse uses for n > 4): factor once, then solve for
// the unit columns that are needed.  Both routines keep, for every matrix element, exactly the sequence
// of multiply-then-subtract operations of the textbook scalar algorithm; they are only arranged so that
// independent elements sit in the innermost loop (rows of the trailing block; the right-hand-side columns
// of the triangular solves) -- that breaks the dependent-add latency chain and lets the compiler
// vectorise without changing a single rounding.
S2M_CPU_CLONES bool lu_factor(const Mat24 &in, Lu24 &f)
{
    constexpr int N = kDim;
    f.lu = in;
    double *lu = f.lu.data();
    for (int i = 0; i < N; ++i) f.perm[i] = i;
    for (int k = 0; k < N; ++k) {
        int piv = k;
        double mag = std::fabs(lu[k * N + k]);
        for (int r = k + 1; r < N; ++r) {
            const double v = std::fabs(lu[r * N + k]);
            if (v > mag) { mag = v; piv = r; }
        }
        if (!(mag > 0.0)) return false;
        if (piv != k) {
            for (int c = 0; c < N; ++c) std::swap(lu[k * N + c], lu[piv * N + c]);
            std::swap(f.perm[k], f.perm[piv]);
        }
        const double d = lu[k * N + k];
        const double *__restrict__ rk = lu + k * N;
        for (int r = k + 1; r < N; ++r) {
            double *__restrict__ rr = lu + r * N;
            const double m = rr[k] / d;
            rr[k] = m;
            for (int c = k + 1; c < N; ++c) rr[c] -= m * rk[c];
        }
    }
    return true;
}
// first NC columns of the inverse, row-major N x NC
template <int NC>
inline __attribute__((always_inline)) void lu_inverse_cols_impl(const Lu24 &f, double *__restrict__ out)
{
    constexpr int N = kDim;
    const double *lu = f.lu.data();
    double y[N][NC];
    for (int r = 0; r < N; ++r) {  // L y = P e_col, all columns side by side
        double acc[NC];
        for (int col = 0; col < NC; ++col) acc[col] = (f.perm[r] == col) ? 1.0 : 0.0;
        for (int c = 0; c < r; ++c) {
            const double l = lu[r * N + c];
            for (int col = 0; col < NC; ++col) acc[col] -= l * y[c][col];
        }
        for (int col = 0; col < NC; ++col) y[r][col] = acc[col];
    }
    for (int r = N - 1; r >= 0; --r) {  // U x = y
        double acc[NC];
        for (int col = 0; col < NC; ++col) acc[col] = y[r][col];
        for (int c = r + 1; c < N; ++c) {
            const double u = lu[r * N + c];
            for (int col = 0; col < NC; ++col) acc[col] -= u * out[c * NC + col];
        }
        const double d = lu[r * N + r];
        for (int col = 0; col < NC; ++col) out[r * NC + col] = acc[col] / d;
    }
}

// (multiversioning does not apply to templates: two cloned wrappers, the body inlined into each clone)
S2M_CPU_CLONES void lu_inverse_24(const Lu24 &f, double *__restrict__ out) { lu_inverse_cols_impl<kDim>(f, out); }
S2M_CPU_CLONES void lu_inverse_12(const Lu24 &f, double *__restrict__ out) { lu_inverse_cols_impl<12>(f, out); }

}  // namespace

void so3_exp(double v1, double v2, double v3, double R[9])
{
    // so3_math.h:55-72: identity unless |v| > 1e-5, Rodrigues on the unit axis
    const double norm = std::sqrt(v1 * v1 + v2 * v2 + v3 * v3);
    M3 E{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
    if (norm > 0.00001) {
        const double r[3] = {v1 / norm, v2 / norm, v3 / norm};
        const M3 K{{0.0, -r[2], r[1], r[2], 0.0, -r[0], -r[1], r[0], 0.0}};
        const double s = std::sin(norm), c1 = 1.0 - std::cos(norm);
        M3 cK;
        for (int i = 0; i < 9; ++i) cK.a[i] = c1 * K.a[i];
        const M3 cKK = mul(cK, K);
        for (int i = 0; i < 9; ++i) E.a[i] = (E.a[i] + s * K.a[i]) + cKK.a[i];
    }
    store3(E, R);
}

void so3_log(const double R[9], double out[3])
{
    // so3_math.h:76-81
    const double tr = R[0] + R[4] + R[8];
    const double theta = (tr > 3.0 - 1e-6) ? 0.0 : std::acos(0.5 * (tr - 1));
    const double K[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
    const double f = (std::fabs(theta) < 0.001) ? 0.5 : (0.5 * theta / std::sin(theta));
    for (int i = 0; i < 3; ++i) out[i] = f * K[i];
}

void boxplus(State &x, const Vec24 &d)
{
    double E[9];
    so3_exp(d[0], d[1], d[2], E);
    store3(mul(load3(x.rot), load3(E)), x.rot);
    so3_exp(d[6], d[7], d[8], E);
    store3(mul(load3(x.R_LI), load3(E)), x.R_LI);
    for (int i = 0; i < 3; ++i) {
        x.pos[i] += d[3 + i];
        x.T_LI[i] += d[9 + i];
        x.vel[i] += d[12 + i];
        x.bg[i] += d[15 + i];
        x.ba[i] += d[18 + i];
        x.grav[i] += d[21 + i];
    }
}

Vec24 boxminus(const State &a, const State &b)
{
    Vec24 o{};
    M3 rd = mul(transposed(load3(b.rot)), load3(a.rot));
    so3_log(rd.a, &o[0]);
    rd = mul(transposed(load3(b.R_LI)), load3(a.R_LI));
    so3_log(rd.a, &o[6]);
    for (int i = 0; i < 3; ++i) {
        o[3 + i] = a.pos[i] - b.pos[i];
        o[9 + i] = a.T_LI[i] - b.T_LI[i];
        o[12 + i] = a.vel[i] - b.vel[i];
        o[15 + i] = a.bg[i] - b.bg[i];
        o[18 + i] = a.ba[i] - b.ba[i];
        o[21 + i] = a.grav[i] - b.grav[i];
    }
    return o;
}

bool eskf_prepare(const EskfParams &p, const Mat24 &P, EskfWork &work)
{
    constexpr int N = kDim;
    if (work.pinv_valid && work.Rkey == p.laser_point_cov &&
        std::memcmp(work.Pkey.data(), P.data(), sizeof(double) * N * N) == 0)
        return true;
    Mat24 S;
    for (int i = 0; i < N * N; ++i) S[i] = P[i] / p.laser_point_cov;
    Lu24 f;
    work.pinv_valid = false;
    if (!lu_factor(S, f)) return false;                            // (state.cov / LASER_POINT_COV).inverse()
    lu_inverse_24(f, work.Pinv.data());
    work.Pkey = P;
    work.Rkey = p.laser_point_cov;
    work.pinv_valid = true;
    return true;
}

bool eskf_update(const EskfParams &p, State &x, const State &x_prop, const Mat24 &P, const double HtH[144],
                 const double Htz[12], Vec24 &solution, bool &converged, EskfWork &work)
{
    constexpr int N = kDim;
    work.valid = false;
    converged = false;
    if (!eskf_prepare(p, P, work)) return false;
    Mat24 A = work.Pinv;
    for (int r = 0; r < 12; ++r)
        for (int c = 0; c < 12; ++c) A[r * N + c] += HtH[r * 12 + c];    // + H_T_H (12x12 block)
    Lu24 f;
    if (!lu_factor(A, f)) return false;                            // K_1 (:1017-1018); only K_1[:, :12] is used
    lu_inverse_12(f, work.K1c.data());
    std::memcpy(work.HtH.data(), HtH, sizeof(double) * 144);

    const Vec24 vec = boxminus(x_prop, x);                         // :1028
    for (int r = 0; r < N; ++r) {
        double kz = 0.0;
        for (int a = 0; a < 12; ++a) kz += work.K1c[r * 12 + a] * Htz[a];
        double khv = 0.0;
        for (int b = 0; b < 12; ++b) {
            double kh = 0.0;
            for (int a = 0; a < 12; ++a) kh += work.K1c[r * 12 + a] * HtH[a * 12 + b];
            khv += kh * vec[b];
        }
        solution[r] = (kz + vec[r]) - khv;                         // :1032
    }
    boxplus(x, solution);                                          // :1033
    const double rn = std::sqrt(solution[0] * solution[0] + solution[1] * solution[1] + solution[2] * solution[2]);
    const double tn = std::sqrt(solution[3] * solution[3] + solution[4] * solution[4] + solution[5] * solution[5]);
    converged = (rn * 57.3 < p.conv_rot_deg) && (tn * 100 < p.conv_pos_cm);  // :1040
    work.valid = true;
    return true;
}

namespace {
// Cholesky factor of the symmetric positive definite n x n matrix M (row-major, n <= 12) in place, lower triangle
bool chol_factor(double *M, int n)
{
    for (int j = 0; j < n; ++j) {
        double d = M[j * n + j];
        for (int k = 0; k < j; ++k) d -= M[j * n + k] * M[j * n + k];
        if (!(d > 0.0)) return false;
        d = std::sqrt(d);
        M[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double v = M[i * n + j];
            for (int k = 0; k < j; ++k) v -= M[i * n + k] * M[j * n + k];
            M[i * n + j] = v / d;
        }
    }
    return true;
}
// x <- (L L^T)^-1 x
void chol_solve(const double *L, int n, double *x)
{
    for (int i = 0; i < n; ++i) {
        double v = x[i];
        for (int k = 0; k < i; ++k) v -= L[i * n + k] * x[k];
        x[i] = v / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double v = x[i];
        for (int k = i + 1; k < n; ++k) v -= L[k * n + i] * x[k];
        x[i] = v / L[i * n + i];
    }
}
}  // namespace

bool loop_prepare(double laser_point_cov, const double *P, int nc, double *G, double *Cinv)
{
    constexpr int N = kDim;
    double L[144];
    for (int r = 0; r < nc; ++r)
        for (int c = 0; c < nc; ++c) L[r * nc + c] = P[r * N + c] / laser_point_cov;
    if (!chol_factor(L, nc)) return false;
    for (int c = 0; c < nc; ++c) {  // C^-1 column by column (symmetric: stored as is)
        double e[12];
        for (int r = 0; r < nc; ++r) e[r] = r == c ? 1.0 : 0.0;
        chol_solve(L, nc, e);
        for (int r = 0; r < nc; ++r) Cinv[r * nc + c] = e[r];
    }
    for (int r = 0; r < nc; ++r)    // exactly symmetric (the device adds it to the symmetric H^T H and factors the sum)
        for (int c = r + 1; c < nc; ++c) Cinv[r * nc + c] = Cinv[c * nc + r] = 0.5 * (Cinv[r * nc + c] + Cinv[c * nc + r]);
    for (int r = 0; r < N; ++r)
        for (int c = 0; c < nc; ++c) {
            double s = 0.0;
            for (int k = 0; k < nc; ++k) s += (P[r * N + k] / laser_point_cov) * Cinv[k * nc + c];
            G[r * nc + c] = s;
        }
    return true;
}

bool loop_cov_update(const double *G, const double *Cinv, const double *HtH, int nc, double *P)
{
    constexpr int N = kDim;
    double M[144];
    for (int r = 0; r < nc; ++r)
        for (int c = 0; c < nc; ++c) M[r * nc + c] = Cinv[r * nc + c] + HtH[r * 12 + c];
    if (!chol_factor(M, nc)) return false;
    double Y[12 * N];  // M^-1 (A P[0:nc, :]), column by column
    for (int c = 0; c < N; ++c) {
        double y[12];
        for (int r = 0; r < nc; ++r) {
            double s = 0.0;
            for (int k = 0; k < nc; ++k) s += HtH[r * 12 + k] * P[k * N + c];
            y[r] = s;
        }
        chol_solve(M, nc, y);
        for (int r = 0; r < nc; ++r) Y[r * N + c] = y[r];
    }
    for (int r = 0; r < N; ++r)
        for (int c = 0; c < N; ++c) {
            double s = 0.0;
            for (int k = 0; k < nc; ++k) s += G[r * nc + k] * Y[k * N + c];
            P[r * N + c] -= s;
        }
    return true;
}

S2M_CPU_CLONES void cov_update(const EskfWork &work, Mat24 &P)
{
    constexpr int N = kDim;
    Mat24 ImG{};
    for (int r = 0; r < N; ++r) {
        for (int c = 0; c < N; ++c) ImG[r * N + c] = (r == c) ? 1.0 : 0.0;
        for (int b = 0; b < 12; ++b) {
            double kh = 0.0;
            for (int a = 0; a < 12; ++a) kh += work.K1c[r * 12 + a] * work.HtH[a * 12 + b];
            ImG[r * N + b] -= kh;
        }
    }
    Mat24 Pn;
    for (int r = 0; r < N; ++r)
        for (int c = 0; c < N; ++c) {
            double s = 0.0;
            for (int k = 0; k < N; ++k) s += ImG[r * N + k] * P[k * N + c];
            Pn[r * N + c] = s;
        }
    P = Pn;
}

}  // namespace s2m
