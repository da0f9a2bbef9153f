/*
 * s2m_oracle.c -- CPU restatement of the eskf_lio scan-to-map hot path (see s2m_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (no reference tests/fixtures exist, reference not
 * buildable in this image).  Never linked into or called from the product.
 *
 * Floating-point contract (what the HIP path reproduces bit-for-bit per point):
 *   - built with -ffp-contract=off, no -march flags: plain IEEE float/double mul/add, like the
 *     reference's x86-64 "-O3" build (eskf_lio/CMakeLists.txt:9);
 *   - sums are evaluated left to right in index order;
 *   - float division and sqrtf are correctly rounded.
 * Eigen's own evaluation order inside colPivHouseholderQr / GEMM is not knowable here (Eigen is
 * not installed); where it matters the sequential order below is the definition.
 */
#include "s2m_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void orc_cfg_default(orc_cfg *c)
{
    c->plane_thr = 0.1f;
    c->knn_d2_gate = 5.0f;
    c->s_gate = 0.9;
    c->res_gate = 2.0;
    c->laser_point_cov = 0.0015;
    c->conv_rot_deg = 0.01;
    c->conv_pos_cm = 0.015;
    c->extrinsic_est_en = 0;
    c->max_iter = 5;
    c->feat_threshold = 100;
    c->nthreads = 1;
}

/* ======================================================================================== */
/* small dense helpers (row-major)                                                          */
/* ======================================================================================== */
static void mat3_mul(const double A[9], const double B[9], double C[9])
{
    double T[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            T[i * 3 + j] = A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j] +
                           A[i * 3 + 2] * B[2 * 3 + j];
    memcpy(C, T, sizeof(T));
}
static void mat3_tmul(const double A[9], const double B[9], double C[9]) /* A^T * B */
{
    double T[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            T[i * 3 + j] = A[0 * 3 + i] * B[0 * 3 + j] + A[1 * 3 + i] * B[1 * 3 + j] +
                           A[2 * 3 + i] * B[2 * 3 + j];
    memcpy(C, T, sizeof(T));
}
/* ---- ASSUMED evaluation orders (parity unpinned) ------------------------------------------------------------
 * Three kinds of sums on the per-point path are Eigen fixed-size reductions whose order is decided inside Eigen
 * (3.3.7's Redux.h / ProductEvaluators.h), which is not installed here: the default below is left to right
 * ((a0 + a1) + a2); Eigen's non-vectorised complete unroller (redux_novec_unroller) is a halving tree
 * (a0 + (a1 + a2) for three terms, (a0 + a1) + (a2 + (a3 + a4)) for five).  orc_set_sum_order switches each kind to
 * the tree form so that tests can MEASURE how many gates the unverifiable choice can flip
 * (tests/test_oracle.py::test_assumed_summation_orders_flip_no_gate); the product follows the default.
 *   bit 0: the 3-term double dot products of rot_end * (R_L_I * p + T_L_I), rot_end^T * n, (. ) * C
 *          (laserMapping.cpp:836, 954, 966-971)
 *   bit 1: normvec.norm() of esti_plane (common_lib.h:285), three floats
 *   bit 2: the column norms of colPivHouseholderQr (colwise().norm(), five floats; common_lib.h:283) */
static int g_sum_order = 0;
void orc_set_sum_order(int mask) { g_sum_order = mask; }
int orc_get_sum_order(void) { return g_sum_order; }
static inline double dot3(double a0, double b0, double a1, double b1, double a2, double b2)
{
    if (g_sum_order & 1) return a0 * b0 + (a1 * b1 + a2 * b2);
    return a0 * b0 + a1 * b1 + a2 * b2;
}
static void mat3_vec(const double A[9], const double v[3], double o[3])
{
    double t[3];
    for (int i = 0; i < 3; ++i) t[i] = dot3(A[i * 3 + 0], v[0], A[i * 3 + 1], v[1], A[i * 3 + 2], v[2]);
    o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
}
static void mat3_tvec(const double A[9], const double v[3], double o[3]) /* A^T v */
{
    double t[3];
    for (int i = 0; i < 3; ++i) t[i] = dot3(A[0 * 3 + i], v[0], A[1 * 3 + i], v[1], A[2 * 3 + i], v[2]);
    o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
}
/* SKEW_SYM_MATRX(v) * w   (so3_math.h:9) */
static void skew_mul(const double v[3], const double w[3], double o[3])
{
    double t0 = 0.0 * w[0] + -v[2] * w[1] + v[1] * w[2];
    double t1 = v[2] * w[0] + 0.0 * w[1] + -v[0] * w[2];
    double t2 = -v[1] * w[0] + v[0] * w[1] + 0.0 * w[2];
    o[0] = t0; o[1] = t1; o[2] = t2;
}

/* In-place inverse of an n x n row-major matrix by LU with partial pivoting (what Eigen's
 * fixed-size .inverse() does for n > 4; laserMapping.cpp:1017-1018).  Returns 0 on a zero
 * pivot. */
static int mat_inverse(int n, const double *Ain, double *Ainv)
{
    double *A = (double *)malloc(sizeof(double) * (size_t)n * n);
    int *piv = (int *)malloc(sizeof(int) * (size_t)n);
    memcpy(A, Ain, sizeof(double) * (size_t)n * n);
    int ok = 1;
    for (int i = 0; i < n; ++i) piv[i] = i;
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = fabs(A[k * n + k]);
        for (int i = k + 1; i < n; ++i) {
            double v = fabs(A[i * n + k]);
            if (v > best) { best = v; p = i; }
        }
        if (best == 0.0) { ok = 0; break; }
        if (p != k) {
            for (int j = 0; j < n; ++j) { double t = A[k * n + j]; A[k * n + j] = A[p * n + j]; A[p * n + j] = t; }
            int t = piv[k]; piv[k] = piv[p]; piv[p] = t;
        }
        for (int i = k + 1; i < n; ++i) {
            double f = A[i * n + k] / A[k * n + k];
            A[i * n + k] = f;
            for (int j = k + 1; j < n; ++j) A[i * n + j] -= f * A[k * n + j];
        }
    }
    if (ok) {
        double *y = (double *)malloc(sizeof(double) * (size_t)n);
        for (int c = 0; c < n; ++c) {
            /* solve L U x = P e_c */
            for (int i = 0; i < n; ++i) {
                double s = (piv[i] == c) ? 1.0 : 0.0;
                for (int j = 0; j < i; ++j) s -= A[i * n + j] * y[j];
                y[i] = s;
            }
            for (int i = n - 1; i >= 0; --i) {
                double s = y[i];
                for (int j = i + 1; j < n; ++j) s -= A[i * n + j] * Ainv[j * n + c];
                Ainv[i * n + c] = s / A[i * n + i];
            }
        }
        free(y);
    }
    free(A);
    free(piv);
    return ok;
}

/* ======================================================================================== */
/* SO(3), manifold ops                                                                      */
/* ======================================================================================== */
void orc_so3_exp(double v1, double v2, double v3, double R[9])
{
    /* so3_math.h:55-72: identity unless norm > 1e-5; Rodrigues with the normalised axis */
    double norm = sqrt(v1 * v1 + v2 * v2 + v3 * v3);
    double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (norm > 0.00001) {
        double r[3] = {v1 / norm, v2 / norm, v3 / norm};
        double K[9] = {0.0, -r[2], r[1], r[2], 0.0, -r[0], -r[1], r[0], 0.0};
        double s = sin(norm), c1 = 1.0 - cos(norm);
        double cK[9], cKK[9];
        for (int i = 0; i < 9; ++i) cK[i] = c1 * K[i];
        mat3_mul(cK, K, cKK);
        for (int i = 0; i < 9; ++i) R[i] = (I[i] + s * K[i]) + cKK[i];
    } else {
        memcpy(R, I, sizeof(I));
    }
}

void orc_so3_log(const double R[9], double out[3])
{
    /* so3_math.h:76-81 */
    double tr = R[0] + R[4] + R[8];
    double theta = (tr > 3.0 - 1e-6) ? 0.0 : acos(0.5 * (tr - 1));
    double K[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
    if (fabs(theta) < 0.001) {
        for (int i = 0; i < 3; ++i) out[i] = 0.5 * K[i];
    } else {
        double f = 0.5 * theta / sin(theta);
        for (int i = 0; i < 3; ++i) out[i] = f * K[i];
    }
}

void orc_state_boxplus(orc_state *x, const double d[ORC_DIM])
{
    /* common_lib.h:146-157 */
    double E[9];
    orc_so3_exp(d[0], d[1], d[2], E);
    mat3_mul(x->rot, E, x->rot);
    for (int i = 0; i < 3; ++i) x->pos[i] += d[3 + i];
    orc_so3_exp(d[6], d[7], d[8], E);
    mat3_mul(x->R_LI, E, x->R_LI);
    for (int i = 0; i < 3; ++i) x->T_LI[i] += d[9 + i];
    for (int i = 0; i < 3; ++i) x->vel[i] += d[12 + i];
    for (int i = 0; i < 3; ++i) x->bg[i] += d[15 + i];
    for (int i = 0; i < 3; ++i) x->ba[i] += d[18 + i];
    for (int i = 0; i < 3; ++i) x->grav[i] += d[21 + i];
}

void orc_state_boxminus(const orc_state *a, const orc_state *b, double out[ORC_DIM])
{
    /* common_lib.h:173-187: a - b */
    double Rd[9];
    mat3_tmul(b->rot, a->rot, Rd);
    orc_so3_log(Rd, out + 0);
    for (int i = 0; i < 3; ++i) out[3 + i] = a->pos[i] - b->pos[i];
    mat3_tmul(b->R_LI, a->R_LI, Rd);
    orc_so3_log(Rd, out + 6);
    for (int i = 0; i < 3; ++i) out[9 + i] = a->T_LI[i] - b->T_LI[i];
    for (int i = 0; i < 3; ++i) out[12 + i] = a->vel[i] - b->vel[i];
    for (int i = 0; i < 3; ++i) out[15 + i] = a->bg[i] - b->bg[i];
    for (int i = 0; i < 3; ++i) out[18 + i] = a->ba[i] - b->ba[i];
    for (int i = 0; i < 3; ++i) out[21 + i] = a->grav[i] - b->grav[i];
}

/* ======================================================================================== */
/* exact 5-NN: static k-d tree, one point per node, AABB pruning                             */
/* ======================================================================================== */
struct orc_kdtree {
    int64_t m;
    float *x, *y, *z;  /* points in tree order: node of range [l,r] is (l+r)>>1 */
    int32_t *orig;     /* original index of each tree-order point */
    uint32_t *rank;    /* tie order of each tree-order point among candidates at exactly the same d2 (default: orig) */
    float *box;        /* 6 floats per node: xmin xmax ymin ymax zmin zmax of its subtree */
};

static inline float kd_coord(const orc_kdtree *t, int axis, int64_t i)
{
    return axis == 0 ? t->x[i] : (axis == 1 ? t->y[i] : t->z[i]);
}
static inline void kd_swap(orc_kdtree *t, int64_t a, int64_t b)
{
    float f;
    int32_t o;
    uint32_t r;
    f = t->x[a]; t->x[a] = t->x[b]; t->x[b] = f;
    f = t->y[a]; t->y[a] = t->y[b]; t->y[b] = f;
    f = t->z[a]; t->z[a] = t->z[b]; t->z[b] = f;
    o = t->orig[a]; t->orig[a] = t->orig[b]; t->orig[b] = o;
    r = t->rank[a]; t->rank[a] = t->rank[b]; t->rank[b] = r;
}
/* nth_element on [l, r] by one coordinate (ikd_Tree.cpp:707-722) */
static void kd_select(orc_kdtree *t, int axis, int64_t l, int64_t r, int64_t k)
{
    while (l < r) {
        int64_t mid = l + ((r - l) >> 1);
        /* median of three as pivot */
        float a = kd_coord(t, axis, l), b = kd_coord(t, axis, mid), c = kd_coord(t, axis, r);
        int64_t pi = (a < b) ? ((b < c) ? mid : (a < c ? r : l)) : ((a < c) ? l : (b < c ? r : mid));
        float pv = kd_coord(t, axis, pi);
        kd_swap(t, pi, r);
        int64_t s = l;
        for (int64_t i = l; i < r; ++i)
            if (kd_coord(t, axis, i) < pv) { kd_swap(t, i, s); ++s; }
        /* second partition point: skip the run of elements equal to the pivot */
        kd_swap(t, s, r);
        int64_t e = s + 1;
        for (int64_t i = s + 1; i <= r; ++i)
            if (kd_coord(t, axis, i) == pv) { kd_swap(t, i, e); ++e; }
        if (k < s) r = s - 1;
        else if (k >= e) l = e;
        else return;
    }
}
static void kd_build(orc_kdtree *t, int64_t l, int64_t r)
{
    if (l > r) return;
    int64_t mid = (l + r) >> 1;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = l; i <= r; ++i) {
        if (t->x[i] < mn[0]) mn[0] = t->x[i];
        if (t->x[i] > mx[0]) mx[0] = t->x[i];
        if (t->y[i] < mn[1]) mn[1] = t->y[i];
        if (t->y[i] > mx[1]) mx[1] = t->y[i];
        if (t->z[i] < mn[2]) mn[2] = t->z[i];
        if (t->z[i] > mx[2]) mx[2] = t->z[i];
    }
    /* longest-extent axis, first wins on ties (ikd_Tree.cpp:699-703) */
    int axis = 0;
    float rng[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
    for (int i = 1; i < 3; ++i)
        if (rng[i] > rng[axis]) axis = i;
    kd_select(t, axis, l, r, mid);
    float *b = t->box + 6 * mid; /* subtree AABB == range AABB (ikd_Tree.cpp:1457-1624) */
    b[0] = mn[0]; b[1] = mx[0]; b[2] = mn[1]; b[3] = mx[1]; b[4] = mn[2]; b[5] = mx[2];
    kd_build(t, l, mid - 1);
    kd_build(t, mid + 1, r);
}

orc_kdtree *orc_kdtree_build(const float *xyz, int64_t m)
{
    orc_kdtree *t = (orc_kdtree *)calloc(1, sizeof(*t));
    t->m = m;
    size_t mm = (size_t)(m > 0 ? m : 1);
    t->x = (float *)malloc(sizeof(float) * mm);
    t->y = (float *)malloc(sizeof(float) * mm);
    t->z = (float *)malloc(sizeof(float) * mm);
    t->orig = (int32_t *)malloc(sizeof(int32_t) * mm);
    t->rank = (uint32_t *)malloc(sizeof(uint32_t) * mm);
    t->box = (float *)malloc(sizeof(float) * 6 * mm);
    for (int64_t i = 0; i < m; ++i) {
        t->x[i] = xyz[3 * i + 0];
        t->y[i] = xyz[3 * i + 1];
        t->z[i] = xyz[3 * i + 2];
        t->orig[i] = (int32_t)i;
        t->rank[i] = (uint32_t)i;
    }
    kd_build(t, 0, m - 1);
    return t;
}
/* rank[i] = position of original index i in the total order that breaks ties between candidates at exactly the same
 * float d2 (a permutation of 0..m-1).  The GPU engine ranks such candidates by its sorted position -- (brick, cell,
 * caller index), s2m_map_get_order -- which tests compute independently from s2m_map_info and hand over here. */
void orc_kdtree_set_rank(orc_kdtree *t, const uint32_t *rank)
{
    for (int64_t i = 0; i < t->m; ++i) t->rank[i] = rank ? rank[t->orig[i]] : (uint32_t)t->orig[i];
}
void orc_kdtree_free(orc_kdtree *t)
{
    if (!t) return;
    free(t->x); free(t->y); free(t->z); free(t->orig); free(t->rank); free(t->box);
    free(t);
}
int64_t orc_kdtree_size(const orc_kdtree *t) { return t->m; }

typedef struct {
    float d2[ORC_K], x[ORC_K], y[ORC_K], z[ORC_K];
    int32_t idx[ORC_K];
    uint32_t rank[ORC_K];
    int n;
} top5;

/* strict total order on candidates: (d2, rank).  The reference orders by d2 and breaks d2 ties by x
 * (ikd_Tree.h:102-108); which of two candidates tied at the 5th place survives there depends on its
 * traversal order.  Any fixed total order on the map points is an equally valid choice; the rank is the
 * original index unless orc_kdtree_set_rank installed another one (the GPU engine's sorted position, which
 * it carries in the low half of a 64-bit search key: one compare per candidate). */
static inline int cand_less(float d2a, uint32_t ra, float d2b, uint32_t rb)
{
    if (d2a != d2b) return d2a < d2b;
    return ra < rb;
}
static inline void top5_offer(top5 *h, float d2, float x, float y, float z, int32_t idx, uint32_t rank)
{
    if (h->n == ORC_K && !cand_less(d2, rank, h->d2[ORC_K - 1], h->rank[ORC_K - 1])) return;
    int p = (h->n < ORC_K) ? h->n : ORC_K - 1;
    while (p > 0 && cand_less(d2, rank, h->d2[p - 1], h->rank[p - 1])) {
        h->d2[p] = h->d2[p - 1]; h->x[p] = h->x[p - 1]; h->y[p] = h->y[p - 1]; h->z[p] = h->z[p - 1];
        h->idx[p] = h->idx[p - 1]; h->rank[p] = h->rank[p - 1];
        --p;
    }
    h->d2[p] = d2; h->x[p] = x; h->y[p] = y; h->z[p] = z; h->idx[p] = idx; h->rank[p] = rank;
    if (h->n < ORC_K) h->n++;
}
/* float squared L2, ((dx*dx + dy*dy) + dz*dz)   (ikd_Tree.cpp:1682-1688) */
static inline float dist2f(float ax, float ay, float az, float bx, float by, float bz)
{
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    float d = dx * dx + dy * dy;
    d = d + dz * dz;
    return d;
}
/* ikd_Tree.cpp:1690-1709 */
static inline float box_dist2f(const float *b, float qx, float qy, float qz)
{
    float md = 0.0f;
    if (qx < b[0]) md += (qx - b[0]) * (qx - b[0]);
    if (qx > b[1]) md += (qx - b[1]) * (qx - b[1]);
    if (qy < b[2]) md += (qy - b[2]) * (qy - b[2]);
    if (qy > b[3]) md += (qy - b[3]) * (qy - b[3]);
    if (qz < b[4]) md += (qz - b[4]) * (qz - b[4]);
    if (qz > b[5]) md += (qz - b[5]) * (qz - b[5]);
    return md;
}
/* ikd_Tree.cpp:1061-1244 without the delete/rebuild machinery: visit the node's own point,
 * then the nearer child first and the farther child only if its box can still hold a better
 * candidate.  "<=" instead of the reference's "<" keeps equal-d2 candidates reachable so the
 * (d2, index) order is exact. */
static void kd_search(const orc_kdtree *t, int64_t l, int64_t r, float qx, float qy, float qz, top5 *h)
{
    if (l > r) return;
    int64_t mid = (l + r) >> 1;
    top5_offer(h, dist2f(qx, qy, qz, t->x[mid], t->y[mid], t->z[mid]), t->x[mid], t->y[mid],
               t->z[mid], t->orig[mid], t->rank[mid]);
    int64_t ll = l, lr = mid - 1, rl = mid + 1, rr = r;
    float dl = (ll <= lr) ? box_dist2f(t->box + 6 * ((ll + lr) >> 1), qx, qy, qz) : INFINITY;
    float dr = (rl <= rr) ? box_dist2f(t->box + 6 * ((rl + rr) >> 1), qx, qy, qz) : INFINITY;
    if (dl <= dr) {
        if (h->n < ORC_K || dl <= h->d2[ORC_K - 1]) kd_search(t, ll, lr, qx, qy, qz, h);
        if (h->n < ORC_K || dr <= h->d2[ORC_K - 1]) kd_search(t, rl, rr, qx, qy, qz, h);
    } else {
        if (h->n < ORC_K || dr <= h->d2[ORC_K - 1]) kd_search(t, rl, rr, qx, qy, qz, h);
        if (h->n < ORC_K || dl <= h->d2[ORC_K - 1]) kd_search(t, ll, lr, qx, qy, qz, h);
    }
}
static void top5_store(const top5 *h, int32_t *idx, float *d2, int32_t *cnt)
{
    for (int k = 0; k < ORC_K; ++k) {
        idx[k] = (k < h->n) ? h->idx[k] : -1;
        d2[k] = (k < h->n) ? h->d2[k] : INFINITY;
    }
    *cnt = h->n;
}

void orc_knn5(const orc_kdtree *t, const float *q, int64_t n, int32_t *idx, float *d2, int32_t *cnt,
              int nthreads)
{
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int64_t i = 0; i < n; ++i) {
        top5 h;
        h.n = 0;
        kd_search(t, 0, t->m - 1, q[3 * i], q[3 * i + 1], q[3 * i + 2], &h);
        top5_store(&h, idx + ORC_K * i, d2 + ORC_K * i, cnt + i);
    }
}

void orc_knn5_brute_ranked(const float *xyz, int64_t m, const uint32_t *rank, const float *q, int64_t n, int32_t *idx,
                           float *d2, int32_t *cnt);
void orc_knn5_brute(const float *xyz, int64_t m, const float *q, int64_t n, int32_t *idx, float *d2,
                    int32_t *cnt)
{
    orc_knn5_brute_ranked(xyz, m, NULL, q, n, idx, d2, cnt);
}
void orc_knn5_brute_ranked(const float *xyz, int64_t m, const uint32_t *rank, const float *q, int64_t n, int32_t *idx,
                           float *d2, int32_t *cnt)
{
    for (int64_t i = 0; i < n; ++i) {
        top5 h;
        h.n = 0;
        for (int64_t j = 0; j < m; ++j)
            top5_offer(&h, dist2f(q[3 * i], q[3 * i + 1], q[3 * i + 2], xyz[3 * j], xyz[3 * j + 1], xyz[3 * j + 2]),
                       xyz[3 * j], xyz[3 * j + 1], xyz[3 * j + 2], (int32_t)j, rank ? rank[j] : (uint32_t)j);
        top5_store(&h, idx + ORC_K * i, d2 + ORC_K * i, cnt + i);
    }
}

/* ======================================================================================== */
/* plane fit                                                                                */
/* ======================================================================================== */
/* esti_plane<float> (common_lib.h:267-299): solve A x = -1 (A = 5x3 neighbour coordinates) in
 * the least-squares sense with a column-pivoted Householder QR in float (the algorithm of
 * Eigen's ColPivHouseholderQR: pivot on the largest remaining column norm with LAPACK-style
 * norm downdating, Householder reflectors, rank threshold), then n = x/|x|, d = 1/|x| and the
 * 5-point inlier check. */
static int esti_plane_impl(const float nb[15], float thr, float pabcd[4], int32_t *perm_out, float *rdiag_out,
                           int32_t *rank_out);
int orc_esti_plane(const float nb[15], float thr, float pabcd[4])
{
    return esti_plane_impl(nb, thr, pabcd, NULL, NULL, NULL);
}
/* the factorisation behind the fit, for cross-checks against LAPACK's sgeqp3 (tests): perm[k] = original column in
 * position k, rdiag[k] = R(k,k), *rank = Eigen's rank decision */
int orc_esti_plane_qr(const float nb[15], float thr, float pabcd[4], int32_t perm[3], float rdiag[3], int32_t *rank)
{
    return esti_plane_impl(nb, thr, pabcd, perm, rdiag, rank);
}
static int esti_plane_impl(const float nb[15], float thr, float pabcd[4], int32_t *perm_out, float *rdiag_out,
                           int32_t *rank_out)
{
    enum { R = ORC_K, C = 3 };
    float A[R][C], c[R];
    float tau[C], nrm_upd[C], nrm_dir[C];
    int trans[C];
    for (int i = 0; i < R; ++i) {
        A[i][0] = nb[3 * i + 0]; A[i][1] = nb[3 * i + 1]; A[i][2] = nb[3 * i + 2];
        c[i] = -1.0f;
    }
    for (int k = 0; k < C; ++k) {
        float s = 0.0f;
        if (g_sum_order & 4) {  /* halving tree of five terms: (a0 + a1) + (a2 + (a3 + a4)) */
            s = (A[0][k] * A[0][k] + A[1][k] * A[1][k]) + (A[2][k] * A[2][k] + (A[3][k] * A[3][k] + A[4][k] * A[4][k]));
        } else {
            for (int i = 0; i < R; ++i) s = s + A[i][k] * A[i][k];
        }
        nrm_dir[k] = sqrtf(s);
        nrm_upd[k] = nrm_dir[k];
    }
    float nmax = nrm_upd[0];
    for (int k = 1; k < C; ++k) if (nrm_upd[k] > nmax) nmax = nrm_upd[k];
    float th = nmax * FLT_EPSILON;
    const float threshold_helper = (th * th) / (float)R;
    const float downdate_thr = sqrtf(FLT_EPSILON);
    int nonzero = C;

    for (int k = 0; k < C; ++k) {
        int big = k;
        for (int j = k + 1; j < C; ++j) if (nrm_upd[j] > nrm_upd[big]) big = j;
        float big_sq = nrm_upd[big] * nrm_upd[big];
        if (nonzero == C && big_sq < threshold_helper * (float)(R - k)) nonzero = k;
        trans[k] = big;
        if (big != k) {
            for (int i = 0; i < R; ++i) { float t = A[i][k]; A[i][k] = A[i][big]; A[i][big] = t; }
            float t = nrm_upd[k]; nrm_upd[k] = nrm_upd[big]; nrm_upd[big] = t;
            t = nrm_dir[k]; nrm_dir[k] = nrm_dir[big]; nrm_dir[big] = t;
        }
        /* Householder vector of A[k..R-1][k]; essential part stored below the diagonal */
        float tail = 0.0f;
        for (int i = k + 1; i < R; ++i) tail = tail + A[i][k] * A[i][k];
        float c0 = A[k][k], beta;
        if (tail <= FLT_MIN) {
            tau[k] = 0.0f;
            beta = c0;
            for (int i = k + 1; i < R; ++i) A[i][k] = 0.0f;
        } else {
            beta = sqrtf(c0 * c0 + tail);
            if (c0 >= 0.0f) beta = -beta;
            float den = c0 - beta;
            for (int i = k + 1; i < R; ++i) A[i][k] = A[i][k] / den;
            tau[k] = (beta - c0) / beta;
        }
        A[k][k] = beta;
        /* apply H_k = I - tau v v^T (v[k] = 1) to the trailing columns */
        if (tau[k] != 0.0f) {
            for (int j = k + 1; j < C; ++j) {
                float tmp = 0.0f;
                for (int i = k + 1; i < R; ++i) tmp = tmp + A[i][k] * A[i][j];
                tmp = tmp + A[k][j];
                A[k][j] = A[k][j] - tau[k] * tmp;
                for (int i = k + 1; i < R; ++i) A[i][j] = A[i][j] - (tau[k] * A[i][k]) * tmp;
            }
        }
        /* norm downdate (LAPACK xGEQPF rule) */
        for (int j = k + 1; j < C; ++j) {
            if (nrm_upd[j] != 0.0f) {
                float t = fabsf(A[k][j]) / nrm_upd[j];
                t = (1.0f + t) * (1.0f - t);
                if (t < 0.0f) t = 0.0f;
                float q = nrm_upd[j] / nrm_dir[j];
                float t2 = t * (q * q);
                if (t2 <= downdate_thr) {
                    float s = 0.0f;
                    for (int i = k + 1; i < R; ++i) s = s + A[i][j] * A[i][j];
                    nrm_dir[j] = sqrtf(s);
                    nrm_upd[j] = nrm_dir[j];
                } else {
                    nrm_upd[j] = nrm_upd[j] * sqrtf(t);
                }
            }
        }
    }
    /* column permutation: identity with the transpositions applied on the right */
    int perm[C] = {0, 1, 2};
    for (int k = 0; k < C; ++k) { int t = perm[k]; perm[k] = perm[trans[k]]; perm[trans[k]] = t; }

    float xs[C] = {0.0f, 0.0f, 0.0f};
    if (nonzero > 0) {
        /* c <- Q^T c : apply H_0, H_1, ... in order */
        for (int k = 0; k < nonzero; ++k) {
            if (tau[k] != 0.0f) {
                float tmp = 0.0f;
                for (int i = k + 1; i < R; ++i) tmp = tmp + A[i][k] * c[i];
                tmp = tmp + c[k];
                c[k] = c[k] - tau[k] * tmp;
                for (int i = k + 1; i < R; ++i) c[i] = c[i] - (tau[k] * A[i][k]) * tmp;
            }
        }
        /* upper-triangular solve, column oriented */
        for (int i = nonzero - 1; i >= 0; --i) {
            c[i] = c[i] / A[i][i];
            for (int r = 0; r < i; ++r) c[r] = c[r] - c[i] * A[r][i];
        }
        for (int i = 0; i < nonzero; ++i) xs[perm[i]] = c[i];
    }
    if (perm_out) for (int k = 0; k < C; ++k) perm_out[k] = perm[k];
    if (rdiag_out) for (int k = 0; k < C; ++k) rdiag_out[k] = A[k][k];
    if (rank_out) *rank_out = nonzero;
    float n = (g_sum_order & 2) ? sqrtf(xs[0] * xs[0] + (xs[1] * xs[1] + xs[2] * xs[2]))
                                : sqrtf((xs[0] * xs[0] + xs[1] * xs[1]) + xs[2] * xs[2]);
    pabcd[0] = xs[0] / n;
    pabcd[1] = xs[1] / n;
    pabcd[2] = xs[2] / n;
    pabcd[3] = (float)(1.0 / (double)n); /* "1.0 / n" is a double division (common_lib.h:289) */
    for (int j = 0; j < R; ++j) {
        float v = ((pabcd[0] * nb[3 * j] + pabcd[1] * nb[3 * j + 1]) + pabcd[2] * nb[3 * j + 2]) + pabcd[3];
        if (fabsf(v) > thr) return 0;
    }
    return 1;
}

/* ======================================================================================== */
/* residual / Jacobian pass                                                                 */
/* ======================================================================================== */
void orc_body_to_world(const orc_state *x, const float pb[3], float pw[3])
{
    /* laserMapping.cpp:835-841 */
    double p[3] = {(double)pb[0], (double)pb[1], (double)pb[2]}, t[3], g[3];
    mat3_vec(x->R_LI, p, t);
    for (int i = 0; i < 3; ++i) t[i] = t[i] + x->T_LI[i];
    mat3_vec(x->rot, t, g);
    for (int i = 0; i < 3; ++i) pw[i] = (float)(g[i] + x->pos[i]);
}

/* one Jacobian row (laserMapping.cpp:948-978) */
static void jac_row(const orc_cfg *cfg, const orc_state *x, const float pb[3], const float pl[4],
                    double h[12], double *z)
{
    double pbe[3] = {(double)pb[0], (double)pb[1], (double)pb[2]};
    double pI[3];
    mat3_vec(x->R_LI, pbe, pI);
    for (int i = 0; i < 3; ++i) pI[i] = pI[i] + x->T_LI[i];
    double nv[3] = {(double)pl[0], (double)pl[1], (double)pl[2]};
    double Cc[3], Aa[3];
    mat3_tvec(x->rot, nv, Cc);
    skew_mul(pI, Cc, Aa);
    h[0] = Aa[0]; h[1] = Aa[1]; h[2] = Aa[2];
    h[3] = nv[0]; h[4] = nv[1]; h[5] = nv[2];
    if (cfg->extrinsic_est_en) {
        /* (point_be_crossmat * R_L_I^T) * C, evaluated left to right as written (:970) */
        double S[9] = {0.0, -pbe[2], pbe[1], pbe[2], 0.0, -pbe[0], -pbe[1], pbe[0], 0.0};
        double Rt[9], M[9], Bb[3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) Rt[i * 3 + j] = x->R_LI[j * 3 + i];
        mat3_mul(S, Rt, M);
        mat3_vec(M, Cc, Bb);
        h[6] = Bb[0]; h[7] = Bb[1]; h[8] = Bb[2];
        h[9] = Cc[0]; h[10] = Cc[1]; h[11] = Cc[2];
    } else {
        for (int i = 6; i < 12; ++i) h[i] = 0.0;
    }
    *z = -(double)pl[3];
}

void orc_residual_pass(const orc_cfg *cfg, const orc_kdtree *tree, const float *map_xyz,
                       const float *scan_xyz, int64_t n, const orc_state *x, int rematch,
                       uint8_t *selected, int32_t *nn_idx, float *nn_d2, int32_t *nn_cnt,
                       float *plane, uint8_t *plane_ok, float *pd2, uint8_t *eff, double *HtH,
                       double *Htz, int32_t *effct, double *total_res, double *Hsub, double *meas)
{
    int nthreads = cfg->nthreads > 0 ? cfg->nthreads : 1;
    (void)nthreads;
    /* laserMapping.cpp:829-882, per point, independent */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads)
#endif
    for (int64_t i = 0; i < n; ++i) {
        const float *pb = scan_xyz + 3 * i;
        float pw[3];
        orc_body_to_world(x, pb, pw);
        plane_ok[i] = 0;
        eff[i] = 0;
        if (rematch) {
            top5 h;
            h.n = 0;
            kd_search(tree, 0, tree->m - 1, pw[0], pw[1], pw[2], &h);
            top5_store(&h, nn_idx + ORC_K * i, nn_d2 + ORC_K * i, nn_cnt + i);
            selected[i] = (nn_cnt[i] < ORC_K) ? 0 : (nn_d2[ORC_K * i + ORC_K - 1] > cfg->knn_d2_gate ? 0 : 1);
        }
        if (!selected[i]) continue;
        selected[i] = 0; /* sticky: only a successful fit + s-gate re-selects (:862) */
        float nb[15], pl[4];
        for (int k = 0; k < ORC_K; ++k) {
            const float *mp = map_xyz + 3 * (int64_t)nn_idx[ORC_K * i + k];
            nb[3 * k] = mp[0]; nb[3 * k + 1] = mp[1]; nb[3 * k + 2] = mp[2];
        }
        if (orc_esti_plane(nb, cfg->plane_thr, pl)) {
            plane_ok[i] = 1;
            float r = ((pl[0] * pw[0] + pl[1] * pw[1]) + pl[2] * pw[2]) + pl[3];        /* :866 */
            /* p_body.norm(): a 3-term double reduction inside Eigen (order assumed: dot3 / orc_set_sum_order bit 0) */
            double pbn = sqrt(dot3((double)pb[0], (double)pb[0], (double)pb[1], (double)pb[1], (double)pb[2], (double)pb[2]));
            /* "float s = 1 - 0.9 * fabs(pd2) / sqrt(p_body.norm())" (:868): the double expression is ROUNDED
             * TO FLOAT before "s > 0.9" (:870) promotes it back, so s_double in (0.9, 0.9000000059604645] --
             * which rounds to float(0.9) = 0.89999997615... -- is rejected */
            const float s = (float)(1 - 0.9 * fabs((double)r) / sqrt(pbn));
            plane[4 * i] = pl[0]; plane[4 * i + 1] = pl[1]; plane[4 * i + 2] = pl[2]; plane[4 * i + 3] = pl[3];
            pd2[i] = r;
            if ((double)s > cfg->s_gate) {
                selected[i] = 1;
                if (fabs((double)r) <= cfg->res_gate) eff[i] = 1;                        /* :889 */
            }
        }
    }
    /* compaction + rows + normal block, index order (laserMapping.cpp:887-896, 942-979, 1015) */
    for (int i = 0; i < 144; ++i) HtH[i] = 0.0;
    for (int i = 0; i < 12; ++i) Htz[i] = 0.0;
    int32_t m = 0;
    double tot = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        if (!eff[i]) continue;
        float pl[4] = {plane[4 * i], plane[4 * i + 1], plane[4 * i + 2], pd2[i]};
        double h[12], z;
        jac_row(cfg, x, scan_xyz + 3 * i, pl, h, &z);
        for (int a = 0; a < 12; ++a) {
            for (int b = 0; b < 12; ++b) HtH[a * 12 + b] += h[a] * h[b];
            Htz[a] += h[a] * z;
        }
        if (Hsub) memcpy(Hsub + 12 * (int64_t)m, h, sizeof(h));
        if (meas) meas[m] = z;
        tot += fabs((double)pd2[i]);
        ++m;
    }
    *effct = m;
    *total_res = tot;
}

/* ======================================================================================== */
/* Kalman update                                                                            */
/* ======================================================================================== */
static int eskf_gain(const orc_cfg *cfg, const double P[ORC_DIM * ORC_DIM], const double HtH[144],
                     double K1[ORC_DIM * ORC_DIM])
{
    enum { D = ORC_DIM };
    double S[D * D], Si[D * D];
    for (int i = 0; i < D * D; ++i) S[i] = P[i] / cfg->laser_point_cov;
    if (!mat_inverse(D, S, Si)) return 0;
    for (int a = 0; a < 12; ++a)
        for (int b = 0; b < 12; ++b) Si[a * D + b] += HtH[a * 12 + b];
    return mat_inverse(D, Si, K1);
}
static int eskf_apply(const orc_cfg *cfg, orc_state *x, const double vec[ORC_DIM],
                      const double Kz[ORC_DIM], const double KH[ORC_DIM * 12],
                      double solution[ORC_DIM])
{
    enum { D = ORC_DIM };
    for (int i = 0; i < D; ++i) {
        double khv = 0.0;
        for (int b = 0; b < 12; ++b) khv += KH[i * 12 + b] * vec[b];
        solution[i] = (Kz[i] + vec[i]) - khv; /* :1032 */
    }
    orc_state_boxplus(x, solution); /* :1033 */
    double rn = sqrt(solution[0] * solution[0] + solution[1] * solution[1] + solution[2] * solution[2]);
    double tn = sqrt(solution[3] * solution[3] + solution[4] * solution[4] + solution[5] * solution[5]);
    return (rn * 57.3 < cfg->conv_rot_deg) && (tn * 100 < cfg->conv_pos_cm); /* :1040 */
}

int orc_eskf_update(const orc_cfg *cfg, orc_state *x, const orc_state *x_prop,
                    const double P[ORC_DIM * ORC_DIM], const double HtH[144], const double Htz[12],
                    double solution[ORC_DIM], double K1[ORC_DIM * ORC_DIM])
{
    enum { D = ORC_DIM };
    double vec[D], Kz[D], KH[D * 12];
    if (!eskf_gain(cfg, P, HtH, K1)) {
        for (int i = 0; i < D; ++i) solution[i] = NAN;
        return 0;
    }
    orc_state_boxminus(x_prop, x, vec); /* :1028 */
    for (int i = 0; i < D; ++i) {
        double s = 0.0;
        for (int a = 0; a < 12; ++a) s += K1[i * D + a] * Htz[a];
        Kz[i] = s;
        for (int b = 0; b < 12; ++b) {
            double t = 0.0;
            for (int a = 0; a < 12; ++a) t += K1[i * D + a] * HtH[a * 12 + b];
            KH[i * 12 + b] = t;
        }
    }
    return eskf_apply(cfg, x, vec, Kz, KH, solution);
}

int orc_eskf_update_dense(const orc_cfg *cfg, orc_state *x, const orc_state *x_prop,
                          const double P[ORC_DIM * ORC_DIM], const double *Hsub, const double *meas,
                          int32_t m, double solution[ORC_DIM], double K1[ORC_DIM * ORC_DIM])
{
    enum { D = ORC_DIM };
    double HtH[144], vec[D], Kz[D], KH[D * 12];
    memset(HtH, 0, sizeof(HtH));
    for (int32_t r = 0; r < m; ++r) /* Hsub_T * Hsub (:1015) */
        for (int a = 0; a < 12; ++a)
            for (int b = 0; b < 12; ++b) HtH[a * 12 + b] += Hsub[12 * (int64_t)r + a] * Hsub[12 * (int64_t)r + b];
    if (!eskf_gain(cfg, P, HtH, K1)) {
        for (int i = 0; i < D; ++i) solution[i] = NAN;
        return 0;
    }
    /* K = K_1.block<24,12>(0,0) * Hsub_T  (24 x m) (:1019) */
    double *K = (double *)malloc(sizeof(double) * D * (size_t)(m > 0 ? m : 1));
    for (int i = 0; i < D; ++i)
        for (int32_t r = 0; r < m; ++r) {
            double s = 0.0;
            for (int a = 0; a < 12; ++a) s += K1[i * D + a] * Hsub[12 * (int64_t)r + a];
            K[(int64_t)i * m + r] = s;
        }
    orc_state_boxminus(x_prop, x, vec);
    for (int i = 0; i < D; ++i) {
        double s = 0.0;
        for (int32_t r = 0; r < m; ++r) s += K[(int64_t)i * m + r] * meas[r];
        Kz[i] = s;
        for (int b = 0; b < 12; ++b) {
            double t = 0.0;
            for (int32_t r = 0; r < m; ++r) t += K[(int64_t)i * m + r] * Hsub[12 * (int64_t)r + b];
            KH[i * 12 + b] = t;
        }
    }
    free(K);
    return eskf_apply(cfg, x, vec, Kz, KH, solution);
}

void orc_cov_update(const double K1[ORC_DIM * ORC_DIM], const double HtH[144], double P[ORC_DIM * ORC_DIM])
{
    enum { D = ORC_DIM };
    double G[D * D], Pn[D * D];
    memset(G, 0, sizeof(G));
    for (int i = 0; i < D; ++i)
        for (int b = 0; b < 12; ++b) {
            double t = 0.0;
            for (int a = 0; a < 12; ++a) t += K1[i * D + a] * HtH[a * 12 + b];
            G[i * D + b] = t; /* G.block<24,12>(0,0) = K * Hsub (:1084) */
        }
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) {
            double s = 0.0;
            for (int k = 0; k < D; ++k) s += ((i == k ? 1.0 : 0.0) - G[i * D + k]) * P[k * D + j];
            Pn[i * D + j] = s;
        }
    memcpy(P, Pn, sizeof(Pn));
}

/* ======================================================================================== */
/* the iterated update (laserMapping.cpp:820-1102)                                          */
/* ======================================================================================== */
void orc_iterated_update(const orc_cfg *cfg, const orc_kdtree *tree, const float *map_xyz,
                         const float *scan_xyz, int64_t n, orc_state *x, const orc_state *x_prop,
                         double P[ORC_DIM * ORC_DIM], int32_t *feat_queue, int32_t *feat_queue_len,
                         int use_dense, int32_t *log_effct, double *log_total_res,
                         int32_t *log_rematch, int32_t *log_converged, double *log_solution,
                         int32_t *nn_idx_out, orc_iter_result *res)
{
    enum { D = ORC_DIM, QN = 10 };
    size_t nn = (size_t)(n > 0 ? n : 1);
    uint8_t *selected = (uint8_t *)malloc(nn), *plane_ok = (uint8_t *)malloc(nn), *eff = (uint8_t *)malloc(nn);
    int32_t *nn_idx = (int32_t *)malloc(sizeof(int32_t) * ORC_K * nn), *nn_cnt = (int32_t *)malloc(sizeof(int32_t) * nn);
    float *nn_d2 = (float *)malloc(sizeof(float) * ORC_K * nn), *plane = (float *)malloc(sizeof(float) * 4 * nn);
    float *pd2 = (float *)malloc(sizeof(float) * nn);
    double *Hsub = use_dense ? (double *)malloc(sizeof(double) * 12 * nn) : NULL;
    double *meas = use_dense ? (double *)malloc(sizeof(double) * nn) : NULL;
    memset(selected, 1, nn); /* point_selected_surf(feats_down_size, true) (:812) */
    memset(nn_idx, 0xff, sizeof(int32_t) * ORC_K * nn);

    int rematch_num = 0, rematch_en = 0, converged = 0, stop = 0, it = 0, passes = 0;
    int32_t effct = 0;
    double total = 0.0, HtH[144], Htz[12], K1[D * D], sol[D];
    memset(K1, 0, sizeof(K1));
    memset(HtH, 0, sizeof(HtH));
    for (it = 0; it < cfg->max_iter; ++it) {
        int rematch = (it == 0) || rematch_en; /* :847 */
        passes += rematch;
        orc_residual_pass(cfg, tree, map_xyz, scan_xyz, n, x, rematch, selected, nn_idx, nn_d2, nn_cnt,
                          plane, plane_ok, pd2, eff, HtH, Htz, &effct, &total, Hsub, meas);
        /* degeneracy queue (:899-918) */
        if (*feat_queue_len < QN + 1) feat_queue[(*feat_queue_len)++] = effct;
        if (*feat_queue_len > QN) {
            memmove(feat_queue, feat_queue + 1, sizeof(int32_t) * QN);
            *feat_queue_len = QN;
        }
        stop = 0;
        for (int q = 0; q < *feat_queue_len; ++q)
            if (feat_queue[q] <= cfg->feat_threshold) { stop = 1; break; }
        for (int i = 0; i < D; ++i) sol[i] = 0.0;
        if (!stop) { /* flg_EKF_inited is always true: INIT_TIME == 0 (:75, :762) */
            converged = use_dense ? orc_eskf_update_dense(cfg, x, x_prop, P, Hsub, meas, effct, sol, K1)
                                  : orc_eskf_update(cfg, x, x_prop, P, HtH, Htz, sol, K1);
        }
        if (log_effct) log_effct[it] = effct;
        if (log_total_res) log_total_res[it] = total;
        if (log_rematch) log_rematch[it] = rematch;
        if (log_converged) log_converged[it] = converged;
        if (log_solution) memcpy(log_solution + D * it, sol, sizeof(sol));
        /* rematch judgement (:1070-1076) */
        rematch_en = 0;
        if (converged || (rematch_num == 0 && it == cfg->max_iter - 2)) {
            rematch_en = 1;
            rematch_num++;
        }
        /* exit + covariance update (:1079-1101) */
        if (rematch_num >= 2 || it == cfg->max_iter - 1) {
            if (!stop) orc_cov_update(K1, HtH, P);
            ++it;
            break;
        } else if (stop) {
            ++it;
            break;
        }
    }
    if (nn_idx_out) memcpy(nn_idx_out, nn_idx, sizeof(int32_t) * ORC_K * nn);
    if (res) {
        res->iters = it;
        res->rematch_passes = passes;
        res->converged = converged;
        res->ekf_stop = stop;
        res->effct_last = effct;
        res->total_res_last = total;
    }
    free(selected); free(plane_ok); free(eff); free(nn_idx); free(nn_cnt); free(nn_d2);
    free(plane); free(pd2); free(Hsub); free(meas);
}

/* ======================================================================================== */
/* incremental map maintenance                                             */
/* ======================================================================================== */
struct orc_map {
    float *xyz;        /* 3 floats per slot, insertion order */
    uint8_t *alive;
    int64_t n, cap, n_alive;
};

orc_map *orc_map_create(const float *xyz, int64_t m)
{
    orc_map *mp = (orc_map *)calloc(1, sizeof(*mp));
    mp->cap = m > 16 ? 2 * m : 32;
    mp->xyz = (float *)malloc(sizeof(float) * 3 * (size_t)mp->cap);
    mp->alive = (uint8_t *)malloc((size_t)mp->cap);
    if (m > 0) memcpy(mp->xyz, xyz, sizeof(float) * 3 * (size_t)m);
    memset(mp->alive, 1, (size_t)(m > 0 ? m : 0));
    mp->n = m;
    mp->n_alive = m;
    return mp;
}
void orc_map_free(orc_map *mp)
{
    if (!mp) return;
    free(mp->xyz); free(mp->alive); free(mp);
}
int64_t orc_map_size(const orc_map *mp) { return mp->n_alive; }
void orc_map_points(const orc_map *mp, float *out)
{
    int64_t k = 0;
    for (int64_t i = 0; i < mp->n; ++i)
        if (mp->alive[i]) { memcpy(out + 3 * k, mp->xyz + 3 * i, 3 * sizeof(float)); ++k; }
}
static void map_push(orc_map *mp, const float p[3])
{
    if (mp->n == mp->cap) {
        mp->cap *= 2;
        mp->xyz = (float *)realloc(mp->xyz, sizeof(float) * 3 * (size_t)mp->cap);
        mp->alive = (uint8_t *)realloc(mp->alive, (size_t)mp->cap);
    }
    memcpy(mp->xyz + 3 * mp->n, p, 3 * sizeof(float));
    mp->alive[mp->n] = 1;
    mp->n++;
    mp->n_alive++;
}
/* Search_by_range / Delete_by_range membership: min <= p < max (ikd_Tree.cpp:1259, 794) */
static inline int in_box(const float *p, const float mn[3], const float mx[3])
{
    return mn[0] <= p[0] && mx[0] > p[0] && mn[1] <= p[1] && mx[1] > p[1] && mn[2] <= p[2] && mx[2] > p[2];
}
static inline float dist2_pts(const float *a, const float *b)
{
    /* calc_dist (ikd_Tree.cpp:1682-1688) */
    float d = (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]);
    d = d + (a[2] - b[2]) * (a[2] - b[2]);
    return d;
}

int64_t orc_map_add(orc_map *mp, const float *xyz, int64_t n, int downsample_on, float ds)
{
    if (!downsample_on) {
        for (int64_t i = 0; i < n; ++i) map_push(mp, xyz + 3 * i);
        return n;
    }
    int64_t counter = 0;
    for (int64_t i = 0; i < n; ++i) {
        const float *P = xyz + 3 * i;
        float mn[3], mx[3], mid[3];
        for (int k = 0; k < 3; ++k) {
            mn[k] = floorf(P[k] / ds) * ds;                                   /* :491-496 */
            mx[k] = mn[k] + ds;
            mid[k] = (float)((double)mn[k] + (double)(mx[k] - mn[k]) / 2.0);  /* :497-499 */
        }
        /* Search_by_range over the current set (brute force: this is the oracle) */
        float min_dist = dist2_pts(P, mid);
        int64_t best = -1, cnt = 0;
        for (int64_t j = 0; j < mp->n; ++j) {
            if (!mp->alive[j] || !in_box(mp->xyz + 3 * j, mn, mx)) continue;
            ++cnt;
            const float td = dist2_pts(mp->xyz + 3 * j, mid);
            if (td < min_dist) { min_dist = td; best = j; }                   /* :503-511, strict */
        }
        /* :514-521: rewrite the voxel when it held several points or the result "is" the new point
         * (same_point: every coordinate within EPSS = 1e-6, ikd_Tree.cpp:1676-1680) */
        const float *res = best >= 0 ? mp->xyz + 3 * best : P;
        const int same = fabs((double)(P[0] - res[0])) < 1e-6 && fabs((double)(P[1] - res[1])) < 1e-6 &&
                         fabs((double)(P[2] - res[2])) < 1e-6;
        if (cnt > 1 || same) {
            float keep[3];
            memcpy(keep, best >= 0 ? mp->xyz + 3 * best : P, sizeof(keep));
            for (int64_t j = 0; j < mp->n; ++j)
                if (mp->alive[j] && in_box(mp->xyz + 3 * j, mn, mx)) { mp->alive[j] = 0; mp->n_alive--; }
            map_push(mp, keep);
            ++counter;
        }
    }
    return counter;
}

int64_t orc_map_delete_box(orc_map *mp, const float box[6])
{
    int64_t c = 0;
    for (int64_t j = 0; j < mp->n; ++j)
        if (mp->alive[j] && in_box(mp->xyz + 3 * j, box, box + 3)) { mp->alive[j] = 0; mp->n_alive--; ++c; }
    return c;
}

void orc_map_incremental_lists(const float *scan_xyz, int64_t n, const orc_state *x, const float *nn_xyz,
                               const int32_t *nn_cnt, int ekf_inited, double fs, float *to_add, int32_t *n_add,
                               float *no_down, int32_t *n_no_down)
{
    int32_t na = 0, nd = 0;
    for (int64_t i = 0; i < n; ++i) {
        float pw[3];
        orc_body_to_world(x, scan_xyz + 3 * i, pw);                            /* :591 */
        if (nn_cnt[i] > 0 && ekf_inited) {                                     /* :593 */
            const float *nb = nn_xyz + 15 * i;
            float mid[3];
            for (int k = 0; k < 3; ++k) mid[k] = (float)(floor((double)pw[k] / fs) * fs + 0.5 * fs);  /* :599-601 */
            const float dist = dist2_pts(pw, mid);                             /* :602 */
            /* float differences, compared in double with 0.5 * fs (:603) */
            if (fabs((double)(nb[0] - mid[0])) > 0.5 * fs && fabs((double)(nb[1] - mid[1])) > 0.5 * fs &&
                fabs((double)(nb[2] - mid[2])) > 0.5 * fs) {
                memcpy(no_down + 3 * nd, pw, sizeof(pw));
                ++nd;
                continue;
            }
            int need_add = 1;
            for (int r = 0; r < ORC_K; ++r) {                                  /* :608-617 */
                if (nn_cnt[i] < ORC_K) break;
                if (dist2_pts(nb + 3 * r, mid) < dist) { need_add = 0; break; }
            }
            if (need_add) { memcpy(to_add + 3 * na, pw, sizeof(pw)); ++na; }
        } else {
            memcpy(to_add + 3 * na, pw, sizeof(pw));                           /* :623 */
            ++na;
        }
    }
    *n_add = na;
    *n_no_down = nd;
}

/* ======================================================================================== */
/* scan voxel down-sampling                                                */
/* ======================================================================================== */
typedef struct { int64_t idx; int64_t pt; } vx_pair;
static int vx_cmp(const void *a, const void *b)
{
    const vx_pair *x = (const vx_pair *)a, *y = (const vx_pair *)b;
    if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
    return x->pt < y->pt ? -1 : (x->pt > y->pt ? 1 : 0); /* stable: ascending input index */
}
int64_t orc_voxel_downsample(const float *xyz, int64_t n, float leaf, float *out)
{
    if (n <= 0) return 0;
    const float inv = 1.0f / leaf;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            if (xyz[3 * i + k] < mn[k]) mn[k] = xyz[3 * i + k];
            if (xyz[3 * i + k] > mx[k]) mx[k] = xyz[3 * i + k];
        }
    int min_b[3];
    int64_t div[3];
    for (int k = 0; k < 3; ++k) {
        min_b[k] = (int)floorf(mn[k] * inv);
        div[k] = (int64_t)(int)floorf(mx[k] * inv) - min_b[k] + 1;
    }
    if (div[0] * div[1] * div[2] > 2147483647LL) return -1;
    vx_pair *v = (vx_pair *)malloc(sizeof(vx_pair) * (size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        const int i0 = (int)(floorf(xyz[3 * i] * inv) - (float)min_b[0]);
        const int i1 = (int)(floorf(xyz[3 * i + 1] * inv) - (float)min_b[1]);
        const int i2 = (int)(floorf(xyz[3 * i + 2] * inv) - (float)min_b[2]);
        v[i].idx = (int64_t)i0 + (int64_t)i1 * div[0] + (int64_t)i2 * div[0] * div[1];
        v[i].pt = i;
    }
    qsort(v, (size_t)n, sizeof(vx_pair), vx_cmp);
    int64_t m = 0;
    for (int64_t s = 0; s < n;) {
        float c[3] = {0.0f, 0.0f, 0.0f};
        int64_t t = s;
        for (; t < n && v[t].idx == v[s].idx; ++t)
            for (int k = 0; k < 3; ++k) c[k] = c[k] + xyz[3 * v[t].pt + k];
        const float fn = (float)(t - s);
        for (int k = 0; k < 3; ++k) out[3 * m + k] = c[k] / fn;
        ++m;
        s = t;
    }
    free(v);
    return m;
}

/* ======================================================================================== */
/* scan undistortion                                                       */
/* ======================================================================================== */
typedef struct { float t; int64_t i; } ud_pair;
static int ud_cmp(const void *a, const void *b)
{
    const ud_pair *x = (const ud_pair *)a, *y = (const ud_pair *)b;
    if (x->t != y->t) return x->t < y->t ? -1 : 1;
    return x->i < y->i ? -1 : (x->i > y->i ? 1 : 0);
}
/* Exp(ang_vel, dt), so3_math.h:31-52 */
static void so3_exp_rate(const double w[3], double dt, double R[9])
{
    double n = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (n > 0.0000001) {
        double r[3] = {w[0] / n, w[1] / n, w[2] / n};
        double K[9] = {0.0, -r[2], r[1], r[2], 0.0, -r[0], -r[1], r[0], 0.0};
        double ang = n * dt, sn = sin(ang), c1 = 1.0 - cos(ang), cK[9], cKK[9];
        for (int i = 0; i < 9; ++i) cK[i] = c1 * K[i];
        mat3_mul(cK, K, cKK);
        for (int i = 0; i < 9; ++i) R[i] = (I[i] + sn * K[i]) + cKK[i];
    } else {
        memcpy(R, I, sizeof(I));
    }
}
void orc_undistort(const float *rec, int64_t stride, int64_t n, int off_a, int off_b, const double *poses, int K,
                   const orc_state *end, int sort, float *out, uint32_t *perm)
{
    if (n <= 0) return;
    ud_pair *v = (ud_pair *)malloc(sizeof(ud_pair) * (size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        const float *r = rec + i * stride;
        v[i].t = off_b >= 0 ? r[off_a] * r[off_b] : r[off_a];
        v[i].i = i;
    }
    if (sort) qsort(v, (size_t)n, sizeof(ud_pair), ud_cmp);                     /* :216 */
    for (int64_t s = 0; s < n; ++s) {                                           /* pcl_out = *(meas.lidar) */
        const float *r = rec + v[s].i * stride;
        out[3 * s] = r[0]; out[3 * s + 1] = r[1]; out[3 * s + 2] = r[2];
        if (perm) perm[s] = (uint32_t)v[s].i;
    }
    if (!sort) {
        /* the reference loop walks the cloud backwards in time order; without the sort each point is
         * still handled by the same head, so process in time order and write back in input order */
        ud_pair *w = (ud_pair *)malloc(sizeof(ud_pair) * (size_t)n);
        memcpy(w, v, sizeof(ud_pair) * (size_t)n);
        qsort(w, (size_t)n, sizeof(ud_pair), ud_cmp);
        free(v);
        v = w;
    }
    /* :333-370.  The `break` at it_pcl == begin() (:366-367) leaves only the INNER loop: the outer loop goes on
     * over the earlier heads, and for each of them whose offset time is still below the first point's time
     * that point is compensated AGAIN, from its already compensated (float) coordinates.  Restated as is for
     * the sorted cloud (the only form the reference runs); with sort == 0 (an extension) every point is
     * handled once by its own head. */
    int64_t it = n - 1;
    int stop = 0;
    for (int kp = K - 1; kp >= 1 && !stop; --kp) {
        const double *head = poses + 22 * (kp - 1);
        const double *acc = head + 1, *gyr = head + 4, *vel = head + 7, *pos = head + 10, *R = head + 13;
        for (; (double)v[it].t > head[0]; --it) {
            const double dt = (double)v[it].t - head[0];
            const int64_t o = sort ? it : v[it].i;   /* output slot */
            double E[9], Ri[9], Tei[3], Pi[3], a[3], b[3], c[3], d[3];
            so3_exp_rate(gyr, dt, E);
            mat3_mul(R, E, Ri);
            for (int k = 0; k < 3; ++k) Tei[k] = ((pos[k] + vel[k] * dt) + (0.5 * acc[k]) * dt * dt) - end->pos[k];
            for (int k = 0; k < 3; ++k) Pi[k] = (double)out[3 * o + k];
            mat3_vec(end->R_LI, Pi, a);
            for (int k = 0; k < 3; ++k) a[k] = a[k] + end->T_LI[k];
            mat3_vec(Ri, a, b);
            for (int k = 0; k < 3; ++k) b[k] = b[k] + Tei[k];
            mat3_tvec(end->rot, b, c);
            for (int k = 0; k < 3; ++k) c[k] = c[k] - end->T_LI[k];
            mat3_tvec(end->R_LI, c, d);
            for (int k = 0; k < 3; ++k) out[3 * o + k] = (float)d[k];
            if (it == 0) { stop = !sort; break; }
        }
    }
    free(v);
}
