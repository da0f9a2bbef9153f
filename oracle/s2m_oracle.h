/*
 * s2m_oracle.h -- CPU restatement of the eskf_lio scan-to-map hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle and the timed CPU baseline.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * (daliti_amd/, include/) never includes, links or calls anything in this directory.
 *
 * PARITY UNPINNED: the reference (HITSZ-NRSL/DaLiTI) ships no tests, golden vectors or
 * fixtures for this path and its sources cannot be built in this image
 * (they need ROS, PCL, Eigen and OpenCV headers, none of which are installed; writing stand-in
 * headers is not allowed).  Every function below therefore restates the reference algorithm
 * from its source, citing file:line, and is cross-checked only against independent
 * mathematical definitions (brute-force kNN, numpy lstsq / linalg) in tests/.
 *
 * All reference citations are relative to /root/reference/.
 */
#ifndef S2M_ORACLE_H
#define S2M_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_K 5          /* NUM_MATCH_POINTS, eskf_lio/src/laserMapping.cpp:77 */
#define ORC_DIM 24       /* DIM_OF_STATES,    eskf_lio/include/common_lib.h:23 */

/* StatesGroup without the covariance (eskf_lio/include/common_lib.h:219-227).
 * Matrices are row-major.  36 doubles, in the reference's member order. */
typedef struct {
    double rot[9];   /* rot_end */
    double pos[3];   /* pos_end */
    double R_LI[9];  /* R_L_I   */
    double T_LI[3];  /* T_L_I   */
    double vel[3];   /* vel_end */
    double bg[3];    /* bias_g  */
    double ba[3];    /* bias_a  */
    double grav[3];  /* gravity */
} orc_state;

/* Gates and constants of the path, with the reference values as defaults (orc_cfg_default). */
typedef struct {
    float  plane_thr;        /* 0.1f   esti_plane threshold, laserMapping.cpp:863            */
    float  knn_d2_gate;      /* 5.0f   d2[4] gate, laserMapping.cpp:853                      */
    double s_gate;           /* 0.9    laserMapping.cpp:870                                  */
    double res_gate;         /* 2.0    laserMapping.cpp:889                                  */
    double laser_point_cov;  /* 0.0015 LASER_POINT_COV, laserMapping.cpp:76                  */
    double conv_rot_deg;     /* 0.01   laserMapping.cpp:1040                                 */
    double conv_pos_cm;      /* 0.015  laserMapping.cpp:1040                                 */
    int    extrinsic_est_en; /* feat.yaml:50 ships false                                      */
    int    max_iter;         /* mapping/max_iteration; BASELINE uses 5                        */
    int    feat_threshold;   /* dynamic_effect_featurepoints_threshold, laserMapping.cpp:97   */
    int    nthreads;         /* 1 = faithful (OpenMP pragmas are commented out, :827-828)     */
} orc_cfg;

void orc_cfg_default(orc_cfg *cfg);

/* ---- SO(3) and the 24-dim manifold ops ------------------------------------------------- */
void orc_so3_exp(double v1, double v2, double v3, double R[9]);     /* so3_math.h:55-72 */
void orc_so3_log(const double R[9], double out[3]);                 /* so3_math.h:76-81 */
void orc_state_boxplus(orc_state *x, const double d[ORC_DIM]);      /* common_lib.h:146-157 */
void orc_state_boxminus(const orc_state *a, const orc_state *b, double out[ORC_DIM]); /* a - b, :173-187 */

/* ---- exact 5-NN --------------------------------------------------------------------------
 * Replaces KD_TREE::Build / Nearest_Search (eskf_lio/include/ikd-Tree/ikd_Tree.cpp:408-461,
 * 678-733, 1061-1244).  Same tree shape rule (median on the longest-extent axis, one point per
 * node, per-node AABB pruning in float) and the same float squared-L2 (ikd_Tree.cpp:1682-1688).
 * Result order: ascending (d2, rank) -- the reference orders by d2 and breaks d2 ties
 * by x (ikd_Tree.h:102-108) with a traversal-dependent choice at the 5th place; a fixed rank makes
 * the order total and independent of the search structure.  rank = original index unless
 * orc_kdtree_set_rank installs another permutation (tests pass the GPU engine's sorted position,
 * computed independently from its grid parameters: (brick, cell, caller index)). */
typedef struct orc_kdtree orc_kdtree;
orc_kdtree *orc_kdtree_build(const float *xyz, int64_t m);   /* xyz: m x 3 floats, AoS */
void        orc_kdtree_set_rank(orc_kdtree *t, const uint32_t *rank); /* rank[original index]; NULL = index order */
void        orc_kdtree_free(orc_kdtree *t);
int64_t     orc_kdtree_size(const orc_kdtree *t);
/* queries: n x 3 floats.  idx: n x 5 (index into the caller's xyz, -1 = missing), d2: n x 5
 * (INFINITY = missing), cnt: n. */
void orc_knn5(const orc_kdtree *t, const float *q, int64_t n, int32_t *idx, float *d2,
              int32_t *cnt, int nthreads);
void orc_knn5_brute(const float *xyz, int64_t m, const float *q, int64_t n, int32_t *idx,
                    float *d2, int32_t *cnt);
void orc_knn5_brute_ranked(const float *xyz, int64_t m, const uint32_t *rank, const float *q, int64_t n,
                           int32_t *idx, float *d2, int32_t *cnt);

/* ---- plane fit: esti_plane<float> (common_lib.h:267-299) -------------------------------- */
int orc_esti_plane(const float nb[15], float thr, float pabcd[4]);
/* the same fit plus the factorisation behind it: perm[k] = original column in position k of the column-pivoted QR,
 * rdiag[k] = R(k,k), *rank = the rank decision (tests compare them with LAPACK's sgeqp3 through scipy) */
int orc_esti_plane_qr(const float nb[15], float thr, float pabcd[4], int32_t perm[3], float rdiag[3], int32_t *rank);
/* ASSUMED summation orders (s2m_oracle.c): bit 0 = 3-term double dot products, bit 1 = normvec.norm(), bit 2 = QR
 * column norms evaluated as Eigen's halving tree instead of left to right.  Default 0.  Test instrument only. */
void orc_set_sum_order(int mask);
int  orc_get_sum_order(void);

/* ---- body -> world transform (laserMapping.cpp:835-841), double math, float result ------ */
void orc_body_to_world(const orc_state *x, const float pb[3], float pw[3]);

/* ---- one residual / Jacobian pass (laserMapping.cpp:829-979 + :1015) -----------------------
 * Persistent per-point state (caller-owned, kept across iterations of one scan):
 *   selected[n] (init 1), nn_idx[n*5], nn_d2[n*5], nn_cnt[n].
 * Per-pass outputs (all caller-owned; optional ones may be NULL):
 *   plane[n*4]  (n, d) of the fit, meaningful where plane_ok[i]
 *   plane_ok[n] esti_plane returned true this pass (0 when the point was skipped)
 *   pd2[n]      point-to-plane residual (float), meaningful where plane_ok[i]
 *   eff[n]      1 iff the point enters laserCloudOri this pass (:889)
 *   HtH[144]    row-major 12x12 sum of h h^T, Htz[12] sum of h z, *effct, *total_res
 *   Hsub[m*12], meas[m]  optional dense rows in index order (m = *effct) (:942-979)
 */
void orc_residual_pass(const orc_cfg *cfg, const orc_kdtree *tree, const float *map_xyz,
                       const float *scan_xyz, int64_t n, const orc_state *x, int rematch,
                       uint8_t *selected, int32_t *nn_idx, float *nn_d2, int32_t *nn_cnt,
                       float *plane, uint8_t *plane_ok, float *pd2, uint8_t *eff,
                       double *HtH, double *Htz, int32_t *effct, double *total_res,
                       double *Hsub, double *meas);

/* ---- Kalman update from the normal block (laserMapping.cpp:1012-1046) ----------------------
 * x is updated in place; solution[24] is the applied delta; K1 (24x24 row-major) is
 * (H_T_H + (P/R)^-1)^-1 and is what the covariance update needs.  Returns flg_EKF_converged. */
int orc_eskf_update(const orc_cfg *cfg, orc_state *x, const orc_state *x_prop,
                    const double P[ORC_DIM * ORC_DIM], const double HtH[144],
                    const double Htz[12], double solution[ORC_DIM],
                    double K1[ORC_DIM * ORC_DIM]);
/* The literal form of the same lines: materialises K = K_1[:, :12] * Hsub^T (24 x m) and
 * evaluates K*z + vec - (K*Hsub)*vec12 as the reference does. */
int orc_eskf_update_dense(const orc_cfg *cfg, orc_state *x, const orc_state *x_prop,
                          const double P[ORC_DIM * ORC_DIM], const double *Hsub,
                          const double *meas, int32_t m, double solution[ORC_DIM],
                          double K1[ORC_DIM * ORC_DIM]);
/* P <- (I - K1[:, :12] * HtH (+) 0) * P  (laserMapping.cpp:1084-1085) */
void orc_cov_update(const double K1[ORC_DIM * ORC_DIM], const double HtH[144],
                    double P[ORC_DIM * ORC_DIM]);

/* ---- the iterated update of one scan (laserMapping.cpp:820-1102) ---------------------------
 * feat_queue / feat_queue_len: the sliding queue of the last <=10 effct_feat_num values
 * (:899-918), updated in place.  log_* arrays have max_iter entries.  use_dense != 0 selects
 * the literal K-materialising update.  Returns the number of iterations executed. */
typedef struct {
    int32_t iters;          /* iterations executed (iterCount + 1 at exit)          */
    int32_t rematch_passes; /* passes that ran the kNN (iter 0 + rematch iterations) */
    int32_t converged;      /* flg_EKF_converged at exit                             */
    int32_t ekf_stop;       /* EKF_stop_flg at exit                                  */
    int32_t effct_last;
    double  total_res_last;
} orc_iter_result;

void orc_iterated_update(const orc_cfg *cfg, const orc_kdtree *tree, const float *map_xyz,
                         const float *scan_xyz, int64_t n, orc_state *x,
                         const orc_state *x_prop, double P[ORC_DIM * ORC_DIM],
                         int32_t *feat_queue, int32_t *feat_queue_len, int use_dense,
                         int32_t *log_effct, double *log_total_res, int32_t *log_rematch,
                         int32_t *log_converged, double *log_solution /* max_iter x 24 */,
                         int32_t *nn_idx_out /* n*5 or NULL */, orc_iter_result *res);

/* ---- incremental map maintenance --------------------------------------------
 * A dynamic point set with the semantics of the ikd-Tree calls the node makes:
 *   orc_map_add          KD_TREE::Add_Points(points, downsample_on)   ikd_Tree.cpp:477-573
 *                        (sequential; with downsample_on the point's voxel [min, max) of edge
 *                        downsample_size keeps only the point closest to the voxel centre;
 *                        Search_by_range / Delete_by_range box tests :1245-1265, 766-800)
 *   orc_map_delete_box   KD_TREE::Delete_Point_Boxes                  ikd_Tree.cpp:631-658
 *   orc_map_incremental  map_incremental()                            laserMapping.cpp:582-630
 * Points keep their insertion order; among several existing points tied for the smallest
 * centre distance the earliest inserted wins (the reference takes the first in its traversal
 * order, which is not knowable here). */
typedef struct orc_map orc_map;
orc_map *orc_map_create(const float *xyz, int64_t m);
void     orc_map_free(orc_map *mp);
int64_t  orc_map_size(const orc_map *mp);
void     orc_map_points(const orc_map *mp, float *xyz_out); /* alive points, insertion order */
/* returns the reference's tmp_counter (voxels rewritten) for downsample_on, n otherwise */
int64_t  orc_map_add(orc_map *mp, const float *xyz, int64_t n, int downsample_on, float downsample_size);
int64_t  orc_map_delete_box(orc_map *mp, const float box[6]); /* min xyz, max xyz; min <= p < max */
/* nn_xyz: n x 5 x 3 neighbour coordinates of the last rematch (Nearest_Points), nn_cnt: n.
 * ekf_inited: flg_EKF_inited at the call (laserMapping.cpp:593); zero sends every point to PointToAdd
 * (:623).  The shipped node only calls map_incremental with the flag set (:762, :1062, :1165).  Writes the two lists the reference builds (world points, index order) and their lengths;
 * to_add / no_down must hold n x 3 floats each. */
void orc_map_incremental_lists(const float *scan_xyz, int64_t n, const orc_state *x, const float *nn_xyz,
                               const int32_t *nn_cnt, int ekf_inited, double filter_size_map, float *to_add,
                               int32_t *n_add, float *no_down, int32_t *n_no_down);

/* ---- scan voxel down-sampling ------------------------------------------------
 * pcl::VoxelGrid<PointType>::applyFilter as called at laserMapping.cpp:775-776 (PCL 1.10,
 * filters/impl/voxel_grid.hpp; not in the reference tree): one centroid per occupied voxel of edge
 * `leaf`, ascending voxel index idx = ijk0 + ijk1*div0 + ijk2*div0*div1, centroid = float sum of the
 * voxel's points / count.  PCL's std::sort leaves the summation order inside a voxel unspecified;
 * ascending input index is used here.  out must hold n x 3 floats; returns the output size, or -1
 * when the voxel index would overflow int32 (PCL then returns the input cloud unchanged). */
int64_t orc_voxel_downsample(const float *xyz, int64_t n, float leaf, float *out);

/* ---- scan undistortion -------------------------------------------------------
 * The time sort and the backward-propagation loop of ImuProcess::UndistortPcl
 * (IMU_Processing.hpp:215-216, 333-370), loop structure as written there.  rec: n records of `stride`
 * floats starting with x, y, z; offset time = rec[off_a] * rec[off_b] (float) or rec[off_a] when
 * off_b < 0.  poses: K x 22 doubles {offset_time, acc3, gyr3, vel3, pos3, rot9} = IMUpose.
 * out: n x 3, in time order when sort != 0 (ties keep input order; std::sort leaves them unspecified);
 * perm[n] = input index of each output point. */
void orc_undistort(const float *rec, int64_t stride, int64_t n, int off_a, int off_b, const double *poses, int K,
                   const orc_state *end, int sort, float *out, uint32_t *perm);

#ifdef __cplusplus
}
#endif
#endif
