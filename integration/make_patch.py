#!/usr/bin/env python3
"""Regenerates integration/laserMapping_s2m.patch from the reference tree (build container only).

The patch is what a DaLiTI maintainer applies to `eskf_lio/` to run the scan-to-map update on the MI355X
engine: every change sits under `#ifdef DALITI_S2M` (CMake option of the same name), the original code
stays the default.  This script holds only the inserted code and the one-line anchors it hangs on; the
reference source itself never enters this repository (it is read from /root/reference, edited in a
temporary directory and diffed).

usage: python integration/make_patch.py [/root/reference]
"""
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))

HELPERS = r'''
#ifdef DALITI_S2M
// MI355X scan-to-map engine (include/daliti_s2m.h of daliti_amd): replaces the ikd-Tree search, esti_plane, the
// residual / Jacobian loops and the iterated Kalman update below with HIP kernels behind a C ABI.
#include "daliti_s2m.h"
#include "daliti_s2m_mirror.hpp"
static s2m_engine *s2m_eng = nullptr;
static s2m_map_mirror s2m_mirror;   // featsFromMap, followed through the engine's change log
static void s2m_check(int rc, const char *what)
{
    if (rc == S2M_OK) return;
    ROS_FATAL("%s failed: %s (%s)", what, s2m_strerror(rc), s2m_eng ? s2m_last_error(s2m_eng) : "");
    ros::shutdown();
    std::exit(1);
}
// StatesGroup <-> the 36 doubles of the ABI: rot_end(9, row-major) pos_end R_L_I(9) T_L_I vel_end bias_g bias_a gravity
static void s2m_to_flat(const StatesGroup &s, double x[S2M_STATE_DOUBLES])
{
    Eigen::Map<Eigen::Matrix<double, 3, 3, Eigen::RowMajor>>(x + 0) = s.rot_end;
    Eigen::Map<Eigen::Vector3d>(x + 9) = s.pos_end;
    Eigen::Map<Eigen::Matrix<double, 3, 3, Eigen::RowMajor>>(x + 12) = s.R_L_I;
    Eigen::Map<Eigen::Vector3d>(x + 21) = s.T_L_I;
    Eigen::Map<Eigen::Vector3d>(x + 24) = s.vel_end;
    Eigen::Map<Eigen::Vector3d>(x + 27) = s.bias_g;
    Eigen::Map<Eigen::Vector3d>(x + 30) = s.bias_a;
    Eigen::Map<Eigen::Vector3d>(x + 33) = s.gravity;
}
static void s2m_from_flat(const double x[S2M_STATE_DOUBLES], StatesGroup &s)
{
    s.rot_end = Eigen::Map<const Eigen::Matrix<double, 3, 3, Eigen::RowMajor>>(x + 0);
    s.pos_end = Eigen::Map<const Eigen::Vector3d>(x + 9);
    s.R_L_I = Eigen::Map<const Eigen::Matrix<double, 3, 3, Eigen::RowMajor>>(x + 12);
    s.T_L_I = Eigen::Map<const Eigen::Vector3d>(x + 21);
    s.vel_end = Eigen::Map<const Eigen::Vector3d>(x + 24);
    s.bias_g = Eigen::Map<const Eigen::Vector3d>(x + 27);
    s.bias_a = Eigen::Map<const Eigen::Vector3d>(x + 30);
    s.gravity = Eigen::Map<const Eigen::Vector3d>(x + 33);
}
#endif
'''

FOV_DELETE = r'''#ifdef DALITI_S2M
    if (cub_needrm.size() > 0)
    {   // BoxPointType is exactly {float min[3], max[3]}: the array goes through as it is
        int64_t s2m_m = 0, s2m_deleted = 0;
        s2m_map_size(s2m_eng, &s2m_m);
        if (s2m_m > 0)
            s2m_check(s2m_map_delete_boxes(s2m_eng, &cub_needrm[0].vertex_min[0], (int64_t)cub_needrm.size(), &s2m_deleted),
                      "s2m_map_delete_boxes");
        kdtree_delete_counter = (int)s2m_deleted;
    }
#else
'''

MAP_INCREMENTAL = r'''#ifdef DALITI_S2M
    {   // classification from the cached Nearest_Points + Add_Points(.., true) + Add_Points(.., false) on the device
        double s2m_x[S2M_STATE_DOUBLES];
        int64_t s2m_add = 0, s2m_nodown = 0;
        s2m_to_flat(state, s2m_x);
        s2m_check(s2m_map_incremental(s2m_eng, s2m_x, filter_size_map_min, flg_EKF_inited ? 1 : 0, &s2m_add, &s2m_nodown),
                  "s2m_map_incremental");
        add_point_size = (int)(s2m_add + s2m_nodown);
        return;
    }
#endif
'''

CREATE = r'''
#ifdef DALITI_S2M
    {
        s2m_config s2m_cfg;
        s2m_config_default(&s2m_cfg);
        s2m_cfg.max_iter = NUM_MAX_ITERATIONS;
        s2m_cfg.extrinsic_est_en = extrinsic_est_en ? 1 : 0;
        s2m_cfg.feat_threshold = dynamic_effect_featurepoints_threshold;
        s2m_check(s2m_create(&s2m_cfg, &s2m_eng), "s2m_create");
    }
#endif
'''

SEED_TEST = r'''#ifdef DALITI_S2M
            int64_t s2m_map_points = 0;
            s2m_map_size(s2m_eng, &s2m_map_points);
            if (s2m_map_points == 0)
#else
'''

SEED_BUILD = r'''#ifdef DALITI_S2M
                    s2m_check(s2m_map_build(s2m_eng, &feats_down_world->points[0].x, sizeof(PointType) / sizeof(float),
                                            feats_down_size, 0), "s2m_map_build");
#else
'''

VALIDNUM = r'''#ifdef DALITI_S2M
            int featsFromMapNum = (int)s2m_map_points;
#else
'''

UPDATE = r'''#ifdef DALITI_S2M
                {   // the whole iterated update (the loop below) in one call
                    s2m_config s2m_cfg;
                    s2m_config_default(&s2m_cfg);
                    s2m_cfg.max_iter = NUM_MAX_ITERATIONS;
                    s2m_cfg.extrinsic_est_en = extrinsic_est_en ? 1 : 0;
                    s2m_cfg.feat_threshold = dynamic_effect_featurepoints_threshold;   // changes after frame 100
                    s2m_check(s2m_set_config(s2m_eng, &s2m_cfg), "s2m_set_config");
                    s2m_check(s2m_scan_set(s2m_eng, &feats_down->points[0].x, sizeof(PointType) / sizeof(float), feats_down_size, 0),
                              "s2m_scan_set");
                    double s2m_x[S2M_STATE_DOUBLES], s2m_xp[S2M_STATE_DOUBLES];
                    Eigen::Matrix<double, DIM_OF_STATES, DIM_OF_STATES, Eigen::RowMajor> s2m_P = state.cov;
                    s2m_to_flat(state, s2m_x);
                    s2m_to_flat(state_propagat, s2m_xp);
                    s2m_iter_log s2m_log;
                    s2m_check(s2m_iterated_update(s2m_eng, s2m_x, s2m_xp, s2m_P.data(), &s2m_log), "s2m_iterated_update");
                    for (int it = 0; it < s2m_log.iters; it++)
                    {   // Log/mat_out.txt, one row per iteration as before
                        res_mean_last = s2m_log.total_residual[it] / s2m_log.effct[it];
                        fout_out << std::setw(10) << Measures.lidar_beg_time - first_lidar_time << " " << s2m_log.effct[it]
                                 << " " << res_mean_last << " " << (it ? s2m_log.conv[it - 1] != 0 : false) << " "
                                 << (it == s2m_log.iters - 1 && s2m_log.ekf_stop) << " " << feats_down_size << " "
                                 << RECV_LIO_FAIL_FLAG << " " << recv_n << std::endl;
                    }
                    iterCount = s2m_log.iters - 1;
                    effct_feat_num = s2m_log.effct[s2m_log.iters - 1];
                    flg_EKF_converged = s2m_log.converged != 0;
                    EKF_stop_flg = s2m_log.ekf_stop != 0;
                    if (!EKF_stop_flg)
                    {
                        s2m_from_flat(s2m_x, state);
                        state.cov = s2m_P;
                        last_nodegared_state = state;
                        deltaOdomSetZero(g_tis_odom_delta);
                        total_distance += (state.pos_end - position_last).norm();
                    }
                    else
                    {   // degenerate scan: thermal-inertial fallback, unchanged
                        state = last_nodegared_state + odomToStateGruop(g_tis_odom_delta);
                        flg_EKF_inited = false;
                    }
                    position_last = state.pos_end;
                    // laserCloudOri (the effective points, index order) for /cloud_effected
                    std::vector<int32_t> s2m_rows(feats_down_size > 0 ? feats_down_size : 1);
                    int64_t s2m_m = 0;
                    s2m_check(s2m_get_rows(s2m_eng, nullptr, nullptr, s2m_rows.data(), feats_down_size, &s2m_m), "s2m_get_rows");
                    laserCloudOri->clear();
                    for (int64_t i = 0; i < s2m_m; i++)
                        laserCloudOri->push_back(feats_down->points[s2m_rows[i]]);
                }
#else
'''

FLATTEN = r'''#ifdef DALITI_S2M
                if (pubLaserCloudMap.getNumSubscribers() > 0)
                {   // only when somebody listens to /Laser_map; the host mirror takes this frame's changes from the
                    // engine (daliti_s2m_mirror.hpp: a few thousand points cross PCIe, not the map)
                    s2m_check(s2m_mirror.update(s2m_eng), "s2m_map_get_changes");
                    featsFromMap->clear();
                    featsFromMap->points.resize(s2m_mirror.size());
                    int64_t s2m_i = 0;
                    s2m_mirror.for_each([&](uint32_t, float px, float py, float pz)
                    {
                        featsFromMap->points[s2m_i].x = px;
                        featsFromMap->points[s2m_i].y = py;
                        featsFromMap->points[s2m_i].z = pz;
                        s2m_i++;
                    });
                }
#else
'''

CMAKE = r'''
# MI355X scan-to-map engine (daliti_amd): -DDALITI_S2M=ON -DDALITI_S2M_ROOT=<checkout of daliti_amd>
option(DALITI_S2M "run the scan-to-map update on the MI355X engine (libdaliti_s2m.so)" OFF)
if(DALITI_S2M)
  set(DALITI_S2M_ROOT "/opt/daliti_amd" CACHE PATH "checkout / install prefix of daliti_amd")
  find_library(DALITI_S2M_LIB daliti_s2m PATHS ${DALITI_S2M_ROOT}/daliti_amd/_lib ${DALITI_S2M_ROOT}/lib NO_DEFAULT_PATH)
  target_compile_definitions(lio_laserMapping PRIVATE DALITI_S2M)
  target_include_directories(lio_laserMapping PRIVATE ${DALITI_S2M_ROOT}/include)
  target_link_libraries(lio_laserMapping ${DALITI_S2M_LIB})
endif()
'''


def insert_after(lines, anchor, text, nth=0):
    hits = [i for i, ln in enumerate(lines) if ln.rstrip("\n") == anchor]
    assert len(hits) > nth, "anchor not found: %r" % anchor
    i = hits[nth]
    lines[i + 1:i + 1] = text.splitlines(keepends=True)


def insert_before(lines, anchor, text, nth=0):
    hits = [i for i, ln in enumerate(lines) if ln.rstrip("\n") == anchor]
    assert len(hits) > nth, "anchor not found: %r" % anchor
    i = hits[nth]
    lines[i:i] = text.splitlines(keepends=True)


def wrap(lines, first, last, head, first_nth=0, last_nth=0):
    """#ifdef-head before `first`, #endif after `last` (both anchors are whole lines)."""
    insert_before(lines, first, head, first_nth)
    insert_after(lines, last, "#endif\n", last_nth)


def patch_lasermapping(path):
    lines = open(path).read().splitlines(keepends=True)
    insert_after(lines, "#include <cmath>", HELPERS)
    wrap(lines, "    if (cub_needrm.size() > 0)", "        kdtree_delete_counter = ikdtree.Delete_Point_Boxes(cub_needrm);", FOV_DELETE)
    insert_after(lines, "    PointVector PointNoNeedDownsample;", MAP_INCREMENTAL)
    insert_after(lines, '    nh.param<double>("common/beta", beta, 0.05);', CREATE)
    wrap(lines, "            if (ikdtree.Root_Node == nullptr)", "            if (ikdtree.Root_Node == nullptr)", SEED_TEST)
    wrap(lines, "                    ikdtree.Build(feats_down_world->points);", "                    ikdtree.Build(feats_down_world->points);",
         SEED_BUILD)
    wrap(lines, "            int featsFromMapNum = ikdtree.validnum();", "            int featsFromMapNum = ikdtree.validnum();", VALIDNUM)
    # the iteration loop: from its `for` line to the brace that closes it (the line before the zeta blend)
    insert_before(lines, "                for (iterCount = 0; iterCount < NUM_MAX_ITERATIONS; iterCount++)", UPDATE)
    insert_before(lines, "                if ( ( lidar_cnt < 100 ) || ( !tis_online ) || ( tis_online && lidar_cnt % 2 == 1 ) )", "#endif\n\n")
    wrap(lines, "                    PointVector().swap(ikdtree.PCL_Storage);", "                    featsFromMap->points = ikdtree.PCL_Storage;",
         FLATTEN)
    open(path, "w").write("".join(lines))


def patch_cmake(path):
    lines = open(path).read().splitlines(keepends=True)
    insert_after(lines, "target_link_libraries(lio_laserMapping ${catkin_LIBRARIES} ${PCL_LIBRARIES} ${OpenCV_LIBS} ${PYTHON_LIBRARIES})",
                 CMAKE)
    open(path, "w").write("".join(lines))


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    src = os.path.join(ref, "eskf_lio")
    with tempfile.TemporaryDirectory() as tmp:
        for side in ("a", "b"):
            os.makedirs(os.path.join(tmp, side, "src"))
            shutil.copy(os.path.join(src, "src", "laserMapping.cpp"), os.path.join(tmp, side, "src"))
            shutil.copy(os.path.join(src, "CMakeLists.txt"), os.path.join(tmp, side))
        patch_lasermapping(os.path.join(tmp, "b", "src", "laserMapping.cpp"))
        patch_cmake(os.path.join(tmp, "b", "CMakeLists.txt"))
        out = subprocess.run(["diff", "-U1", "-r", "a", "b"], cwd=tmp, capture_output=True, text=True)
        assert out.returncode == 1, out.stderr
        keep = []
        for ln in out.stdout.splitlines():
            if ln.startswith("diff -U1"):
                continue
            if ln.startswith("--- a/") or ln.startswith("+++ b/"):
                ln = ln.split("\t")[0]          # no timestamps: the file is reproducible
            keep.append(ln)
        body = "\n".join(keep) + "\n"
    head = ("DaLiTI eskf_lio -> MI355X scan-to-map engine (daliti_amd).  Apply inside eskf_lio/:  patch -p1 < laserMapping_s2m.patch\n"
            "Everything is guarded by DALITI_S2M (cmake -DDALITI_S2M=ON -DDALITI_S2M_ROOT=...); without it the node builds as before.\n"
            "Generated by integration/make_patch.py; see INTEGRATION.md.\n\n")
    dst = os.path.join(HERE, "laserMapping_s2m.patch")
    open(dst, "w").write(head + body)
    print("wrote", dst, len(body.splitlines()), "lines")


if __name__ == "__main__":
    main()
