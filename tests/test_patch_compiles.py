"""The text of integration/laserMapping_s2m.patch through a compiler.  The node itself cannot be built here
(laserMapping.cpp needs ROS, PCL and Eigen: /root/reference/eskf_lio/src/laserMapping.cpp:43-66), so the 18 DALITI_S2M hunks had
only ever been through `patch --dry-run`.  Here the `+` lines between `#ifdef DALITI_S2M` and the matching `#else` / `#endif`
are lifted out of the patch into one translation unit -- the file-scope helpers at file scope, the fragments of each patched
function inside one function, in order -- behind a declaration-only stand-in for the node's types and globals, and compiled
with `g++ -fsyntax-only -I include`.
What it proves: the inserted text parses, every engine call matches include/daliti_s2m.h (argument count and types), the
mirror is used the way daliti_s2m_mirror.hpp declares it, no identifier is misspelt.  What it does not: that the stand-in's
types behave like Eigen / PCL / ROS (they are declarations written for this test and pin nothing), or that the hunks sit at the
right lines -- `patch --dry-run` against the reference (tests/test_replay.py) covers that in the build container."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = r'''
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <vector>
// ---- declaration-only stand-ins for what the hunks touch (test infrastructure) ----
namespace Eigen {
enum { RowMajor = 1 };
template <class T, int R, int C, int O = 0> struct Matrix {
    Matrix();
    template <class U> Matrix(const U &);
    template <class U> Matrix &operator=(const U &);
    Matrix operator-(const Matrix &) const;
    T *data();
    T norm() const;
};
typedef Matrix<double, 3, 1> Vector3d;
template <class M> struct Map {
    Map(double *);
    template <class U> Map &operator=(const U &);
};
template <class M> struct Map<const M> { Map(const double *); };
}  // namespace Eigen
#define DIM_OF_STATES 24
struct StatesGroup {
    Eigen::Matrix<double, 3, 3> rot_end, R_L_I;
    Eigen::Vector3d pos_end, T_L_I, vel_end, bias_g, bias_a, gravity;
    Eigen::Matrix<double, DIM_OF_STATES, DIM_OF_STATES> cov;
    StatesGroup operator+(const Eigen::Matrix<double, DIM_OF_STATES, 1> &) const;
};
struct PointType { float x, y, z, intensity, normal_x, normal_y, normal_z, curvature, pad[4]; };
struct BoxPointType { float vertex_min[3], vertex_max[3]; };
struct Cloud {
    std::vector<PointType> points;
    void clear();
    void push_back(const PointType &);
};
struct Publisher { int getNumSubscribers() const; };
struct MeasureGroup { double lidar_beg_time; };
struct OdomDelta {};
namespace ros { void shutdown(); }
void ROS_FATAL(const char *, ...);
extern std::vector<BoxPointType> cub_needrm;
extern int kdtree_delete_counter, add_point_size, NUM_MAX_ITERATIONS, dynamic_effect_featurepoints_threshold, feats_down_size, iterCount,
    effct_feat_num, recv_n, RECV_LIO_FAIL_FLAG;
extern bool flg_EKF_inited, extrinsic_est_en, flg_EKF_converged, EKF_stop_flg;
extern double filter_size_map_min, res_mean_last, first_lidar_time, total_distance;
extern StatesGroup state, state_propagat, last_nodegared_state;
extern Cloud *feats_down_world, *feats_down, *laserCloudOri, *featsFromMap;
extern std::ofstream fout_out;
extern MeasureGroup Measures;
extern OdomDelta g_tis_odom_delta;
extern Eigen::Vector3d position_last;
extern Publisher pubLaserCloudMap;
void deltaOdomSetZero(OdomDelta &);
Eigen::Matrix<double, DIM_OF_STATES, 1> odomToStateGruop(const OdomDelta &);
'''


def _fragments(patch_text):
    """[(first line of the hunk in the original file, text)] for every `#ifdef DALITI_S2M` block among the + lines of the
    laserMapping.cpp part of the patch (up to the matching #else / #endif)."""
    part = patch_text.split("+++ b/src/laserMapping.cpp", 1)[1]
    out, at, cur, depth = [], 0, None, 0
    for ln in part.splitlines():
        m = re.match(r"@@ -(\d+)", ln)
        if m:
            at = int(m.group(1))
            continue
        if not ln.startswith("+"):
            continue
        body = ln[1:]
        if cur is None:
            if body.strip() == "#ifdef DALITI_S2M":
                cur, depth = [], 0
            continue
        s = body.strip()
        if s.startswith("#if"):
            depth += 1
        if depth == 0 and (s == "#else" or s == "#endif"):
            out.append((at, "\n".join(cur)))
            cur = None
            continue
        if s == "#endif":
            depth -= 1
        cur.append(body)
    return out


def test_the_patch_text_compiles_against_the_abi(tmp_path):
    patch = open(os.path.join(ROOT, "integration", "laserMapping_s2m.patch")).read()
    frags = _fragments(patch)
    assert len(frags) >= 8, len(frags)                      # helpers, fov, map_incremental, create, seed x3, update, flatten
    assert "s2m_iterated_update(" in "".join(t for _, t in frags) and "s2m_mirror.update(" in "".join(t for _, t in frags)
    # the hunks by the function of laserMapping.cpp they sit in (its line numbers: helpers :70, lasermap_fov_segment :313-369,
    # map_incremental :582-630, main :641-)
    file_scope = [t for a, t in frags if a < 300]
    fov = [t for a, t in frags if 300 <= a < 500]
    incr = [t for a, t in frags if 500 <= a < 640]
    main = [t for a, t in frags if a >= 640]
    assert file_scope and fov and incr and len(main) >= 5
    tu = [STUB, "#define DALITI_S2M 1"] + file_scope
    tu.append("void patched_lasermap_fov_segment()\n{\n" + "\n".join(fov) + "\n}")
    tu.append("void patched_map_incremental()\n{\n" + "\n".join(incr) + "\n}")
    # (a fragment may end in the `if (...)` that governs the original's next statement: an empty block closes it)
    tu.append("int patched_main()\n{\n" + "\n{}\n".join(main) + "\n{}\nreturn 0;\n}")
    src = tmp_path / "patch_tu.cpp"
    src.write_text("\n".join(tu))
    r = subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-Wno-unused-variable", "-Wno-unused-function", "-I",
                        os.path.join(ROOT, "include"), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    # the check has teeth: a call with an argument too few and a misspelt mirror member are refused
    for good, bad in (("&s2m_add, &s2m_nodown)", "&s2m_add)"), ("s2m_mirror.size()", "s2m_mirror.ids.size()")):
        text = "\n".join(tu)
        assert good in text
        src.write_text(text.replace(good, bad))
        r = subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)], capture_output=True, text=True)
        assert r.returncode != 0, bad
