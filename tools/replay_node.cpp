// replay_node.cpp -- ROS-free replay of the lio_laserMapping per-scan loop through the C ABI.
//
// Links only libdaliti_s2m.so (plain g++; no HIP, ROS, PCL or Eigen headers).  It stands where
// eskf_lio/src/laserMapping.cpp's `while (sync_packages(Measures))` body stands (:731-1290) and consumes,
// per frame, exactly what that body has in hand after `p_imu->Process` up to the undistortion loop:
//   * the serialised sensor_msgs/PointCloud2 of /laser_cloud_surf (pcl::PointXYZINormal records,
//     feature_extract.cpp:335-346) -- parsed with include/daliti_s2m_wire.h,
//   * IMUpose (IMU_Processing.hpp:224, :310) and the propagated state + covariance (the sequential IMU
//     forward propagation :226-323 stays on the host side of the boundary and is an input here),
// and writes what the node writes:
//   * Log/mat_out.txt rows, one per ESKF iteration (:936-937), with the same iostream formatting,
//   * /cloud_effected (laserCloudOri in the world frame, :1213-1227) as serialised PointCloud2 (PointXYZI),
//   * the map after every frame (/Laser_map, :1170-1175, 1229-1235) as PointCloud2 (PointXYZINormal): a host mirror kept
//     up to date from the engine's change log (include/daliti_s2m_mirror.hpp), not a flatten of the device map per frame,
//   * odometry.txt: time, position, rotation, iterations, last effct_feat_num, map size (full precision).
// The TIS / zeta fusion (:1105-1129), the TIS fallback (:1054-1063) and every publisher stay in the node.
//
// Stream format ("S2MREPL1", little-endian):
//   char magic[8]; uint32 n_frames; int32 max_iter; int32 extrinsic_est_en; int32 feat_threshold;
//   double filter_size_surf, filter_size_map, cube_len;
//   per frame: uint32 n_imu; uint32 msg_len; double state[36]; double P[576]; double imu[n_imu][22];
//              uint8 msg[msg_len] (padded with zeros to a multiple of 8 bytes)
//
// usage: replay_node <stream.bin> <out_dir>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <string>
#include <vector>

#include "daliti_s2m.h"
#include "daliti_s2m_mirror.hpp"
#include "daliti_s2m_wire.h"

#define CK(call)                                                                                     \
    do {                                                                                             \
        const int rc_ = (call);                                                                      \
        if (rc_ != S2M_OK) {                                                                         \
            std::fprintf(stderr, "%s -> %d (%s) %s\n", #call, rc_, s2m_strerror(rc_), eng ? s2m_last_error(eng) : ""); \
            return rc_ == S2M_ERR_NO_DEVICE ? 2 : 1;                                                 \
        }                                                                                            \
    } while (0)

namespace {
struct Reader {
    std::vector<uint8_t> buf;
    size_t at = 0;
    bool ok = true;
    template <class T>
    T get()
    {
        T v{};
        if (at + sizeof(T) > buf.size()) { ok = false; return v; }
        std::memcpy(&v, buf.data() + at, sizeof(T));
        at += sizeof(T);
        return v;
    }
    const uint8_t *take(size_t n)
    {
        if (at + n > buf.size()) { ok = false; return nullptr; }
        const uint8_t *p = buf.data() + at;
        at += n;
        return p;
    }
};

// pointBodyToWorld / RGBpointBodyToWorld (laserMapping.cpp:260-269, 283-291): double math, float result
void body_to_world(const double x[S2M_STATE_DOUBLES], const float pb[3], float pw[3])
{
    const double *R = x, *t = x + 9, *RLI = x + 12, *TLI = x + 21;
    const double p[3] = {(double)pb[0], (double)pb[1], (double)pb[2]};
    double q[3], g[3];
    for (int i = 0; i < 3; ++i) q[i] = ((RLI[3 * i] * p[0] + RLI[3 * i + 1] * p[1]) + RLI[3 * i + 2] * p[2]) + TLI[i];
    for (int i = 0; i < 3; ++i) g[i] = ((R[3 * i] * q[0] + R[3 * i + 1] * q[1]) + R[3 * i + 2] * q[2]) + t[i];
    for (int i = 0; i < 3; ++i) pw[i] = (float)g[i];
}

// one frame of the stream, parsed: what the loop body has in hand after `p_imu->Process` (:731-773)
struct Frame {
    double state[S2M_STATE_DOUBLES], P[S2M_DIM * S2M_DIM];
    std::vector<s2m_imu_pose> poses;
    std::vector<float> blob;  // the point records, realigned when the message body starts at an odd offset
    s2m_pc2_view v;
    const float *pts = nullptr;
    int64_t stride = 0;
    int32_t oa = 0, ob = 0;
};
// 0 = parsed, 65 = malformed (message on stderr)
int parse_frame(Reader &rd, uint32_t f, Frame &fr)
{
    const uint32_t n_imu = rd.get<uint32_t>(), msg_len = rd.get<uint32_t>();
    const uint8_t *ps = rd.take(sizeof(fr.state)), *pP = rd.take(sizeof(fr.P));
    const uint8_t *pimu = rd.take((size_t)n_imu * sizeof(s2m_imu_pose));
    const uint8_t *pmsg = rd.take(((size_t)msg_len + 7) & ~(size_t)7);
    if (!rd.ok) { std::fprintf(stderr, "truncated frame %u\n", f); return 65; }
    std::memcpy(fr.state, ps, sizeof(fr.state));
    std::memcpy(fr.P, pP, sizeof(fr.P));
    fr.poses.resize(n_imu);
    std::memcpy(fr.poses.data(), pimu, (size_t)n_imu * sizeof(s2m_imu_pose));
    // pcl::fromROSMsg(*(lidar_buffer.front()), *(meas.lidar)) (:545) without PCL
    if (s2m_pc2_parse(pmsg, msg_len, &fr.v, nullptr) != S2M_WIRE_OK) { std::fprintf(stderr, "bad PointCloud2 in frame %u\n", f); return 65; }
    int wrc = s2m_pc2_scan_args(&fr.v, &fr.pts, &fr.stride, &fr.oa, &fr.ob);
    if (wrc == S2M_WIRE_UNALIGNED) {  // the blob starts wherever the header ends: realign it once
        fr.blob.resize((fr.v.data_len + 3) / 4);
        std::memcpy(fr.blob.data(), fr.v.data, fr.v.data_len);
        fr.v.data = reinterpret_cast<const uint8_t *>(fr.blob.data());
        wrc = s2m_pc2_scan_args(&fr.v, &fr.pts, &fr.stride, &fr.oa, &fr.ob);
    }
    if (wrc != S2M_WIRE_OK) {
        std::fprintf(stderr, "frame %u: /laser_cloud_surf layout not usable (need float32 x,y,z,normal_x,normal_z)\n", f);
        return 65;
    }
    return 0;
}

void append_msg(std::vector<uint8_t> &out, int kind, uint32_t seq, double stamp, const void *rec, uint32_t n)
{
    const size_t sz = s2m_pc2_serialized_size(kind, n, "camera_init");
    const size_t at = out.size();
    out.resize(at + 4 + sz);
    s2m_wire_put_u32(out.data() + at, (uint32_t)sz);
    s2m_pc2_write(out.data() + at + 4, sz, kind, seq, stamp, "camera_init", rec, n);
}
}  // namespace

int main(int argc, char **argv)
{
    s2m_engine *eng = nullptr;
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <stream.bin> <out_dir>\n", argv[0]);
        return 64;
    }
    Reader rd;
    {
        std::ifstream f(argv[1], std::ios::binary);
        if (!f) { std::fprintf(stderr, "cannot open %s\n", argv[1]); return 66; }
        rd.buf.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    }
    const uint8_t *magic = rd.take(8);
    if (!magic || std::memcmp(magic, "S2MREPL1", 8) != 0) { std::fprintf(stderr, "not an S2MREPL1 stream\n"); return 65; }
    const uint32_t n_frames = rd.get<uint32_t>();
    const int32_t max_iter = rd.get<int32_t>(), extrinsic = rd.get<int32_t>(), feat_thr = rd.get<int32_t>();
    const double fs_surf = rd.get<double>(), fs_map = rd.get<double>(), cube_len = rd.get<double>();
    if (!rd.ok) { std::fprintf(stderr, "truncated header\n"); return 65; }

    s2m_config cfg;
    s2m_config_default(&cfg);
    cfg.max_iter = max_iter;                 // mapping/max_iteration (:656)
    cfg.extrinsic_est_en = extrinsic;        // mapping/extrinsic_est_en (:662)
    cfg.feat_threshold = feat_thr;           // dynamic_effect_featurepoints_threshold (:427-430)
    CK(s2m_create(&cfg, &eng));

    const std::string dir = argv[2];
    std::ofstream fout_out(dir + "/mat_out.txt"), fodom(dir + "/odometry.txt");
    std::vector<uint8_t> effected_msgs;
    if (!fout_out || !fodom) { std::fprintf(stderr, "cannot write into %s\n", argv[2]); return 73; }
    fodom << std::setprecision(17);

    bool first_scan = true, have_map = false;
    s2m_map_mirror mirror;
    // /Laser_map is streamed to its file as it is produced (one whole-map message per frame: kept in memory it would be
    // frames x map points x 48 bytes), through buffers that are reused
    std::vector<uint8_t> map_msg;
    std::vector<float> map_rec;
    std::ofstream fm;
    double first_lidar_time = 0.0;
    std::vector<float> xyz, world;
    std::vector<int32_t> ridx;
    // One frame ahead: a replay has the next message (and the IMU poses propagated for it, an input of this stream) in hand
    // while the current one is registered, so its records travel (s2m_scan_prefetch_raw) and its undistortion + voxel filter
    // run (s2m_scan_prepare_raw, once this frame's update is done) beside this frame's map update; the next iteration's
    // s2m_scan_set_from_raw picks the prepared scan up.  Same results as computing every frame in place.
    Frame frames[2];
    bool have_next = false;
    if (n_frames > 0) {
        const int prc = parse_frame(rd, 0, frames[0]);
        if (prc) return prc;
    }
    for (uint32_t f = 0; f < n_frames; ++f) {
        Frame &cur = frames[f & 1], &nxt = frames[(f + 1) & 1];
        have_next = false;
        if (f + 1 < n_frames) {
            const int prc = parse_frame(rd, f + 1, nxt);
            if (prc) return prc;
            have_next = true;
        }
        double *state = cur.state, *P = cur.P;
        std::vector<s2m_imu_pose> &poses = cur.poses;
        const uint32_t n_imu = (uint32_t)poses.size();
        const s2m_pc2_view &v = cur.v;
        const float *pts = cur.pts;
        const int64_t stride = cur.stride;
        const int32_t oa = cur.oa, ob = cur.ob;
        const double lidar_beg_time = (double)v.stamp_sec + 1e-9 * (double)v.stamp_nsec;  // :546
        if (first_scan) { first_lidar_time = lidar_beg_time; first_scan = false; }          // :737-741
        // observation_end_time = lidar_beg_time + points.back().normal_z (:547): the stamp of everything published
        const double obs_end = v.n_points ? lidar_beg_time + (double)pts[(int64_t)(v.n_points - 1) * stride + ob] : lidar_beg_time;

        // UndistortPcl's sort + backward loop (IMU_Processing.hpp:215-216, 333-370) and the surf voxel filter
        // (:775-776) on the device; the result is feats_down
        int64_t n_down = 0;
        CK(s2m_scan_set_from_raw(eng, pts, stride, v.n_points, oa, ob, poses.data(), (int32_t)n_imu, state, (float)fs_surf,
                                 0, &n_down));
        if (have_next) CK(s2m_scan_prefetch_raw(eng, nxt.pts, nxt.stride, nxt.v.n_points, nxt.oa, nxt.ob));
        // lasermap_fov_segment (:772): pos_lid = pos_end + rot_end * T_L_I (:753)
        double pos_lid[3];
        for (int i = 0; i < 3; ++i)
            pos_lid[i] = state[9 + i] + ((state[3 * i] * state[21] + state[3 * i + 1] * state[22]) + state[3 * i + 2] * state[23]);
        CK(s2m_fov_segment(eng, pos_lid, cube_len, nullptr, nullptr, nullptr));
        xyz.resize((size_t)std::max<int64_t>(n_down, 1) * 3);
        int64_t got = 0;
        CK(s2m_scan_get(eng, xyz.data(), n_down, &got));
        if (!have_map) {  // :780-793: the first usable scan seeds the map with its world-frame points
            if (n_down > 5) {
                world.resize((size_t)n_down * 3);
                for (int64_t i = 0; i < n_down; ++i) body_to_world(state, &xyz[3 * i], &world[3 * i]);
                CK(s2m_map_build(eng, world.data(), 3, n_down, 0));
                have_map = true;
            }
            fodom << lidar_beg_time - first_lidar_time << " seed " << n_down << "\n";
            continue;
        }
        double x[S2M_STATE_DOUBLES];
        std::memcpy(x, state, sizeof(x));  // state_propagat(state) (:752)
        s2m_iter_log log;
        CK(s2m_iterated_update(eng, x, state, P, &log));
        // Log/mat_out.txt (:936-937), one row per iteration; flg_EKF_converged is the value BEFORE this iteration's
        // update (0 at iteration 0, :815), RECV_LIO_FAIL_FLAG / recv_n belong to the node (0 here)
        for (int it = 0; it < log.iters; ++it) {
            const int effct = log.effct[it];
            const double res_mean_last = log.total_residual[it] / effct;  // :932 (NaN when effct == 0, like the node)
            const int conv_before = it ? log.conv[it - 1] : 0;
            const int stop = (it == log.iters - 1) ? log.ekf_stop : 0;     // a stop ends the loop (:1095-1101)
            fout_out << std::setw(10) << lidar_beg_time - first_lidar_time << " " << effct << " " << res_mean_last << " "
                     << (conv_before != 0) << " " << (stop != 0) << " " << n_down << " " << 0 << " " << 0 << std::endl;
        }
        // /cloud_effected (:1213-1227): laserCloudOri = the effective points of the LAST pass in index order
        int64_t m_eff = 0;
        ridx.resize((size_t)std::max<int64_t>(n_down, 1));
        CK(s2m_get_rows(eng, nullptr, nullptr, ridx.data(), n_down, &m_eff));
        std::vector<float> rec((size_t)m_eff * S2M_PXYZI_FLOATS, 0.0f);
        for (int64_t i = 0; i < m_eff; ++i) body_to_world(x, &xyz[3 * (size_t)ridx[i]], &rec[(size_t)i * S2M_PXYZI_FLOATS]);
        append_msg(effected_msgs, 0, f, obs_end, rec.data(), (uint32_t)m_eff);
        // map_incremental (:1165-1168)
        int64_t n_add = 0, n_nodown = 0, m_map = 0;
        if (have_next)  // the next frame's front half beside this frame's map update
            CK(s2m_scan_prepare_raw(eng, nxt.pts, nxt.stride, nxt.v.n_points, nxt.oa, nxt.ob, nxt.poses.data(), (int32_t)nxt.poses.size(),
                                    nxt.state, (float)fs_surf));
        if (!log.ekf_stop) CK(s2m_map_incremental(eng, x, fs_map, 1, &n_add, &n_nodown));
        CK(s2m_map_size(eng, &m_map));
        {   // ikdtree.flatten -> featsFromMap -> /Laser_map (:1170-1175, 1229-1235), every frame: the mirror takes this frame's
            // changes (a few thousand points) and is published as it stands
            CK(mirror.update(eng));
            const size_t m = (size_t)mirror.size();
            map_rec.assign(m * S2M_PXYZIN_FLOATS, 0.0f);
            size_t i = 0;
            mirror.for_each([&](uint32_t, float px, float py, float pz) {
                float *r = &map_rec[i++ * S2M_PXYZIN_FLOATS];
                r[0] = px; r[1] = py; r[2] = pz;
            });
            map_msg.clear();
            append_msg(map_msg, 1, f, obs_end, map_rec.data(), (uint32_t)m);
            if (!fm.is_open()) fm.open(dir + "/laser_map.pc2s", std::ios::binary);
            fm.write(reinterpret_cast<const char *>(map_msg.data()), (std::streamsize)map_msg.size());
        }
        fodom << lidar_beg_time - first_lidar_time;
        for (int i = 0; i < 3; ++i) fodom << " " << x[9 + i];
        for (int i = 0; i < 9; ++i) fodom << " " << x[i];
        fodom << " " << log.iters << " " << log.effct[log.iters - 1] << " " << (int)log.ekf_stop << " " << n_down << " " << m_map
              << " " << (n_add + n_nodown) << "\n";
    }
    {
        std::ofstream fe(dir + "/cloud_effected.pc2s", std::ios::binary);
        fe.write(reinterpret_cast<const char *>(effected_msgs.data()), (std::streamsize)effected_msgs.size());
    }
    if (have_map) {
        if (!fm.is_open()) fm.open(dir + "/laser_map.pc2s", std::ios::binary);   // (no frame behind the seed: an empty file, as before)
        fm.close();
        std::fprintf(stderr, "replay_node: /Laser_map followed through the change log: %lld whole-map fetches in %u frames\n",
                     (long long)mirror.resyncs, n_frames);
    }
    s2m_destroy(eng);
    return 0;
}
