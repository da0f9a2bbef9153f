/* daliti_s2m_wire.h -- the byte formats either side of the scan-to-map path.
 *
 * Header-only C (C99 / C++), no dependencies: what a ROS-side shim needs to hand the node's messages
 * to the C ABI of daliti_s2m.h without PCL, and to serialise the path's outputs again.
 *
 *   in   /laser_cloud_surf : sensor_msgs/PointCloud2 of pcl::PointXYZINormal, point_step 48, written by
 *                            feature_extract.cpp:335-346 (normal_x = time ratio t / timespan, normal_y =
 *                            ring, normal_z = timespan in seconds) and consumed at laserMapping.cpp:544-546
 *                            (lidar_beg_time = header.stamp, observation_end = beg + back().normal_z) and
 *                            IMU_Processing.hpp:352 (offset time = normal_x * normal_z)
 *   out  /cloud_effected, /cloud_registered : PointCloud2 of pcl::PointXYZI, point_step 32
 *                            (laserMapping.cpp:1183-1227)
 *   out  /Laser_map        : PointCloud2 of pcl::PointXYZINormal (laserMapping.cpp:1229-1235)
 *
 * The serialised form is ROS 1's: little-endian, string = uint32 length + bytes, array = uint32 count +
 * elements; sensor_msgs/PointCloud2 = Header{uint32 seq, uint32 secs, uint32 nsecs, string frame_id},
 * uint32 height, uint32 width, PointField[]{string name, uint32 offset, uint8 datatype, uint32 count},
 * uint8 is_bigendian, uint32 point_step, uint32 row_step, uint8[] data, uint8 is_dense.
 */
#ifndef DALITI_S2M_WIRE_H
#define DALITI_S2M_WIRE_H

#include <stddef.h>
#include <stdint.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

/* pcl::PointXYZINormal in memory and on the wire: {x y z pad | normal_x normal_y normal_z pad |
 * intensity curvature pad pad}, 48 bytes; indices in floats */
enum {
    S2M_PXYZIN_BYTES = 48, S2M_PXYZIN_FLOATS = 12,
    S2M_PXYZIN_X = 0, S2M_PXYZIN_Y = 1, S2M_PXYZIN_Z = 2,
    S2M_PXYZIN_NORMAL_X = 4,  /* time ratio within the sweep  (feature_extract.cpp:345) */
    S2M_PXYZIN_NORMAL_Y = 5,  /* ring                          (feature_extract.cpp:343) */
    S2M_PXYZIN_NORMAL_Z = 6,  /* sweep time span in seconds    (feature_extract.cpp:344) */
    S2M_PXYZIN_INTENSITY = 8, S2M_PXYZIN_CURVATURE = 9
};
/* pcl::PointXYZI: {x y z pad | intensity pad pad pad}, 32 bytes */
enum { S2M_PXYZI_BYTES = 32, S2M_PXYZI_FLOATS = 8, S2M_PXYZI_INTENSITY = 4 };
enum { S2M_PC2_FLOAT32 = 7 };  /* sensor_msgs/PointField.FLOAT32 */

enum { S2M_WIRE_OK = 0, S2M_WIRE_TRUNCATED = -1, S2M_WIRE_BAD = -2, S2M_WIRE_UNSUPPORTED = -3, S2M_WIRE_UNALIGNED = -4 };

typedef struct {
    uint32_t seq, stamp_sec, stamp_nsec;
    const char *frame_id; uint32_t frame_id_len;  /* not NUL-terminated: points into the message */
    uint32_t height, width, point_step, row_step;
    uint8_t is_bigendian, is_dense;
    const uint8_t *data; uint32_t data_len;
    uint32_t n_points;                            /* width * height */
    /* byte offsets of the FLOAT32 fields this path reads, -1 when the message lacks them */
    int32_t off_x, off_y, off_z, off_normal_x, off_normal_y, off_normal_z, off_intensity, off_curvature;
} s2m_pc2_view;

static inline int s2m_wire_u32(const uint8_t **p, const uint8_t *end, uint32_t *v)
{
    if ((size_t)(end - *p) < 4) return S2M_WIRE_TRUNCATED;
    *v = (uint32_t)(*p)[0] | ((uint32_t)(*p)[1] << 8) | ((uint32_t)(*p)[2] << 16) | ((uint32_t)(*p)[3] << 24);
    *p += 4;
    return S2M_WIRE_OK;
}
static inline int s2m_wire_bytes(const uint8_t **p, const uint8_t *end, uint32_t n, const uint8_t **out)
{
    if ((size_t)(end - *p) < (size_t)n) return S2M_WIRE_TRUNCATED;
    *out = *p;
    *p += n;
    return S2M_WIRE_OK;
}

/* Parses one serialised sensor_msgs/PointCloud2; the view points into `buf`.  *consumed (optional) = bytes
 * of this message, so that concatenated messages can be walked. */
static inline int s2m_pc2_parse(const uint8_t *buf, size_t len, s2m_pc2_view *v, size_t *consumed)
{
    const uint8_t *p = buf, *end = buf + len, *s = NULL;
    uint32_t nfields = 0;
    uint64_t n = 0;
    int rc;
    if (!buf || !v) return S2M_WIRE_BAD;
    memset(v, 0, sizeof(*v));
    v->off_x = v->off_y = v->off_z = v->off_normal_x = v->off_normal_y = v->off_normal_z = -1;
    v->off_intensity = v->off_curvature = -1;
    if ((rc = s2m_wire_u32(&p, end, &v->seq)) || (rc = s2m_wire_u32(&p, end, &v->stamp_sec)) ||
        (rc = s2m_wire_u32(&p, end, &v->stamp_nsec)) || (rc = s2m_wire_u32(&p, end, &v->frame_id_len)) ||
        (rc = s2m_wire_bytes(&p, end, v->frame_id_len, &s)))
        return rc;
    v->frame_id = (const char *)s;
    if ((rc = s2m_wire_u32(&p, end, &v->height)) || (rc = s2m_wire_u32(&p, end, &v->width)) ||
        (rc = s2m_wire_u32(&p, end, &nfields)))
        return rc;
    if (nfields > 1024) return S2M_WIRE_BAD;
    for (uint32_t f = 0; f < nfields; ++f) {
        uint32_t nl = 0, off = 0, cnt = 0;
        const uint8_t *name = NULL, *dt = NULL;
        if ((rc = s2m_wire_u32(&p, end, &nl)) || (rc = s2m_wire_bytes(&p, end, nl, &name)) ||
            (rc = s2m_wire_u32(&p, end, &off)) || (rc = s2m_wire_bytes(&p, end, 1, &dt)) ||
            (rc = s2m_wire_u32(&p, end, &cnt)))
            return rc;
        if (*dt != S2M_PC2_FLOAT32 || cnt != 1) continue;  /* ring / t of the raw sensor clouds etc. */
#define S2M_WIRE_FIELD(lit, member) \
        if (nl == sizeof(lit) - 1 && memcmp(name, lit, sizeof(lit) - 1) == 0) v->member = (int32_t)off
        S2M_WIRE_FIELD("x", off_x); S2M_WIRE_FIELD("y", off_y); S2M_WIRE_FIELD("z", off_z);
        S2M_WIRE_FIELD("normal_x", off_normal_x); S2M_WIRE_FIELD("normal_y", off_normal_y);
        S2M_WIRE_FIELD("normal_z", off_normal_z); S2M_WIRE_FIELD("intensity", off_intensity);
        S2M_WIRE_FIELD("curvature", off_curvature);
#undef S2M_WIRE_FIELD
    }
    if ((rc = s2m_wire_bytes(&p, end, 1, &s))) return rc;
    v->is_bigendian = *s;
    if ((rc = s2m_wire_u32(&p, end, &v->point_step)) || (rc = s2m_wire_u32(&p, end, &v->row_step)) ||
        (rc = s2m_wire_u32(&p, end, &v->data_len)) || (rc = s2m_wire_bytes(&p, end, v->data_len, &v->data)) ||
        (rc = s2m_wire_bytes(&p, end, 1, &s)))
        return rc;
    v->is_dense = *s;
    n = (uint64_t)v->width * (uint64_t)v->height;          /* 64-bit: width * height may not wrap */
    if (n > 0xffffffffull) return S2M_WIRE_BAD;
    if (v->point_step == 0 ? n != 0 : n * v->point_step > v->data_len) return S2M_WIRE_BAD;
    {   /* every named float32 field lies inside a record: the offsets come from the message and are handed on as
         * pointers (s2m_pc2_scan_args), so a field that sticks out of point_step would read past the blob */
        const int32_t offs[8] = {v->off_x, v->off_y, v->off_z, v->off_normal_x, v->off_normal_y, v->off_normal_z,
                                 v->off_intensity, v->off_curvature};
        for (int k = 0; k < 8; ++k)
            if (offs[k] >= 0 && (uint64_t)(uint32_t)offs[k] + 4u > (uint64_t)v->point_step) return S2M_WIRE_BAD;
    }
    v->n_points = (uint32_t)n;
    if (consumed) *consumed = (size_t)(p - buf);
    return S2M_WIRE_OK;
}

/* Arguments for s2m_scan_set_from_raw / s2m_undistort / s2m_scan_set from a /laser_cloud_surf view: records
 * start at x (y, z follow), stride in floats, off_a / off_b = normal_x / normal_z relative to x, in floats.
 * The data blob must be 4-byte aligned in memory: S2M_WIRE_UNALIGNED asks the caller to copy v->data into an
 * aligned buffer, point v->data at it and call again (inside a serialised message the blob starts wherever
 * the header and field list end). */
static inline int s2m_pc2_scan_args(const s2m_pc2_view *v, const float **points, int64_t *stride_floats,
                                    int32_t *off_a, int32_t *off_b)
{
    if (!v || !points || !stride_floats) return S2M_WIRE_BAD;
    if (v->is_bigendian || v->point_step % 4 != 0) return S2M_WIRE_UNSUPPORTED;
    if (v->off_x < 0 || v->off_y != v->off_x + 4 || v->off_z != v->off_x + 8 || v->off_x % 4 != 0) return S2M_WIRE_UNSUPPORTED;
    if (((uintptr_t)(v->data + v->off_x)) % 4 != 0) return S2M_WIRE_UNALIGNED;
    *points = (const float *)(const void *)(v->data + v->off_x);
    *stride_floats = v->point_step / 4;
    if (off_a) {
        if (v->off_normal_x < v->off_x || (v->off_normal_x - v->off_x) % 4 != 0) return S2M_WIRE_UNSUPPORTED;
        *off_a = (v->off_normal_x - v->off_x) / 4;
    }
    if (off_b) {
        if (v->off_normal_z < v->off_x || (v->off_normal_z - v->off_x) % 4 != 0) return S2M_WIRE_UNSUPPORTED;
        *off_b = (v->off_normal_z - v->off_x) / 4;
    }
    return S2M_WIRE_OK;
}

/* ---- writers: what pcl::toROSMsg produces for the two point types the node publishes ------------- */
static inline uint8_t *s2m_wire_put_u32(uint8_t *p, uint32_t v)
{
    p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
    return p + 4;
}
static inline uint8_t *s2m_wire_put_field(uint8_t *p, const char *name, uint32_t off)
{
    const uint32_t nl = (uint32_t)strlen(name);
    p = s2m_wire_put_u32(p, nl);
    memcpy(p, name, nl); p += nl;
    p = s2m_wire_put_u32(p, off);
    *p++ = S2M_PC2_FLOAT32;
    return s2m_wire_put_u32(p, 1);
}
/* kind 0: pcl::PointXYZI records (32 B), kind 1: pcl::PointXYZINormal records (48 B) */
static inline size_t s2m_pc2_serialized_size(int kind, uint32_t n_points, const char *frame_id)
{
    const size_t fields = kind ? (3 * (4 + 1 + 9) + 3 * (4 + 8 + 9) + (4 + 9 + 9) + (4 + 9 + 9))
                               : (3 * (4 + 1 + 9) + (4 + 9 + 9));
    return 12 + 4 + strlen(frame_id) + 8 + 4 + fields + 1 + 8 + 4 +
           (size_t)n_points * (size_t)(kind ? (int)S2M_PXYZIN_BYTES : (int)S2M_PXYZI_BYTES) + 1;
}
/* stamp as ros::Time().fromSec(t) (laserMapping.cpp:1199 etc.); returns bytes written, 0 if cap is too small */
static inline size_t s2m_pc2_write(uint8_t *dst, size_t cap, int kind, uint32_t seq, double stamp_sec,
                                   const char *frame_id, const void *records, uint32_t n_points)
{
    const size_t need = s2m_pc2_serialized_size(kind, n_points, frame_id);
    const uint32_t step = (uint32_t)(kind ? (int)S2M_PXYZIN_BYTES : (int)S2M_PXYZI_BYTES);
    const uint32_t fl = (uint32_t)strlen(frame_id);
    uint32_t sec = (uint32_t)stamp_sec;
    double frac = (stamp_sec - (double)sec) * 1e9 + 0.5;
    uint32_t nsec = (uint32_t)frac;
    uint8_t *p = dst;
    if (!dst || cap < need) return 0;
    if (nsec >= 1000000000u) { nsec -= 1000000000u; ++sec; }
    p = s2m_wire_put_u32(p, seq); p = s2m_wire_put_u32(p, sec); p = s2m_wire_put_u32(p, nsec);
    p = s2m_wire_put_u32(p, fl); memcpy(p, frame_id, fl); p += fl;
    p = s2m_wire_put_u32(p, 1); p = s2m_wire_put_u32(p, n_points);   /* unorganised cloud: height 1 */
    p = s2m_wire_put_u32(p, kind ? 8u : 4u);
    p = s2m_wire_put_field(p, "x", 0); p = s2m_wire_put_field(p, "y", 4); p = s2m_wire_put_field(p, "z", 8);
    if (kind) {
        p = s2m_wire_put_field(p, "normal_x", 16); p = s2m_wire_put_field(p, "normal_y", 20);
        p = s2m_wire_put_field(p, "normal_z", 24); p = s2m_wire_put_field(p, "intensity", 32);
        p = s2m_wire_put_field(p, "curvature", 36);
    } else {
        p = s2m_wire_put_field(p, "intensity", 16);
    }
    *p++ = 0;                                                   /* little-endian */
    p = s2m_wire_put_u32(p, step); p = s2m_wire_put_u32(p, step * n_points);
    p = s2m_wire_put_u32(p, step * n_points);
    if (n_points) memcpy(p, records, (size_t)step * n_points);
    p += (size_t)step * n_points;
    *p++ = 1;                                                   /* is_dense */
    return (size_t)(p - dst);
}

#ifdef __cplusplus
}
#endif
#endif /* DALITI_S2M_WIRE_H */
