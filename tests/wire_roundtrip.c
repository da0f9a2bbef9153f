/* Test helper (plain C99): parse a serialised PointCloud2 with include/daliti_s2m_wire.h, print what the
 * shim would pass to the C ABI, and serialise the records again.  usage: wire_roundtrip <in> <out> <kind> */
#include <stdio.h>
#include <stdlib.h>

#include "daliti_s2m_wire.h"

int main(int argc, char **argv)
{
    if (argc < 4) return 64;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 66;
    fseek(f, 0, SEEK_END);
    long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *buf = (uint8_t *)malloc((size_t)len + 8);
    if (fread(buf, 1, (size_t)len, f) != (size_t)len) return 66;
    fclose(f);
    s2m_pc2_view v;
    size_t used = 0;
    int rc = s2m_pc2_parse(buf, (size_t)len, &v, &used);
    if (rc) { printf("parse rc %d\n", rc); return 1; }
    const float *pts = NULL;
    int64_t stride = 0;
    int32_t oa = -1, ob = -1;
    float *aligned = NULL;
    rc = s2m_pc2_scan_args(&v, &pts, &stride, &oa, &ob);
    if (rc == S2M_WIRE_UNALIGNED) {
        aligned = (float *)malloc(v.data_len + 4);
        memcpy(aligned, v.data, v.data_len);
        v.data = (const uint8_t *)aligned;
        rc = s2m_pc2_scan_args(&v, &pts, &stride, &oa, &ob);
    }
    printf("used %zu n %u step %u stride %lld oa %d ob %d rc %d sec %u nsec %u frame %.*s\n", used, v.n_points, v.point_step,
           (long long)stride, oa, ob, rc, v.stamp_sec, v.stamp_nsec, (int)v.frame_id_len, v.frame_id);
    if (rc == 0 && v.n_points) printf("first %.9g %.9g %.9g t %.9g span %.9g\n", pts[0], pts[1], pts[2], pts[oa], pts[ob]);
    const int kind = atoi(argv[3]);
    char frame[64] = {0};
    memcpy(frame, v.frame_id, v.frame_id_len < 63 ? v.frame_id_len : 63);
    const size_t sz = s2m_pc2_serialized_size(kind, v.n_points, frame);
    uint8_t *out = (uint8_t *)malloc(sz);
    const size_t w = s2m_pc2_write(out, sz, kind, v.seq, (double)v.stamp_sec + 1e-9 * (double)v.stamp_nsec, frame, v.data, v.n_points);
    f = fopen(argv[2], "wb");
    fwrite(out, 1, w, f);
    fclose(f);
    /* a truncated message must be reported, never read past */
    s2m_pc2_view t;
    printf("truncated rc %d\n", s2m_pc2_parse(buf, (size_t)len / 2, &t, NULL));
    return 0;
}
