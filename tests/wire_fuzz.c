/* Test helper (plain C99, built with -fsanitize=address): the PointCloud2 parser of include/daliti_s2m_wire.h must
 * never read past the bytes it was given -- every truncation of a valid message and a few thousand corrupted copies go
 * through it in exactly-sized heap buffers, so an over-read is an ASan report.  usage: wire_fuzz <message file> */
#include <stdio.h>
#include <stdlib.h>

#include "daliti_s2m_wire.h"

static int check(const uint8_t *buf, size_t len)
{
    s2m_pc2_view v;
    size_t used = 0;
    const int rc = s2m_pc2_parse(buf, len, &v, &used);
    if (rc != 0) return 0;
    /* accepted: everything it points at must lie inside the buffer */
    if (used > len) return 1;
    if (v.data < buf || v.data + v.data_len > buf + len) return 2;
    if ((size_t)v.n_points * v.point_step > v.data_len) return 3;
    if (v.frame_id && ((const uint8_t *)v.frame_id < buf || (const uint8_t *)v.frame_id + v.frame_id_len > buf + len)) return 4;
    const float *pts = NULL;
    int64_t stride = 0;
    int32_t oa = -1, ob = -1;
    const int rs = s2m_pc2_scan_args(&v, &pts, &stride, &oa, &ob);
    if (rs == 0 && v.n_points) {
        /* the offsets handed to the C ABI address floats inside one record */
        if (oa < 0 || ob < 0 || (uint32_t)(oa + 1) * 4u > v.point_step || (uint32_t)(ob + 1) * 4u > v.point_step) return 5;
        /* ... and the pointers themselves: read what s2m_scan_set_from_raw / s2m_undistort would read of the FIRST and
         * the LAST record (x, y, z and the two time fields) -- an offset that sticks out of the blob is an ASan report */
        volatile float sink = 0.0f;
        const int64_t recs[2] = {0, (int64_t)v.n_points - 1};
        for (int r = 0; r < 2; ++r) {
            const float *rec = pts + recs[r] * stride;
            sink += rec[0]; sink += rec[1]; sink += rec[2]; sink += rec[oa]; sink += rec[ob];
        }
        (void)sink;
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 2) return 64;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 66;
    fseek(f, 0, SEEK_END);
    const long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *msg = (uint8_t *)malloc((size_t)len);
    if (fread(msg, 1, (size_t)len, f) != (size_t)len) return 66;
    fclose(f);
    if (check(msg, (size_t)len) != 0) { printf("the valid message fails its own checks\n"); return 1; }
    long accepted_cuts = 0, accepted_flips = 0;
    for (long cut = 0; cut < len; ++cut) {            /* every truncation, in a buffer of exactly that size */
        uint8_t *b = (uint8_t *)malloc((size_t)cut ? (size_t)cut : 1);
        memcpy(b, msg, (size_t)cut);
        s2m_pc2_view v;
        size_t used = 0;
        if (s2m_pc2_parse(b, (size_t)cut, &v, &used) == 0) {
            ++accepted_cuts;
            const int bad = check(b, (size_t)cut);
            if (bad) { printf("cut %ld accepted inconsistently (%d)\n", cut, bad); return 2; }
        }
        free(b);
    }
    uint32_t s = 12345u;
    const long head = len < 400 ? len : 400;           /* the header is where the lengths and counts live */
    for (int it = 0; it < 4000; ++it) {
        uint8_t *b = (uint8_t *)malloc((size_t)len);
        memcpy(b, msg, (size_t)len);
        const int flips = 1 + (int)((s >> 28) & 3u);
        for (int k = 0; k < flips; ++k) {
            s = s * 1664525u + 1013904223u;
            const long pos = (long)((s >> 8) % (uint32_t)head);
            s = s * 1664525u + 1013904223u;
            b[pos] = (uint8_t)(s >> 16);
        }
        const int bad = check(b, (size_t)len);
        if (bad) { printf("corrupted copy %d accepted inconsistently (%d)\n", it, bad); return 3; }
        s2m_pc2_view v;
        if (s2m_pc2_parse(b, (size_t)len, &v, NULL) == 0) ++accepted_flips;
        free(b);
    }
    printf("ok: %ld truncations (%ld accepted), 4000 corrupted copies (%ld accepted)\n", len, accepted_cuts, accepted_flips);
    free(msg);
    return 0;
}
