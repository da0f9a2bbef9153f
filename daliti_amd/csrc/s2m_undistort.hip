// s2m_undistort.hip -- per-point motion compensation of a scan on the GPU.
//
// Replaces the backward-propagation loop of ImuProcess::UndistortPcl
// (eskf_lio/src/IMU_Processing.hpp:333-370) and the time sort in front of it (:215-216).  The
// sequential IMU forward / covariance propagation (:226-308) stays on the host; it produces the
// list IMUpose (Pose6D: offset_time, acc, gyr, vel, pos, rot; :224, :310) that is the input here.
//
// Reference loop, restated per point with offset time t = normal_x * normal_z (float product):
//   head = the last IMU pose h <= K-2 with offset_time[h] < t   (the last pose is never a head)
//   no such pose (t <= offset_time[0])  ->  the point is left untouched
//   dt  = t - offset_time[head]
//   R_i = R_head * Exp(gyr_head, dt)                                  (so3_math.h:31-52)
//   T_ei = pos_head + vel_head*dt + 0.5*acc_head*dt*dt - pos_end
//   P   = R_L_I^T * (rot_end^T * (R_i * (R_L_I*P_i + T_L_I) + T_ei) - T_L_I)      (:358)
// Every point is independent given IMUpose, so this is one lane per point with a binary search over
// the <= few dozen poses.  fp64 like the reference; sin/cos come from the device math library and can
// differ from glibc in the last bit, which moves a float output by at most one ulp (tested).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "s2m_device.h"
#include "s2m_kernels.h"

namespace s2m {

#define S2M_TRY(x)                       \
    do {                                 \
        hipError_t e_ = (x);             \
        if (e_ != hipSuccess) return e_; \
    } while (0)

__device__ __forceinline__ float point_time(const float *rec, int off_a, int off_b)
{
    return off_b >= 0 ? rec[off_a] * rec[off_b] : rec[off_a];  // normal_x * normal_z (:352)
}

__device__ __forceinline__ uint32_t time_key(const float *rec, int off_a, int off_b)
{
    const uint32_t b = __float_as_uint(point_time(rec, off_a, off_b));
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);  // order-preserving for any sign
}
// val2 = the identity (the time order of records that ARE in time order: a stable sort leaves them where they stand);
// *unsorted += 1 per wave that holds a record earlier than the one in front of it
__global__ __launch_bounds__(256) void undist_key_kernel(const float *__restrict__ pts, int64_t stride, int64_t n,
                                                         int off_a, int off_b, uint32_t *__restrict__ key,
                                                         uint32_t *__restrict__ val, uint32_t *__restrict__ val2,
                                                         uint32_t *__restrict__ unsorted)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool broken = false;
    if (i < n) {
        const uint32_t k = time_key(pts + i * stride, off_a, off_b);
        key[i] = k;
        val[i] = (uint32_t)i;
        val2[i] = (uint32_t)i;
        broken = i > 0 && time_key(pts + (i - 1) * stride, off_a, off_b) > k;
    }
    const unsigned long long any = __ballot(broken);
    if (any != 0ull && (threadIdx.x & 63) == 0) atomicAdd(unsorted, 1u);
}

__device__ __forceinline__ void m3v(const double *A, const double *v, double *o)
{
    const double t0 = (A[0] * v[0] + A[1] * v[1]) + A[2] * v[2];
    const double t1 = (A[3] * v[0] + A[4] * v[1]) + A[5] * v[2];
    const double t2 = (A[6] * v[0] + A[7] * v[1]) + A[8] * v[2];
    o[0] = t0; o[1] = t1; o[2] = t2;
}
__device__ __forceinline__ void m3tv(const double *A, const double *v, double *o)
{
    const double t0 = (A[0] * v[0] + A[3] * v[1]) + A[6] * v[2];
    const double t1 = (A[1] * v[0] + A[4] * v[1]) + A[7] * v[2];
    const double t2 = (A[2] * v[0] + A[5] * v[1]) + A[8] * v[2];
    o[0] = t0; o[1] = t1; o[2] = t2;
}

// one application of :347-363 to the point (x, y, z) with time t under the head pose ph (22 doubles)
__device__ __forceinline__ void compensate(const double *__restrict__ ph, double t, const Pose &end, float &x, float &y,
                                           float &z)
{
    {
        const double dt = t - ph[0];
        const double *acc = ph + 1, *gyr = ph + 4, *vel = ph + 7, *pos = ph + 10, *R = ph + 13;
        // Exp(gyr, dt), so3_math.h:31-52
        double E[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        const double wn = sqrt((gyr[0] * gyr[0] + gyr[1] * gyr[1]) + gyr[2] * gyr[2]);
        if (wn > 0.0000001) {
            const double ax[3] = {gyr[0] / wn, gyr[1] / wn, gyr[2] / wn};
            const double Kx[9] = {0.0, -ax[2], ax[1], ax[2], 0.0, -ax[0], -ax[1], ax[0], 0.0};
            const double ang = wn * dt;
            const double sn = sin(ang), c1 = 1.0 - cos(ang);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double kk = ((c1 * Kx[r * 3 + 0]) * Kx[0 * 3 + c] + (c1 * Kx[r * 3 + 1]) * Kx[1 * 3 + c]) +
                                      (c1 * Kx[r * 3 + 2]) * Kx[2 * 3 + c];
                    E[r * 3 + c] = (E[r * 3 + c] + sn * Kx[r * 3 + c]) + kk;
                }
        }
        double Ri[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                Ri[r * 3 + c] = (R[r * 3 + 0] * E[0 * 3 + c] + R[r * 3 + 1] * E[1 * 3 + c]) + R[r * 3 + 2] * E[2 * 3 + c];
        double Tei[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) Tei[k] = ((pos[k] + vel[k] * dt) + (0.5 * acc[k]) * dt * dt) - end.t[k];
        const double Pi[3] = {(double)x, (double)y, (double)z};
        double a[3], b[3], c[3], d[3];
        m3v(end.RLI, Pi, a);
#pragma unroll
        for (int k = 0; k < 3; ++k) a[k] = a[k] + end.TLI[k];
        m3v(Ri, a, b);
#pragma unroll
        for (int k = 0; k < 3; ++k) b[k] = b[k] + Tei[k];
        m3tv(end.R, b, c);
#pragma unroll
        for (int k = 0; k < 3; ++k) c[k] = c[k] - end.TLI[k];
        m3tv(end.RLI, c, d);
        x = (float)d[0]; y = (float)d[1]; z = (float)d[2];
    }
}

// poses: K records of 22 doubles {offset_time, acc[3], gyr[3], vel[3], pos[3], rot[9]} in device memory.
// order: optional permutation (sorted position -> input index); out is written in sorted position.
__global__ __launch_bounds__(256) void undistort_kernel(const float *__restrict__ pts, int64_t stride, int64_t n,
                                                        int off_a, int off_b, const uint32_t *__restrict__ order,
                                                        const double *__restrict__ poses, int K, Pose end,
                                                        float *__restrict__ out, uint32_t *__restrict__ perm_out)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const int64_t i = order ? (int64_t)order[s] : s;
    const float *rec = pts + i * stride;
    const float tf = point_time(rec, off_a, off_b);
    const double t = (double)tf;
    float ox = rec[0], oy = rec[1], oz = rec[2];
    // head = last h in [0, K-2] with offset_time[h] < t  (binary search, offset times ascend)
    int lo = 0, hi = K - 2, head = -1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        if (poses[22 * mid] < t) { head = mid; lo = mid + 1; } else { hi = mid - 1; }
    }
    if (head >= 0) {
        compensate(poses + 22 * head, t, end, ox, oy, oz);
        // The reference's `break` at it_pcl == begin() (:366-367) leaves only the inner loop; the outer loop
        // carries on over the earlier heads and compensates the first point of the sorted cloud again -- from
        // its already compensated float coordinates -- for every earlier head whose offset time is below the
        // point's time.  Restated as is (sorted cloud only; sort_by_time == 0 is an extension without it).
        if (s == 0 && order)
            for (int h = head - 1; h >= 0; --h)
                if (t > poses[22 * h]) compensate(poses + 22 * h, t, end, ox, oy, oz);
    }
    out[3 * s] = ox; out[3 * s + 1] = oy; out[3 * s + 2] = oz;
    if (perm_out) perm_out[s] = (uint32_t)i;
}

void free_undist(UndistBuffers &u)
{
    void *ptrs[] = {u.key, u.key2, u.val, u.val2, u.tmp, u.poses, u.out, u.perm, u.unsorted};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    free_mailbox(u.mail);
    u = UndistBuffers();
}

static hipError_t undist_reserve(UndistBuffers &u, int64_t n)
{
    if (u.cap >= n) return hipSuccess;
    void **ps[] = {(void **)&u.key, (void **)&u.key2, (void **)&u.val, (void **)&u.val2, (void **)&u.out};
    const size_t es[] = {4, 4, 4, 4, 12};
    for (int k = 0; k < 5; ++k) {
        if (*ps[k]) S2M_TRY(hipFree(*ps[k]));
        *ps[k] = nullptr;
        S2M_TRY(hipMalloc(ps[k], (size_t)n * es[k]));
    }
    u.cap = n;
    return hipSuccess;
}

// the time order of the records -- std::sort(pcl_out.points.begin(), pcl_out.points.end(), time_list) (:216), made stable
// -- into u.val2 (sorted position -> input index).  It needs the records only, not the poses: the side thread computes
// it as soon as a prefetched sweep has arrived (s2m_scan_prefetch_raw)
hipError_t undistort_order(UndistBuffers &u, const float *pts, int64_t stride, int64_t n, int off_a, int off_b, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    S2M_TRY(undist_reserve(u, n));
    const int nb = (int)((n + 255) / 256);
    if (!u.unsorted) {
        S2M_TRY(hipMalloc((void **)&u.unsorted, sizeof(uint32_t)));
        S2M_TRY(hipMemsetAsync(u.unsorted, 0, sizeof(uint32_t), st));
        u.unsorted_seen = 0;
    }
    hipLaunchKernelGGL(undist_key_kernel, dim3(nb), dim3(256), 0, st, pts, stride, n, off_a, off_b, u.key, u.val, u.val2, u.unsorted);
    if (!u.always_sort) {
        const uint32_t *src[1] = {u.unsorted};
        uint32_t v = 0;
        S2M_TRY(mail_fetch(u.mail, src, 1, &v, st));
        const bool in_order = v == u.unsorted_seen;
        u.unsorted_seen = v;
        if (in_order) { ++u.n_sorted_input; return hipGetLastError(); }  // (val2 is the order)
        ++u.n_unsorted_input;
    }
    size_t bytes = 0;
    S2M_TRY(rocprim::radix_sort_pairs(nullptr, bytes, u.key, u.key2, u.val, u.val2, (size_t)n, 0, 32, st));
    if (bytes > u.tmp_bytes) {
        if (u.tmp) S2M_TRY(hipFree(u.tmp));
        u.tmp = nullptr;
        S2M_TRY(hipMalloc(&u.tmp, bytes));
        u.tmp_bytes = bytes;
    }
    size_t b2 = u.tmp_bytes;
    S2M_TRY(rocprim::radix_sort_pairs(u.tmp, b2, u.key, u.key2, u.val, u.val2, (size_t)n, 0, 32, st));
    return hipGetLastError();
}

// pts: device records; poses_host: K x 22 doubles; result: u.out (n x 3 floats, device), sorted by time when asked
// (order_ready: u.val2 already holds the time order of exactly these records, undistort_order)
hipError_t undistort(UndistBuffers &u, const float *pts, int64_t stride, int64_t n, int off_a, int off_b,
                     const double *poses_host, int K, const Pose &end, bool sort_by_time, uint32_t *perm_dev,
                     hipStream_t st, bool order_ready)
{
    if (n <= 0) return hipSuccess;
    if (!(sort_by_time && order_ready)) S2M_TRY(undist_reserve(u, n));  // (a reserve would drop the order that is ready)
    if (u.pose_cap < K) {
        if (u.poses) S2M_TRY(hipFree(u.poses));
        u.poses = nullptr;
        S2M_TRY(hipMalloc((void **)&u.poses, (size_t)std::max(K, 64) * 22 * sizeof(double)));
        u.pose_cap = std::max(K, 64);
    }
    S2M_TRY(hipMemcpyAsync(u.poses, poses_host, (size_t)K * 22 * sizeof(double), hipMemcpyHostToDevice, st));
    const int nb = (int)((n + 255) / 256);
    const uint32_t *order = nullptr;
    if (sort_by_time) {
        if (!order_ready) S2M_TRY(undistort_order(u, pts, stride, n, off_a, off_b, st));
        order = u.val2;
    }
    hipLaunchKernelGGL(undistort_kernel, dim3(nb), dim3(256), 0, st, pts, stride, n, off_a, off_b, order, u.poses, K,
                       end, u.out, perm_dev);
    return hipGetLastError();
}

}  // namespace s2m
