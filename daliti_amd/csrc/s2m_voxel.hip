// s2m_voxel.hip -- scan voxel down-sampling on the GPU.
//
// Replaces downSizeFilterSurf.filter(*feats_down) (eskf_lio/src/laserMapping.cpp:153, 703, 775-776):
// pcl::VoxelGrid<PointType> with leaf = mapping/filter_size_surf (0.5 m).  PCL is not in the reference
// tree (libpcl-dev 1.10 from Ubuntu focal, Dockerfile:17); its published algorithm
// (filters/impl/voxel_grid.hpp, applyFilter) is restated here:
//   min/max of the cloud -> min_b = floor(min * inv_leaf), div_b = max_b - min_b + 1;
//   per point ijk = (int)(floor(p * inv_leaf) - (float)min_b), idx = ijk0 + ijk1*div0 + ijk2*div0*div1;
//   sort by idx; one output point per occupied voxel, in ascending idx, = float sum of its points
//   divided by their count (CentroidPoint / AccumulatorXYZ).
// PCL sorts with std::sort, which is not stable, so the order in which a voxel's points are summed --
// and hence the last bit of the centroid -- is implementation-defined there; here the sum runs in
// ascending input index (stable radix sort), and the oracle does the same.  Only x, y, z are produced:
// nothing else of feats_down is read on the registration path.
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

struct VoxelDims {
    float inv_leaf;
    int min_b[3];
    int64_t mul1, mul2;  // divb_mul_[1], divb_mul_[2]
};

__global__ __launch_bounds__(256) void vx_key_kernel(const float *__restrict__ xyz, int64_t stride, int64_t n,
                                                     VoxelDims d, uint32_t *__restrict__ key, uint32_t *__restrict__ val)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int i0 = (int)(floorf(xyz[i * stride] * d.inv_leaf) - (float)d.min_b[0]);
    const int i1 = (int)(floorf(xyz[i * stride + 1] * d.inv_leaf) - (float)d.min_b[1]);
    const int i2 = (int)(floorf(xyz[i * stride + 2] * d.inv_leaf) - (float)d.min_b[2]);
    key[i] = (uint32_t)((int64_t)i0 + (int64_t)i1 * d.mul1 + (int64_t)i2 * d.mul2);  // < 2^31 (checked on the host)
    val[i] = (uint32_t)i;
}

// the same with the grid's numbers read from the device (bbox_final_kernel left them there)
__global__ __launch_bounds__(256) void vx_key_dev_kernel(const float *__restrict__ xyz, int64_t stride, int64_t n, float inv_leaf,
                                                         const VoxelDimsDev *__restrict__ dims, uint32_t *__restrict__ key,
                                                         uint32_t *__restrict__ val)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int m0 = dims->min_b[0], m1 = dims->min_b[1], m2 = dims->min_b[2];
    const int64_t mul1 = dims->mul1, mul2 = dims->mul2;
    const int i0 = (int)(floorf(xyz[i * stride] * inv_leaf) - (float)m0);
    const int i1 = (int)(floorf(xyz[i * stride + 1] * inv_leaf) - (float)m1);
    const int i2 = (int)(floorf(xyz[i * stride + 2] * inv_leaf) - (float)m2);
    key[i] = (uint32_t)((int64_t)i0 + (int64_t)i1 * mul1 + (int64_t)i2 * mul2);  // (< 2^31 unless too_fine: then nobody reads it)
    val[i] = (uint32_t)i;
}

// head[s] = the sorted position starts a voxel; blk[b] = the voxels that start in workgroup b (the centroid kernel adds up
// the workgroups in front of its own: no device-wide scan -- two launches of rocprim -- between the two)
__global__ __launch_bounds__(256) void vx_head_kernel(const uint32_t *__restrict__ skey, int64_t n, uint32_t *__restrict__ head,
                                                      uint32_t *__restrict__ blk)
{
    __shared__ uint32_t s_c[4];
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool h = s < n && (s == 0 || skey[s - 1] != skey[s]);
    if (s < n) head[s] = h ? 1u : 0u;
    const uint32_t c = (uint32_t)__popcll(__ballot(h));
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blk[blockIdx.x] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
}

// one lane per voxel: float sum of its points in ascending input index, divided by the count
__global__ __launch_bounds__(256) void vx_centroid_kernel(const float *__restrict__ xyz, int64_t stride, int64_t n,
                                                          const uint32_t *__restrict__ skey,
                                                          const uint32_t *__restrict__ sval,
                                                          const uint32_t *__restrict__ head,
                                                          const uint32_t *__restrict__ blk, uint32_t *__restrict__ count2,
                                                          float *__restrict__ ox, float *__restrict__ oy, float *__restrict__ oz)
{
    __shared__ uint32_t s_before[4], s_wave[4];
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // the voxel's place in the output: the voxels of the workgroups in front, then the ones in front inside this one
    uint32_t before = 0u;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) before += blk[b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off, 64);
    const bool is_head = s < n && head[s] != 0u;
    const unsigned long long hm = __ballot(is_head);
    if (lane == 0) { s_before[wave] = before; s_wave[wave] = (uint32_t)__popcll(hm); }
    __syncthreads();
    uint32_t o = s_before[0] + s_before[1] + s_before[2] + s_before[3] + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) o += s_wave[w];
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) {  // (the last thread of the launch: every voxel lies in front of it)
        count2[0] = o + (is_head ? 1u : 0u);
        count2[1] = 0u;
    }
    if (!is_head) return;
    float cx = 0.0f, cy = 0.0f, cz = 0.0f;
    int cnt = 0;
    // four positions per trip: their keys, indices and points are requested together and added in order (a near-field
    // voxel holds dozens of points; one dependent key -> index -> point chain per point made this the longest kernel of
    // the front half)
    const uint32_t k0 = skey[s];
    for (int64_t t = s; t < n; t += 4) {
        uint32_t kk[4];
        uint32_t ii[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t tu = t + u < n ? t + u : n - 1;
            kk[u] = skey[tu];
            ii[u] = sval[tu];
        }
        float px[4], py[4], pz[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            px[u] = xyz[(int64_t)ii[u] * stride];
            py[u] = xyz[(int64_t)ii[u] * stride + 1];
            pz[u] = xyz[(int64_t)ii[u] * stride + 2];
        }
        bool more = true;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            more = more && t + u < n && kk[u] == k0;
            if (more) {
                cx = cx + px[u];
                cy = cy + py[u];
                cz = cz + pz[u];
                ++cnt;
            }
        }
        if (!more) break;
    }
    const float fn = (float)cnt;
    ox[o] = cx / fn;
    oy[o] = cy / fn;
    oz[o] = cz / fn;
}

void free_voxel(VoxelBuffers &v)
{
    void *ptrs[] = {v.key, v.key2, v.val, v.val2, v.head, v.pos, v.tmp, v.box, v.dims};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    free_mailbox(v.mail);
    v = VoxelBuffers();
}

// xyz: device AoS cloud; writes the down-sampled cloud as SoA into ox/oy/oz (capacity >= n each)
hipError_t voxel_downsample(VoxelBuffers &v, const float *xyz, int64_t stride, int64_t n, float leaf, float *ox,
                            float *oy, float *oz, int64_t *n_out, bool *too_fine, hipStream_t st)
{
    *n_out = 0;
    *too_fine = false;
    if (n <= 0) return hipSuccess;
    if (v.cap < n) {
        void **ps[] = {(void **)&v.key, (void **)&v.key2, (void **)&v.val, (void **)&v.val2, (void **)&v.head, (void **)&v.pos};
        const size_t es[] = {4, 4, 4, 4, 4, 4};
        for (int k = 0; k < 6; ++k) {
            if (*ps[k]) S2M_TRY(hipFree(*ps[k]));
            *ps[k] = nullptr;
            S2M_TRY(hipMalloc(ps[k], (size_t)(n + 64) * es[k]));  // (+64: v.pos also holds the two count words and the per-workgroup counts)
        }
        v.cap = n;
    }
    if (!v.box) S2M_TRY(hipMalloc((void **)&v.box, kBboxScratchFloats * sizeof(float)));
    if (!v.dims) S2M_TRY(hipMalloc((void **)&v.dims, sizeof(VoxelDimsDev)));
    const int nb = (int)((n + 255) / 256);
    // The grid's numbers follow from the cloud's box.  A stream of sweeps from one sensor keeps the number of BITS of its voxel
    // indices for thousands of frames: with a hint from the last cloud the box stays on the device (bbox_final_kernel derives
    // the numbers there), the sort runs on as many bits as last time, and the one hand-back at the end says whether that was
    // enough -- a round trip (25 us of the front half, which bounds a pipelined frame) less.  When it was not, the cloud is
    // done again the classic way below.
    if (v.kbits_hint > 0 && !v.no_hint) {
        const float inv_leaf = 1.0f / leaf;
        S2M_TRY(cloud_bbox_launch(xyz, stride, n, v.box, inv_leaf, v.dims, st));
        hipLaunchKernelGGL(vx_key_dev_kernel, dim3(nb), dim3(256), 0, st, xyz, stride, n, inv_leaf, v.dims, v.key, v.val);
        const unsigned kbits = (unsigned)v.kbits_hint;
        size_t bytes = 0;
        S2M_TRY(rocprim::radix_sort_pairs(nullptr, bytes, v.key, v.key2, v.val, v.val2, (size_t)n, 0, kbits, st));
        if (bytes > v.tmp_bytes) {
            if (v.tmp) S2M_TRY(hipFree(v.tmp));
            v.tmp = nullptr;
            S2M_TRY(hipMalloc(&v.tmp, bytes));
            v.tmp_bytes = bytes;
        }
        size_t t1 = v.tmp_bytes;
        S2M_TRY(rocprim::radix_sort_pairs(v.tmp, t1, v.key, v.key2, v.val, v.val2, (size_t)n, 0, kbits, st));
        hipLaunchKernelGGL(vx_head_kernel, dim3(nb), dim3(256), 0, st, v.key2, n, v.head, v.pos + 2);
        hipLaunchKernelGGL(vx_centroid_kernel, dim3(nb), dim3(256), 0, st, xyz, stride, n, v.key2, v.val2, v.head, v.pos + 2, v.pos, ox, oy, oz);
        const uint32_t *src[4] = {v.pos, v.pos + 1, &v.dims->bits, &v.dims->too_fine};
        uint32_t ab[4] = {0, 0, 0, 0};
        S2M_TRY(mail_fetch(v.mail, src, 4, ab, st));
        if (ab[3] != 0u) { *too_fine = true; return hipSuccess; }
        const bool enough = ab[2] <= kbits;
        v.kbits_hint = (int)ab[2];
        if (enough) {
            *n_out = (int64_t)ab[0] + ab[1];
            return hipGetLastError();
        }
        ++v.n_respeculated;  // (the cloud's extent crossed a power of two: sorted on too few bits)
    }
    float blo[3], bhi[3];
    S2M_TRY(cloud_bbox(xyz, stride, n, v.box, v.mail, blo, bhi, st));
    VoxelDims d;
    d.inv_leaf = 1.0f / leaf;  // inverse_leaf_size_ = 1 / leaf_size_
    int64_t div[3];
    for (int k = 0; k < 3; ++k) {
        const float mn = blo[k], mx = bhi[k];
        d.min_b[k] = (int)std::floor(mn * d.inv_leaf);
        const int max_b = (int)std::floor(mx * d.inv_leaf);
        div[k] = (int64_t)max_b - d.min_b[k] + 1;
    }
    // PCL refuses (returns the input) when the voxel index would overflow int32; report it instead
    if (div[0] * div[1] * div[2] > (int64_t)2147483647) { *too_fine = true; return hipSuccess; }
    d.mul1 = div[0];
    d.mul2 = div[0] * div[1];
    hipLaunchKernelGGL(vx_key_kernel, dim3(nb), dim3(256), 0, st, xyz, stride, n, d, v.key, v.val);
    // the voxel index is below div0 * div1 * div2: sort only the bits it can have
    unsigned kbits = 1;
    while (kbits < 32 && ((int64_t)1 << kbits) < div[0] * div[1] * div[2]) ++kbits;
    v.kbits_hint = (int)kbits;
    size_t bytes = 0;
    S2M_TRY(rocprim::radix_sort_pairs(nullptr, bytes, v.key, v.key2, v.val, v.val2, (size_t)n, 0, kbits, st));
    if (bytes > v.tmp_bytes) {
        if (v.tmp) S2M_TRY(hipFree(v.tmp));
        v.tmp = nullptr;
        S2M_TRY(hipMalloc(&v.tmp, bytes));
        v.tmp_bytes = bytes;
    }
    size_t t1 = v.tmp_bytes;
    S2M_TRY(rocprim::radix_sort_pairs(v.tmp, t1, v.key, v.key2, v.val, v.val2, (size_t)n, 0, kbits, st));
    hipLaunchKernelGGL(vx_head_kernel, dim3(nb), dim3(256), 0, st, v.key2, n, v.head, v.pos + 2);
    hipLaunchKernelGGL(vx_centroid_kernel, dim3(nb), dim3(256), 0, st, xyz, stride, n, v.key2, v.val2, v.head, v.pos + 2, v.pos,
                       ox, oy, oz);
    const uint32_t *src[2] = {v.pos, v.pos + 1};
    uint32_t ab[2] = {0, 0};
    S2M_TRY(mail_fetch(v.mail, src, 2, ab, st));
    *n_out = (int64_t)ab[0] + ab[1];
    return hipGetLastError();
}

}  // namespace s2m
