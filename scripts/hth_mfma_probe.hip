// hth_mfma_probe.hip -- north star: "MFMA considered only for the tall-skinny J^T J contraction and taken only if
// rocprof shows it beating the shuffle reduce".  This probe times, on the device, the wave-level contraction of the
// reduce kernel both ways:
//   (S) what s2m_reduce.hip does: every lane forms the upper triangle of h h^T (+ h z, |r|, 1) of its own point in
//       registers and the 64 lanes are summed by the halving butterfly (v_permlane32/16_swap + DPP), one 32-term group
//       per butterfly: one group for 6 Jacobian columns (29 terms), three for 12 (92 terms);
//   (M) the same sums as ONE matrix product per wave: the 64 x 16 matrix [h | z | |r| | 1 | 0] staged through LDS and
//       multiplied with its own transpose by sixteen v_mfma_f64_16x16x4_f64 (K = 4 points per instruction, the same
//       register as A and B operand: A[i][k] = B[k][i]); H^T H, H^T z, sum |r| and the count are entries of the
//       16 x 16 result.
// Both are checked against a host sum.  Build and run on the MI355X box:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I daliti_amd/csrc -I include scripts/hth_mfma_probe.hip -o /tmp/hth && /tmp/hth
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../daliti_amd/csrc/s2m_reduce.hip"  // wave_sum32, Terms<> (the product's own butterfly)

using namespace s2m;

constexpr int kReps = 64;  // contractions per wave inside the timed region (different data every time)

// per point: 12 Jacobian entries, z, |r|; layout [rep][point][16] doubles (columns 14 = 1.0, 15 = 0.0)
template <int NC>
__global__ __launch_bounds__(512) void shuffle_kernel(const double *__restrict__ in, double *__restrict__ out,
                                                      unsigned long long *__restrict__ ticks)
{
    typedef Terms<NC> T;
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const double *src = in + ((size_t)wave * 64 + lane) * 16;
    double acc[T::kSlots / 32];
#pragma unroll
    for (int g = 0; g < T::kSlots / 32; ++g) acc[g] = 0.0;
    double h0[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) h0[c] = src[c];
    const long long t0 = wall_clock64();
    for (int rep = 0; rep < kReps; ++rep) {
        // the point's values of this repetition, made in registers (no memory traffic inside the timed region)
        const double f = 1.0 + 0.001 * rep;
        double h[16];
#pragma unroll
        for (int c = 0; c < 14; ++c) h[c] = h0[c] * f;
        h[14] = 1.0; h[15] = 0.0;
        double term[T::kSlots];
#pragma unroll
        for (int t = 0; t < T::kSlots; ++t) term[t] = 0.0;
#pragma unroll
        for (int r = 0; r < NC; ++r)
#pragma unroll
            for (int c = r; c < NC; ++c) term[T::tri(r, c)] = h[r] * h[c];
#pragma unroll
        for (int r = 0; r < NC; ++r) term[T::kHtz + r] = h[r] * h[12];
        term[T::kRes] = h[13];
        term[T::kCnt] = 1.0;
#pragma unroll
        for (int g = 0; g < T::kSlots / 32; ++g) {
            double v[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) v[i] = term[g * 32 + i];
            wave_sum32(v, lane);
            acc[g] += v[0];  // lane l: running sum of term g * 32 + ((l >> 1) & 31)
        }
    }
    const long long t1 = wall_clock64();
#pragma unroll
    for (int g = 0; g < T::kSlots / 32; ++g)
        if ((lane & 1) == 0) out[(size_t)wave * 96 + g * 32 + (lane >> 1)] = acc[g];
    if (lane == 0) ticks[wave] = (unsigned long long)(t1 - t0);
}

typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void mfma_kernel(const double *__restrict__ in, double *__restrict__ out,
                                                   unsigned long long *__restrict__ ticks)
{
    __shared__ double stage[8][64 * 17];  // one 64 x 16 matrix per wave, rows padded to 17 doubles
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const double *src = in + ((size_t)wave * 64 + lane) * 16;
    double *my = stage[w];
    double4_t c = {0.0, 0.0, 0.0, 0.0};
    double h0[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) h0[k] = src[k];
    const long long t0 = wall_clock64();
    for (int rep = 0; rep < kReps; ++rep) {
        const double f = 1.0 + 0.001 * rep;
        double h[16];
#pragma unroll
        for (int k = 0; k < 14; ++k) h[k] = h0[k] * f;
        h[14] = 1.0; h[15] = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) my[lane * 17 + k] = h[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            // A[i = lane % 16][k = lane / 16] = B[k][j = lane % 16] = H[point 4 s + lane / 16][column lane % 16]
            const double a = my[(4 * s + (lane >> 4)) * 17 + (lane & 15)];
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, c, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    const long long t1 = wall_clock64();
    // raw accumulators, lane-major; the host maps them to D[i][j] (i = lane / 16 + 4 r, j = lane % 16)
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(size_t)wave * 256 + lane * 4 + r] = c[r];
    if (lane == 0) ticks[wave] = (unsigned long long)(t1 - t0);
}

int main()
{
    const int blocks = 256, waves = blocks * 8;  // the reduce kernel's shape: 512-thread workgroups, here one per CU
    std::vector<double> h((size_t)waves * 64 * 16);
    srand(7);
    for (size_t p = 0; p < h.size() / 16; ++p) {
        for (int k = 0; k < 14; ++k) h[p * 16 + k] = (double)rand() / RAND_MAX - 0.5;
        h[p * 16 + 13] = fabs(h[p * 16 + 13]);
        h[p * 16 + 14] = 1.0;
        h[p * 16 + 15] = 0.0;
    }
    double *d_in, *d_out;
    unsigned long long *d_t;
    hipMalloc(&d_in, h.size() * sizeof(double));
    hipMalloc(&d_out, (size_t)waves * 256 * sizeof(double));
    hipMalloc(&d_t, waves * sizeof(unsigned long long));
    hipMemcpy(d_in, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice);
    std::vector<double> out((size_t)waves * 256);
    std::vector<unsigned long long> t(waves);
    auto report = [&](const char *name) {
        hipDeviceSynchronize();
        hipMemcpy(t.data(), d_t, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        hipMemcpy(out.data(), d_out, out.size() * sizeof(double), hipMemcpyDeviceToHost);
        double s = 0;
        for (auto v : t) s += (double)v;
        // wall_clock64 ticks at 100 MHz
        printf("%-28s %7.1f ns per wave-level contraction (64 points), %d waves in flight, %d per SIMD\n", name,
               s / waves / kReps * 10.0, waves, 2);
    };
    // host reference for wave 0: sums over all reps
    auto ref = [&](int i, int j) {
        double s = 0;
        for (int rep = 0; rep < kReps; ++rep) {
            const double f = 1.0 + 0.001 * rep;
            for (int p = 0; p < 64; ++p) {
                const double a = i < 14 ? h[(size_t)p * 16 + i] * f : h[(size_t)p * 16 + i];
                const double b = j < 14 ? h[(size_t)p * 16 + j] * f : h[(size_t)p * 16 + j];
                s += a * b;
            }
        }
        return s;
    };
    for (int rounds = 0; rounds < 2; ++rounds) {  // second round: warm
        hipLaunchKernelGGL(shuffle_kernel<6>, dim3(blocks), dim3(512), 0, 0, d_in, d_out, d_t);
        report("shuffle butterfly, 6 columns");
        double e6 = 0;
        for (int r = 0; r < 6; ++r)
            for (int c = r; c < 6; ++c) e6 = fmax(e6, fabs(out[Terms<6>::tri(r, c)] - ref(r, c)));
        for (int r = 0; r < 6; ++r) e6 = fmax(e6, fabs(out[Terms<6>::kHtz + r] - ref(r, 12)));
        e6 = fmax(e6, fabs(out[Terms<6>::kRes] - ref(13, 14)));
        e6 = fmax(e6, fabs(out[Terms<6>::kCnt] - ref(14, 14)));
        hipLaunchKernelGGL(shuffle_kernel<12>, dim3(blocks), dim3(512), 0, 0, d_in, d_out, d_t);
        report("shuffle butterfly, 12 columns");
        double e12 = 0;
        for (int r = 0; r < 12; ++r)
            for (int c = r; c < 12; ++c) e12 = fmax(e12, fabs(out[Terms<12>::tri(r, c)] - ref(r, c)));
        for (int r = 0; r < 12; ++r) e12 = fmax(e12, fabs(out[Terms<12>::kHtz + r] - ref(r, 12)));
        hipLaunchKernelGGL(mfma_kernel, dim3(blocks), dim3(512), 0, 0, d_in, d_out, d_t);
        report("v_mfma_f64_16x16x4, 16 cols");
        double em = 0;   // lane l holds D[l / 16 + 4 r][l % 16], r = 0..3
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                const int j = l & 15, i = (l >> 4) + 4 * r;
                if (j < 15 && i < 15) em = fmax(em, fabs(out[l * 4 + r] - ref(i, j)));
            }
        printf("max |device - host| over wave 0: shuffle6 %.2e  shuffle12 %.2e  mfma %.2e\n", e6, e12, em);
    }
    return 0;
}
