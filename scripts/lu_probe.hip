// lu_probe.hip -- how long would the 24x24 ESKF solve take ON the device?  ("device-side update")
//
// One wave does what s2m_eskf.cpp does on the host every iteration: partial-pivot LU of the 24x24 matrix
// H_T_H + (P/R)^-1 with the first 12 unit columns as right-hand sides (K_1[:, :12]), i.e. Gaussian elimination of a
// 24 x 36 augmented matrix held in LDS, then the back substitution.  The probe times it both inside the kernel
// (wall_clock64, 100 MHz) and as back-to-back launches between HIP events, and checks the result against a host
// elimination of the same matrix.  Build and run on the MI355X box:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/lu_probe.hip -o /tmp/lu_probe && /tmp/lu_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int N = 24, W = 36;  // 24 x 24 system, 12 right-hand sides

__global__ __launch_bounds__(64) void lu_kernel(const double *__restrict__ A_in, double *__restrict__ X_out,
                                                unsigned long long *__restrict__ ticks)
{
    __shared__ double a[N][W + 1];
    __shared__ int piv;
    const int lane = threadIdx.x;
    const long long t0 = wall_clock64();
    for (int e = lane; e < N * W; e += 64) {
        const int r = e / W, c = e % W;
        a[r][c] = c < N ? A_in[r * N + c] : ((c - N) == r ? 1.0 : 0.0);
    }
    __syncthreads();
    for (int k = 0; k < N; ++k) {
        // partial pivoting: the largest |a[r][k]|, r >= k (lanes 0..23 hold one row each)
        double v = (lane >= k && lane < N) ? fabs(a[lane][k]) : -1.0;
        int idx = lane;
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) {
            const double ov = __shfl_xor(v, off, 32);
            const int oi = __shfl_xor(idx, off, 32);
            if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
        }
        if (lane == 0) piv = idx;
        __syncthreads();
        const int p = piv;
        if (p != k && lane < W) { const double t = a[k][lane]; a[k][lane] = a[p][lane]; a[p][lane] = t; }
        __syncthreads();
        const double inv = 1.0 / a[k][k];
        // eliminate column k from the rows below: element (r, c), r > k, c > k, dealt to the 64 lanes
        const int rows = N - 1 - k, cols = W - 1 - k;
        for (int e = lane; e < rows * cols; e += 64) {
            const int r = k + 1 + e / cols, c = k + 1 + e % cols;
            a[r][c] -= (a[r][k] * inv) * a[k][c];
        }
        __syncthreads();
    }
    // back substitution, one right-hand side per lane
    if (lane < W - N) {
        const int c = N + lane;
        for (int r = N - 1; r >= 0; --r) {
            double s = a[r][c];
            for (int j = r + 1; j < N; ++j) s -= a[r][j] * a[j][c];
            a[r][c] = s / a[r][r];
        }
    }
    __syncthreads();
    for (int e = lane; e < N * (W - N); e += 64) X_out[e] = a[e / (W - N)][N + e % (W - N)];
    if (lane == 0) ticks[0] = (unsigned long long)(wall_clock64() - t0);
}

int main()
{
    std::vector<double> A(N * N), X(N * 12), R(N * 12);
    srand(1);
    for (int r = 0; r < N; ++r)
        for (int c = 0; c < N; ++c) A[r * N + c] = (r == c ? 700.0 : 0.0) + (rand() / (double)RAND_MAX - 0.5) * 50.0;
    // host reference: the same elimination
    {
        std::vector<double> a(N * W);
        for (int r = 0; r < N; ++r)
            for (int c = 0; c < W; ++c) a[r * W + c] = c < N ? A[r * N + c] : ((c - N) == r ? 1.0 : 0.0);
        for (int k = 0; k < N; ++k) {
            int p = k;
            for (int r = k + 1; r < N; ++r)
                if (std::fabs(a[r * W + k]) > std::fabs(a[p * W + k])) p = r;
            if (p != k)
                for (int c = 0; c < W; ++c) std::swap(a[k * W + c], a[p * W + c]);
            const double inv = 1.0 / a[k * W + k];
            for (int r = k + 1; r < N; ++r)
                for (int c = k + 1; c < W; ++c) a[r * W + c] -= (a[r * W + k] * inv) * a[k * W + c];
        }
        for (int c = N; c < W; ++c)
            for (int r = N - 1; r >= 0; --r) {
                double s = a[r * W + c];
                for (int j = r + 1; j < N; ++j) s -= a[r * W + j] * a[j * W + c];
                a[r * W + c] = s / a[r * W + r];
            }
        for (int r = 0; r < N; ++r)
            for (int c = 0; c < 12; ++c) R[r * 12 + c] = a[r * W + N + c];
    }
    double *dA = nullptr, *dX = nullptr;
    unsigned long long *dT = nullptr, ticks = 0;
    hipMalloc((void **)&dA, sizeof(double) * N * N);
    hipMalloc((void **)&dX, sizeof(double) * N * 12);
    hipMalloc((void **)&dT, 8);
    hipMemcpy(dA, A.data(), sizeof(double) * N * N, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(lu_kernel, dim3(1), dim3(64), 0, 0, dA, dX, dT);
    hipDeviceSynchronize();
    const int reps = 2000;
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(lu_kernel, dim3(1), dim3(64), 0, 0, dA, dX, dT);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(X.data(), dX, sizeof(double) * N * 12, hipMemcpyDeviceToHost);
    hipMemcpy(&ticks, dT, 8, hipMemcpyDeviceToHost);
    double err = 0.0, scale = 0.0;
    for (int i = 0; i < N * 12; ++i) { err = std::fmax(err, std::fabs(X[i] - R[i])); scale = std::fmax(scale, std::fabs(R[i])); }
    std::printf("one-wave 24x24 partial-pivot LU + 12 right-hand sides: %.2f us inside the kernel (wall_clock64), "
                "%.2f us per launch back to back (incl. the kernel boundary); max |dev - host| = %.2e (scale %.2e)\n",
                ticks / 100.0, 1e3 * ms / reps, err, scale);
    return err <= 1e-12 * std::fmax(scale, 1.0) ? 0 : 1;
}
