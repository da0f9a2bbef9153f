// Dev probe (companion of ta_rates.hip): cost of one wave-level 16-byte load instruction as a function of how many lanes
// read ADJACENT points (G = 1, 2, 4, 8, 16, 64: 64, 32, 16, 8, 4, 1 distinct runs per instruction), L1-resident (2 k points),
// L2-resident (100 k) and Infinity-Cache-resident (5 M).  Eight loads per lane and batch walk on through the run like
// load_batch<G>: point i + j + G u.
// build: hipcc --offload-arch=gfx950 -O3 scripts/ta_lines.hip -o /tmp/ta_lines
#include <hip/hip_runtime.h>

#include <cstdio>

template <int G>
__global__ __launch_bounds__(256, 3) void probe(const float4 *__restrict__ base, uint32_t m, int trips, float *__restrict__ out)
{
    const uint32_t grp = (blockIdx.x * 256 + threadIdx.x) / G, j = threadIdx.x % G;
    uint32_t s = grp * 2654435761u + 12345u;
    float acc = 0.0f;
    for (int t = 0; t < trips; ++t) {
        s = s * 1664525u + 1013904223u;
        const uint32_t i = (s >> 4) % (m - 16u * G - 8u);
        float4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = base[i + j + (uint32_t)(G * u)];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += (v[u].x + v[u].y) + (v[u].z + v[u].w);
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int G>
static void run(const float4 *d, uint32_t m, float *out, int blocks, int trips)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(probe<G>, dim3(blocks), dim3(256), 0, 0, d, m, trips, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(probe<G>, dim3(blocks), dim3(256), 0, 0, d, m, trips, out);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    ms /= 5;
    const double wave_loads = (double)blocks * 4 * trips * 16;
    std::printf("  %2d lanes per run: %7.1f us per launch, %6.2f ns = %5.1f cycles (2.4 GHz) per wave load instruction and CU\n", G, ms * 1e3,
                ms * 1e6 / (wave_loads / 256.0), ms * 1e6 / (wave_loads / 256.0) * 2.4);
}

int main()
{
    const uint32_t m = 5000000;
    float4 *d;
    float *out;
    (void)hipMalloc(&d, (size_t)m * 16 + 65536);
    (void)hipMalloc(&out, 64);
    (void)hipMemset(d, 0, (size_t)m * 16 + 65536);
    for (uint32_t mm : {2000u, 100000u, m}) {
        std::printf("%u points (16 bytes each), 4096 workgroups, 12 trips of 16 loads per lane\n", mm);
        run<1>(d, mm, out, 4096, 12);
        run<2>(d, mm, out, 4096, 12);
        run<4>(d, mm, out, 4096, 12);
        run<8>(d, mm, out, 4096, 12);
        run<16>(d, mm, out, 4096, 12);
        run<64>(d, mm, out, 4096, 12);
    }
    return 0;
}
