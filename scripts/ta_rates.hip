// Dev probe: what does the vector-memory path charge for a 12-byte point against a 16-byte point in the access pattern of
// the first-shell kernel (s2m_match.hip, load_batch<G = 2>): the two lanes of a pair read adjacent points of a run, eight
// points per lane and batch, two batches in flight, runs at random places of a 5 M-point array (Infinity-Cache resident)?
// build: hipcc --offload-arch=gfx950 -O3 scripts/ta_rates.hip -o /tmp/ta_rates ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

struct __attribute__((packed, aligned(4))) P3 { float x, y, z; };

template <int W>  // 4: float4 points, 3: packed float3 points, 2: float2 (8 bytes: what a quantised candidate would cost)
__global__ __launch_bounds__(256, 3) void probe(const float *__restrict__ base, uint32_t m, int trips, float *__restrict__ out)
{
    const uint32_t pair = (blockIdx.x * 256 + threadIdx.x) >> 1, j = threadIdx.x & 1;
    uint32_t s = pair * 2654435761u + 12345u;
    float acc = 0.0f;
    for (int t = 0; t < trips; ++t) {
        s = s * 1664525u + 1013904223u;
        const uint32_t i = (s >> 4) % (m - 64u);
        float v[16][4];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const uint32_t p = i + j + 2u * (uint32_t)u;
            if (W == 4) { const float4 q = reinterpret_cast<const float4 *>(base)[p]; v[u][0] = q.x; v[u][1] = q.y; v[u][2] = q.z; v[u][3] = q.w; }
            else if (W == 3) { const P3 q = reinterpret_cast<const P3 *>(base)[p]; v[u][0] = q.x; v[u][1] = q.y; v[u][2] = q.z; v[u][3] = 0.f; }
            else { const float2 q = reinterpret_cast<const float2 *>(base)[p]; v[u][0] = q.x; v[u][1] = q.y; v[u][2] = 0.f; v[u][3] = 0.f; }
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int W>
static void run(const float *d, uint32_t m, float *out, int blocks, int trips)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(probe<W>, dim3(blocks), dim3(256), 0, 0, d, m, trips, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(probe<W>, dim3(blocks), dim3(256), 0, 0, d, m, trips, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    ms /= 5;
    const double wave_loads = (double)blocks * 4 * trips * 16;
    std::printf("%2d-byte points: %7.1f us per launch, %.2f ns per wave load instruction per CU-equivalent (256 CUs), %.1f G points/s\n",
                4 * W, ms * 1e3, ms * 1e6 / (wave_loads / 256.0), wave_loads * 64 / (ms * 1e-3) * 1e-9);
}

int main()
{
    const uint32_t m = 5000000;
    float *d, *out;
    hipMalloc(&d, (size_t)m * 16 + 4096);
    hipMalloc(&out, 64);
    hipMemset(d, 0, (size_t)m * 16 + 4096);
    // 5 M points: beyond the L2s (4 MB x 8), inside the Infinity Cache; 100 k / 20 k points: L2- / mostly L1-resident
    for (uint32_t mm : {m, 100000u, 20000u, 2000u})
        for (int blocks : {4096, 16384}) {
            std::printf("%u points, grid %d workgroups, 12 trips of 16 loads per lane\n", mm, blocks);
            run<4>(d, mm, out, blocks, 12);
            run<3>(d, mm, out, blocks, 12);
            run<2>(d, mm, out, blocks, 12);
        }
    return 0;
}
