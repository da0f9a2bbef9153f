// Dev probe: issue cost (cycles per wave64 instruction on one SIMD) of the VALU operations the search kernels' top-5
// network and the plane fit are made of.  One workgroup of 256 threads per CU x W waves per SIMD, each wave running a
// dependent-free stream of N copies of the instruction on 8 independent register sets; cycles = s_memtime delta / N.
// build: hipcc --offload-arch=gfx950 -O3 scripts/valu_rates.hip -o /tmp/valu_rates ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP>
__global__ __launch_bounds__(256) void probe(unsigned long long *out, int iters)
{
    double d0 = threadIdx.x, d1 = 1.5 + threadIdx.x, d2 = 2.5, d3 = 3.5, d4 = 4.5, d5 = 5.5, d6 = 6.5, d7 = 7.5, dk = 0.75 + blockIdx.x;
    float f0 = threadIdx.x, f1 = 1.5f, f2 = 2.5f, f3 = 3.5f, f4 = 4.5f, f5 = 5.5f, f6 = 6.5f, f7 = 7.5f, fk = 0.75f + blockIdx.x;
    unsigned u0 = threadIdx.x, u1 = 11, u2 = 12, u3 = 13, u4 = 14, u5 = 15, u6 = 16, u7 = 17, uk = 3 + blockIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) {  // v_min_f64
            REP8(asm volatile("v_min_f64 %0, %0, %8\n v_min_f64 %1, %1, %8\n v_min_f64 %2, %2, %8\n v_min_f64 %3, %3, %8\n"
                              "v_min_f64 %4, %4, %8\n v_min_f64 %5, %5, %8\n v_min_f64 %6, %6, %8\n v_min_f64 %7, %7, %8"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dk));)
        } else if (OP == 1) {  // v_fma_f64
            REP8(asm volatile("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n"
                              "v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dk));)
        } else if (OP == 2) {  // v_fma_f32
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                              "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"
                              : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fk));)
        } else if (OP == 3) {  // v_min_u32
            REP8(asm volatile("v_min_u32 %0, %0, %8\n v_min_u32 %1, %1, %8\n v_min_u32 %2, %2, %8\n v_min_u32 %3, %3, %8\n"
                              "v_min_u32 %4, %4, %8\n v_min_u32 %5, %5, %8\n v_min_u32 %6, %6, %8\n v_min_u32 %7, %7, %8"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(uk));)
        } else if (OP == 4) {  // v_cmp_lt_u32 + v_cndmask (pair)
            REP8(asm volatile("v_cmp_lt_u32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %8, vcc\n v_cmp_lt_u32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %8, vcc\n"
                              "v_cmp_lt_u32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_cmp_lt_u32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %8, vcc"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(uk) : "vcc");)
        } else if (OP == 5) {  // v_rcp_f32
            REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                              "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
                              : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fk));)
        } else if (OP == 6) {  // v_sqrt_f32
            REP8(asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                              "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7"
                              : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fk));)
        } else if (OP == 7) {  // v_cmp_lt_u64 + 2 x v_cndmask
            REP8(asm volatile("v_cmp_lt_u64 vcc, %0, %4\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %2, vcc\n"
                              "v_cmp_lt_u64 vcc, %1, %4\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %2, vcc\n"
                              : "+v"(d0), "+v"(d1), "+v"(u2), "+v"(u3) : "v"(dk) : "vcc");)
        } else if (OP == 8) {  // v_add_f64
            REP8(asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                              "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dk));)
        } else if (OP == 9) {  // v_mul_f64
            REP8(asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                              "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dk));)
        } else if (OP == 10) {  // v_pk_mul_f32 (2 floats per lane)
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                              "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dk));)
        } else if (OP == 11) {  // v_max3_u32 / v_min3: three-input integer min
            REP8(asm volatile("v_min3_u32 %0, %0, %8, %1\n v_min3_u32 %1, %1, %8, %2\n v_min3_u32 %2, %2, %8, %3\n v_min3_u32 %3, %3, %8, %4\n"
                              "v_min3_u32 %4, %4, %8, %5\n v_min3_u32 %5, %5, %8, %6\n v_min3_u32 %6, %6, %8, %7\n v_min3_u32 %7, %7, %8, %0"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(uk));)
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
    // keep everything alive
    if (d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7 == -1.234) out[0] = 0;
}

template <int OP>
static void run(const char *name, int per_rep, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd, iters = 200;
    unsigned long long *d;
    hipMalloc(&d, blocks * 4 * sizeof(*d));
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), d, h.size() * sizeof(*d), hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    s /= h.size();
    // s_memtime counts at a fixed 100 MHz on this part: convert with the shader clock measured by the fp32 row
    std::printf("%-28s waves/SIMD %d: %8.1f ticks per wave for %d instr -> %.3f ticks/instr/wave, x waves = %.3f ticks per SIMD-instr\n",
                name, waves_per_simd, s, iters * 8 * per_rep, s / (iters * 8.0 * per_rep), s / (iters * 8.0 * per_rep) / waves_per_simd);
    hipFree(d);
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<2>("v_fma_f32", 8, w);
        run<0>("v_min_f64", 8, w);
        run<1>("v_fma_f64", 8, w);
        run<8>("v_add_f64", 8, w);
        run<9>("v_mul_f64", 8, w);
        run<3>("v_min_u32", 8, w);
        run<11>("v_min3_u32", 8, w);
        run<4>("v_cmp_lt_u32+v_cndmask", 8, w);
        run<7>("v_cmp_lt_u64+2 cndmask", 6, w);
        run<5>("v_rcp_f32", 8, w);
        run<6>("v_sqrt_f32", 8, w);
        run<10>("v_pk_mul_f32", 8, w);
    }
    return 0;
}
