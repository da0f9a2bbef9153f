// Test helper: the product's plane fit (daliti_amd/csrc/s2m_plane.h -- the code reduce<FIT> runs per point) called on
// the HOST (it is __host__ __device__), built with hipcc -ffp-contract=off like the library.  No GPU is touched.
// usage: plane_check <file of n x 15 floats: five neighbours xyz> <n>; prints per case: ok a b c d (hex bits)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "s2m_plane.h"

int main(int argc, char **argv)
{
    if (argc < 3) return 64;
    const int n = std::atoi(argv[2]);
    std::vector<float> buf((size_t)n * 15);
    FILE *f = std::fopen(argv[1], "rb");
    if (!f || std::fread(buf.data(), sizeof(float), buf.size(), f) != buf.size()) return 66;
    std::fclose(f);
    for (int k = 0; k < n; ++k) {
        const float *p = buf.data() + (size_t)k * 15;
        float nx[s2m::kK], ny[s2m::kK], nz[s2m::kK];
        for (int i = 0; i < s2m::kK; ++i) { nx[i] = p[3 * i]; ny[i] = p[3 * i + 1]; nz[i] = p[3 * i + 2]; }
        float4 pl = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool ok = s2m::fit_plane(nx, ny, nz, 0.1f, pl);
        uint32_t b[4];
        std::memcpy(b, &pl, sizeof(b));
        std::printf("%d %08x %08x %08x %08x\n", ok ? 1 : 0, b[0], b[1], b[2], b[3]);
    }
    return 0;
}
