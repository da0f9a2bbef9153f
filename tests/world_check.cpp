// Test helper: the product's body->world transform (daliti_amd/csrc/s2m_device.h, __host__ __device__) on the HOST,
// built with hipcc -ffp-contract=off like the library.  File: 36 doubles (flat state: rot9 pos3 R_LI9 T_LI3 ...) then
// n x 3 floats; prints the world points as hex bits.  usage: world_check <file> <n>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "s2m_device.h"

int main(int argc, char **argv)
{
    if (argc < 3) return 64;
    const int n = std::atoi(argv[2]);
    FILE *f = std::fopen(argv[1], "rb");
    double x[36];
    std::vector<float> pts((size_t)n * 3);
    if (!f || std::fread(x, sizeof(double), 36, f) != 36 || std::fread(pts.data(), sizeof(float), pts.size(), f) != pts.size()) return 66;
    std::fclose(f);
    s2m::Pose P;
    std::memcpy(P.R, x, sizeof(P.R));
    std::memcpy(P.t, x + 9, sizeof(P.t));
    std::memcpy(P.RLI, x + 12, sizeof(P.RLI));
    std::memcpy(P.TLI, x + 21, sizeof(P.TLI));
    for (int k = 0; k < n; ++k) {
        float w[3];
        s2m::body_to_world(P, pts[3 * k], pts[3 * k + 1], pts[3 * k + 2], w[0], w[1], w[2]);
        uint32_t b[3];
        std::memcpy(b, w, sizeof(b));
        std::printf("%08x %08x %08x\n", b[0], b[1], b[2]);
    }
    return 0;
}
