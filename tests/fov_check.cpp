// Test helper: the product's local-map cube logic (daliti_amd/csrc/s2m_fov.h) replayed on the CPU.  Reads
//   cube_len n  x y z (n times)          and prints per position:  nb  cube[6]  boxes[nb][6]   (floats as %.9g)
#include <cstdio>

#include "s2m_fov.h"

int main()
{
    double cube = 0;
    int n = 0;
    if (std::scanf("%lf %d", &cube, &n) != 2) return 64;
    float lm[6] = {0, 0, 0, 0, 0, 0};
    bool init = false;
    for (int k = 0; k < n; ++k) {
        double p[3];
        if (std::scanf("%lf %lf %lf", &p[0], &p[1], &p[2]) != 3) return 64;
        float boxes[3][6];
        const int nb = s2m::fov_step(lm, init, p, cube, boxes);
        std::printf("%d", nb);
        for (int i = 0; i < 6; ++i) std::printf(" %.9g", lm[i]);
        for (int b = 0; b < nb; ++b)
            for (int i = 0; i < 6; ++i) std::printf(" %.9g", boxes[b][i]);
        std::printf("\n");
    }
    return 0;
}
