// s2m_fov.h -- the local-map cube of lasermap_fov_segment() (eskf_lio/src/laserMapping.cpp:304-369), free of HIP so
// that it can be tested on the CPU.  The engine keeps the cube (min xyz, max xyz, floats like the reference's
// BoxPointType) and hands the slabs this returns to Delete_Point_Boxes (s2m_map_delete_boxes).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>

namespace s2m {

// One call per scan.  local_map[6] = {min xyz, max xyz}; init = Localmap_Initialized.  Returns the number of slabs
// (0..3) written to boxes: the parts of the old cube that fall out when it moves.
inline int fov_step(float local_map[6], bool &init, const double pos_lid[3], double cube_len, float boxes[3][6])
{
    const float DET_RANGE = 300.0f, MOV_THRESHOLD = 1.5f;  // laserMapping.cpp:304-305
    float *mn = local_map, *mx = local_map + 3;
    if (!init) {  // :320-328
        for (int i = 0; i < 3; ++i) {
            mn[i] = (float)(pos_lid[i] - cube_len / 2.0);
            mx[i] = (float)(pos_lid[i] + cube_len / 2.0);
        }
        init = true;
        return 0;
    }
    float dist[3][2];
    bool need_move = false;
    for (int i = 0; i < 3; ++i) {  // :331-337
        dist[i][0] = (float)std::fabs(pos_lid[i] - (double)mn[i]);
        dist[i][1] = (float)std::fabs(pos_lid[i] - (double)mx[i]);
        if (dist[i][0] <= MOV_THRESHOLD * DET_RANGE || dist[i][1] <= MOV_THRESHOLD * DET_RANGE) need_move = true;
    }
    if (!need_move) return 0;
    int nb = 0;
    float new_map[6];
    std::memcpy(new_map, local_map, sizeof(new_map));
    const float mov_dist = (float)std::max((cube_len - 2.0 * MOV_THRESHOLD * DET_RANGE) * 0.5 * 0.9,
                                           (double)(DET_RANGE * (MOV_THRESHOLD - 1)));  // :345
    for (int i = 0; i < 3; ++i) {  // :346-363
        float tmp[6];
        std::memcpy(tmp, local_map, sizeof(tmp));
        if (dist[i][0] <= MOV_THRESHOLD * DET_RANGE) {
            new_map[3 + i] -= mov_dist;
            new_map[i] -= mov_dist;
            tmp[i] = mx[i] - mov_dist;
            std::memcpy(boxes[nb++], tmp, sizeof(tmp));
        } else if (dist[i][1] <= MOV_THRESHOLD * DET_RANGE) {
            new_map[3 + i] += mov_dist;
            new_map[i] += mov_dist;
            tmp[3 + i] = mn[i] + mov_dist;
            std::memcpy(boxes[nb++], tmp, sizeof(tmp));
        }
    }
    std::memcpy(local_map, new_map, sizeof(new_map));
    return nb;
}

}  // namespace s2m
