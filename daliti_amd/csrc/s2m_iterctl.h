// s2m_iterctl.h -- the loop control of the iterated update, free of HIP so that it can be tested on the CPU.
//
// Mirrors eskf_lio/src/laserMapping.cpp:
//   effct_feat_numQueue / EKF_stop_flg      :192-193, 899-918   (degeneracy queue)
//   rematch judgement                        :1070-1076
//   exit test, covariance update, stop exit  :1079-1101
#pragma once
#include <cstdint>
#include <cstring>

#include "../../include/daliti_s2m.h"

namespace s2m {

// loop variables of the iterated update (laserMapping.cpp:813-818, 820)
struct IterCtl {
    int it, rematch, rematch_num, rematch_en;
    int32_t conv, stop;
};

// push effct_feat_num of the pass that just ended; returns EKF_stop_flg: any of the last QUEUE_SIZE counts at or
// below the threshold (:899-918).  queue holds S2M_FEAT_QUEUE + 1 slots.
inline int32_t degeneracy_push(int32_t *queue, int32_t &len, int32_t effct, int32_t threshold)
{
    queue[len++] = effct;
    if (len > S2M_FEAT_QUEUE) {
        std::memmove(queue, queue + 1, sizeof(int32_t) * S2M_FEAT_QUEUE);
        len = S2M_FEAT_QUEUE;
    }
    for (int q = 0; q < len; ++q)
        if (queue[q] <= threshold) return 1;
    return 0;
}

// after the Kalman update of iteration c.it (c.conv, c.stop set): will the next pass search again, does the loop end
// here, and is the covariance updated on the way out (:1070-1101)
inline void iter_judge(IterCtl &c, int max_iter, bool &finished, bool &update_cov)
{
    c.rematch_en = 0;
    if (c.conv || (c.rematch_num == 0 && c.it == max_iter - 2)) {
        c.rematch_en = 1;
        c.rematch_num++;
    }
    finished = false;
    update_cov = false;
    if (c.rematch_num >= 2 || c.it == max_iter - 1) {
        update_cov = !c.stop;
        finished = true;
    } else if (c.stop) {
        finished = true;
    }
}

}  // namespace s2m
