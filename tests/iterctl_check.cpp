// Test helper: the product's loop control (daliti_amd/csrc/s2m_iterctl.h) replayed on the CPU over the per-iteration
// (effct_feat_num, converged) sequence of an oracle run.  Reads one line of integers:
//   max_iter feat_threshold qlen q[0..qlen) n  effct[0] conv[0] ... effct[n-1] conv[n-1]
// (n = iterations the oracle executed; on a stop iteration the reference skips the update, so conv keeps its value)
// and prints: iters stop qlen_after q_after... | rematch[0..iters)
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "s2m_iterctl.h"

using namespace s2m;

int main()
{
    int max_iter = 0, thr = 0, qlen = 0, n = 0;
    if (std::scanf("%d %d %d", &max_iter, &thr, &qlen) != 3) return 64;
    int32_t queue[S2M_FEAT_QUEUE + 1] = {0};
    for (int i = 0; i < qlen; ++i)
        if (std::scanf("%d", &queue[i]) != 1) return 64;
    if (std::scanf("%d", &n) != 1) return 64;
    std::vector<int> effct(n), conv(n);
    for (int i = 0; i < n; ++i)
        if (std::scanf("%d %d", &effct[i], &conv[i]) != 2) return 64;
    IterCtl c{0, 1, 0, 0, 0, 0};
    std::vector<int> rematch;
    int32_t ql = qlen;
    int it = 0;
    for (it = 0; it < max_iter; ++it) {
        if (it >= n) { std::printf("ran past the oracle's %d iterations\n", n); return 1; }
        c.it = it;
        c.rematch = (it == 0) || c.rematch_en;   // :847
        rematch.push_back(c.rematch);
        c.stop = degeneracy_push(queue, ql, effct[it], thr);
        if (!c.stop) c.conv = conv[it];          // the Kalman update (and its convergence test) only runs without a stop
        bool finished = false, update_cov = false;
        iter_judge(c, max_iter, finished, update_cov);
        if (finished) { ++it; break; }
    }
    std::printf("%d %d %d", it, (int)c.stop, (int)ql);
    for (int i = 0; i < ql; ++i) std::printf(" %d", queue[i]);
    std::printf(" |");
    for (int r : rematch) std::printf(" %d", r);
    std::printf("\n");
    return 0;
}
