// mirror_replay.cpp -- the reports of a real drive (scripts/dump_reports.py) applied to the follower's mirror on a host without a
// device: what a report costs, phase by phase.  g++ -std=c++14 -O2 -I include scripts/mirror_replay.cpp -o /tmp/mirror_replay
// usage: mirror_replay <file> [repeat]
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define private public
#include "daliti_s2m_mirror.hpp"
#undef private

extern "C" {
int s2m_map_get_changes(s2m_engine *, uint64_t *, s2m_map_changes *) { return S2M_ERR_STATE; }
int s2m_map_get_points(s2m_engine *, float *, int64_t, int64_t *) { return S2M_ERR_STATE; }
int s2m_map_get_ids(s2m_engine *, uint32_t *, int64_t, int64_t *) { return S2M_ERR_STATE; }
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct Frame { std::vector<float> ax, rx, box; std::vector<uint32_t> ai, ri; std::vector<int64_t> ba, br; };
template <class T> static bool rd(FILE *f, std::vector<T> &v, size_t n) { v.resize(n); return n == 0 || std::fread(v.data(), sizeof(T), n, f) == n; }

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    int64_t m = 0;
    if (std::fread(&m, 8, 1, f) != 1) return 2;
    struct P { float x, y, z; uint32_t id; };
    std::vector<P> map0((size_t)m);
    if (std::fread(map0.data(), sizeof(P), (size_t)m, f) != (size_t)m) return 2;
    std::vector<Frame> frames;
    for (;;) {
        int64_t h[3];
        if (std::fread(h, 8, 3, f) != 3) break;
        Frame fr;
        if (!rd(f, fr.ax, 3 * (size_t)h[0]) || !rd(f, fr.ai, (size_t)h[0]) || !rd(f, fr.rx, 3 * (size_t)h[1]) || !rd(f, fr.ri, (size_t)h[1]) ||
            !rd(f, fr.box, 6 * (size_t)h[2]) || !rd(f, fr.ba, (size_t)h[2]) || !rd(f, fr.br, (size_t)h[2])) return 2;
        frames.push_back(std::move(fr));
    }
    std::fclose(f);
    const int repeat = argc > 2 ? std::atoi(argv[2]) : 1;
    for (int rep = 0; rep < repeat; ++rep) {
        s2m_map_mirror mir;
        std::sort(map0.begin(), map0.end(), [](const P &a, const P &b) { return a.id < b.id; });
        for (const P &p : map0) mir.add(p.id, &p.x);
        std::vector<double> t_all, t_add, t_rem;
        int64_t na = 0, nr = 0;
        for (const Frame &fr : frames) {
            s2m_map_mirror::IO &io = mir.io_[0];
            mir.ap_ = &io;
            io.add_xyz = fr.ax; io.add_ids = fr.ai; io.rem_xyz = fr.rx; io.rem_ids = fr.ri; io.box = fr.box; io.box_a = fr.ba; io.box_r = fr.br;
            io.c.n_added = (int64_t)fr.ai.size(); io.c.n_removed = (int64_t)fr.ri.size(); io.c.n_boxes = (int64_t)fr.ba.size();
            io.resync = false;
            if (fr.ba.empty()) {   // (the phases on their own)
                mir.give_back(64);
                double t = now_us();
                mir.add_many(0, io.c.n_added);
                const double ta = now_us() - t;
                t = now_us();
                mir.remove_many(0, io.c.n_removed);
                const double tr = now_us() - t;
                t_add.push_back(ta); t_rem.push_back(tr); t_all.push_back(ta + tr);
                io.c.n_added = io.c.n_removed = 0;
            } else {
                const double t = now_us();
                mir.apply_report(0);
                std::printf("   report with %zu box(es): %.3f ms\n", fr.ba.size(), (now_us() - t) * 1e-3);
            }
            na += (int64_t)fr.ai.size(); nr += (int64_t)fr.ri.size();
        }
        auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
        auto p99 = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[(size_t)(0.99 * (v.size() - 1))]; };
        std::printf("%zu frames, %.0f added %.0f removed per frame: report median %.3f p99 %.3f ms (add %.3f = %.1f ns each, remove %.3f = %.1f ns each); live %lld, missed %lld, %zu buckets\n",
                    frames.size(), (double)na / frames.size(), (double)nr / frames.size(), med(t_all) * 1e-3, p99(t_all) * 1e-3, med(t_add) * 1e-3,
                    med(t_add) * 1e3 * frames.size() / std::max<double>(1, na), med(t_rem) * 1e-3, med(t_rem) * 1e3 * frames.size() / std::max<double>(1, nr), (long long)mir.size(),
                    (long long)mir.missed, mir.buckets_.size());
    }
    return 0;
}
