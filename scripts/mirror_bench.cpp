// mirror_bench.cpp -- what applying a report costs the follower's mirror (include/daliti_s2m_mirror.hpp), without a device:
// a hall's surfaces at the bench's seed density (12 points per 0.5 m cell, ~5 M points), then frames of a sensor that advances
// 1 m per frame: (a) the transient of the dense seed -- every voxel a new point falls into loses all its old points but one
// (the voxel rule of laserMapping.cpp:590-640 on a map that was not built by it): ~4 k additions, ~45 k removals per frame;
// (b) the steady state -- 5.5 k additions, 2.2 k removals; (c) a field-of-view trim: one box with a quarter of the map in it.
// Measurement tool, not product: g++ -std=c++14 -O2 -I include scripts/mirror_bench.cpp -o /tmp/mirror_bench && /tmp/mirror_bench
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <unordered_map>
#include <vector>

#define private public
#include "daliti_s2m_mirror.hpp"
#undef private

extern "C" {
int s2m_map_get_changes(s2m_engine *, uint64_t *, s2m_map_changes *) { return S2M_ERR_STATE; }
int s2m_map_get_points(s2m_engine *, float *, int64_t, int64_t *) { return S2M_ERR_STATE; }
int s2m_map_get_ids(s2m_engine *, uint32_t *, int64_t, int64_t *) { return S2M_ERR_STATE; }
}

struct Pt3 { float x, y, z; uint32_t id; };
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static uint64_t cell_key(float x, float y, float z)
{
    const int64_t a = (int64_t)std::floor(x * 2.0f) + (1 << 20), b = (int64_t)std::floor(y * 2.0f) + (1 << 20), c = (int64_t)std::floor(z * 2.0f) + (1 << 20);
    return ((uint64_t)a << 42) | ((uint64_t)b << 21) | (uint64_t)c;
}
static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; }
static double pct(std::vector<double> v, double p) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[std::min(v.size() - 1, (size_t)(p * v.size()))]; }

int main(int argc, char **argv)
{
    const double W = 100.0, LEN = 470.0, H = 10.0;
    const int per_cell = argc > 1 ? std::atoi(argv[1]) : 12;
    std::mt19937_64 rng(11);
    std::uniform_real_distribution<float> u(0.0f, 1.0f);
    std::vector<Pt3> pts;
    // floor, ceiling, two walls: every 0.5 m cell of a surface holds per_cell points
    auto surface = [&](int axis, float at, float a0, float a1, float b0, float b1) {
        for (float a = a0; a < a1; a += 0.5f)
            for (float b = b0; b < b1; b += 0.5f)
                for (int k = 0; k < per_cell; ++k) {
                    const float pa = a + 0.5f * u(rng), pb = b + 0.5f * u(rng), pn = at + 0.04f * (u(rng) - 0.5f) + 0.25f;
                    Pt3 p;
                    if (axis == 2) { p.x = pa; p.y = pb; p.z = pn; } else { p.x = pa; p.y = pn; p.z = pb; }
                    pts.push_back(p);
                }
    };
    surface(2, 0.0f, 0.0f, (float)LEN, (float)(-W / 2), (float)(W / 2));
    surface(2, (float)H, 0.0f, (float)LEN, (float)(-W / 2), (float)(W / 2));
    surface(1, (float)(-W / 2), 0.0f, (float)LEN, 0.0f, (float)H);
    surface(1, (float)(W / 2 - 0.5), 0.0f, (float)LEN, 0.0f, (float)H);
    std::shuffle(pts.begin(), pts.end(), rng);   // (the engine's ids follow the caller's order of the seed cloud, not space)
    for (size_t i = 0; i < pts.size(); ++i) pts[i].id = (uint32_t)i;
    uint32_t next_id = (uint32_t)pts.size();
    std::fprintf(stderr, "%zu points\n", pts.size());

    s2m_map_mirror m;
    double t0 = now_us();
    for (const Pt3 &p : pts) m.add(p.id, &p.x);
    std::fprintf(stderr, "whole map into buckets: %.1f ms; %zu buckets, %.1f MB\n", (now_us() - t0) * 1e-3, m.buckets_.size(), m.memory_bytes() * 1e-6);
    // cell -> the live points in it
    std::unordered_map<uint64_t, std::vector<Pt3>> cells;
    cells.reserve(pts.size() / 8);
    for (const Pt3 &p : pts) cells[cell_key(p.x, p.y, p.z)].push_back(p);

    auto frame = [&](double xs, int n_vox, bool thin, std::vector<double> &t_add, std::vector<double> &t_rem, int64_t &na, int64_t &nr) {
        // the voxels a scan reaches: cells of the surfaces within 60 m of the sensor, n_vox of them at random

        s2m_map_mirror::IO &io = m.io_[0];
        m.ap_ = &io;
        if (io.add_ids.size() < 200000) m.reserve_io(200000, 400000);
        int64_t a = 0, r = 0;
        std::vector<std::pair<uint64_t, Pt3>> fresh;
        for (int k = 0; k < n_vox; ++k) {
            const float x = (float)(xs + 120.0 * u(rng) - 40.0);
            Pt3 p;
            const int s = (int)(u(rng) * 2.2f);
            if (s == 0) { p.x = x; p.y = (float)(W * (u(rng) - 0.5)); p.z = 0.25f + 0.02f * u(rng); }
            else if (s == 1) { p.x = x; p.y = (float)(W * (u(rng) - 0.5)); p.z = (float)H + 0.25f + 0.02f * u(rng); }
            else { p.x = x; p.y = (float)(-W / 2) + 0.25f; p.z = (float)(H * u(rng)); }
            if (p.x < 0.0f) continue;
            auto &c = cells[cell_key(p.x, p.y, p.z)];
            if (thin) {   // the voxel keeps one old point at most ... and here: none, the new one is nearer the centre
                for (const Pt3 &q : c) { io.rem_ids[(size_t)r] = q.id; io.rem_xyz[3 * (size_t)r] = q.x; io.rem_xyz[3 * (size_t)r + 1] = q.y; io.rem_xyz[3 * (size_t)r + 2] = q.z; ++r; }
                c.clear();
            } else if (!c.empty() && u(rng) < 0.4f) {
                const Pt3 q = c.back(); c.pop_back();
                io.rem_ids[(size_t)r] = q.id; io.rem_xyz[3 * (size_t)r] = q.x; io.rem_xyz[3 * (size_t)r + 1] = q.y; io.rem_xyz[3 * (size_t)r + 2] = q.z; ++r;
            }
            p.id = next_id++;
            io.add_ids[(size_t)a] = p.id; io.add_xyz[3 * (size_t)a] = p.x; io.add_xyz[3 * (size_t)a + 1] = p.y; io.add_xyz[3 * (size_t)a + 2] = p.z; ++a;
            c.push_back(p);
        }

        double t = now_us();
        m.add_many(0, a);
        t_add.push_back(now_us() - t);
        t = now_us();
        m.remove_many(0, r);
        t_rem.push_back(now_us() - t);
        na += a; nr += r;
    };
    for (int phase = 0; phase < 2; ++phase) {
        std::vector<double> ta, tr;
        int64_t na = 0, nr = 0;
        const int frames = phase == 0 ? 60 : 300;
        for (int f = 0; f < frames; ++f) frame(40.0 + f * (phase == 0 ? 1.0 : 1.0), phase == 0 ? 4000 : 5500, phase == 0, ta, tr, na, nr);
        std::printf("%s: %d frames, per frame %.0f added %.0f removed: add median %.3f p99 %.3f ms (%.1f ns each), remove median %.3f p99 %.3f ms (%.1f ns each); missed %lld\n",
                    phase == 0 ? "dense seed being thinned" : "steady state", frames, (double)na / frames, (double)nr / frames, med(ta) * 1e-3, pct(ta, 0.99) * 1e-3,
                    med(ta) * 1e3 / std::max(1.0, (double)na / frames), med(tr) * 1e-3, pct(tr, 0.99) * 1e-3, med(tr) * 1e3 / std::max(1.0, (double)nr / frames), (long long)m.missed);
    }
    {   // the table growing (or shedding emptied buckets): every bucket is looked at
        // (cold: walk something large first)
        std::vector<char> big(256 << 20, 1);
        size_t sum = 0;
        for (size_t i = 0; i < big.size(); i += 64) sum += big[i];
        for (int k = 0; k < 3; ++k) {   // (the first one writes a table the allocator has just handed out; the later ones the spare)
            for (size_t i = 0; i < big.size(); i += 64) sum += big[i];
            const double t = now_us();
            m.rehash();
            std::printf("rehash %d with %zu buckets: %.3f ms (%zu)\n", k, m.buckets_.size(), (now_us() - t) * 1e-3, sum & 1);
        }
    }
    // the trim: everything behind x = 150
    const float box[6] = {-1000.0f, -1000.0f, -1000.0f, 150.0f, 1000.0f, 1000.0f};
    const int64_t before = m.size();
    t0 = now_us();
    m.delete_box(box);
    std::printf("trim: %lld points of %lld in the box: %.3f ms\n", (long long)(before - m.size()), (long long)before, (now_us() - t0) * 1e-3);
    return 0;
}
