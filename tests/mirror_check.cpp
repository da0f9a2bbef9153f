// mirror_check.cpp -- include/daliti_s2m_mirror.hpp without a device: the pieces update() is made of (add / remove / delete_box)
// against a plain id -> point map, over a long random history whose ids run past 2^30 while the live set stays at ~10^5 points.
// Checks: the mirror holds exactly the reference's points after every round; a box delete drops whole buckets and filters
// the cut ones with min <= p < max; memory follows the live points, not the ids ever issued.
// Test infrastructure: g++ -std=c++14 -O2 -I include tests/mirror_check.cpp
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <vector>

#define private public   /* (the batched passes update() applies a report with are private: the check drives them directly) */
#include "daliti_s2m_mirror.hpp"
#undef private

// the ABI entries update() calls are not reached here; give the linker something
extern "C" {
int s2m_map_get_changes(s2m_engine *, uint64_t *, s2m_map_changes *) { return S2M_ERR_STATE; }
int s2m_map_get_points(s2m_engine *, float *, int64_t, int64_t *) { return S2M_ERR_STATE; }
int s2m_map_get_ids(s2m_engine *, uint32_t *, int64_t, int64_t *) { return S2M_ERR_STATE; }
}

struct P { float x, y, z; };

static bool same(const s2m_map_mirror &m, const std::map<uint32_t, P> &ref)
{
    if (m.size() != (int64_t)ref.size()) { std::fprintf(stderr, "size %lld vs %zu\n", (long long)m.size(), ref.size()); return false; }
    std::vector<float> xyz((size_t)m.size() * 3 + 3);
    std::vector<uint32_t> ids((size_t)m.size() + 1);
    m.copy_points(xyz.data(), ids.data());
    for (int64_t i = 0; i < m.size(); ++i) {
        const auto it = ref.find(ids[(size_t)i]);
        if (it == ref.end() || it->second.x != xyz[3 * (size_t)i] || it->second.y != xyz[3 * (size_t)i + 1] || it->second.z != xyz[3 * (size_t)i + 2]) {
            std::fprintf(stderr, "point %u differs\n", ids[(size_t)i]);
            return false;
        }
    }
    int64_t seen = 0;
    m.for_each([&](uint32_t, float, float, float) { ++seen; });
    return seen == m.size();
}

int main()
{
    std::mt19937_64 rng(5);
    s2m_map_mirror m;
    std::map<uint32_t, P> ref;
    std::vector<uint32_t> live;   // ids in the reference (for random removal)
    uint32_t next_id = 0;
    double x0 = 0.0;              // the window the points are drawn in moves along x, like a drive
    const int rounds = 400;
    size_t peak_bytes = 0;
    int64_t expect_missed = 0;
    for (int r = 0; r < rounds; ++r) {
        // ids leap ahead: the engine's ids ascend for ever (here by 2^30 over the history)
        next_id += (uint32_t)(((uint64_t)1 << 30) / rounds);
        // add ~3000 points in the window [x0, x0 + 200) x [-50, 50) x [0, 10): every other round through the batched pass a report
        // is applied with (new buckets open -- and the table grows -- in the middle of a stretch)
        m.give_back(64);   // (what apply_report does at the start of a report without boxes: a trim's emptied buckets return their memory over the next reports)
        m.reserve_io(65536, 65536);
        for (int i = 0; i < 3000; ++i) {
            P p{(float)(x0 + 200.0 * (rng() % 100000) / 100000.0), (float)(-50.0 + 100.0 * (rng() % 100000) / 100000.0), (float)(10.0 * (rng() % 100000) / 100000.0)};
            const float q[3] = {p.x, p.y, p.z};
            if (r & 1) { m.io_[0].add_ids[(size_t)i] = next_id; m.io_[0].add_xyz[3 * (size_t)i] = q[0]; m.io_[0].add_xyz[3 * (size_t)i + 1] = q[1]; m.io_[0].add_xyz[3 * (size_t)i + 2] = q[2]; }
            else m.add(next_id, q);
            ref[next_id] = p;
            live.push_back(next_id);
            ++next_id;
        }
        if (r & 1) m.add_many(0, 3000);
        // remove ~1500 one by one (the voxel rule), some of them twice (the second one must miss)
        for (int i = 0; i < 1500 && !live.empty(); ++i) {
            const size_t k = rng() % live.size();
            const uint32_t id = live[k];
            live[k] = live.back();
            live.pop_back();
            const auto it = ref.find(id);
            if (it == ref.end()) continue;   // (went with a box)
            const float q[3] = {it->second.x, it->second.y, it->second.z};
            if (!m.remove(id, q)) { std::fprintf(stderr, "remove of %u missed\n", id); return 1; }
            if (i % 100 == 0) {
                if (m.remove(id, q)) { std::fprintf(stderr, "second remove of %u hit\n", id); return 1; }
                ++expect_missed;
            }
            ref.erase(it);
        }
        // every tenth round: the trim behind the window, with a face that cuts through buckets and one ON a coordinate
        if (r % 10 == 9) {
            float cut = (float)(x0 + 37.3);
            if (!ref.empty() && r % 20 == 19) cut = ref.begin()->second.x;   // max face on a point's coordinate: that point stays (p < max)
            const float box[6] = {-1e6f, -1e6f, -1e6f, cut, 1e6f, 1e6f};
            m.delete_box(box);
            for (auto it = ref.begin(); it != ref.end();)
                if (it->second.x >= box[0] && it->second.x < box[3]) it = ref.erase(it); else ++it;
            const float slab[6] = {(float)(x0 + 60.0), -10.0f, 2.0f, (float)(x0 + 90.0), 10.0f, 5.0f};   // a box in the middle: all six faces cut
            m.delete_box(slab);
            for (auto it = ref.begin(); it != ref.end();)
                if (it->second.x >= slab[0] && it->second.x < slab[3] && it->second.y >= slab[1] && it->second.y < slab[4] && it->second.z >= slab[2] &&
                    it->second.z < slab[5]) it = ref.erase(it); else ++it;
            x0 += 40.0;
        }
        if (r % 10 == 9 || r == rounds - 1) {
            if (!same(m, ref)) { std::fprintf(stderr, "round %d: the mirror differs from the reference\n", r); return 1; }
        }
        peak_bytes = std::max(peak_bytes, m.memory_bytes());
    }
    for (int k = 0; k < 64; ++k) m.give_back(64);   // (the reports after the last trim)
    const double per_point = (double)m.memory_bytes() / (double)std::max<int64_t>(m.size(), 1);
    std::printf("ids issued up to %u, live %lld, mirror %zu bytes (peak %zu) = %.1f bytes per live point, missed %lld\n", next_id, (long long)m.size(),
                m.memory_bytes(), peak_bytes, per_point, (long long)m.missed);
    if (next_id < (1u << 30)) return 1;
    if (m.missed != expect_missed) { std::fprintf(stderr, "missed %lld, expected %lld (the deliberate double removals)\n", (long long)m.missed, (long long)expect_missed); return 1; }
    // 16 bytes per point are the payload; marks, vector slack and the table may double or triple that -- an id-indexed table would
    // be 4 bytes x 2^30 = 4 GB here, 40 000 bytes per live point
    if (per_point > 96.0) { std::fprintf(stderr, "memory does not follow the live points\n"); return 1; }
    return 0;
}
