/* daliti_s2m_mirror.hpp -- the node's copy of the map, kept up to date from the engine's change log.
 *
 * The reference flattens the whole ikd-Tree and publishes it every frame (ikdtree.flatten -> featsFromMap ->
 * /Laser_map, eskf_lio/src/laserMapping.cpp:1170-1175, 1229-1235): O(map) per frame on the host.  With the engine the
 * map lives on the device; this mirror holds it on the host, keyed by point id, and
 * applies what s2m_map_get_changes reports -- the points an update added, the ids it removed -- so that a frame moves
 * a few thousand points across PCIe instead of the map.  Header-only C++ over the C ABI (include/daliti_s2m.h). */
#ifndef DALITI_S2M_MIRROR_HPP
#define DALITI_S2M_MIRROR_HPP

#include <algorithm>
#include <cstdint>
#include <vector>

#include "daliti_s2m.h"

struct s2m_map_mirror {
    std::vector<uint32_t> ids;   /* the ids of the points held, in no particular order after the first change ... */
    std::vector<float> xyz;      /* ... 3 floats per point, same order */
    uint64_t token = 0;
    int64_t resyncs = 0;         /* times the whole map had to be fetched (first call, a rebuild, a log overflow) */
    int64_t last_added = 0, last_removed = 0;

    /* brings the mirror to the engine's current map; returns an S2M_* code.  Cost: the changes, not the map -- a removed id is
     * found through a direct-address table (id -> slot) and its slot refilled from the end of the arrays. */
    int update(s2m_engine *e)
    {
        if (add_xyz_.empty()) { add_xyz_.resize(3 * 65536); add_ids_.resize(65536); rem_ids_.resize(65536); }
        for (;;) {
            int64_t na = 0, nr = 0;
            int32_t resync = 0;
            int rc = s2m_map_get_changes(e, &token, add_xyz_.data(), add_ids_.data(), (int64_t)add_ids_.size(), &na, rem_ids_.data(),
                                         (int64_t)rem_ids_.size(), &nr, &resync);
            if (rc == S2M_ERR_CAPACITY) {  /* the counts came back: make room and ask again (the changes were kept) */
                if ((int64_t)add_ids_.size() < na) { add_ids_.resize((size_t)na * 2); add_xyz_.resize(add_ids_.size() * 3); }
                if ((int64_t)rem_ids_.size() < nr) rem_ids_.resize((size_t)nr * 2);
                continue;
            }
            if (rc != S2M_OK) return rc;
            if (resync) {
                int64_t m = 0;
                rc = s2m_map_get_points(e, nullptr, 0, &m);
                if (rc != S2M_OK) return rc;
                xyz.resize((size_t)std::max<int64_t>(m, 1) * 3);
                ids.resize((size_t)std::max<int64_t>(m, 1));
                rc = s2m_map_get_points(e, xyz.data(), m, &m);
                if (rc == S2M_OK) rc = s2m_map_get_ids(e, ids.data(), m, &m);
                if (rc != S2M_OK) return rc;
                xyz.resize((size_t)m * 3);
                ids.resize((size_t)m);
                slot_.assign(slot_.size(), kNone);
                for (size_t i = 0; i < ids.size(); ++i) place(ids[i], (uint32_t)i);
                ++resyncs;
                last_added = last_removed = 0;
                return S2M_OK;
            }
            last_added = na;
            last_removed = nr;
            /* additions first: a point that came and went between two calls is in both lists */
            for (int64_t i = 0; i < na; ++i) {
                place(add_ids_[(size_t)i], (uint32_t)ids.size());
                ids.push_back(add_ids_[(size_t)i]);
                xyz.insert(xyz.end(), add_xyz_.begin() + 3 * i, add_xyz_.begin() + 3 * i + 3);
            }
            for (int64_t k = 0; k < nr; ++k) {
                const uint32_t id = rem_ids_[(size_t)k];
                if (id >= slot_.size() || slot_[id] == kNone) continue;
                const uint32_t at = slot_[id], last = (uint32_t)ids.size() - 1u;
                slot_[id] = kNone;
                if (at != last) {
                    ids[at] = ids[last];
                    xyz[3 * (size_t)at] = xyz[3 * (size_t)last]; xyz[3 * (size_t)at + 1] = xyz[3 * (size_t)last + 1];
                    xyz[3 * (size_t)at + 2] = xyz[3 * (size_t)last + 2];
                    slot_[ids[at]] = at;
                }
                ids.pop_back();
                xyz.resize(xyz.size() - 3);
            }
            return S2M_OK;
        }
    }

  private:
    static constexpr uint32_t kNone = 0xffffffffu;
    void place(uint32_t id, uint32_t at)
    {
        if (id >= slot_.size()) slot_.resize((size_t)id + (size_t)id / 2 + 1024, kNone);
        slot_[id] = at;
    }
    std::vector<float> add_xyz_;
    std::vector<uint32_t> add_ids_, rem_ids_, slot_;   /* slot_[id] = where the point sits in ids / xyz */
};

#endif /* DALITI_S2M_MIRROR_HPP */
