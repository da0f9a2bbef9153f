/* daliti_s2m_mirror.hpp -- the node's copy of the map, kept up to date from the engine's change log.
 *
 * The reference flattens the whole ikd-Tree and publishes it every frame (ikdtree.flatten -> featsFromMap ->
 * /Laser_map, eskf_lio/src/laserMapping.cpp:1170-1175, 1229-1235): O(map) per frame on the host.  With the engine the
 * map lives on the device; this mirror holds it on the host and applies what s2m_map_get_changes reports -- the points an
 * update added, the points it removed one by one, the boxes a field-of-view trim deleted -- so that a frame moves a few
 * thousand points across PCIe instead of the map, and a trim of millions of points costs what the BUCKETS it empties cost.
 *
 * Layout: the points sit in buckets of 4 m x 4 m x 4 m (an open-addressing table from the bucket's integer coordinates to its
 * arrays); inside a bucket ids ascend (the engine hands ids out in ascending order and reports additions in that order), so a
 * removed point is found by its place (the bucket) and a binary search over a few hundred ids, and is only marked (its x
 * becomes NaN; a bucket is compacted when more than half of it is marks, and gives its memory back when nothing is left).  A box delete drops the buckets that lie inside the
 * box whole -- no point is looked at -- and filters the ones its faces cut with the engine's own test (min <= p < max per
 * axis, float compares).  Memory: the live points (+ marks, at most as many), nothing that grows with the ids ever issued.
 * Publishing reads the buckets in turn (copy_points / for_each): the one O(map) pass a whole-map message needs anyway.
 * Header-only C++14 over the C ABI (include/daliti_s2m.h). */
#ifndef DALITI_S2M_MIRROR_HPP
#define DALITI_S2M_MIRROR_HPP

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "daliti_s2m.h"

struct s2m_map_mirror {
    uint64_t token = 0;
    int lag = 0;                 /* s2m_map_changes.lag: 1 = take the report of the previous call (nobody waits for the device) */
    int64_t resyncs = 0;         /* times the whole map had to be fetched (first call, a rebuild, a log overflow) */
    int64_t last_added = 0, last_removed = 0, last_boxes = 0;
    int64_t missed = 0;          /* removals that found no point (0 unless the follower and the engine are out of step) */
    std::vector<uint32_t> missed_ids;   /* ... the first few of them (diagnostic) */
    std::vector<float> missed_xyz;      /* ... with the coordinates the report gave */
    /* diagnostic, O(map): where the mirror holds point `id`, if anywhere */
    bool find_anywhere(uint32_t id, float *xyz) const
    {
        for (const Bucket &b : buckets_)
            for (const Pt &q : b.pts)
                if (q.id == id && q.x == q.x) { xyz[0] = q.x; xyz[1] = q.y; xyz[2] = q.z; return true; }
        return false;
    }

    int64_t size() const { return live_; }
    /* the points held, bucket by bucket (no particular order); ids optional.  xyz: 3 * size() floats */
    void copy_points(float *xyz, uint32_t *ids = nullptr) const
    {
        size_t at = 0;
        for (const Bucket &b : buckets_)
            for (const Pt &q : b.pts) {
                if (q.x != q.x) continue;   /* a mark */
                xyz[3 * at] = q.x; xyz[3 * at + 1] = q.y; xyz[3 * at + 2] = q.z;
                if (ids) ids[at] = q.id;
                ++at;
            }
    }
    template <class F>
    void for_each(F f) const   /* f(id, x, y, z) */
    {
        for (const Bucket &b : buckets_)
            for (const Pt &q : b.pts)
                if (q.x == q.x) f(q.id, q.x, q.y, q.z);
    }
    /* bytes the mirror holds (tests: it follows the live points, not the ids ever issued) */
    size_t memory_bytes() const
    {
        size_t s = (table_.capacity() + spare_.capacity()) * sizeof(Slot) + buckets_.capacity() * sizeof(Bucket) + free_.capacity() * sizeof(int32_t);
        for (const Bucket &b : buckets_) s += b.pts.capacity() * sizeof(Pt);
        s += later_.capacity() * sizeof(int32_t) + (group_.capacity() + touched_.capacity()) * sizeof(uint32_t);
        for (const IO &io : io_)
            s += (io.add_xyz.capacity() + io.rem_xyz.capacity() + io.box.capacity()) * sizeof(float) + (io.add_ids.capacity() + io.rem_ids.capacity()) * sizeof(uint32_t);
        return s;
    }

    /* brings the mirror to the engine's map (lag = 1: to the map as it was at the previous call); returns an S2M_* code.
     * = fetch + apply.  fetch talks to the engine (the handle's own thread) and fills one of two report buffers; apply works on
     * the mirror alone.  A node that publishes from its own thread calls fetch + hand_over on the engine's thread and
     * apply_report(k) on the publisher's -- the two must not run for the SAME buffer at once, and copy_points / for_each belong to
     * the thread that applies. */
    int update(s2m_engine *e)
    {
        const int rc = fetch(e);
        if (rc != S2M_OK) return rc;
        apply();
        return S2M_OK;
    }
    int fetch(s2m_engine *e)
    {
        IO &io = io_[cur_];
        if (io.add_ids.empty()) reserve_io(65536, 65536);
        io.resync = false;
        for (;;) {
            s2m_map_changes &c = io.c;
            std::memset(&c, 0, sizeof(c));
            c.added_xyz = io.add_xyz.data(); c.added_ids = io.add_ids.data(); c.capacity_added = (int64_t)io.add_ids.size();
            c.removed_xyz = io.rem_xyz.data(); c.removed_ids = io.rem_ids.data(); c.capacity_removed = (int64_t)io.rem_ids.size();
            c.boxes = io.box.data(); c.box_after_added = io.box_a.data(); c.box_after_removed = io.box_r.data(); c.capacity_boxes = (int64_t)io.box_a.size();
            c.lag = lag;
            int rc = s2m_map_get_changes(e, &token, &c);
            if (rc == S2M_ERR_CAPACITY) {  /* the counts came back: make room and ask again (the changes were kept) */
                reserve_io(std::max<int64_t>(c.n_added * 2, (int64_t)io.add_ids.size()), std::max<int64_t>(c.n_removed * 2, (int64_t)io.rem_ids.size()));
                if ((int64_t)io.box_a.size() < c.n_boxes) { io.box.resize((size_t)c.n_boxes * 6); io.box_a.resize((size_t)c.n_boxes); io.box_r.resize((size_t)c.n_boxes); }
                continue;
            }
            if (rc != S2M_OK) { c.n_added = c.n_removed = c.n_boxes = 0; return rc; }
            if (c.resync) {   /* the whole map, once: fetched here (the engine's thread), put into buckets by apply */
                c.n_added = c.n_removed = c.n_boxes = 0;
                int64_t m = 0;
                rc = s2m_map_get_points(e, nullptr, 0, &m);
                if (rc != S2M_OK) return rc;
                io.full_xyz.resize((size_t)std::max<int64_t>(m, 1) * 3);
                io.full_ids.resize((size_t)std::max<int64_t>(m, 1));
                rc = s2m_map_get_points(e, io.full_xyz.data(), m, &m);
                if (rc == S2M_OK) rc = s2m_map_get_ids(e, io.full_ids.data(), m, &m);
                if (rc != S2M_OK) return rc;
                io.full_xyz.resize((size_t)m * 3);
                io.full_ids.resize((size_t)m);
                io.resync = true;
            }
            return S2M_OK;
        }
    }
    void apply() { apply_report(cur_); }
    /* the report just fetched is handed to whoever applies it; the next fetch fills the other buffer.  Returns its number */
    int hand_over() { const int k = cur_; cur_ ^= 1; return k; }
    /* what fetch brought into buffer k, stretch by stretch: additions, then removals, then the box that closes the stretch */
    void apply_report(int k)
    {
        IO &io = io_[k];
        ap_ = &io;
        if (io.resync) {
            clear();
            for (size_t i = 0; i < io.full_ids.size(); ++i) add(io.full_ids[i], &io.full_xyz[3 * i]);   /* ascending ids: every bucket's ids ascend */
            std::vector<float>().swap(io.full_xyz);
            std::vector<uint32_t>().swap(io.full_ids);
            io.resync = false;
            ++resyncs;
            last_added = last_removed = last_boxes = 0;
            return;
        }
        const s2m_map_changes &c = io.c;
        last_added = c.n_added; last_removed = c.n_removed; last_boxes = c.n_boxes;
        if (c.n_boxes == 0) give_back(64);
        int64_t a0 = 0, r0 = 0;
        for (int64_t q = 0; q <= c.n_boxes; ++q) {
            const int64_t a1 = q < c.n_boxes ? io.box_a[(size_t)q] : c.n_added, r1 = q < c.n_boxes ? io.box_r[(size_t)q] : c.n_removed;
            add_many(a0, a1);
            remove_many(r0, r1);
            if (q < c.n_boxes) delete_box(&io.box[6 * (size_t)q]);
            a0 = a1; r0 = r1;
        }
        io.c.n_added = io.c.n_removed = io.c.n_boxes = 0;
    }

    /* ---- the pieces update() is made of, public so that they can be tested without a device ---- */
    void clear()
    {
        buckets_.clear(); free_.clear(); later_.clear(); table_.clear(); spare_.clear(); used_ = 0; gone_ = 0; live_ = 0; last_ = -1;
    }
    /* the arrays of up to n buckets that a box delete emptied go back to the allocator (unless points have moved in since) */
    void give_back(size_t n)
    {
        for (; n > 0 && !later_.empty(); --n) {
            Bucket &b = buckets_[(size_t)later_.back()];
            later_.pop_back();
            if (b.pts.empty() && b.pts.capacity() != 0) std::vector<Pt>().swap(b.pts);
        }
    }
    void add(uint32_t id, const float *p)
    {
        Bucket &b = bucket_of(p, true);
        b.pts.push_back(Pt{p[0], p[1], p[2], id});
        ++live_;
    }
    bool remove(uint32_t id, const float *p)
    {
        const int32_t at = find_bucket(key_of(p));
        if (at < 0) { ++missed; return false; }
        Bucket &b = buckets_[(size_t)at];
        size_t lo = 0, n = b.pts.size();   /* ids ascend inside a bucket: the first entry with id >= the one asked for */
        while (n > 0) {
            const size_t half = n >> 1;
            if (b.pts[lo + half].id < id) { lo += half + 1; n -= half + 1; } else n = half;
        }
        if (lo == b.pts.size() || b.pts[lo].id != id || b.pts[lo].x != b.pts[lo].x) { ++missed; return false; }
        mark(b, lo);
        settle(at);
        return true;
    }
    /* Delete_Point_Boxes for one box {min xyz, max xyz}: a point goes when min <= p < max on every axis */
    void delete_box(const float *box)
    {
        for (size_t at = 0; at < buckets_.size(); ++at) {
            Bucket &b = buckets_[at];
            if (b.pts.empty()) continue;
            bool inside = true, apart = false;
            for (int k = 0; k < 3; ++k) {
                const double lo = kEdge * (double)b.c[k], hi = lo + kEdge;   /* the bucket holds lo <= p < hi exactly */
                inside = inside && (double)box[k] <= lo && hi <= (double)box[3 + k];
                apart = apart || hi <= (double)box[k] || lo >= (double)box[3 + k];
            }
            if (apart) continue;
            if (inside) {   /* (its memory is given back over the next reports: a thousand free() calls are most of a trim otherwise) */
                live_ -= (int64_t)b.pts.size() - (int64_t)b.dead;
                b.pts.clear();
                b.dead = 0;
                mark_gone((int32_t)at);
                later_.push_back((int32_t)at);
                continue;
            }
            for (size_t i = 0, n = b.pts.size(); i < n; ++i) {
                const Pt &q = b.pts[i];
                if (q.x >= box[0] && q.x < box[3] && q.y >= box[1] && q.y < box[4] && q.z >= box[2] && q.z < box[5]) mark(b, i);   /* (a mark's NaN fails) */
            }
            settle((int32_t)at);
        }
    }

  private:
    static constexpr double kEdge = 4.0;   /* metres; a power of two: floor(p / 4) and 4 * c are exact in double */
    struct Pt { float x, y, z; uint32_t id; };
    struct Bucket {
        int32_t c[3] = {0, 0, 0};
        uint32_t dead = 0;
        uint32_t want = 0, off = 0; /* remove_many: removals of the stretch that fall into this bucket; their place in group_ */
        std::vector<Pt> pts;      /* ids ascending; a removed point stays as a mark (x = NaN) until the bucket is compacted */
    };
    struct Slot { uint64_t key; int32_t at; int32_t gone; };   /* at < 0: empty; gone: the bucket holds nothing (it leaves at the next rehash) */
    static constexpr uint64_t kBias = (uint64_t)1 << 20;

    static uint64_t key_of(const float *p)
    {
        uint64_t k = 0;
        for (int q = 0; q < 3; ++q) {
            const double v = (double)p[q] * (1.0 / kEdge);
            int64_t c = (int64_t)v;          /* floor without the library call */
            c -= (v < (double)c) ? 1 : 0;
            k = (k << 21) | (uint64_t)(c + (int64_t)kBias);
        }
        return k;
    }
    static size_t hash(uint64_t k)
    {
        k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
        return (size_t)k;
    }
    int32_t find_bucket(uint64_t key)
    {
        if (last_ >= 0 && last_key_ == key) return last_;
        if (table_.empty()) return -1;
        const size_t mask = table_.size() - 1;
        for (size_t i = hash(key) & mask;; i = (i + 1) & mask) {
            if (table_[i].at < 0) return -1;
            if (table_[i].key == key) { last_ = table_[i].at; last_key_ = key; last_slot_ = i; return last_; }
        }
    }
    Bucket &bucket_of(const float *p, bool)
    {
        const uint64_t key = key_of(p);
        int32_t at = find_bucket(key);
        if (at >= 0) {   /* (possibly one that a trim emptied: it kept its place in the table) */
            if (table_[last_slot_].gone) { table_[last_slot_].gone = 0; --gone_; }
            return buckets_[(size_t)at];
        }
        if ((used_ + 1) * 2 > table_.size()) rehash();
        if (!free_.empty()) { at = free_.back(); free_.pop_back(); }
        else {
            if (buckets_.capacity() == 0) buckets_.reserve((size_t)1 << 14);   /* (a drive opens a few dozen buckets per frame: not a reallocation of the pool every few hundred frames) */
            at = (int32_t)buckets_.size();
            buckets_.emplace_back();
        }
        Bucket &b = buckets_[(size_t)at];
        for (int q = 0; q < 3; ++q) b.c[q] = (int32_t)((int64_t)((key >> (21 * (2 - q))) & 0x1fffff) - (int64_t)kBias);
        b.dead = 0;
        last_slot_ = insert_slot(key, at);
        ++used_;
        last_ = at; last_key_ = key;
        return b;
    }
    size_t insert_slot(uint64_t key, int32_t at)
    {
        const size_t mask = table_.size() - 1;
        size_t i = hash(key) & mask;
        while (table_[i].at >= 0) i = (i + 1) & mask;
        table_[i].key = key; table_[i].at = at; table_[i].gone = 0;
        return i;
    }
    /* the bucket holds nothing any more: its slot says so (the rehash sheds it without looking at the bucket) */
    void mark_gone(int32_t at)
    {
        const Bucket &b = buckets_[(size_t)at];
        uint64_t key = 0;
        for (int q = 0; q < 3; ++q) key = (key << 21) | (uint64_t)((int64_t)b.c[q] + (int64_t)kBias);
        const size_t mask = table_.size() - 1;
        for (size_t i = hash(key) & mask;; i = (i + 1) & mask) {
            if (table_[i].at < 0) return;   /* (not in the table: cannot happen) */
            if (table_[i].key == key) {
                if (!table_[i].gone) { table_[i].gone = 1; ++gone_; }
                return;
            }
        }
    }
    /* the table grows -- or only sheds the buckets that trims have emptied since (they keep their slot until now, so that a
     * trim does not pay for thousands of deletions from the table) */
    void rehash()
    {
        /* (the slots know which buckets hold nothing: no bucket is looked at -- tens of thousands of them at random places were a
         * millisecond of cache misses on the publishing thread every few hundred frames of a drive) */
        const size_t keep = used_ - gone_;
        size_t cap = 1024;
        while (cap < 4 * (keep + 1)) cap *= 2;
        /* into the spare table: the two change places at every rehash, so that one that only sheds emptied buckets (the usual
         * case on a drive: the table keeps its size) writes into memory it has written before -- a fresh megabyte from the
         * allocator is a few hundred page faults */
        spare_.assign(cap, Slot{0, -1, 0});
        spare_.swap(table_);
        used_ = 0;
        gone_ = 0;
        for (size_t i = 0; i < spare_.size(); ++i) {
            const Slot &s = spare_[i];
            if (s.at < 0) continue;
            if (s.gone) { free_.push_back(s.at); continue; }
            insert_slot(s.key, s.at);
            ++used_;
        }
        last_ = -1;
    }
    /* the bucket gives its memory back; its slot in the table (marked) and its place in the pool are shed by the next rehash */
    void release(int32_t at)
    {
        Bucket &b = buckets_[(size_t)at];
        std::vector<Pt>().swap(b.pts);
        b.dead = 0;
        mark_gone(at);
    }
    void mark(Bucket &b, size_t i)
    {
        b.pts[i].x = std::numeric_limits<float>::quiet_NaN();
        ++b.dead;
        --live_;
    }
    /* after marks: an empty bucket leaves, one that is more marks than points is compacted (ids stay ascending) */
    void settle(int32_t at)
    {
        Bucket &b = buckets_[(size_t)at];
        if (b.dead == 0) return;
        if ((size_t)b.dead == b.pts.size()) { release(at); return; }
        if ((size_t)b.dead * 2 <= b.pts.size()) return;
        size_t w = 0;
        for (size_t i = 0, n = b.pts.size(); i < n; ++i)
            if (b.pts[i].x == b.pts[i].x) b.pts[w++] = b.pts[i];
        b.pts.resize(w);
        b.dead = 0;
    }
    /* The fetched additions [a0, a1) and removals [r0, r1).  A point's way to its place is a chain of dependent loads (table
     * slot -> bucket -> the end of its array, or a binary search): done point by point the loads of one point wait for each
     * other and the next point waits for them all.  In passes over the whole stretch -- find every bucket, then touch every
     * bucket's memory, then write -- the misses of different points are in flight together. */
    void add_many(int64_t a0, int64_t a1)
    {
        const size_t n = (size_t)(a1 - a0);
        if (n == 0) return;
        where_.resize(n);
        for (size_t i = 0; i < n; ++i) {
            const float *p = &ap_->add_xyz[3 * ((size_t)a0 + i)];
            Bucket &b = bucket_of(p, true);
            where_[i] = (int32_t)(&b - buckets_.data());
        }
        for (size_t i = 0; i < n; ++i) {
            const Bucket &b = buckets_[(size_t)where_[i]];
            if (!b.pts.empty()) __builtin_prefetch(&b.pts.back() + 1, 1);
        }
        for (size_t i = 0; i < n; ++i) {
            const float *p = &ap_->add_xyz[3 * ((size_t)a0 + i)];
            Bucket &b = buckets_[(size_t)where_[i]];
            b.pts.push_back(Pt{p[0], p[1], p[2], ap_->add_ids[(size_t)a0 + i]});
        }
        live_ += (int64_t)n;
    }
    void remove_many(int64_t r0, int64_t r1)
    {
        const size_t n = (size_t)(r1 - r0);
        if (n == 0) return;
        where_.resize(n);
        touched_.clear();
        for (size_t i = 0; i < n; ++i) {
            where_[i] = find_bucket(key_of(&ap_->rem_xyz[3 * ((size_t)r0 + i)]));
            if (where_[i] >= 0 && buckets_[(size_t)where_[i]].want++ == 0) touched_.push_back((uint32_t)where_[i]);
        }
        /* A bucket that loses a good part of its points in this stretch (the voxel rule thinning a dense map, laserMapping.cpp:
         * 590-640: a dozen old points per new one) is walked ONCE beside its removals in id order -- sequential memory -- instead
         * of one binary search per removal; group_ collects those removals bucket by bucket */
        size_t grouped = 0;
        for (uint32_t at : touched_) {
            Bucket &b = buckets_[at];
            if (b.want >= 8 && (size_t)b.want * 96 >= b.pts.size()) { b.off = (uint32_t)grouped; grouped += b.want; }
            else b.off = 0xffffffffu;
            b.want = 0;   /* (from here on: how many of the group are in) */
        }
        lo_.assign(n, 0);
        len_.resize(n);
        size_t longest = 0;
        if (grouped) group_.resize(grouped);
        for (size_t i = 0; i < n; ++i) {
            len_[i] = 0;
            if (where_[i] < 0) continue;
            Bucket &b = buckets_[(size_t)where_[i]];
            if (b.off != 0xffffffffu) { group_[b.off + b.want++] = (uint32_t)i; lo_[i] = 0xffffffffu; continue; }
            len_[i] = (uint32_t)b.pts.size();
            longest = std::max<size_t>(longest, len_[i]);
        }
        for (uint32_t at : touched_) {
            Bucket &b = buckets_[at];
            if (b.off == 0xffffffffu) continue;
            uint32_t *g = &group_[b.off];
            const size_t k = b.want;
            b.want = 0;
            std::sort(g, g + k, [&](uint32_t x, uint32_t y) { return ap_->rem_ids[(size_t)r0 + x] < ap_->rem_ids[(size_t)r0 + y]; });
            size_t j = 0;
            const size_t m = b.pts.size();
            for (size_t q = 0; q < k; ++q) {
                const uint32_t id = ap_->rem_ids[(size_t)r0 + g[q]];
                while (j < m && b.pts[j].id < id) ++j;
                lo_[g[q]] = (j < m && b.pts[j].id == id && b.pts[j].x == b.pts[j].x) ? (uint32_t)j : (uint32_t)m;   /* (m: not there) */
            }
        }
        /* the binary searches level by level, a few hundred points abreast (lo_ / len_ are every search's window): the probes of
         * one level are independent loads, the next level's are requested while this one's are compared -- and a few hundred
         * windows' lines stay in the first-level cache, which a whole report's do not */
        base_.resize(n);
        (void)longest;
        for (size_t c0 = 0; c0 < n; c0 += kAbreast) {
            const size_t c1 = std::min(n, c0 + kAbreast);
            uint32_t deepest = 0;
            for (size_t i = c0; i < c1; ++i) {
                if (len_[i] == 0) continue;
                base_[i] = buckets_[(size_t)where_[i]].pts.data();
                deepest = std::max(deepest, len_[i]);
                __builtin_prefetch(&base_[i][len_[i] >> 1]);
            }
            for (; deepest > 0; deepest >>= 1) {
                for (size_t i = c0; i < c1; ++i) {
                    if (len_[i] == 0) continue;
                    const uint32_t half = len_[i] >> 1;
                    if (base_[i][lo_[i] + half].id < ap_->rem_ids[(size_t)r0 + i]) { lo_[i] += half + 1; len_[i] -= half + 1; } else len_[i] = half;
                    if (len_[i] > 0) __builtin_prefetch(&base_[i][lo_[i] + (len_[i] >> 1)]);
                }
            }
        }
        for (size_t i = 0; i < n; ++i) {
            if (where_[i] < 0) {
                if (missed_ids.size() < 64) { missed_ids.push_back(ap_->rem_ids[(size_t)r0 + i]); missed_xyz.insert(missed_xyz.end(), &ap_->rem_xyz[3 * ((size_t)r0 + i)], &ap_->rem_xyz[3 * ((size_t)r0 + i)] + 3); }
                ++missed;
                continue;
            }
            Bucket &b = buckets_[(size_t)where_[i]];
            const size_t at = lo_[i];
            if (at >= b.pts.size() || b.pts[at].id != ap_->rem_ids[(size_t)r0 + i] || b.pts[at].x != b.pts[at].x) {
                if (missed_ids.size() < 64) { missed_ids.push_back(ap_->rem_ids[(size_t)r0 + i]); missed_xyz.insert(missed_xyz.end(), &ap_->rem_xyz[3 * ((size_t)r0 + i)], &ap_->rem_xyz[3 * ((size_t)r0 + i)] + 3); }
                ++missed;
                continue;
            }
            mark(b, at);
        }
        for (uint32_t at : touched_) settle((int32_t)at);
    }
    void reserve_io(int64_t na, int64_t nr)
    {
        IO &io = io_[cur_];
        io.add_ids.resize((size_t)na); io.add_xyz.resize((size_t)na * 3);
        io.rem_ids.resize((size_t)nr); io.rem_xyz.resize((size_t)nr * 3);
        if (io.box_a.empty()) { io.box.resize(6 * 64); io.box_a.resize(64); io.box_r.resize(64); }
    }

    std::vector<Bucket> buckets_;
    std::vector<int32_t> free_;
    std::vector<Slot> table_, spare_;
    size_t used_ = 0, gone_ = 0;   /* slots in use; of those, buckets that hold nothing */
    size_t last_slot_ = 0;
    int64_t live_ = 0;
    int32_t last_ = -1;
    uint64_t last_key_ = 0;
    struct IO {   /* one report: what fetch filled and apply has not applied yet */
        std::vector<float> add_xyz, rem_xyz, box, full_xyz;
        std::vector<uint32_t> add_ids, rem_ids, full_ids;
        std::vector<int64_t> box_a, box_r;
        s2m_map_changes c = {};
        bool resync = false;
    };
    IO io_[2];
    int cur_ = 0;
    IO *ap_ = &io_[0];   /* the report being applied */
    std::vector<int32_t> where_;
    std::vector<uint32_t> lo_, len_, group_, touched_;
    std::vector<const Pt *> base_;
    static constexpr size_t kAbreast = 256;
    std::vector<int32_t> later_;   /* buckets a box delete emptied whose arrays are still to be given back */
};

#endif /* DALITI_S2M_MIRROR_HPP */
