// s2m_mapedit.hip -- how the brick-grid map changes after a scan (the verdicts come from s2m_mapupd.hip):
//   slab_update   the touched bricks are rewritten where they stand, new bricks take their stretch from the room of the brick
//                 in front of them -- cost proportional to the update (the reference inserts per point in O(log M):
//                 KD_TREE::Add_Points / Delete_Point_Boxes, eskf_lio/include/ikd-Tree/ikd_Tree.cpp:477-573, 631-658)
//   merge_update  the whole map is re-laid out (nothing re-sorted), with room behind every brick for the next in-place updates
// Both leave the map in the order a fresh build of the same point list would produce: (brick, cell, point id).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "s2m_map_internal.h"

namespace s2m {

// ---- merge update: the map after an incremental update WITHOUT a new sort ----------------------------------
// Input: the current map (sorted points, their ids pidx, their sorted keys keys_alt; holes allowed), the update's
// verdicts (alive_s[sorted position]: 0 = removed by this update, or a hole) and its staged new points (caller order of
// the new map: survivors in id order, then the staged points -- the same list update_finish would hand to a full build).
// The new points are sorted by their key in the CURRENT grid (thousands, not millions), every one finds its place among
// the old keys by binary search -- BEHIND the old points of its own cell (upper bound): the new points carry the highest
// ids, so the merged array is ordered by (brick, cell, id) exactly like a fresh build of the same list, and the search's
// tie order (sorted position) does not depend on which way the map was produced.  A surviving old point j moves to
// j - removed-before(j) + new-in-front-of(j): the first term from a rank over the removed positions (a bit mask per 64
// positions + the number removed before every word: L2-resident), the second from one scalar binary search per wave in
// the new points' sorted places plus the few entries inside the wave's 64 positions -- no map-sized scan.  Old points
// move with one coalesced read and one scattered-but-monotone write and keep their ids.  ~0.3 GB of traffic at 5 M
// points instead of a 5 M-key radix sort, a bounding-box pass and a gather.
__global__ __launch_bounds__(256) void dead_words_kernel(int64_t m, const uint8_t *__restrict__ alive_s,
                                                         unsigned long long *__restrict__ word_s, uint32_t *__restrict__ cnt,
                                                         uint32_t *__restrict__ outside_flag)
{
    // the removed points by sorted position: where a survivor lands.  (Point ids are stable -- s2m_kernels.h, MapBuffers::pidx
    // -- so nothing is renumbered: rounds 2-3 also ranked the removed CALLER INDICES here and every survivor gathered its
    // new index from that rank.)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 8) outside_flag[i] = 0u;  // merge_newkey_kernel's flag and excess words, tail_key_kernel's count
    if (i == 0) cnt[(m + 63) >> 6] = 0u;  // the scan's spare element
    const unsigned long long ws = __ballot(i < m && alive_s[i] == 0);
    if ((threadIdx.x & 63) == 0 && (i >> 6) <= ((m - 1) >> 6)) {
        word_s[i >> 6] = ws;
        cnt[i >> 6] = (uint32_t)__popcll(ws);
    }
}
struct DeadRank {  // per 64 positions: mask of the removed ones, number removed before the word (one 16-byte gather)
    unsigned long long word;
    uint32_t prefix, pad;
};
// entry `words` (one past the last word) is an empty mask with the total as its prefix: a position may equal m
__global__ __launch_bounds__(256) void dead_pack_kernel(int64_t words, const unsigned long long *__restrict__ word_s,
                                                        const uint32_t *__restrict__ prefix, DeadRank *__restrict__ out_s)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w > words) return;
    out_s[w] = DeadRank{w < words ? word_s[w] : 0ull, prefix[w], 0u};
}
__device__ __forceinline__ uint32_t dead_before(uint32_t ci, const DeadRank *__restrict__ rank)
{
    const uint4 r = *reinterpret_cast<const uint4 *>(rank + (ci >> 6));
    const unsigned long long w = ((unsigned long long)r.y << 32) | r.x;
    return r.z + (uint32_t)__popcll(w & ((1ull << (ci & 63u)) - 1ull));
}

// How far a batch of new points reaches beyond the bricks in use: ext6[0..3) = bricks below blo, ext6[3..6) = bricks above bhi
// (maxima over the batch; zero-initialised by the caller).  Called by every lane of a wave; `valid` lanes contribute.
__device__ __forceinline__ void report_excess(const Grid &g, int bx, int by, int bz, bool valid, uint32_t *__restrict__ ext6)
{
    const int b[3] = {bx, by, bz};
    uint32_t e[6];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        e[k] = valid ? (uint32_t)max(g.blo[k] - b[k], 0) : 0u;
        e[3 + k] = valid ? (uint32_t)max(b[k] - g.bhi[k], 0) : 0u;
    }
    wave_max6_to(e, ext6);
}

// outside[0] |= 1 when a point's cell is beyond the representable range (nothing can hold it: the caller rebuilds around a new
// origin); outside[1..7) = how far the batch reaches beyond the bricks in use (report_excess)
__global__ __launch_bounds__(256) void merge_newkey_kernel(const float4 *__restrict__ stage, int n, Grid g, int n_tail,
                                                           uint64_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                           uint32_t *__restrict__ outside)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    keys += n_tail; vals += n_tail;  // (the staged points follow the tail's in the list that is sorted)
    bool out = false;
    int cx = 0, cy = 0, cz = 0;
    if (i < n) {
        const float4 p = stage[i];
        cx = cell_coord(p.x, g.ox, g.inv_c); cy = cell_coord(p.y, g.oy, g.inv_c); cz = cell_coord(p.z, g.oz, g.inv_c);
        out = !cell_representable(cx, cy, cz);
        keys[i] = out ? ~0ull : point_key(cx, cy, cz);
        vals[i] = (uint32_t)(n_tail + i);
    }
    report_excess(g, cx >> 3, cy >> 3, cz >> 3, i < n && !out, outside + 1);
    if (__syncthreads_or(out ? 1 : 0) && threadIdx.x == 0) atomicOr(outside, 1u);
}

// upper bound of every new key among the old keys (the first old point of a LATER cell): the new point goes in front
// of that old point, behind the old points of its own cell.  The keys are sorted, so lb is ascending.
__global__ __launch_bounds__(256) void merge_lb_kernel(int n, const uint64_t *__restrict__ nkeys, const uint64_t *__restrict__ okeys,
                                                       int64_t m, uint32_t *__restrict__ lb)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = nkeys[i];
    int64_t lo = 0, hi = m;  // first j with okeys[j] > k
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (okeys[mid] <= k) lo = mid + 1; else hi = mid;
    }
    lb[i] = (uint32_t)lo;
}

// surviving old point j -> (survivors before j) + (new points that go in front of j or of an earlier old point): no
// map-sized scan and no per-position counter array -- the survivors before j come from the removed-by-position rank, the
// new points from one scalar binary search per wave in the sorted lb[] (a few thousand entries, L2-resident) plus the few
// entries inside the wave's 64 positions
__global__ __launch_bounds__(256) void merge_old_kernel(int64_t m, const float4 *__restrict__ pts, const uint32_t *__restrict__ pidx,
                                                        const uint64_t *__restrict__ okeys,
                                                        const uint8_t *__restrict__ alive_s,
                                                        const DeadRank *__restrict__ rank_s, const uint32_t *__restrict__ lb, int n_new,
                                                        float4 *__restrict__ npts, uint32_t *__restrict__ npidx,
                                                        uint64_t *__restrict__ nkeys_out)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // wave-uniform: number of new points with lb < the wave's first position
    const uint32_t j0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(j - (threadIdx.x & 63)));
    int lo = 0, hi = n_new;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (lb[mid] < j0) lo = mid + 1; else hi = mid;
    }
    if (j >= m || !alive_s[j]) return;
    int k = lo;
    while (k < n_new && lb[k] <= (uint32_t)j) ++k;
    const uint32_t pos = ((uint32_t)j - dead_before((uint32_t)j, rank_s)) + (uint32_t)k;
    const float4 p = pts[j];
    npts[pos] = make_map_point(p.x, p.y, map_point_z(p), pos);
    npidx[pos] = pidx[j];  // the point keeps its id
    nkeys_out[pos] = okeys[j];
}

// new point i (sorted order) -> (survivors before its lb) + i.  The "new" points of a merge are the live points of the TAIL
// (bricks that in-place updates moved or opened behind the key-ordered part of the array: they keep their ids) followed by
// the staged points of the update (ids next_id, next_id + 1, ...); value t < n_tail names tail position main_ext + t, the
// others staged point t - n_tail.  Dead tail positions carry the key ~0: sorted last, skipped here.
__global__ __launch_bounds__(256) void merge_new_kernel(int n, const uint64_t *__restrict__ nkeys, const uint32_t *__restrict__ nvals,
                                                        const uint32_t *__restrict__ lb, const DeadRank *__restrict__ rank_s,
                                                        const float4 *__restrict__ stage, uint32_t next_id, int n_tail, int64_t main_ext,
                                                        const float4 *__restrict__ pts, const uint32_t *__restrict__ pidx,
                                                        float4 *__restrict__ npts, uint64_t *__restrict__ nkeys_out,
                                                        uint32_t *__restrict__ npidx)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = nkeys[i];
    if (key == ~0ull) return;
    const uint32_t l = lb[i];
    const uint32_t pos = (l - dead_before(l, rank_s)) + (uint32_t)i;
    const uint32_t t = nvals[i];
    if ((int)t < n_tail) {
        const float4 p = pts[main_ext + t];
        npts[pos] = make_map_point(p.x, p.y, map_point_z(p), pos);
        npidx[pos] = pidx[main_ext + t];
    } else {
        const float4 p = stage[t - (uint32_t)n_tail];
        npts[pos] = make_map_point(p.x, p.y, p.z, pos);
        npidx[pos] = next_id + (t - (uint32_t)n_tail);  // staged order = caller order of the new points
    }
    nkeys_out[pos] = key;
}
// keys of the tail's positions for that sort: the point's own key, ~0 for a position that holds no point (counted in *dead)
__global__ __launch_bounds__(256) void tail_key_kernel(int n_tail, int64_t main_ext, const uint64_t *__restrict__ okeys,
                                                       const uint8_t *__restrict__ alive_s, uint64_t *__restrict__ keys,
                                                       uint32_t *__restrict__ vals, uint32_t *__restrict__ dead)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool d = false;
    if (i < n_tail) {
        d = alive_s[main_ext + i] == 0;
        keys[i] = d ? ~0ull : okeys[main_ext + i];
        vals[i] = (uint32_t)i;
    }
    const int c = __syncthreads_count(d ? 1 : 0);
    if (threadIdx.x == 0 && c) atomicAdd(dead, (uint32_t)c);
}

// ---- slack for the in-place updates -------------------------------------------------------------------------
// A merge (or a rebuild inside an update) leaves the points densely packed; slab_update can only rewrite a brick where it
// stands while the brick fits between its first position and the next brick's.  So a map that is being maintained gets
// room behind every brick: an eighth of its points, 16 to 512 positions.  One more pass over the map on the frames that
// merge anyway (the frames in between are the ones that stay in place): the points move to `position + slack of the bricks
// before`, the holes are filled with sentinel points (never a neighbour: +inf distance), id ~0 and the brick's largest key
// (the key array stays sorted), prefix words and brick starts are shifted.
__global__ __launch_bounds__(256) void slack_size_kernel(int64_t bound, const uint32_t *__restrict__ bricks_dev, int64_t m,
                                                         const uint32_t *__restrict__ bstart, const uint64_t *__restrict__ bkey, Grid g,
                                                         uint32_t *__restrict__ grow, int by_growth, uint32_t *__restrict__ slack)
{
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id > bound) return;
    const int64_t bricks = (int64_t)*bricks_dev;
    uint32_t v = 0u;
    if (id < bricks) {
        const uint32_t cnt = (id + 1 < bricks ? bstart[id + 1] : (uint32_t)m) - bstart[id];
        // an eighth of the brick -- or four times what it has gained since the room was last laid out, if that is more: the
        // bricks that grow (a frontier, a surface the sensor keeps refining) are the ones that would force the next merge
        const uint32_t slot = top_slot_of_key(g, bkey[id]);
        const uint32_t gained = grow[slot];
        v = max(min(max(cnt >> 3, 16u), 512u), by_growth ? min(4u * gained, 4096u) : 0u);
        grow[slot] = gained >> 1;  // the history fades: half of it counts towards the next layout
    }
    slack[id] = v;
}
__global__ __launch_bounds__(256) void slack_move_kernel(int64_t m, const float4 *__restrict__ pts, const uint32_t *__restrict__ pidx,
                                                         const uint64_t *__restrict__ keys, Grid g,
                                                         const uint32_t *__restrict__ shift, float4 *__restrict__ npts,
                                                         uint32_t *__restrict__ npidx, uint64_t *__restrict__ nkeys)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const uint64_t k = keys[j];
    const uint32_t dst = (uint32_t)j + shift[g.top[top_slot_of_key(g, k >> 9)].x - 1u];
    const float4 p = pts[j];
    npts[dst] = make_map_point(p.x, p.y, map_point_z(p), dst);
    npidx[dst] = pidx[j];
    nkeys[dst] = k;
}
// one wave per brick: the holes behind it, its prefix words and its start
__global__ __launch_bounds__(256) void slack_brick_kernel(const uint32_t *__restrict__ bricks_dev, int64_t m, uint32_t *__restrict__ bstart,
                                                          uint32_t *__restrict__ bend,
                                                          const uint64_t *__restrict__ bkey, const uint32_t *__restrict__ shift,
                                                          uint32_t *__restrict__ tab, float4 *__restrict__ npts,
                                                          uint32_t *__restrict__ npidx, uint64_t *__restrict__ nkeys)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t bricks = (int64_t)*bricks_dev;
    const int64_t id = (int64_t)blockIdx.x * 4 + wave;
    if (id >= bricks) return;
    const uint32_t e0 = id + 1 < bricks ? bstart[id + 1] : (uint32_t)m;  // the brick's end in the dense layout (starts are shifted by a later launch)
    const uint32_t sh = shift[id], room = shift[id + 1] - sh;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    uint32_t *t = tab + id * kBrickStride;
    for (int c = lane; c <= kBrickCells; c += 64) t[c] += sh;
    const uint64_t filler = (bkey[id] << 9) | 511ull;
    if (lane == 0) bend[id] = e0 + sh + room;  // = the next brick's new start
    for (uint32_t h = e0 + sh + (uint32_t)lane; h < e0 + sh + room; h += 64u) {
        npts[h] = make_map_point(3.0e38f, 3.0e38f, 3.0e38f, 0xffffffffu);
        npidx[h] = 0xffffffffu;
        nkeys[h] = filler;
    }
}
// (a separate launch: a wave of slack_brick_kernel reads its successor's start)
__global__ __launch_bounds__(256) void slack_start_kernel(const uint32_t *__restrict__ bricks_dev, const uint32_t *__restrict__ shift,
                                                          uint32_t *__restrict__ bstart)
{
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id < (int64_t)*bricks_dev) bstart[id] += shift[id];
}

// every position of the tail holds nothing: far coordinates (never a neighbour), id ~0, key ~0
__global__ __launch_bounds__(256) void tail_fill_kernel(int64_t from, int64_t to, float4 *__restrict__ pts, uint32_t *__restrict__ pidx,
                                                        uint64_t *__restrict__ keys)
{
    const int64_t j = from + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= to) return;
    pts[j] = make_map_point(3.0e38f, 3.0e38f, 3.0e38f, 0xffffffffu);
    pidx[j] = 0xffffffffu;
    keys[j] = ~0ull;
}

// How many positions the tail of a layout of m points gets: room for the bricks that in-place updates open or move until the
// next merge.  Generous (HBM is not what this engine is short of -- 29 bytes per position): twice the map's points, at
// least four million.  Measured on the drive of DESIGN section 6: ~13 K positions per frame at 0.5 m cells (4 m bricks), ~6 K at
// the reference's map density (1.25 m cells, 10 m bricks that are moved whole): 600+ frames per merge at the smallest size.
static int64_t tail_size_for(int64_t m) { return std::max<int64_t>(2 * m, (int64_t)1 << 22); }

// dense layout of m points (buf.pts / pidx / keys_alt, tables built) -> layout with slack behind every brick and the tail
// behind the last one; g.m becomes the new extent, buf.main_ext the extent of the key-ordered part
static hipError_t spread_with_slack(MapBuffers &buf, Grid &g, int64_t bricks_bound, hipStream_t st)
{
    const int64_t m = g.m;
    buf.main_ext = m;
    buf.tail_used = 0;
    if (m <= 0 || bricks_bound <= 0) return hipSuccess;
    // slack <= max(cnt / 8, 16) + 4 x growth per brick; the host knows the sum of the growth as a bound
    int64_t room_bound = m / 8 + 16 * bricks_bound + 64 + 4 * buf.added_since_layout;
    int by_growth = 1;
    // room for the slack and a full-sized tail (a map that started small -- one scan -- and has grown since)
    S2M_TRY(map_grow_scratch(buf, m + room_bound + tail_size_for(m) + 64, m, st));
    if (m + room_bound > buf.scratch_cap) {  // no space for the growth-sized part: the plain eighth then
        room_bound = m / 8 + 16 * bricks_bound + 64;
        by_growth = 0;
    }
    const int64_t ext_bound = m + room_bound;
    if (ext_bound > buf.scratch_cap || ext_bound >= ((int64_t)1 << 31) || bricks_bound + 2 > buf.scratch_cap) return hipSuccess;  // stays dense
    const int64_t tail = std::max<int64_t>(std::min<int64_t>(std::min<int64_t>(tail_size_for(m), buf.scratch_cap - ext_bound), ((int64_t)1 << 31) - 64 - ext_bound), 0);
    S2M_TRY(map_ensure((void **)&buf.pts2, &buf.pts2_cap, ext_bound + tail + kSentinelPoints, sizeof(float4), map_headroom_for(ext_bound)));
    S2M_TRY(map_ensure((void **)&buf.pidx2, &buf.pidx2_cap, ext_bound + tail + 1, sizeof(uint32_t), map_headroom_for(ext_bound)));
    const uint32_t *bricks_dev = buf.counters + kBricksWord;
    uint32_t *slack = buf.work_a, *shift = buf.work_b;
    size_t tmp = 0;
    S2M_TRY(rocprim::exclusive_scan(nullptr, tmp, slack, shift, 0u, (size_t)bricks_bound + 1, rocprim::plus<uint32_t>(), st));
    S2M_TRY(map_ensure_sort_tmp(buf, tmp));
    hipLaunchKernelGGL(slack_size_kernel, dim3((unsigned)((bricks_bound + 256) / 256)), dim3(256), 0, st, bricks_bound, bricks_dev, m,
                       buf.bstart, buf.bkey, g, buf.grow, by_growth, slack);
    buf.added_since_layout = (buf.added_since_layout + 1) / 2;
    size_t t = buf.sort_tmp_bytes;
    S2M_TRY(rocprim::exclusive_scan(buf.sort_tmp, t, slack, shift, 0u, (size_t)bricks_bound + 1, rocprim::plus<uint32_t>(), st));
    {   // the total room: the map's new extent
        const uint32_t *src[1] = {shift + bricks_bound};
        S2M_TRY(mail_post(buf.mail, src, 1, st));
    }
    hipLaunchKernelGGL(slack_move_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, m, buf.pts, buf.pidx, buf.keys_alt,
                       g, shift, buf.pts2, buf.pidx2, buf.keys);
    hipLaunchKernelGGL(slack_brick_kernel, dim3((unsigned)((bricks_bound + 3) / 4)), dim3(256), 0, st, bricks_dev, m, buf.bstart, buf.bend,
                       buf.bkey, shift, buf.tab, buf.pts2, buf.pidx2, buf.keys);
    hipLaunchKernelGGL(slack_start_kernel, dim3((unsigned)((bricks_bound + 255) / 256)), dim3(256), 0, st, bricks_dev, shift, buf.bstart);
    uint32_t room = 0;
    S2M_TRY(mail_collect(buf.mail, 1, &room, st));
    const int64_t main_ext = m + (int64_t)room, ext = main_ext + tail;
    if (tail > 0)
        hipLaunchKernelGGL(tail_fill_kernel, dim3((unsigned)((tail + 255) / 256)), dim3(256), 0, st, main_ext, ext, buf.pts2, buf.pidx2, buf.keys);
    S2M_TRY(map_put_sentinels(buf.pts2, ext, st));
    // the tail's cursor lives on the device (the in-place update's plan kernel hands positions out)
    launch_set_word(buf.counters + kTailWord, (uint32_t)main_ext, st);
    std::swap(buf.pts, buf.pts2); std::swap(buf.pts_cap, buf.pts2_cap);
    std::swap(buf.pidx, buf.pidx2); std::swap(buf.pidx_cap, buf.pidx2_cap);
    std::swap(buf.keys, buf.keys_alt);
    buf.main_ext = main_ext;
    g.m = ext;
    g.sent_off = (ext + kSentinelPoints) < ((int64_t)1 << 28) ? (uint32_t)(ext << 4) : 0u;
    g.pts = buf.pts; g.pidx = buf.pidx;
    return hipGetLastError();
}

hipError_t merge_update(MapBuffers &buf, Grid &g, MapStats &stats, const uint8_t *alive_s,
                        const float4 *stage, int64_t n_new, bool &merged, hipStream_t st, bool with_slack)
{
    merged = false;
    // [0, m): the key-ordered part of the array; [m, m + n_tail): the part of the tail that in-place updates have handed out
    const int64_t m = buf.main_ext > 0 && buf.main_ext <= g.m ? buf.main_ext : g.m;
    const int64_t n_tail = std::min<int64_t>(buf.tail_used, g.m - m);
    if (m <= 0 || !buf.keys_alt || g.pts != buf.pts || g.m > buf.scratch_cap) return hipSuccess;
    if (n_new >= ((int64_t)1 << 30) || n_new + n_tail >= ((int64_t)1 << 31)) return hipSuccess;
    S2M_TRY(map_grow_scratch(buf, m + n_new + n_tail + 64, g.m, st));  // (the merged map, before its room is laid out)
    const int n = (int)(n_new + n_tail), nt = (int)n_tail;  // the points the merge sorts: the tail's, then the staged ones
    const int64_t words = (m + 63) / 64;
    // dword: [masks of the removed positions | packed records], words + 1 each; work_c: [removed per word | exclusive prefix];
    // mv: [stage positions | upper bounds] of the new points
    const int64_t w1 = words + 2;  // (even offsets keep the 16-byte records aligned)
    S2M_TRY(map_ensure((void **)&buf.dword, &buf.dword_cap, 3 * w1 + 2, sizeof(unsigned long long), 3 * (words / 4 + 1024)));
    unsigned long long *word_s = buf.dword;
    DeadRank *rank_s = reinterpret_cast<DeadRank *>(buf.dword + (w1 + (w1 & 1)));
    uint32_t *dcnt = buf.work_c, *dprefix = buf.work_c + (words + 1);
    if (2 * (words + 1) > buf.scratch_cap + 1) return hipSuccess;
    if (buf.next_id + n_new >= ((int64_t)1 << 32) - 2) return hipSuccess;  // ids exhausted: a rebuild makes them dense again
    const unsigned kbits = 64;  // the box-independent key, ~0 for the tail's empty positions (a batch of thousands: the merge sort path)
    // capacity for the merged map before anything is enqueued (the exact size arrives with the hand-back below)
    const int64_t m_bound = m + n;
    if (m_bound > buf.scratch_cap || m_bound >= ((int64_t)1 << 31)) return hipSuccess;
    size_t tmp = 0, tmp3 = 0;
    S2M_TRY(rocprim::exclusive_scan(nullptr, tmp, dcnt, dprefix, 0u, (size_t)words + 1, rocprim::plus<uint32_t>(), st));
    if (n > 0)
        S2M_TRY(rocprim::radix_sort_pairs(nullptr, tmp3, buf.keys, buf.keys, buf.vals, buf.vals, (size_t)n, 0, kbits, st));
    S2M_TRY(map_ensure_sort_tmp(buf, std::max(tmp, tmp3)));
    if (n > 0) {
        S2M_TRY(map_ensure((void **)&buf.mk, &buf.mk_cap, n, sizeof(uint64_t), n / 2 + 4096));
        S2M_TRY(map_ensure((void **)&buf.mv, &buf.mv_cap, 2 * (int64_t)n, sizeof(uint32_t), n + 8192));
    }
    S2M_TRY(map_ensure((void **)&buf.pts2, &buf.pts2_cap, m_bound + kSentinelPoints, sizeof(float4), map_headroom_for(m_bound)));
    S2M_TRY(map_ensure((void **)&buf.pidx2, &buf.pidx2_cap, m_bound + 1, sizeof(uint32_t), map_headroom_for(m_bound)));

    // the rank of the removed positions (element `words` of the counts is zero: the prefix there is the total)
    hipLaunchKernelGGL(dead_words_kernel, dim3((unsigned)((m + 256) / 256)), dim3(256), 0, st, m, alive_s, word_s, dcnt,
                       buf.counters + 8);
    size_t t = buf.sort_tmp_bytes;
    S2M_TRY(rocprim::exclusive_scan(buf.sort_tmp, t, dcnt, dprefix, 0u, (size_t)words + 1, rocprim::plus<uint32_t>(), st));
    hipLaunchKernelGGL(dead_pack_kernel, dim3((unsigned)((words + 256) / 256)), dim3(256), 0, st, words, word_s, dprefix, rank_s);
    uint64_t *nk_sorted = buf.mk;
    uint32_t *nv_sorted = buf.mv, *lb = buf.mv + n;
    // keys of the points to merge in (keys / vals are free until the merge writes them): the tail's own, the staged points' from
    // their cells; the second kernel also says whether one of them cannot be represented and how far they reach beyond the bounds
    if (nt > 0)
        hipLaunchKernelGGL(tail_key_kernel, dim3((nt + 255) / 256), dim3(256), 0, st, nt, m, buf.keys_alt, alive_s, buf.keys, buf.vals,
                           buf.counters + 15);
    if (n_new > 0)
        hipLaunchKernelGGL(merge_newkey_kernel, dim3((unsigned)((n_new + 255) / 256)), dim3(256), 0, st, stage, (int)n_new, g, nt, buf.keys,
                           buf.vals, buf.counters + 8);
    // the one hand-back -- number of dead, "a new point cannot be represented", how far the new points reach beyond the
    // bricks in use, the tail's empty positions -- is posted here and collected after the merge kernels have been enqueued:
    // they write into the spare arrays, which only become the map if the answer allows it
    const uint32_t *dead_dev = dprefix + words;
    {
        const uint32_t *src[9] = {dead_dev, buf.counters + 8, buf.counters + 9, buf.counters + 10, buf.counters + 11, buf.counters + 12,
                                  buf.counters + 13, buf.counters + 14, buf.counters + 15};
        S2M_TRY(mail_post(buf.mail, src, 9, st));
    }
    if (n > 0) {
        t = buf.sort_tmp_bytes;
        S2M_TRY(rocprim::radix_sort_pairs(buf.sort_tmp, t, buf.keys, nk_sorted, buf.vals, nv_sorted, (size_t)n, 0, kbits, st));
        hipLaunchKernelGGL(merge_lb_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, nk_sorted, buf.keys_alt, m, lb);
    }
    hipLaunchKernelGGL(merge_old_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, m, buf.pts, buf.pidx, buf.keys_alt,
                       alive_s, rank_s, lb, n, buf.pts2, buf.pidx2, buf.keys);
    if (n > 0)
        hipLaunchKernelGGL(merge_new_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, nk_sorted, nv_sorted, lb, rank_s, stage,
                           (uint32_t)buf.next_id, nt, m, buf.pts, buf.pidx, buf.pts2, buf.keys, buf.pidx2);
    uint32_t dead = 0, outside = 0, tail_dead = 0;
    int lo[3], hi[3];
    {
        uint32_t v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        S2M_TRY(mail_collect(buf.mail, 9, v, st));
        dead = v[0];
        outside = v[1];
        tail_dead = v[8];
        for (int k = 0; k < 3; ++k) { lo[k] = g.blo[k] - (n_new > 0 ? (int)v[2 + k] : 0); hi[k] = g.bhi[k] + (n_new > 0 ? (int)v[5 + k] : 0); }
    }
    if (outside) return hipSuccess;     // full rebuild around a new origin; what was written to the spare arrays is dropped
    if (dead == 0 && n_new == 0 && nt == 0) {  // nothing was removed, nothing is added, nothing to bring home: the map stands as it is
        merged = true;
        return hipSuccess;
    }
    const int64_t survivors = m - (int64_t)dead;
    const int64_t m_new = survivors + ((int64_t)n - (int64_t)tail_dead);
    if (m_new == 0) return hipSuccess;
    buf.next_id += n_new;
    if (dead > 0 || tail_dead > 0) buf.ids_dense = false;
    S2M_TRY(map_put_sentinels(buf.pts2, m_new, st));
    // the merged arrays become the map
    std::swap(buf.pts, buf.pts2); std::swap(buf.pts_cap, buf.pts2_cap);
    std::swap(buf.pidx, buf.pidx2); std::swap(buf.pidx_cap, buf.pidx2_cap);
    std::swap(buf.keys, buf.keys_alt);
    g.m = m_new;
    g.live = m_new;
    buf.main_ext = m_new;
    buf.tail_used = 0;
    g.sent_off = (m_new + kSentinelPoints) < ((int64_t)1 << 28) ? (uint32_t)(m_new << 4) : 0u;
    // every new point opens at most one brick: no read-back before the tables are built -- unless the box of the bricks in
    // use, grown by the new points, has left the window of the top array: then the exact box of the merged map is fetched
    // (the bounds only ever grow between such events: bricks that the field-of-view trim emptied are dropped here) and the
    // array resized if even that does not fit
    const int64_t brick_bound = std::min<int64_t>(stats.bricks + n_new, m_new);
    const bool fits = (int64_t)hi[0] - lo[0] <= (int64_t)g.tmx && (int64_t)hi[1] - lo[1] <= (int64_t)g.tmy && (int64_t)hi[2] - lo[2] <= (int64_t)g.tmz;
    bool too_large = false;
    if (fits) S2M_TRY(map_set_window(buf, g, lo, hi, too_large, nullptr, st));
    else ++buf.n_relaid;
    S2M_TRY(map_build_tables(buf, g, buf.keys_alt, m_new, stats, st, brick_bound, !fits, &too_large));
    if (too_large) return hipErrorOutOfMemory;  // (a box of bricks beyond 2^28 top entries: not a map this engine can hold)
    g.top = buf.top; g.tab = buf.tab; g.pts = buf.pts; g.pidx = buf.pidx;
    if (with_slack) S2M_TRY(spread_with_slack(buf, g, brick_bound, st));
    merged = true;
    return hipSuccess;
}

// ---- in-place update: only the touched bricks are rewritten -------------------------------------------------
// The merge update above moves every point of the map; what an ordinary frame changes is a few thousand voxels in a few
// hundred bricks.  When every touched brick still fits the stretch of pts it owns -- [its first position, the next brick's
// first position): a fresh layout leaves no slack, but every removal does -- and no new point opens a brick, each touched
// brick is rewritten where it stands by one workgroup: survivors and new points staged in LDS, merged by cell (new points
// behind the old ones of their cell: ascending id, the documented order), written back with their keys and ids, the
// brick's 513 prefix words and row mask rebuilt from the merged cells; what is left of the stretch becomes a hole
// (alive_s 0, id ~0, the brick's largest key so that the key array stays sorted for the next merge).  Positions outside the
// touched bricks do not move, so the cost follows the scan, not the map.  Anything else -- a brick that would overflow, a
// new brick, a point outside the grid, a brick too large to stage -- is decided on the device BEFORE anything is written
// and falls back to the merge update (which lays the map out densely again).
constexpr int kSlabMax = 2048;  // points of one brick the rewrite stages in LDS (47 KB: three workgroups per CU) ...
constexpr int kSlabBig = 6144;  // ... and in the second, rarely launched form for crowded bricks (134 KB of dynamic LDS: one per CU)
// kSlabOutside: a new point's cell cannot be represented (rebuild around a new origin); kSlabWindow: the box of the bricks in
// use, grown by the new points, no longer fits the window of the top array (the host re-lays it and tries again)
enum : uint32_t { kSlabOutside = 1u, kSlabNewBrick = 2u, kSlabOverflow = 4u, kSlabWindow = 8u, kSlabTooBig = 16u,
                  kSlabRefuse = 32u };  // (refuse: the one-workgroup preparation cannot number the batch's bricks, or one of them is crowded: the separate kernels run)
// words of the update's counters behind `flags`: [0] outcome bits, [1] points removed, [2] bricks opened, [3] points gained
// by bricks, [4] crowded bricks among the touched ones, [5..11) how far the new points reach beyond the bricks in use,
// [11] bricks moved to the tail (the opened ones included), [12] bricks touched
constexpr int kSlabWords = 13;

__global__ __launch_bounds__(256) void slab_key_kernel(const float4 *__restrict__ stage, int n, Grid g, uint64_t *__restrict__ keys,
                                                       uint32_t *__restrict__ vals, uint8_t *__restrict__ bmark,
                                                       uint32_t *__restrict__ flags, const uint32_t *__restrict__ n_dev)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t f = 0u;
    int cx = 0, cy = 0, cz = 0;
    // n is the host's bound when the count stayed on the device: the places behind the count sort to the end as ~0 and no
    // later step takes them for points (a real key never has all of its 63 sorted bits set: kCellLimit)
    const int n_act = n_dev ? min((int)*n_dev, n) : n;
    if (i >= n_act && i < n) { keys[i] = ~0ull; vals[i] = (uint32_t)i; }
    if (i < n_act) {
        const float4 p = stage[i];
        cx = cell_coord(p.x, g.ox, g.inv_c); cy = cell_coord(p.y, g.oy, g.inv_c); cz = cell_coord(p.z, g.oz, g.inv_c);
        vals[i] = (uint32_t)i;
        if (!cell_representable(cx, cy, cz)) {
            keys[i] = ~0ull;
            f = kSlabOutside;
        } else {
            keys[i] = point_key(cx, cy, cz);
            if (brick_in_bounds(g, cx >> 3, cy >> 3, cz >> 3)) {
                const uint32_t idp1 = g.top[top_slot(g, cx >> 3, cy >> 3, cz >> 3)].x;
                if (idp1 != 0u) bmark[idp1 - 1u] |= 2u;  // (every writer of this byte in this launch stores the same value)
            }
            // (a point whose brick does not exist yet, inside the bounds or beyond them: slab_newbrick_kernel opens it)
        }
    }
    report_excess(g, cx >> 3, cy >> 3, cz >> 3, i < n_act && f == 0u, flags + 5);
    const unsigned long long any = __ballot(f != 0u);
    if (any != 0ull) {
        uint32_t w = f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) w |= __shfl_xor(w, off, 64);
        if ((threadIdx.x & 63) == 0) atomicOr(flags, w);
    }
}

// the entry of brick key b in the top array, if the brick lies inside the bounds in use (else "no brick": 0)
__device__ __forceinline__ uint32_t slab_brick_id(const Grid &g, uint64_t b)
{
    int bx, by, bz;
    brick_coords(b, bx, by, bz);
    return brick_in_bounds(g, bx, by, bz) ? g.top[top_slot(g, bx, by, bz)].x : 0u;
}

// What a brick that cannot stay where it stands -- one that this update opens (a sensor that moves sees new ground every
// frame), or one whose points no longer fit the stretch it owns -- asks of the TAIL, the part of the array behind the
// key-ordered one: its points and room to grow (a brick at the frontier fills up over the next frames as the sensor comes
// closer).  Handing such bricks a fresh stretch instead of re-laying the map out keeps every frame's cost proportional to
// the scan; the price is that a brick's position no longer says where its key stands among the others -- between two merges
// the order of s2m_map_get_order is (brick, cell, id) only INSIDE every brick (include/daliti_s2m.h).
__device__ __forceinline__ uint32_t reloc_need(uint32_t total, bool opened)
{
    return total + (opened ? max(total, 96u) : max(total >> 1, 64u));
}

// Bricks that this update opens, in three parallel steps over the sorted new keys: (1) the first point of every brick that
// does not exist yet raises a flag -- after the check that the box of the bricks in use, grown by the new points, still
// fits the window of the top array (else kSlabWindow: nothing is written, the host re-lays the array and calls again);
// (2) an exclusive scan numbers the flags in key order; (3) every flagged point opens its brick: top entry, key, an empty
// table row, the marks "new" (4) + "touched" (2) -- the plan gives it its stretch.  More bricks than there are table rows
// to spare: kSlabNewBrick (the merge re-lays the map and its tables out).
__global__ __launch_bounds__(256) void slab_head_kernel(const uint64_t *__restrict__ nk, int n_new, Grid g, uint32_t *__restrict__ head,
                                                        uint32_t *__restrict__ bricks_dev, uint32_t *__restrict__ flags, uint2 *__restrict__ run)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (flags[0] & kSlabOutside) return;  // (uniform; the key kernel has finished)
    {
        bool fits = true;
        const uint32_t tm[3] = {g.tmx, g.tmy, g.tmz};
#pragma unroll
        for (int q = 0; q < 3; ++q)
            fits = fits && ((int64_t)g.bhi[q] + flags[8 + q]) - ((int64_t)g.blo[q] - flags[5 + q]) <= (int64_t)tm[q];
        if (!fits) {  // (uniform: every thread reads the same six words)
            if (k == 0) atomicOr(flags, kSlabWindow);
            return;
        }
    }
    if (k == 0) bricks_dev[1] = bricks_dev[0];  // the bricks before this update: where the new ids start
    if (k >= n_new) return;
    if (nk[k] == ~0ull) { head[k] = 0u; return; }  // (behind the device's count of the staged points: not a point)
    const uint64_t b = nk[k] >> 9;
    const bool first = k == 0 || (nk[k - 1] >> 9) != b, last = k == n_new - 1 || (nk[k + 1] >> 9) != b;
    head[k] = (first && slab_brick_id(g, b) == 0u) ? 1u : 0u;
    // where the brick's new points stand among the sorted ones, left at the brick's slot of the top array: the plan reads
    // two words instead of searching the keys twice (26 dependent loads per touched brick were the plan's critical path)
    if (first || last) {
        uint2 *r = run + top_slot_of_key(g, b);
        if (first) r->x = (uint32_t)k;
        if (last) r->y = (uint32_t)k + 1u;
    }
}
__global__ __launch_bounds__(256) void slab_open_kernel(const uint64_t *__restrict__ nk, int n_new, Grid g, const uint32_t *__restrict__ head,
                                                        const uint32_t *__restrict__ rank, uint4 *__restrict__ top, uint32_t *__restrict__ tab,
                                                        uint32_t *__restrict__ bend, uint64_t *__restrict__ bkey, uint8_t *__restrict__ bmark,
                                                        uint32_t *__restrict__ bricks_dev, int max_new, uint32_t *__restrict__ flags)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_new || (flags[0] & (kSlabOutside | kSlabWindow))) return;
    const uint32_t first_id = bricks_dev[1];
    if (k == n_new - 1) {  // the last point knows how many bricks open
        const uint32_t nh = rank[k] + head[k];
        if (nh > (uint32_t)max_new) atomicOr(flags, kSlabNewBrick);
        else { bricks_dev[0] = first_id + nh; flags[2] = nh; }
    }
    if (!head[k] || rank[k] >= (uint32_t)max_new) return;
    const uint32_t id = first_id + rank[k];
    const uint64_t b = nk[k] >> 9;
    top[top_slot_of_key(g, b)] = make_uint4(id + 1u, 0u, 0u, 0u);
    bkey[id] = b;
    bend[id] = 0u;
    tab[id * kBrickStride] = 0u;
    tab[id * kBrickStride + kBrickCells] = 0u;
    bmark[id] = 6u;
}

// The preparation of an in-place update in ONE workgroup, for the batch of a scan (up to kPrepMax staged points; larger batches
// take the kernels above): keys, their order, the runs of the touched bricks, the bricks that open.  Eleven launches -- the key
// kernel, rocprim's sort of 63-bit keys (six kernels for 6 k pairs, 42 us), head flags, their scan (two), the opening --
// were half of what the host enqueues for a map update and the frame is bound by that, not by the device.
// The order is found without a general sort: a scan's batch falls into a few hundred bricks of the batch's own box of bricks
// (numbered z, y, x like the brick key: at most kPrepBricks of them), a dozen points each -- a histogram over the box in LDS
// (packed 16-bit counters, LDS atomics), its prefix sum, every point dropped into its brick's stretch in arrival order, and
// then every point counts the points of its brick in front of it by (cell, staged index): its place.  The same order as the
// 63-bit keys' (a block-wide radix sort of 32-bit keys, three passes of rocprim::block_radix_sort, took 21 us of the
// kernel's 39).  A brick with more than kPrepCrowd new points, or a box of more bricks, is left to the separate kernels.
constexpr int kPrepThreads = 1024, kPrepItems = 8, kPrepMax = kPrepThreads * kPrepItems;
constexpr int kPrepBricks = 32768, kPrepCrowd = 128;
constexpr size_t kPrepLds = ((size_t)kPrepMax + (size_t)kPrepBricks / 2 + 1) * sizeof(uint32_t);
// q = a / d for a workgroup-uniform divisor (a < 2^24, d < 2^22: exact in float up to one unit, corrected)
__device__ __forceinline__ uint32_t prep_div(uint32_t a, uint32_t d, float inv_d)
{
    uint32_t q = (uint32_t)((float)a * inv_d);
    if (q * d > a) --q;
    else if ((q + 1u) * d <= a) ++q;
    return q;
}
// The points: src_a[0 .. na) -- all of them, or the first *n_dev when the count stayed on the device, or (flag_a given: the
// voxel rule's batch as update_add(defer) left it, StagePending) the ones whose flag is set -- followed by src_b[0 .. n_b).
// With stage_out the kernel writes the staging list itself (the winners in batch order, then src_b) and leaves their number
// in *count_out: the two launches that did that (compact1_kernel, append_kernel: 19 us of one-workgroup kernels) are gone.
__global__ __launch_bounds__(kPrepThreads) void slab_prepare_kernel(const float4 *__restrict__ src_a, const uint32_t *__restrict__ flag_a, int na,
                                                                    const float4 *__restrict__ src_b, int n_b, const uint32_t *__restrict__ n_dev,
                                                                    const uint64_t *__restrict__ vkey, unsigned long long *__restrict__ vtab,
                                                                    float4 *__restrict__ stage_out, uint32_t *__restrict__ count_out,
                                                                    Grid g, uint64_t *__restrict__ nk, uint32_t *__restrict__ nv,
                                                                    uint8_t *__restrict__ bmark, uint32_t *__restrict__ flags,
                                                                    uint2 *__restrict__ run, uint32_t *__restrict__ bricks_dev,
                                                                    uint4 *__restrict__ top, uint32_t *__restrict__ tab,
                                                                    uint32_t *__restrict__ bend, uint64_t *__restrict__ bkey, int max_new)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char prep_lds[];
    uint32_t *seg = reinterpret_cast<uint32_t *>(prep_lds);  // (cell << 13 | staged index) of the points, brick by brick
    uint32_t *hist = seg + kPrepMax;                         // per brick of the box, two to a word: count, then offset
    __shared__ int s_lo[kPrepThreads / 64][3], s_hi[kPrepThreads / 64][3];
    __shared__ uint32_t s_bad[kPrepThreads / 64], s_part[kPrepThreads / 64], s_max[kPrepThreads / 64];
    __shared__ int s_box[6];
    __shared__ uint32_t s_f;
    __shared__ uint32_t s_headbits[kPrepBricks / 32], s_headrank[kPrepBricks / 32];
    __shared__ uint8_t s_flag[kPrepMax];
    __shared__ uint16_t s_rank[kPrepMax];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n = na + n_b;                                      // the host's bound of the number of points
    const int na_act = n_dev ? min((int)*n_dev, na) : na;        // (without flags: the first na_act of src_a)
    const uint32_t first_id = bricks_dev[0];
    auto half = [&](uint32_t e) { return (hist[e >> 1] >> ((e & 1u) * 16u)) & 0xffffu; };
    // 0. with flags: the winners' places in the staging list (their rank in batch order), the winner table emptied
    uint32_t n_win = (uint32_t)na_act;
    if (flag_a) {
        uint32_t fl[kPrepItems];
        uint64_t vk[kPrepItems];
#pragma unroll
        for (int j = 0; j < kPrepItems; ++j) {  // (all the loads first: sixteen round trips in flight, not one after the other)
            const int i = j * kPrepThreads + tid;
            fl[j] = i < na ? flag_a[i] : 0u;
            vk[j] = (i < na && vtab) ? vkey[i] : 0ull;
        }
#pragma unroll
        for (int j = 0; j < kPrepItems; ++j) {
            const int i = j * kPrepThreads + tid;
            s_flag[i] = fl[j] != 0u ? 1 : 0;
            if (i < na && vtab) vtab[vk[j]] = ~0ull;  // the voxel's slot of the winner table back to "empty"
        }
        __syncthreads();
        uint32_t mine = 0u;
#pragma unroll
        for (int j = 0; j < kPrepItems; ++j) mine += s_flag[tid * kPrepItems + j];
        uint32_t inc = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t a = __shfl_up(inc, off, 64);
            if (lane >= off) inc += a;
        }
        if (lane == 63) s_part[wave] = inc;
        __syncthreads();
        uint32_t before = inc - mine;
        n_win = 0u;
        for (int w = 0; w < kPrepThreads / 64; ++w) {
            if (w < wave) before += s_part[w];
            n_win += s_part[w];
        }
        // every candidate's number of winners in front of it (its place in the staging list if it is one)
#pragma unroll
        for (int j = 0; j < kPrepItems; ++j) {
            s_rank[tid * kPrepItems + j] = (uint16_t)before;
            before += s_flag[tid * kPrepItems + j];
        }
        __syncthreads();
    }
    if (tid == 0 && count_out) *count_out = n_win + (uint32_t)n_b;
    // 1. the cells of the points (striped: thread t holds candidates t, t + 1024, ..), the batch's box of bricks
    int bx[kPrepItems], by[kPrepItems], bz[kPrepItems];
    uint32_t cell[kPrepItems], tix[kPrepItems];  // tix: the point's index in the staging list
    int lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
    uint32_t bad = 0u;
#pragma unroll
    for (int j = 0; j < kPrepItems; ++j) {
        const int i = j * kPrepThreads + tid;
        cell[j] = 0xffffffffu;  // (not a point)
        bx[j] = by[j] = bz[j] = 0;
        tix[j] = 0u;
        bool is = false;
        if (i < na) {
            if (flag_a) {
                is = s_flag[i] != 0;
                tix[j] = s_rank[i];
            } else {
                is = i < na_act;
                tix[j] = (uint32_t)i;
            }
        } else if (i < n) {
            is = true;
            tix[j] = n_win + (uint32_t)(i - na);
        }
        if (is) {
            float4 p = i < na ? src_a[i] : src_b[i - na];
            if (stage_out) { p.w = 0.0f; stage_out[tix[j]] = p; }
            const int cx = cell_coord(p.x, g.ox, g.inv_c), cy = cell_coord(p.y, g.oy, g.inv_c), cz = cell_coord(p.z, g.oz, g.inv_c);
            if (!cell_representable(cx, cy, cz)) bad = kSlabOutside;
            else {
                bx[j] = cx >> 3; by[j] = cy >> 3; bz[j] = cz >> 3;
                cell[j] = (uint32_t)((((cz & 7) << 3) | (cy & 7)) << 3 | (cx & 7));
                lo[0] = min(lo[0], bx[j]); lo[1] = min(lo[1], by[j]); lo[2] = min(lo[2], bz[j]);
                hi[0] = max(hi[0], bx[j]); hi[1] = max(hi[1], by[j]); hi[2] = max(hi[2], bz[j]);
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = min(lo[k], __shfl_xor(lo[k], off, 64));
            hi[k] = max(hi[k], __shfl_xor(hi[k], off, 64));
        }
        bad |= __shfl_xor(bad, off, 64);
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { s_lo[wave][k] = lo[k]; s_hi[wave][k] = hi[k]; }
        s_bad[wave] = bad;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t f = 0u;
        int l[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, h[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
        for (int w = 0; w < kPrepThreads / 64; ++w) {
            f |= s_bad[w];
            for (int k = 0; k < 3; ++k) { l[k] = min(l[k], s_lo[w][k]); h[k] = max(h[k], s_hi[w][k]); }
        }
        for (int k = 0; k < 3; ++k) { s_box[k] = l[k]; s_box[3 + k] = h[k]; }
        s_f = f;
    }
    __syncthreads();
    if (s_f != 0u) {  // a cell that cannot be represented: the caller rebuilds around a new origin
        if (tid == 0) atomicOr(flags, s_f);
        return;
    }
    const bool any = s_box[0] <= s_box[3];
    int b0[3] = {0, 0, 0}, nb[3] = {1, 1, 1};
    // 2. how far the batch reaches beyond the bricks in use, and whether the grown box still fits the window of the top array
    // (a batch that does not fit leaves its keys -- the host re-lays the top array around them -- and nothing else)
    bool fits = true;
    {
        const uint32_t tm[3] = {g.tmx, g.tmy, g.tmz};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const uint32_t below = any ? (uint32_t)max(g.blo[k] - s_box[k], 0) : 0u, above = any ? (uint32_t)max(s_box[3 + k] - g.bhi[k], 0) : 0u;
            if (tid == 0) { flags[5 + k] = below; flags[8 + k] = above; }
            fits = fits && ((int64_t)g.bhi[k] + above) - ((int64_t)g.blo[k] - below) <= (int64_t)tm[k];
            if (any) { b0[k] = s_box[k]; nb[k] = s_box[3 + k] - s_box[k] + 1; }
        }
    }
    const uint64_t vol64 = (uint64_t)nb[0] * (uint64_t)nb[1] * (uint64_t)nb[2];
    if (vol64 >= (uint64_t)kPrepBricks) {  // a batch strewn over more bricks than the histogram holds
        if (tid == 0) atomicOr(flags, kSlabRefuse);
        return;
    }
    const uint32_t vol = (uint32_t)vol64, words = (vol + 2u) / 2u;  // entries 0 .. vol (the last one: the end of the points)
    // 3. the histogram over the box, its prefix sum
    for (uint32_t w = tid; w < words; w += kPrepThreads) hist[w] = 0u;
    for (uint32_t w = tid; w < kPrepBricks / 32; w += kPrepThreads) s_headbits[w] = 0u;
    __syncthreads();
    uint32_t bidx[kPrepItems], arrival[kPrepItems];
#pragma unroll
    for (int j = 0; j < kPrepItems; ++j) {
        bidx[j] = 0xffffffffu;
        arrival[j] = 0u;
        if (cell[j] != 0xffffffffu) {
            const uint32_t b = (uint32_t)(((bz[j] - b0[2]) * nb[1] + (by[j] - b0[1])) * nb[0] + (bx[j] - b0[0]));
            const uint32_t sh = (b & 1u) * 16u;
            arrival[j] = (atomicAdd(&hist[b >> 1], 1u << sh) >> sh) & 0xffffu;
            bidx[j] = b;
        }
    }
    __syncthreads();
    const uint32_t per = (words + kPrepThreads - 1) / kPrepThreads;
    const uint32_t w0 = min((uint32_t)tid * per, words), w1 = min(w0 + per, words);
    uint32_t sum = 0u, most = 0u;
    for (uint32_t w = w0; w < w1; ++w) {
        const uint32_t v = hist[w];
        sum += (v & 0xffffu) + (v >> 16);
        most = max(most, max(v & 0xffffu, v >> 16));
    }
    uint32_t in = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t a = __shfl_up(in, off, 64);
        if (lane >= off) in += a;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) most = max(most, __shfl_xor(most, off, 64));
    if (lane == 63) s_part[wave] = in;
    if (lane == 0) s_max[wave] = most;
    __syncthreads();
    uint32_t at = in - sum;
    most = 0u;
    for (int w = 0; w < kPrepThreads / 64; ++w) {
        if (w < wave) at += s_part[w];
        most = max(most, s_max[w]);
    }
    if (most > (uint32_t)kPrepCrowd) {  // (uniform) a crowded brick: ranking by counting would be quadratic in it
        if (tid == 0) atomicOr(flags, kSlabRefuse);
        return;
    }
    for (uint32_t w = w0; w < w1; ++w) {
        const uint32_t v = hist[w], c0 = v & 0xffffu, c1 = v >> 16;
        hist[w] = at | ((at + c0) << 16);
        at += c0 + c1;
    }
    __syncthreads();
    if (tid == 0 && fits) bricks_dev[1] = first_id;  // the bricks before this update: where the new ids start
    // 4. every point into its brick's stretch, then to its place there: (cell, staged index) ascending
#pragma unroll
    for (int j = 0; j < kPrepItems; ++j)
        if (bidx[j] != 0xffffffffu) seg[half(bidx[j]) + arrival[j]] = (cell[j] << 13) | tix[j];
    __syncthreads();
    const uint32_t total = half(vol);
#pragma unroll
    for (int j = 0; j < kPrepItems; ++j) {
        if (bidx[j] == 0xffffffffu) continue;
        const uint32_t off = half(bidx[j]), end = half(bidx[j] + 1u), mine = (cell[j] << 13) | tix[j];
        uint32_t sp = off + arrival[j];
        if (fits) {
            sp = off;
            for (uint32_t q = off; q < end; ++q) sp += seg[q] < mine ? 1u : 0u;
        }
        nk[sp] = (brick_key(bx[j], by[j], bz[j]) << 9) | (uint64_t)cell[j];
        nv[sp] = tix[j];
    }
    for (uint32_t sp = total + (uint32_t)tid; sp < (uint32_t)n; sp += kPrepThreads) {  // behind the device's count: not points
        nk[sp] = ~0ull;
        nv[sp] = sp;
    }
    if (!fits) {  // (uniform)
        if (tid == 0) atomicOr(flags, kSlabWindow);
        return;
    }
    // 5. the bricks of the box that got points: their runs, their marks -- and the ones that do not exist yet
    const uint32_t nbx = (uint32_t)nb[0], nbxy = (uint32_t)nb[0] * (uint32_t)nb[1];
    const float inv_x = 1.0f / (float)nbx, inv_xy = 1.0f / (float)nbxy;
    for (uint32_t e = tid; e < vol; e += kPrepThreads) {
        const uint32_t off = half(e), end = half(e + 1u);
        if (end == off) continue;
        const uint32_t qz = prep_div(e, nbxy, inv_xy), rem = e - qz * nbxy, qy = prep_div(rem, nbx, inv_x);
        const int rx = (int)(rem - qy * nbx) + b0[0], ry = (int)qy + b0[1], rz = (int)qz + b0[2];
        const uint32_t slot = top_slot(g, rx, ry, rz);
        run[slot] = make_uint2(off, end);
        const uint32_t idp1 = brick_in_bounds(g, rx, ry, rz) ? g.top[slot].x : 0u;
        if (idp1 != 0u) {  // a brick that new points fall into: its mark, without waiting for the byte (an atomic OR on its word)
            const uint32_t id = idp1 - 1u;
            atomicOr(reinterpret_cast<uint32_t *>(bmark + (id & ~3u)), 2u << ((id & 3u) * 8u));
        } else atomicOr(&s_headbits[e >> 5], 1u << (e & 31u));
    }
    __syncthreads();
    // the bricks that open, numbered in key order (= the order of the box's entries)
    {
        const uint32_t v = tid < kPrepBricks / 32 ? s_headbits[tid] : 0u;  // (kPrepBricks / 32 = the workgroup's size)
        const uint32_t c = (uint32_t)__popc(v);
        uint32_t inc = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t a = __shfl_up(inc, off, 64);
            if (lane >= off) inc += a;
        }
        __syncthreads();  // (s_part of the prefix sum above has been read)
        if (lane == 63) s_part[wave] = inc;
        __syncthreads();
        uint32_t before = inc - c;
        for (int w = 0; w < wave; ++w) before += s_part[w];
        s_headrank[tid] = before;
    }
    __syncthreads();
    uint32_t nh = 0u;
    for (int w = 0; w < kPrepThreads / 64; ++w) nh += s_part[w];
    if (nh > (uint32_t)max_new) {  // more bricks than there are table rows to spare: the merge re-lays the map and its tables out
        if (tid == 0) atomicOr(flags, kSlabNewBrick);
        return;
    }
    if (tid == 0) { bricks_dev[0] = first_id + nh; flags[2] = nh; }
    for (uint32_t e = tid; e < vol; e += kPrepThreads) {
        const uint32_t bits = s_headbits[e >> 5];
        if (!((bits >> (e & 31u)) & 1u)) continue;
        const uint32_t id = first_id + s_headrank[e >> 5] + (uint32_t)__popc(bits & ((1u << (e & 31u)) - 1u));
        const uint32_t qz = prep_div(e, nbxy, inv_xy), rem = e - qz * nbxy, qy = prep_div(rem, nbx, inv_x);
        const int rx = (int)(rem - qy * nbx) + b0[0], ry = (int)qy + b0[1], rz = (int)qz + b0[2];
        top[top_slot(g, rx, ry, rz)] = make_uint4(id + 1u, 0u, 0u, 0u);
        bkey[id] = brick_key(rx, ry, rz);
        bend[id] = 0u;
        tab[(int64_t)id * kBrickStride] = 0u;
        tab[(int64_t)id * kBrickStride + kBrickCells] = 0u;
        bmark[id] = 6u;  // "new" (4) + "touched" (2)
    }
}

// What the plan found for a touched brick
struct BrickPlan {
    uint32_t lo, n_b;    // its new points: nk[lo .. lo + n_b)
    uint32_t need;       // positions it asks of the tail (0: it stays where it stands)
    uint32_t info;       // bits 0..23 points removed from it, bit 24 crowded (the large staging form), bit 25 cannot be staged at all
    uint32_t gained, pad[3];
};

static_assert(sizeof(BrickPlan) == 32, "BrickPlan is two 16-byte words");

// one wave per brick: does the touched brick fit where it stands -- and if not, how much of the tail it needs.  No shared
// counters here (a few thousand touched bricks adding to the same words cost more than the rest of the kernel: same-line
// atomics serialise at ~11 ns each): every brick leaves its record, slab_alloc_kernel adds them up.
__global__ __launch_bounds__(256) void slab_plan_kernel(const uint32_t *__restrict__ bricks_dev, Grid g,
                                                        const uint32_t *__restrict__ bend, const uint32_t *__restrict__ tab,
                                                        const uint64_t *__restrict__ bkey, const uint8_t *__restrict__ bmark,
                                                        const uint8_t *__restrict__ alive_s, const uint64_t *__restrict__ nk, int n_new,
                                                        uint32_t *__restrict__ grow, BrickPlan *__restrict__ plan,
                                                        const uint32_t *__restrict__ flags, int big_ok, const uint2 *__restrict__ run)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t bricks = (int64_t)*bricks_dev;
    const int64_t id = (int64_t)blockIdx.x * 4 + wave;
    if (id >= bricks || bmark[id] == 0) return;
    if (flags[0] & (kSlabOutside | kSlabWindow | kSlabNewBrick)) return;  // the host deals with those first: count nothing twice
    const uint32_t base = tab[id * kBrickStride], end = tab[id * kBrickStride + kBrickCells];
    const uint32_t cap_end = bend[id];
    int alive = 0;
    for (uint32_t j = base + (uint32_t)lane; j < end; j += 64u) alive += alive_s[j] ? 1 : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) alive += __shfl_xor(alive, off, 64);
    if (lane != 0) return;
    const uint64_t b = bkey[id];
    int lo = 0, n_b = 0;
    if (n_new > 0 && (bmark[id] & 2u)) {  // (the mark of a brick that new points fall into: slab_key_kernel, slab_open_kernel)
        const uint2 r = run[top_slot_of_key(g, b)];
        lo = (int)r.x;
        n_b = (int)(r.y - r.x);
    }
    const uint32_t total = (uint32_t)alive + (uint32_t)n_b;
    // what the rewrite has to stage at most: the brick's positions so far plus its new points
    const uint32_t stage = (end - base) + (uint32_t)n_b;
    BrickPlan p = {(uint32_t)lo, (uint32_t)n_b, 0u, min((end - base) - (uint32_t)alive, 0xffffffu), 0u, {0u, 0u, 0u}};
    if (stage > (uint32_t)(big_ok ? kSlabBig : kSlabMax)) p.info |= 1u << 25;
    else if (stage > (uint32_t)kSlabMax) p.info |= 1u << 24;
    if (total > cap_end - base) p.need = reloc_need(total, (bmark[id] & 4u) != 0u);  // it does not fit where it stands
    // what the brick gains by this update (whether it ends up in place or merged): the next layout sizes its room by it
    if (total > end - base) {
        p.gained = total - (end - base);
        grow[top_slot_of_key(g, b)] += p.gained;
    }
    plan[id] = p;
}

// One workgroup turns the touched bricks' requests into stretches of the tail (a prefix sum in id order: the same positions
// whatever order the waves of the plan finished in), their ids into a compact list for the rewrite, their counts into the
// update's totals.  Two steps: (a) every thread reads the marks of a run of ids (sixteen at a time) and the ids of the
// touched ones go, in order, to the list; (b) the threads share the LIST -- the touched bricks of a drive are the frontier's,
// opened together and so consecutive ids: walked by the owners of their id runs they were a chain of 32 memory latencies in
// the last few threads (20 us of the frame).  flags[0] |= overflow when a brick cannot be staged or the tail is exhausted
// (nothing is handed out then).
constexpr int kAllocThreads = 1024;
__device__ __forceinline__ uint32_t alloc_block_scan(uint32_t v, uint32_t *w_part, uint32_t &total)  // exclusive, over the workgroup
{
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    uint32_t in = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t a = __shfl_up(in, off, 64);
        if (lane >= off) in += a;
    }
    __syncthreads();  // (w_part of the previous use has been read)
    if (lane == 63) w_part[wave] = in;
    __syncthreads();
    uint32_t before = 0u;
    total = 0u;
    for (int w = 0; w < kAllocThreads / 64; ++w) {
        if (w < wave) before += w_part[w];
        total += w_part[w];
    }
    return before + in - v;
}
__global__ __launch_bounds__(kAllocThreads) void slab_alloc_kernel(const uint32_t *__restrict__ bricks_dev, const uint8_t *__restrict__ bmark,
                                                                   const BrickPlan *__restrict__ plan, uint32_t *__restrict__ bmove,
                                                                   uint32_t *__restrict__ blist, uint32_t *__restrict__ tail_cursor,
                                                                   uint32_t tail_end, uint32_t *__restrict__ flags, int64_t mark_cap)
{
    __shared__ uint32_t w_part[kAllocThreads / 64];
    __shared__ uint32_t s_sum[5];
    const int tid = threadIdx.x, lane = tid & 63;
    if (flags[0] & (kSlabOutside | kSlabWindow | kSlabNewBrick)) return;
    if (tid < 5) s_sum[tid] = 0u;
    const int64_t bricks = (int64_t)*bricks_dev;
    const uint32_t cursor0 = *tail_cursor;
    // a. the list
    const int64_t per = (((bricks + kAllocThreads - 1) / kAllocThreads) + 15) & ~(int64_t)15;
    const int64_t id0 = (int64_t)tid * per, id1 = min(id0 + per, bricks);
    auto marks16 = [&](int64_t base) {  // 16 marks from `base` on, as bits; ids from id1 on read as untouched
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        if (base + 16 <= mark_cap) {
            const uint4 v = *reinterpret_cast<const uint4 *>(bmark + base);
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (base + j < mark_cap) w[j >> 2] |= (uint32_t)bmark[base + j] << ((j & 3) * 8);
        }
        uint32_t bits = 0u;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (((w[j >> 2] >> ((j & 3) * 8)) & 0xffu) != 0u && base + j < id1) bits |= 1u << j;
        return bits;
    };
    uint32_t touch = 0u;
    for (int64_t base = id0; base < id1; base += 16) touch += (uint32_t)__popc(marks16(base));
    uint32_t tt = 0u;
    uint32_t li = alloc_block_scan(touch, w_part, tt);
    if (touch)
        for (int64_t base = id0; base < id1; base += 16)
            for (uint32_t bits = marks16(base); bits != 0u; bits &= bits - 1u) blist[li++] = (uint32_t)(base + (__ffs((int)bits) - 1));
    __threadfence();
    __syncthreads();
    // b. the touched bricks, one per thread and round, in list (= id) order
    uint32_t my[5] = {0u, 0u, 0u, 0u, 0u};  // this thread's share of: points removed, points gained, crowded, moved, cannot be staged
    uint64_t carry = (uint64_t)cursor0;
    for (uint32_t base = 0u; base < tt; base += (uint32_t)kAllocThreads) {
        const uint32_t i = base + (uint32_t)tid;
        uint32_t nd = 0u, id = 0u;
        if (i < tt) {
            id = blist[i];
            const uint4 p = reinterpret_cast<const uint4 *>(plan + id)[0];  // {lo, n_b, need, info}
            nd = p.z;
            my[0] += p.w & 0xffffffu;
            my[1] += plan[id].gained;
            my[2] += (p.w >> 24) & 1u;
            my[3] += p.z ? 1u : 0u;
            my[4] += (p.w >> 25) & 1u;
        }
        uint32_t round = 0u;
        const uint32_t before = alloc_block_scan(nd, w_part, round);
        if (i < tt) {
            const uint64_t at = carry + before;
            bmove[id] = nd ? (at + nd <= (uint64_t)tail_end ? (uint32_t)at : 0xfffffffeu) : 0xffffffffu;
        }
        carry += round;
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        uint32_t v = my[q];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0 && v) atomicAdd(&s_sum[q], v);
    }
    __syncthreads();
    if (tid == 0) {
        const bool over = carry > (uint64_t)tail_end;
        if (s_sum[4] || over) atomicOr(flags, kSlabOverflow | (s_sum[4] ? kSlabTooBig : 0u));
        else *tail_cursor = (uint32_t)carry;
        flags[1] = s_sum[0];
        flags[3] = s_sum[1];
        flags[4] = s_sum[2];
        flags[11] = s_sum[3];
        flags[12] = tt;
    }
}

// one workgroup per touched brick (the list of slab_alloc_kernel, dealt round-robin to a fixed grid: a launch of one workgroup
// per brick id spent its time starting workgroups that had nothing to do); does nothing when the plan found a reason not to.
// CAP = 2 048: the bricks whose staging fits that; CAP = 6 144: the crowded ones only (launched when the plan counted any)
template <int CAP>
__global__ __launch_bounds__(256) void slab_rewrite_kernel(const uint32_t *__restrict__ flags, Grid g, const uint32_t *__restrict__ blist,
                                                           const BrickPlan *__restrict__ plan,
                                                           float4 *__restrict__ pts, uint32_t *__restrict__ pidx,
                                                           uint64_t *__restrict__ keys, uint8_t *__restrict__ alive_s,
                                                           uint32_t *__restrict__ tab, uint4 *__restrict__ top,
                                                           const uint64_t *__restrict__ bkey, uint8_t *__restrict__ bmark,
                                                           uint32_t *__restrict__ bend, const uint32_t *__restrict__ bmove,
                                                           const uint64_t *__restrict__ nk, const uint32_t *__restrict__ nv,
                                                           const float4 *__restrict__ stage, uint32_t next_id)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char slab_lds[];
    float4 *l_p = reinterpret_cast<float4 *>(slab_lds);
    uint32_t *l_id = reinterpret_cast<uint32_t *>(l_p + CAP);
    uint32_t *l_t = l_id + CAP;
    uint16_t *l_c = reinterpret_cast<uint16_t *>(l_t + kBrickCells);
    __shared__ int wsum[4];
    if (flags[0] != 0u) return;
    const uint32_t n_touched = flags[12];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (uint32_t li = blockIdx.x; li < n_touched; li += gridDim.x) {
    const int64_t id = blist[li];
    const BrickPlan bp = plan[id];
    if (CAP == kSlabMax ? (bp.info & (1u << 24)) != 0u : (bp.info & (1u << 24)) == 0u) continue;  // (the other launch's brick)
    const uint32_t base = tab[id * kBrickStride], cnt = tab[id * kBrickStride + kBrickCells] - base;
    const uint64_t bk = bkey[id];
    __syncthreads();  // (the previous brick's LDS is no longer read)
    // a. the survivors, in order, into LDS
    int n_old = 0;
    for (uint32_t c0 = 0; c0 < cnt; c0 += 256u) {
        const uint32_t i = c0 + (uint32_t)tid;
        const bool a = i < cnt && alive_s[base + i] != 0;
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        uint32_t pid = 0u, cell = 0u;
        if (a) { p = pts[base + i]; pid = pidx[base + i]; cell = (uint32_t)keys[base + i] & 511u; }
        const unsigned long long bal = __ballot(a);
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int off = n_old;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        const int chunk = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (a) {
            const int r = off + __popcll(bal & ((1ull << lane) - 1ull));
            l_p[r] = p; l_id[r] = pid; l_c[r] = (uint16_t)cell;
        }
        n_old += chunk;
        __syncthreads();
    }
    // b. the brick's new points behind them (sorted by cell; equal cells in staged = id order: the sort is stable)
    const int lo = (int)bp.lo, n_b = (int)bp.n_b;
    for (int k = tid; k < n_b; k += 256) {
        const uint32_t t = nv[lo + k];
        l_p[n_old + k] = stage[t];
        l_id[n_old + k] = next_id + t;
        l_c[n_old + k] = (uint16_t)((uint32_t)nk[lo + k] & 511u);
    }
    for (int c = tid; c < kBrickCells; c += 256) l_t[c] = 0xffffffffu;
    __syncthreads();
    // c. merged places: an old point goes behind the new points of EARLIER cells, a new one behind the old points of its
    // own and earlier cells.  A brick that moves is written to its new stretch (everything it held is in LDS by now).
    const int total = n_old + n_b;
    const uint32_t moved_to = bmove[id];
    const bool moves = moved_to != 0xffffffffu;
    const uint32_t dst = moves ? moved_to : base;
    for (int e = tid; e < total; e += 256) {
        const uint32_t c = l_c[e];
        int dlo = 0, dhi = 0, dest;
        if (e < n_old) {
            dhi = n_b;  // first new k with cell >= c
            while (dlo < dhi) { const int mid = (dlo + dhi) >> 1; if (l_c[n_old + mid] < c) dlo = mid + 1; else dhi = mid; }
            dest = e + dlo;
        } else {
            dhi = n_old;  // first old r with cell > c
            while (dlo < dhi) { const int mid = (dlo + dhi) >> 1; if (l_c[mid] <= c) dlo = mid + 1; else dhi = mid; }
            dest = (e - n_old) + dlo;
        }
        const uint32_t pos = dst + (uint32_t)dest;
        const float4 p = l_p[e];
        // (a survivor holds {x, y, position, z}, a staged point {x, y, z, -})
        pts[pos] = e < n_old ? make_map_point(p.x, p.y, map_point_z(p), pos) : make_map_point(p.x, p.y, p.z, pos);
        pidx[pos] = l_id[e];
        keys[pos] = (bk << 9) | (uint64_t)c;
        alive_s[pos] = 1;
        atomicMin(&l_t[c], pos);
    }
    if (moves) {
        // the new stretch beyond the points, and the whole old one, hold nothing (the old stretch keeps the brick's largest key:
        // the key-ordered part of the array stays sorted for the next merge)
        const uint32_t need = bp.need;
        for (uint32_t h = (uint32_t)total + (uint32_t)tid; h < need; h += 256u) {
            alive_s[dst + h] = 0;
            pidx[dst + h] = 0xffffffffu;
            keys[dst + h] = (bk << 9) | 511ull;
        }
        for (uint32_t h = (uint32_t)tid; h < cnt; h += 256u) {
            alive_s[base + h] = 0;
            pidx[base + h] = 0xffffffffu;
            keys[base + h] = (bk << 9) | 511ull;
        }
        if (tid == 0) bend[id] = dst + need;
    } else {
        // what the brick no longer fills
        for (uint32_t h = (uint32_t)total + (uint32_t)tid; h < cnt; h += 256u) {
            alive_s[base + h] = 0;
            pidx[base + h] = 0xffffffffu;
            keys[base + h] = (bk << 9) | 511ull;
        }
    }
    __syncthreads();
    // d. prefix words and row mask
    if (wave == 0) {
        unsigned long long mask;
        (void)table_from_firsts(l_t, dst + (uint32_t)total, tab + id * kBrickStride, lane, mask);
        if (lane == 0) {
            uint32_t *te = reinterpret_cast<uint32_t *>(&top[top_slot_of_key(g, bk)]);
            te[2] = (uint32_t)mask;
            te[3] = (uint32_t)(mask >> 32);
            bmark[id] = 0;
        }
    }
    }
}

// The top array for a box of bricks that has outgrown (or wandered out of) its window, WITHOUT touching a point: the exact box
// of the bricks that hold points and of the update's new points comes back from the device (the bounds only ever grow
// between such events, so this is also where the bricks the field-of-view trim emptied are let go), the array keeps its
// sizes if that box fits them and is enlarged otherwise, and every brick inside the box moves its entry from its old slot to
// its new one -- a few thousand 16-byte entries (ikd-Tree has no counterpart: its nodes hold absolute coordinates).
__global__ __launch_bounds__(256) void top_relay_kernel(const uint32_t *__restrict__ bricks_dev, const uint64_t *__restrict__ bkey, Grid go,
                                                        Grid gn, uint4 *__restrict__ ntop)
{
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (int64_t)*bricks_dev) return;
    int bx, by, bz;
    brick_coords(bkey[id], bx, by, bz);
    if (!brick_in_bounds(gn, bx, by, bz) || !brick_in_bounds(go, bx, by, bz)) return;
    const uint4 e = go.top[top_slot(go, bx, by, bz)];
    if (e.x == (uint32_t)id + 1u) ntop[top_slot(gn, bx, by, bz)] = e;
}
static hipError_t relay_top(MapBuffers &buf, Grid &g, MapStats &stats, int64_t bricks, const uint64_t *nk, int n, bool &too_large, hipStream_t st)
{
    too_large = false;
    const uint32_t *bricks_dev = buf.counters + kBricksWord;
    launch_brick_box(bricks_dev, buf.bkey, buf.tab, nk, n, reinterpret_cast<int32_t *>(buf.counters + kBoxWords), st);
    const uint32_t *src[6] = {buf.counters + kBoxWords, buf.counters + kBoxWords + 1, buf.counters + kBoxWords + 2,
                              buf.counters + kBoxWords + 3, buf.counters + kBoxWords + 4, buf.counters + kBoxWords + 5};
    uint32_t v[6];
    S2M_TRY(mail_fetch(buf.mail, src, 6, v, st));
    int lo[3], hi[3];
    for (int k = 0; k < 3; ++k) { lo[k] = (int32_t)v[k]; hi[k] = (int32_t)v[3 + k]; }
    if (lo[0] > hi[0]) { lo[0] = lo[1] = lo[2] = 0; hi[0] = hi[1] = hi[2] = -1; }
    const Grid go = g;
    Grid gn = g;
    bool resized = false;
    S2M_TRY(map_window_for(gn, lo, hi, too_large, &resized));  // (sizes are chosen anew only if the box does not fit the current ones)
    if (too_large) return hipSuccess;
    // the new array is written beside the old one (the kernel reads go.top = buf.top)
    const int64_t slots = top_slots(gn);
    S2M_TRY(map_ensure((void **)&buf.top2, &buf.top2_cap, slots + 1, sizeof(uint4), 3 * slots));
    S2M_TRY(hipMemsetAsync(buf.top2, 0, (size_t)(slots + 1) * sizeof(uint4), st));
    hipLaunchKernelGGL(top_relay_kernel, dim3((unsigned)((bricks + 255) / 256)), dim3(256), 0, st, bricks_dev, buf.bkey, go, gn, buf.top2);
    std::swap(buf.top, buf.top2); std::swap(buf.top_cap, buf.top2_cap);
    if (resized) {  // (the growth history is kept per slot: it does not survive a change of the slots)
        S2M_TRY(map_ensure((void **)&buf.grow, &buf.grow_cap, slots + 1, sizeof(uint32_t), 3 * slots));
        S2M_TRY(hipMemsetAsync(buf.grow, 0, (size_t)(slots + 1) * sizeof(uint32_t), st));
    }
    gn.top = buf.top;
    g = gn;
    stats.top_entries = slots;
    ++buf.n_relaid;
    return hipGetLastError();
}

// flags: kSlabWords zeroed words of the update's counters.  done = the map was updated in place; otherwise nothing was
// touched and the caller goes on to merge_update.
// the conditions under which slab_update gets as far as its preparation, and that one workgroup can do it
static bool slab_possible(const MapBuffers &buf, const Grid &g, const MapStats &stats, int64_t n_new)
{
    const int64_t m = g.m;
    if (m <= 0 || !buf.keys_alt || g.pts != buf.pts || !buf.bmark || !buf.bmove || stats.bricks <= 0 || m > buf.scratch_cap) return false;
    if (n_new >= ((int64_t)1 << 30) || n_new > buf.scratch_cap) return false;
    if (buf.next_id + n_new >= ((int64_t)1 << 32) - 2) return false;
    return true;
}
bool slab_fuses(MapBuffers &buf, const Grid &g, const MapStats &stats, int64_t n_bound)
{
    if (!slab_possible(buf, g, stats, n_bound) || n_bound <= 0 || n_bound > kPrepMax || buf.no_fused_prep) return false;
    if (buf.prep_lds == 0) {  // (its histogram wants 96 KB of dynamic LDS: granted once per handle like the crowded-brick rewrite's)
        const hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void *>(&slab_prepare_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                  (int)kPrepLds);
        if (ae != hipSuccess) (void)hipGetLastError();
        buf.prep_lds = ae == hipSuccess ? 1 : -1;
    }
    return buf.prep_lds > 0;
}

hipError_t slab_update(MapBuffers &buf, Grid &g, MapStats &stats, uint8_t *alive_s, const float4 *stage, int64_t &n_new_io,
                       uint32_t *flags, bool &done, hipStream_t st, const uint32_t *n_dev, bool *counted, StagePending *pend,
                       float4 *stage_out, uint32_t *count_out)
{
    done = false;
    if (counted) *counted = false;
    const int64_t n_new = n_new_io;  // (a bound when n_dev is given)
    const int64_t m = g.m;
    if (!slab_possible(buf, g, stats, n_new)) return hipSuccess;
    const int n = (int)n_new;
    uint32_t *bricks_dev = buf.counters + kBricksWord;
    uint64_t *nk_sorted = buf.mk;
    uint32_t *nv_sorted = buf.mv;
    // a scan's batch is prepared by one workgroup (slab_prepare_kernel); larger ones -- and a handle made under S2M_NO_FUSED_PREP=1, for A/B and
    // tests -- by the separate kernels
    bool fused = slab_fuses(buf, g, stats, n_new);
    const bool pending = pend && pend->on;
    if (pending && !fused) return hipErrorInvalidValue;  // (the caller asks slab_fuses first and stages the batches itself otherwise)
    auto sort_keys = [&]() -> hipError_t {  // the separate kernels' keys, sorted
        const unsigned kbits = 9 + 3 * kBrickBits;
        size_t tmp = 0;
        S2M_TRY(rocprim::radix_sort_pairs(nullptr, tmp, buf.keys, buf.keys, buf.vals, buf.vals, (size_t)n_new, 0, kbits, st));
        S2M_TRY(map_ensure_sort_tmp(buf, tmp));
        hipLaunchKernelGGL(slab_key_kernel, dim3((n + 255) / 256), dim3(256), 0, st, stage, n, g, buf.keys, buf.vals, buf.bmark, flags, n_dev);
        size_t t = buf.sort_tmp_bytes;
        return rocprim::radix_sort_pairs(buf.sort_tmp, t, buf.keys, buf.mk, buf.vals, buf.mv, (size_t)n_new, 0, kbits, st);
    };
    if (n > 0) {
        S2M_TRY(map_ensure((void **)&buf.mk, &buf.mk_cap, n_new, sizeof(uint64_t), std::max<int64_t>(n_new / 2 + 4096, kPrepMax)));
        S2M_TRY(map_ensure((void **)&buf.mv, &buf.mv_cap, 2 * n_new, sizeof(uint32_t), std::max<int64_t>(n_new + 8192, 2 * kPrepMax)));
        nk_sorted = buf.mk;
        nv_sorted = buf.mv;
        if (!fused) S2M_TRY(sort_keys());
    }
    // the crowded-brick form of the rewrite needs 134 KB of dynamic LDS: granted once per handle (= per device; the
    // attribute belongs to the kernel object of the current device); a device that refuses leaves those bricks to the merge
    constexpr size_t kLdsSmall = (size_t)kSlabMax * 22 + kBrickCells * 4, kLdsBig = (size_t)kSlabBig * 22 + kBrickCells * 4;
    if (buf.big_slab == 0) {
        const hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void *>(&slab_rewrite_kernel<kSlabBig>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBig);
        if (ae != hipSuccess) (void)hipGetLastError();
        buf.big_slab = ae == hipSuccess ? 1 : -1;
    }
    const int big_ok = buf.big_slab > 0 ? 1 : 0;
    // spare rows for bricks this update opens (the tables were allocated with headroom)
    const int64_t rows = std::min(std::min(buf.tab_cap / kBrickStride, buf.bmove_cap), std::min(std::min(buf.bkey_cap, buf.bmark_cap), buf.bend_cap));
    const int max_new = (int)std::max<int64_t>(std::min<int64_t>(rows - stats.bricks, (int64_t)1 << 20), 0);
    const int64_t bricks = stats.bricks + std::min<int64_t>(max_new, n_new);  // (an upper bound: every new point opens at most one brick)
    S2M_TRY(map_ensure((void **)&buf.bplan, &buf.bplan_cap, rows, 32, 0));
    S2M_TRY(map_ensure((void **)&buf.blist, &buf.blist_cap, rows, sizeof(uint32_t), 0));
    BrickPlan *plan = reinterpret_cast<BrickPlan *>(buf.bplan);
    uint32_t *tail_cursor = buf.counters + kTailWord;
    uint32_t *head = buf.work_a, *rank = buf.work_b;
    if (n > 0) {
        const int64_t slots = top_slots(g);
        S2M_TRY(map_ensure((void **)&buf.run, &buf.run_cap, slots + 1, sizeof(uint2), 3 * slots));
        size_t ts = 0;
        S2M_TRY(rocprim::exclusive_scan(nullptr, ts, head, rank, 0u, (size_t)n, rocprim::plus<uint32_t>(), st));
        S2M_TRY(map_ensure_sort_tmp(buf, ts));
    }
    const unsigned rewrite_grid = (unsigned)std::max<int64_t>(std::min<int64_t>(bricks, 1024), 1);
    uint32_t v[kSlabWords + 2];
    for (int attempt = 0;; ++attempt) {
        if (n > 0 && fused) {
            if (pend && pend->on) {  // the scan's batches as update_add left them: staged here (once: a second attempt finds them staged)
                hipLaunchKernelGGL(slab_prepare_kernel, dim3(1), dim3(kPrepThreads), kPrepLds, st, pend->la, pend->flag, pend->na, pend->lb, pend->nb,
                                   (const uint32_t *)nullptr, pend->vkey, pend->vtab, stage_out, count_out, g, nk_sorted, nv_sorted, buf.bmark, flags,
                                   buf.run, bricks_dev, buf.top, buf.tab, buf.bend, buf.bkey, max_new);
                *pend = StagePending();
            } else {
                hipLaunchKernelGGL(slab_prepare_kernel, dim3(1), dim3(kPrepThreads), kPrepLds, st, stage, (const uint32_t *)nullptr, n, (const float4 *)nullptr, 0,
                                   n_dev, (const uint64_t *)nullptr, (unsigned long long *)nullptr, (float4 *)nullptr, (uint32_t *)nullptr, g, nk_sorted,
                                   nv_sorted, buf.bmark, flags, buf.run, bricks_dev, buf.top, buf.tab, buf.bend, buf.bkey, max_new);
            }
        } else if (n > 0) {
            hipLaunchKernelGGL(slab_head_kernel, dim3((n + 255) / 256), dim3(256), 0, st, nk_sorted, n, g, head, bricks_dev, flags, buf.run);
            size_t ts = buf.sort_tmp_bytes;
            S2M_TRY(rocprim::exclusive_scan(buf.sort_tmp, ts, head, rank, 0u, (size_t)n, rocprim::plus<uint32_t>(), st));
            hipLaunchKernelGGL(slab_open_kernel, dim3((n + 255) / 256), dim3(256), 0, st, nk_sorted, n, g, head, rank, buf.top, buf.tab, buf.bend,
                               buf.bkey, buf.bmark, bricks_dev, max_new, flags);
        }
        hipLaunchKernelGGL(slab_plan_kernel, dim3((unsigned)((bricks + 3) / 4)), dim3(256), 0, st, bricks_dev, g, buf.bend, buf.tab,
                           buf.bkey, buf.bmark, alive_s, nk_sorted, n, buf.grow, plan, flags, big_ok, buf.run);
        hipLaunchKernelGGL(slab_alloc_kernel, dim3(1), dim3(kAllocThreads), 0, st, bricks_dev, buf.bmark, plan, buf.bmove, buf.blist, tail_cursor,
                           (uint32_t)m, flags, buf.bmark_cap);
        {
            const uint32_t *src[kSlabWords + 2];
            for (int k = 0; k < kSlabWords; ++k) src[k] = flags + k;
            src[kSlabWords] = tail_cursor;
            src[kSlabWords + 1] = n_dev ? n_dev : tail_cursor;  // (the staged count rides along)
            S2M_TRY(mail_post(buf.mail, src, kSlabWords + 2, st));
        }
        hipLaunchKernelGGL((slab_rewrite_kernel<kSlabMax>), dim3(rewrite_grid), dim3(256), kLdsSmall, st, flags, g, buf.blist, plan, buf.pts, buf.pidx,
                           buf.keys_alt, alive_s, buf.tab, buf.top, buf.bkey, buf.bmark, buf.bend, buf.bmove, nk_sorted, nv_sorted, stage,
                           (uint32_t)buf.next_id);
        for (int k = 0; k <= kSlabWords + 1; ++k) v[k] = 0u;
        S2M_TRY(mail_collect(buf.mail, kSlabWords + 2, v, st));
        if (n_dev) {
            n_new_io = std::min<int64_t>((int64_t)v[kSlabWords + 1], n_new);
            if (counted) *counted = true;
        }
        if (fused && (v[0] & kSlabRefuse)) {  // (a batch strewn over too many bricks, or a crowded brick: nothing that counts was written, the separate kernels do it)
            fused = false;
            --attempt;
            S2M_TRY(hipMemsetAsync(flags, 0, kSlabWords * sizeof(uint32_t), st));
            S2M_TRY(sort_keys());
            continue;
        }
        if (v[0] != kSlabWindow || attempt > 0) break;
        // the box of the bricks in use has left the window: re-lay the top array (nothing else was written) and go again
        bool too_large = false;
        S2M_TRY(relay_top(buf, g, stats, bricks, nk_sorted, n, too_large, st));
        if (too_large) return hipSuccess;  // (the merge reports it)
        if (n > 0) S2M_TRY(map_ensure((void **)&buf.run, &buf.run_cap, top_slots(g) + 1, sizeof(uint2), 3 * top_slots(g)));
        S2M_TRY(hipMemsetAsync(flags, 0, kSlabWords * sizeof(uint32_t), st));
    }
    // what the tail has handed out so far -- also by an attempt that was given up: the merge brings home whatever lies below
    if (buf.main_ext > 0 && (int64_t)v[kSlabWords] >= buf.main_ext) buf.tail_used = std::min<int64_t>((int64_t)v[kSlabWords], m) - buf.main_ext;
    if (v[0] == 0u && v[4] != 0u) {  // crowded bricks among the touched ones: the form with the large staging area
        hipLaunchKernelGGL((slab_rewrite_kernel<kSlabBig>), dim3(std::min(rewrite_grid, 256u)), dim3(256), kLdsBig, st, flags, g, buf.blist, plan, buf.pts,
                           buf.pidx, buf.keys_alt, alive_s, buf.tab, buf.top, buf.bkey, buf.bmark, buf.bend, buf.bmove, nk_sorted, nv_sorted, stage,
                           (uint32_t)buf.next_id);
        buf.n_big_slab += v[4];  // (diagnostic: bricks that went through the large form)
    }
    buf.added_since_layout += v[3];  // (counted by the plan kernel whether the update stays in place or not)
    if (v[0] != 0u) {  // the rewrite kernel saw the same word and left the points alone
        if (v[0] & kSlabOutside) ++buf.slab_fail[0];
        else if (v[0] & kSlabNewBrick) ++buf.slab_fail[1];
        else if (v[0] & kSlabTooBig) ++buf.slab_fail[2];
        else if (v[0] & kSlabOverflow) ++buf.slab_fail[3];
        return hipSuccess;
    }
    // the bounds follow the bricks this update opened
    for (int k = 0; k < 3; ++k) { g.blo[k] -= (int)v[5 + k]; g.bhi[k] += (int)v[8 + k]; }
    stats.bricks += v[2];
    buf.n_moved += v[11];
    g.live += n_new_io - (int64_t)v[1];
    buf.next_id += n_new_io;
    if (v[1] > 0u) buf.ids_dense = false;
    done = true;
    return hipGetLastError();
}

}  // namespace s2m
