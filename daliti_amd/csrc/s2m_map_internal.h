// s2m_map_internal.h -- what s2m_map.hip (build, tables) and s2m_mapedit.hip (merge, room, in-place update) share.
#pragma once
#include <hip/hip_runtime.h>

#include "s2m_device.h"
#include "s2m_kernels.h"

namespace s2m {

#define S2M_TRY(x)                      \
    do {                                \
        hipError_t e_ = (x);            \
        if (e_ != hipSuccess) return e_; \
    } while (0)

// the cell of a coordinate, clamped to the grid (every point of the map was binned with this expression)
__device__ __forceinline__ int cell_of(float v, float o, float inv_c, int nc)
{
    const int c = (int)floorf((v - o) * inv_c);
    return min(max(c, 0), nc - 1);
}

// One wave: t[512] holds the first position of every non-empty cell of a brick (0xffffffff = empty), e the end of the
// brick's points.  Writes the brick's 513 prefix words (an empty cell takes the start of the next non-empty one) and returns
// this lane's number of non-empty cells; mask = the brick's occupied (z,y) rows (row = lane).
__device__ __forceinline__ int table_from_firsts(const uint32_t *t, uint32_t e, uint32_t *__restrict__ out, int lane, unsigned long long &mask)
{
    int cells = 0;
    uint32_t v[8];
    uint32_t mn = 0xffffffffu;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        v[k] = t[lane * 8 + k];
        mn = min(mn, v[k]);
        cells += v[k] != 0xffffffffu ? 1 : 0;
    }
    // row lane = (z,y) row of the brick: occupied when any of its eight cells is
    mask = __ballot(mn != 0xffffffffu);
    // suffix-min over the lanes behind this one, seeded with the brick end
    uint32_t suf = mn;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_down(suf, off, 64);
        if (lane + off < 64) suf = min(suf, o);
    }
    uint32_t nxt = __shfl_down(suf, 1, 64);
    if (lane == 63) nxt = e;
    nxt = min(nxt, e);
#pragma unroll
    for (int k = 7; k >= 0; --k) {
        if (v[k] == 0xffffffffu) v[k] = nxt;
        nxt = v[k];
        out[lane * 8 + k] = v[k];
    }
    if (lane == 0) out[kBrickCells] = e;
    return cells;
}

// host helpers of s2m_map.hip
hipError_t map_ensure(void **p, int64_t *cap, int64_t need, size_t elem, int64_t headroom = 0);
hipError_t map_ensure_sort_tmp(MapBuffers &buf, size_t bytes);
int64_t map_headroom_for(int64_t m);
hipError_t map_put_sentinels(float4 *pts, int64_t m, hipStream_t st);
// top entries + brick tables of the m points whose sorted keys are `keys` (buf.top zeroed by the caller)
hipError_t map_build_tables(MapBuffers &buf, const uint64_t *keys, int64_t m, int64_t top_entries, MapStats &stats, hipStream_t st,
                            int64_t brick_bound = -1);

}  // namespace s2m
