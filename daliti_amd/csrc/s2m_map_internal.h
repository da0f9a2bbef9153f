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

// Six running maxima kept in device words (zero-initialised by the caller), raised by a whole wave at a time: called by every
// lane of the wave with its own six values.  The usual wave holds nothing but zeros and leaves after one ballot; the others
// reduce over the wave first, and only the words the wave would raise see an atomic (same-line atomics cost ~11 ns each).
__device__ __forceinline__ void wave_max6_to(uint32_t (&e)[6], uint32_t *__restrict__ dst6)
{
    const uint32_t any = e[0] | e[1] | e[2] | e[3] | e[4] | e[5];
    if (__ballot(any != 0u) == 0ull) return;  // wave-uniform
#pragma unroll
    for (int k = 0; k < 6; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) e[k] = max(e[k], (uint32_t)__shfl_xor((int)e[k], off, 64));
        if ((threadIdx.x & 63) == 0 && e[k] > __hip_atomic_load(dst6 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dst6 + k, e[k]);
    }
}

// host helpers of s2m_map.hip
void launch_set_word(uint32_t *dst, uint32_t value, hipStream_t st);  // *dst = value, in stream order
hipError_t map_ensure(void **p, int64_t *cap, int64_t need, size_t elem, int64_t headroom = 0);
hipError_t map_ensure_sort_tmp(MapBuffers &buf, size_t bytes);
// the per-point scratch arrays grown to hold `need` points, the first keys_in_use sorted keys (keys_alt) carried over
hipError_t map_grow_scratch(MapBuffers &buf, int64_t need, int64_t keys_in_use, hipStream_t st);
int64_t map_headroom_for(int64_t m);
hipError_t map_put_sentinels(float4 *pts, int64_t m, hipStream_t st);
// top entries + brick tables of the m points whose sorted keys are `keys` (s2m_map.hip)
hipError_t map_build_tables(MapBuffers &buf, Grid &g, const uint64_t *keys, int64_t m, MapStats &stats, hipStream_t st,
                            int64_t brick_bound, bool find_bounds, bool *too_large);
// the grid's bounds become the box of bricks [lo, hi]; new sizes of the top array are chosen when the box does not fit the
// current ones (map_window_for: the fields of g only; map_set_window: and the map's own array is (re)allocated, not filled)
hipError_t map_window_for(Grid &g, const int lo[3], const int hi[3], bool &too_large, bool *resized);
hipError_t map_set_window(MapBuffers &buf, Grid &g, const int lo[3], const int hi[3], bool &too_large, bool *resized, hipStream_t st);
// box of the bricks that hold points (tab != nullptr: only those whose table says so) and of the brick parts of n sorted keys
void launch_brick_box(const uint32_t *bricks_dev, const uint64_t *bkey, const uint32_t *tab, const uint64_t *nk, int n, int32_t *out6,
                      hipStream_t st);

}  // namespace s2m
