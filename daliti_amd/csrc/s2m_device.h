// s2m_device.h -- data layout shared by the HIP kernels of the scan-to-map engine (gfx950).
//
// Map layout in HBM ("brick grid", replaces the ikd-Tree of eskf_lio/include/ikd-Tree/):
//   pts   : M x float4 {x, y, bitcast(sorted position), z} (make_map_point), sorted by (brick, cell-in-brick,
//           caller index), followed by 32 sentinel points (padding targets of the search kernels' batches);
//           one 16-byte load per candidate, a cell's points are contiguous, the cells of one
//           x-row of a brick are contiguous.  A neighbour is identified by its SORTED POSITION on the whole
//           per-iteration path: the plane fit gathers its five points from this array (the five neighbours of a
//           query sit in the same or adjacent cells, i.e. in a few cache lines), so nothing in caller order is
//           touched between two map updates.
//   pidx  : M x uint32, the point id of every sorted position: the index into the array handed to s2m_map_build,
//           next_id, next_id + 1, ... for points added later -- ascending in caller order (survivors, then the added
//           points), never renumbered; ~0 = a position that holds no point.  Read only when somebody asks for caller
//           indices or caller order (s2m_get_neighbors, s2m_map_get_points: the ids are ranked then) and by the map
//           update; 4 B/point instead of the 16 B/point copy in caller order that round 2 kept.
//   top   : 16-byte entries {brick id + 1 (0 = empty), -, 64-bit mask of the (y,z) rows of the brick that hold points},
//           addressed TOROIDALLY by the brick's integer coordinates: slot = ((bz & mz) << sz) | ((by & my) << sy) | (bx & mx)
//           with power-of-two sizes per axis that exceed the extent of the bricks in use (Grid::blo..bhi, a conservative
//           box: a lookup is valid only inside it, and inside it no two bricks share a slot).  A brick is 8x8x8 cells and
//           its coordinates are (cell >> 3) relative to an origin that is fixed when the map is first built: the map can
//           grow in any direction -- a moving sensor opens bricks at the front while the field-of-view trim empties them
//           at the back -- without a single entry moving; only when the extent outgrows the window is the array re-laid
//           (a few thousand entries).  ikd-Tree inserts anywhere too (ikd_Tree.cpp:477-573).
//   tab   : per occupied brick a 520-entry row (513 used): exclusive prefix of the point counts of
//           its 512 cells (x fastest) as absolute indices into pts; tab[cell] .. tab[cell+1] is
//           the cell, tab[row*8 + x0] .. tab[row*8 + x1 + 1] a run of cells along x.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace s2m {

constexpr int kK = 5;
constexpr int kBrick = 8;
constexpr int kBrickCells = 512;
constexpr int kBrickStride = 520;

enum : uint8_t { kFlagGate = 1, kFlagPlane = 2 };

// cell / brick coordinates are signed and box-independent; a brick coordinate takes kBrickBits bits of a point's sort key
constexpr int kBrickBits = 18;
constexpr int kBrickBias = 1 << (kBrickBits - 1);
constexpr int kCellLimit = (1 << (kBrickBits + 2)) - 16;  // |cell coordinate| below this is representable

struct Grid {
    float ox, oy, oz;   // world coordinate of cell (0,0,0)'s lower corner: fixed by the first build, the map grows around it
    float c, inv_c;     // cell edge and its reciprocal
    float slop;         // safety margin of the termination bound, in cells
    int blo[3], bhi[3]; // bricks that may hold points (inclusive, conservative); empty map: bhi < blo
    uint32_t tmx, tmy, tmz;  // the top array's sizes per axis minus one (powers of two) ...
    uint32_t tsy, tsz;       // ... and the shifts of the y and z parts of a slot
    const uint4 *top;      // {id + 1, -, rowmask lo, rowmask hi}
    const uint32_t *tab;
    const float4 *pts;
    const uint32_t *pidx;
    int64_t m;          // extent of pts (the sentinel block starts here)
    int64_t live;       // points in the map: m after a build or a merge; in-place updates leave holes at the ends of bricks
    uint32_t sent_off;  // byte offset of the sentinel block pts[m..m+32) (valid while it fits 32 bits), else 0
};

// The cell of a world coordinate: floor((v - origin) * (1 / cell)) evaluated in DOUBLE from the float operands (the float
// reciprocal included), so that the binning of a point does not depend on how far the map has grown from its origin (in
// float the rounding error of v - origin grows with the distance and the search bounds would have to give way to it).
// Monotone in v.  Clamped far outside the representable range so that the conversion is defined (NaN -> the lower clamp).
__host__ __device__ __forceinline__ double cell_pos(float v, float o, float inv_c) { return ((double)v - (double)o) * (double)inv_c; }
__host__ __device__ __forceinline__ int cell_coord(float v, float o, float inv_c)
{
    const double f = floor(cell_pos(v, o, inv_c));
    return (int)fmin(fmax(f, -1.0e9), 1.0e9);
}
__host__ __device__ __forceinline__ bool cell_representable(int cx, int cy, int cz)
{
    return cx > -kCellLimit && cx < kCellLimit && cy > -kCellLimit && cy < kCellLimit && cz > -kCellLimit && cz < kCellLimit;
}
// sort key of a point: (brick z, brick y, brick x | 18 bits each, biased) << 9 | cell inside the brick (z, y, x | 3 bits
// each): lexicographic in the signed coordinates, whatever box the map occupies
__host__ __device__ __forceinline__ uint64_t brick_key(int bx, int by, int bz)
{
    return ((uint64_t)(uint32_t)(bz + kBrickBias) << (2 * kBrickBits)) | ((uint64_t)(uint32_t)(by + kBrickBias) << kBrickBits) |
           (uint64_t)(uint32_t)(bx + kBrickBias);
}
__host__ __device__ __forceinline__ uint64_t point_key(int cx, int cy, int cz)
{
    const uint32_t local = (uint32_t)((((cz & 7) << 3) | (cy & 7)) << 3 | (cx & 7));
    return (brick_key(cx >> 3, cy >> 3, cz >> 3) << 9) | local;
}
__host__ __device__ __forceinline__ void brick_coords(uint64_t bk, int &bx, int &by, int &bz)
{
    const uint32_t mask = (1u << kBrickBits) - 1u;
    bx = (int)((uint32_t)bk & mask) - kBrickBias;
    by = (int)((uint32_t)(bk >> kBrickBits) & mask) - kBrickBias;
    bz = (int)((uint32_t)(bk >> (2 * kBrickBits)) & mask) - kBrickBias;
}
__host__ __device__ __forceinline__ bool brick_in_bounds(const Grid &g, int bx, int by, int bz)
{
    return bx >= g.blo[0] && bx <= g.bhi[0] && by >= g.blo[1] && by <= g.bhi[1] && bz >= g.blo[2] && bz <= g.bhi[2];
}
__host__ __device__ __forceinline__ uint32_t top_row(const Grid &g, int by, int bz)
{
    return (((uint32_t)bz & g.tmz) << g.tsz) | (((uint32_t)by & g.tmy) << g.tsy);
}
__host__ __device__ __forceinline__ uint32_t top_slot(const Grid &g, int bx, int by, int bz)
{
    return top_row(g, by, bz) | ((uint32_t)bx & g.tmx);
}
__host__ __device__ __forceinline__ uint32_t top_slot_of_key(const Grid &g, uint64_t bk)
{
    int bx, by, bz;
    brick_coords(bk, bx, by, bz);
    return top_slot(g, bx, by, bz);
}
__host__ __device__ __forceinline__ int64_t top_slots(const Grid &g) { return ((int64_t)g.tmz + 1) << g.tsz; }

// Element of the sorted point array Grid::pts: {x, y, bitcast(sorted position), z}.  The position sits in the
// third word so that a search key (position, d2) is formed in place: the 64-bit key needs an even-aligned
// register pair, the position already is in one, and d2 is written over z once dz has been taken -- no
// register moves per candidate (three with the natural {x, y, z, w} order, one more if the position had to be
// derived from the load address).  Candidates tied at exactly the same float d2 are therefore ranked by sorted
// position = (brick, cell, caller index): a total order both this engine and the oracle can compute.
__host__ __device__ inline float4 make_map_point(float x, float y, float z, uint32_t pos)
{
    float4 p;
    p.x = x; p.y = y; p.w = z;
#if defined(__HIP_DEVICE_COMPILE__)
    p.z = __uint_as_float(pos);
#else
    __builtin_memcpy(&p.z, &pos, 4);
#endif
    return p;
}
__device__ __forceinline__ float map_point_z(const float4 &p) { return p.w; }
__device__ __forceinline__ uint32_t map_point_pos(const float4 &p) { return __float_as_uint(p.z); }

// rot_end, pos_end, R_L_I, T_L_I of StatesGroup (eskf_lio/include/common_lib.h:219-222)
struct Pose {
    double R[9], t[3], RLI[9], TLI[3];
};

struct Gates {
    float plane_thr;
    float knn_d2_gate;
    double s_gate;
    double res_gate;
    int extrinsic;
};

// p_w = rot_end * (R_L_I * p_b + T_L_I) + pos_end in double, rounded to float
// (eskf_lio/src/laserMapping.cpp:835-841).  Sums left to right, no contraction.
__host__ __device__ __forceinline__ void body_to_world(const Pose &P, float bx, float by, float bz, float &wx,
                                              float &wy, float &wz)
{
    const double x = (double)bx, y = (double)by, z = (double)bz;
    const double ix = ((P.RLI[0] * x + P.RLI[1] * y) + P.RLI[2] * z) + P.TLI[0];
    const double iy = ((P.RLI[3] * x + P.RLI[4] * y) + P.RLI[5] * z) + P.TLI[1];
    const double iz = ((P.RLI[6] * x + P.RLI[7] * y) + P.RLI[8] * z) + P.TLI[2];
    wx = (float)(((P.R[0] * ix + P.R[1] * iy) + P.R[2] * iz) + P.t[0]);
    wy = (float)(((P.R[3] * ix + P.R[4] * iy) + P.R[5] * iz) + P.t[1]);
    wz = (float)(((P.R[6] * ix + P.R[7] * iy) + P.R[8] * iz) + P.t[2]);
}

}  // namespace s2m
