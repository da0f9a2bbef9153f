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
//   top   : dense nbx*nby*nbz array of 16-byte entries over the map bounding box:
//           {brick id + 1 (0 = empty), position of the brick's first point + 1, 64-bit mask of the (y,z)
//           rows of the brick that hold points}.  A brick is 8x8x8 cells; at c = 0.5 m this array is 7 K entries (0.1 MB) for
//           a 215 m scene, so it stays L2-resident; the row mask lets the search skip the table
//           lookups of empty rows.
//   tab   : per occupied brick a 520-entry row (513 used): exclusive prefix of the point counts of
//           its 512 cells (x fastest) as absolute indices into pts; tab[cell] .. tab[cell+1] is
//           the cell, tab[row*8 + x0] .. tab[row*8 + x1 + 1] a run of cells along x.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace s2m {

constexpr int kK = 5;
constexpr int kBrick = 8;
constexpr int kBrickCells = 512;
constexpr int kBrickStride = 520;

enum : uint8_t { kFlagGate = 1, kFlagPlane = 2 };

struct Grid {
    float ox, oy, oz;   // world coordinate of cell (0,0,0)'s lower corner
    float c, inv_c;     // cell edge and its reciprocal
    float slop;         // safety margin of the termination bound, in cells
    int ncx, ncy, ncz;  // cells per axis (multiples of 8)
    int nbx, nby, nbz;  // bricks per axis
    const uint4 *top;      // {id + 1, first point + 1 (used by the table builder only), rowmask lo, rowmask hi}
    const uint32_t *tab;
    const float4 *pts;
    const uint32_t *pidx;
    int64_t m;          // extent of pts (the sentinel block starts here)
    int64_t live;       // points in the map: m after a build or a merge; in-place updates leave holes at the ends of bricks
    uint32_t sent_off;  // byte offset of the sentinel block pts[m..m+32) (valid while it fits 32 bits), else 0
};

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
