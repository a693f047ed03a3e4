// hashgrid.h -- key packing and hashing of the voxel hash grid (shared by hashmap.hip and knn.hip)
#pragma once
#include "lio_common.h"

namespace lio {

__device__ __host__ inline unsigned long long pack_key(int x, int y, int z) {
    return ((unsigned long long)((uint32_t)x & 0x1FFFFFu)) | ((unsigned long long)((uint32_t)y & 0x1FFFFFu) << 21) |
           ((unsigned long long)((uint32_t)z & 0x1FFFFFu) << 42);
}

// "brick-coherent" hashing with double hashing BY WINDOW.  The 4x4x4 brick a voxel lies in picks a 64-slot
// (1 KiB) window of the table, the voxel's position inside the brick picks the slot in the window, so the probes
// of one stencil land in a handful of 128-B lines.  On a collision (the slot holds another brick's voxel) the
// probe sequence moves to ANOTHER WINDOW (w + k * step, step odd => full cycle) and keeps the in-brick offset:
// plain linear probing inside a window would walk the occupied runs that brick coherence creates (chains of
// tens of dependent loads, each a DRAM round trip, for the empty-cell probes of every stencil).
// Brick coordinates fit 19 bits, so the mixing uses the full-rate 24-bit multiplier (v_mul_u32_u24).
struct BrickProbe {
    uint32_t win, step, local;
};

__device__ inline BrickProbe brick_probe(int x, int y, int z) {
    const uint32_t bx = (uint32_t)(x >> 2) & 0x7FFFFu, by = (uint32_t)(y >> 2) & 0x7FFFFu, bz = (uint32_t)(z >> 2) & 0x7FFFFu;
    uint32_t h = __umul24(bx, 0x9E3779u) ^ (__umul24(by, 0x85EBCBu) + 0x7F4A7C15u) ^ __umul24(bz, 0xC2B2AFu);
    h ^= h >> 15;
    h = __umul24(h & 0xFFFFFFu, 0x2C1B3Du) ^ (h >> 9);
    h ^= h >> 13;
    BrickProbe p;
    p.win = h;
    p.step = (__umul24((h >> 7) & 0xFFFFFFu, 0x5BD1E9u) ^ (h << 11)) | 1u;
    // in-window position = in-brick offset XOR six hash bits of the brick: still one distinct slot per voxel of the brick, but
    // bricks no longer agree on which of the 64 positions a given offset uses -- a map that is mostly one ground plane (one
    // value of z & 3) would otherwise crowd a quarter of the positions and overflow them at a quarter of the nominal load
    p.local = ((uint32_t)(x & 3) | ((uint32_t)(y & 3) << 2) | ((uint32_t)(z & 3) << 4)) ^ ((h >> 17) & 63u);
    return p;
}
__device__ inline uint32_t brick_slot(const BrickProbe& p, uint32_t mask) { return ((p.win << 6) | p.local) & mask; }
__device__ inline void brick_next(BrickProbe& p) { p.win += p.step; }

// ivox3d.h:258-261: Pos2Grid = round(p * inv_res) per axis (std::round: half away from zero), in f32
__device__ inline void pos2grid(float x, float y, float z, float inv_res, int& kx, int& ky, int& kz) {
    kx = (int)roundf(x * inv_res);
    ky = (int)roundf(y * inv_res);
    kz = (int)roundf(z * inv_res);
}

// fast_gicp's Gaussian-voxel key (vector3_hash.cuh:35-38): (x.array() / resolution - 0.5).floor() in f32 -- cells
// centred on integer multiples of the resolution, different from both iVox (round) and map_incremental (floor)
__device__ inline void pos2grid_ndt(float x, float y, float z, float res, int& kx, int& ky, int& kz) {
    kx = (int)floorf(x / res - 0.5f);
    ky = (int)floorf(y / res - 0.5f);
    kz = (int)floorf(z / res - 0.5f);
}

// fast_gicp's CPU Gaussian-voxel key (fast_vgicp_voxel.hpp:165-167): (x.array() / voxel_resolution - 0.5).floor() with x and the resolution
// in DOUBLE (the point cast from f32)
__device__ inline void pos2grid_vgicp(float x, float y, float z, double res, int& kx, int& ky, int& kz) {
    kx = (int)floor((double)x / res - 0.5);
    ky = (int)floor((double)y / res - 0.5);
    kz = (int)floor((double)z / res - 0.5);
}

// ---- group search: sixteen lanes (one DPP row) per query --------------------------------------------------------------------------
// A cloud of a few ten thousand points is 100-200 workgroups with one lane per query: two waves per CU walking hundreds of dependent hash
// probes each, and a wave waits for its sparsest query (ring r costs 24 r^2 + 2 probes).  With sixteen lanes per query the probes of a
// shell are made sixteen at a time, the points of a hit cell are split over the lanes, and the launch has sixteen times the waves.
constexpr int kGrp = 16;
constexpr int kGrpThreads = 256;  // 16 queries per workgroup

__device__ inline uint32_t grp_ballot(bool p) {
    const unsigned long long b = __ballot(p);
    const int lane = threadIdx.x & 63;
    const uint32_t half = (lane & 32) ? (uint32_t)(b >> 32) : (uint32_t)b;
    return (half >> (lane & 16)) & 0xFFFFu;
}

// cell t of the shell of the (2r + 1)^3 cube around the query's cell: the two z faces, then the perimeter of every layer between them
__device__ inline int shell_cells(int r) { return r == 0 ? 1 : 2 * (2 * r + 1) * (2 * r + 1) + (2 * r - 1) * 8 * r; }
__device__ inline void shell_cell(int r, int t, int& dx, int& dy, int& dz) {
    if (r == 0) { dx = dy = dz = 0; return; }
    const int s = 2 * r + 1, face = s * s;
    if (t < 2 * face) {
        const int f = t >= face, rem = t - f * face, row = rem / s;
        dz = f ? r : -r;
        dy = row - r;
        dx = rem - row * s - r;
        return;
    }
    const int u = t - 2 * face, layer = u / (8 * r), q = u - layer * 8 * r;
    dz = layer - r + 1;
    if (q < s) { dy = -r; dx = q - r; }
    else if (q < 2 * s) { dy = r; dx = q - s - r; }
    else {
        const int w = q - 2 * s, side = w >= s - 2;
        dy = w - side * (s - 2) - r + 1;
        dx = side ? r : -r;
    }
}

// candidate key: f32 bits of the squared distance (non-negative: bit order = numeric order) above the pool index -- ties of exact distance
// go to the lower index, whatever the order of enumeration
__device__ inline unsigned long long cand_key(float d2, uint32_t idx) { return ((unsigned long long)__float_as_uint(d2) << 32) | idx; }


// distance from x to the nearer face of its own cell along one axis (key floor(x / res - 0.5)), shrunk by the rounding of the key arithmetic:
// after ring r of a shell search every unseen point is at least r * res + (the smallest gap of the three axes) away
__device__ inline float cell_gap(float x, float res, int k) {
    const float f = x / res - 0.5f - (float)k;
    const float g = fminf(f, 1.0f - f) - 1e-6f * (1.0f + fabsf(x / res));
    return g > 0.f ? g * res : 0.f;
}
__device__ inline float cell_gap3(float x, float y, float z, float res, int kx, int ky, int kz) {
    return fminf(fminf(cell_gap(x, res, kx), cell_gap(y, res, ky)), cell_gap(z, res, kz));
}

// exact-key lookup in a hash grid: (ptr, cnt) of the cell's points
__device__ inline bool grid_find(const Slot* __restrict__ table, uint32_t mask, int cx, int cy, int cz, uint32_t& ptr, uint32_t& cnt) {
    const unsigned long long want = pack_key(cx, cy, cz);
    BrickProbe bp = brick_probe(cx, cy, cz);
    for (uint32_t probe = 0; probe <= (mask >> 6); probe++) {
        const Slot sl = table[brick_slot(bp, mask)];
        if (sl.key == want) { ptr = sl.ptr; cnt = sl.cnt; return cnt > 0; }
        if (sl.key == kEmptyKey) return false;
        brick_next(bp);
    }
    return false;
}

}  // namespace lio
