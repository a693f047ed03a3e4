// voxelgrid.hip -- per-scan voxel-grid downsample on gfx950 (wave64).
//
// Behaviour restated from pcl::VoxelGrid<PointT>::applyFilter (PCL 1.9.1 voxel_grid.hpp; third-party,
// not in the reference tree) as the reference uses it: downSizeFilterSurf.filter() at
// /root/reference/slam/mapping/fastlio/src/laserMapping.cpp:1206-1207, leaf set at :1073.
//   bbox of the finite points -> ijk = floor(p * inv_leaf) - min_b -> linear voxel index ->
//   points ordered by index -> one centroid (all four fields, f32 running sums) per occupied voxel,
//   output in ascending index; if the voxel count of the bbox overflows int32 the input is returned.
// The in-voxel summation order is unspecified in PCL (unstable std::sort); here it is ascending input
// index, which a stable LSD radix sort of (voxel index, point index) provides.
//
// Pipeline (all sizes stay on the device; nothing is read back between stages):
//   bbox      wave shuffle + LDS reduce, one set of ordered-uint atomics per workgroup
//   keys      voxel index per point (every lane re-derives the grid from the bbox: no setup launch)
//   4 x {hist, scatter}  stable 8-bit LSD radix sort of (key, point index); the scatter workgroups scan the
//             [digit][tile] histogram themselves (no scan launch); passes above the significant key bits
//             exit at once and the consumers pick the ping-pong buffer from the pass count
//   count_heads / heads  voxel occupancy flags -> wave ballot + prefix-sum compaction (fixed order), fused
//             with the gather of the points into sorted order
//   centroid  one lane per voxel, sequential f32 sums; runs of >= 32 points are queued and summed one wave per
//             run (coalesced loads, v_readlane broadcast) in the same sequential order
#include "lio_common.h"

namespace lio {

constexpr int kThreads = 256;
constexpr int kItems = 8;
constexpr int kTile = kThreads * kItems;  // 2048 keys per workgroup
constexpr int kWaves = kThreads / 64;

__device__ inline uint32_t f2ord(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u ^ 0x80000000u) : ~u); }

// the batched chain's form (64 scans per launch: the grid is full either way): 48 workgroups per scan striding over the cloud, one set of
// ordered-uint atomics per workgroup into the scan's device record, which vg_keys reads.  (The per-tile records below cost the batched vg_keys
// 20 -> 35 us per round: 3 776 workgroups folding 59 records each; the atomics' serialisation is hidden there by the other scans' work.)
__device__ __forceinline__ void vg_bbox_atomic_body(const float4* __restrict__ in, uint32_t n, ScanDev* sd) {
    float mn0 = INFINITY, mn1 = INFINITY, mn2 = INFINITY, mx0 = -INFINITY, mx1 = -INFINITY, mx2 = -INFINITY;
    uint32_t cnt = 0;
    // four points of the stride loop per step, requested together (clamped, unconditional: one load per step was one memory round trip per step)
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += 4u * stride) {
        float4 q[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t j = i + (uint32_t)k * stride;
            q[k] = in[j < n ? j : i];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float4 p = q[k];
            if (i + (uint32_t)k * stride < n && isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
                mn0 = fminf(mn0, p.x); mx0 = fmaxf(mx0, p.x);
                mn1 = fminf(mn1, p.y); mx1 = fmaxf(mx1, p.y);
                mn2 = fminf(mn2, p.z); mx2 = fmaxf(mx2, p.z);
                cnt++;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mn0 = fminf(mn0, __shfl_xor(mn0, off)); mx0 = fmaxf(mx0, __shfl_xor(mx0, off));
        mn1 = fminf(mn1, __shfl_xor(mn1, off)); mx1 = fmaxf(mx1, __shfl_xor(mx1, off));
        mn2 = fminf(mn2, __shfl_xor(mn2, off)); mx2 = fmaxf(mx2, __shfl_xor(mx2, off));
        cnt += __shfl_xor(cnt, off);
    }
    // one set of atomics per workgroup: the seven words share one L2 line, every atomic on it serialises
    __shared__ float red[kWaves][6];
    __shared__ uint32_t redc[kWaves];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        red[wave][0] = mn0; red[wave][1] = mn1; red[wave][2] = mn2;
        red[wave][3] = mx0; red[wave][4] = mx1; red[wave][5] = mx2;
        redc[wave] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kWaves; w++) {
            mn0 = fminf(mn0, red[w][0]); mn1 = fminf(mn1, red[w][1]); mn2 = fminf(mn2, red[w][2]);
            mx0 = fmaxf(mx0, red[w][3]); mx1 = fmaxf(mx1, red[w][4]); mx2 = fmaxf(mx2, red[w][5]);
            cnt += redc[w];
        }
        if (cnt > 0) {
            atomicMin(&sd->bbox_min[0], f2ord(mn0)); atomicMax(&sd->bbox_max[0], f2ord(mx0));
            atomicMin(&sd->bbox_min[1], f2ord(mn1)); atomicMax(&sd->bbox_max[1], f2ord(mx1));
            atomicMin(&sd->bbox_min[2], f2ord(mn2)); atomicMax(&sd->bbox_max[2], f2ord(mx2));
            atomicAdd(&sd->n_valid, cnt);
        }
    }
}

// bounding box of the finite points, stage 1: one workgroup per sort tile (eight independent 16-byte loads per thread in flight), its box and
// count stored as ONE 32-byte record {min x, y, z, max x, y, z (order-preserving uint codes), finite points, -}.  vg_keys folds the <= 128 records
// itself.  (Until round 4: 48 workgroups striding over the cloud with seven same-line atomics each -- 9 us for 120 000 points, all of it the
// dependent loads of the stride loop and the atomics' serialisation in one L2 channel.)
__device__ __forceinline__ void vg_bbox_body(const float4* __restrict__ in, uint32_t n, uint32_t* __restrict__ parts) {
    float mn0 = INFINITY, mn1 = INFINITY, mn2 = INFINITY, mx0 = -INFINITY, mx1 = -INFINITY, mx2 = -INFINITY;
    uint32_t cnt = 0;
    const uint32_t base = blockIdx.x * kTile;
    float4 p[kItems];
#pragma unroll
    for (int r = 0; r < kItems; r++) {
        const uint32_t i = base + r * kThreads + threadIdx.x;
        if (i < n) p[r] = in[i];
    }
#pragma unroll
    for (int r = 0; r < kItems; r++) {
        const uint32_t i = base + r * kThreads + threadIdx.x;
        if (i < n && isfinite(p[r].x) && isfinite(p[r].y) && isfinite(p[r].z)) {
            mn0 = fminf(mn0, p[r].x); mx0 = fmaxf(mx0, p[r].x);
            mn1 = fminf(mn1, p[r].y); mx1 = fmaxf(mx1, p[r].y);
            mn2 = fminf(mn2, p[r].z); mx2 = fmaxf(mx2, p[r].z);
            cnt++;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mn0 = fminf(mn0, __shfl_xor(mn0, off)); mx0 = fmaxf(mx0, __shfl_xor(mx0, off));
        mn1 = fminf(mn1, __shfl_xor(mn1, off)); mx1 = fmaxf(mx1, __shfl_xor(mx1, off));
        mn2 = fminf(mn2, __shfl_xor(mn2, off)); mx2 = fmaxf(mx2, __shfl_xor(mx2, off));
        cnt += __shfl_xor(cnt, off);
    }
    __shared__ float red[kWaves][6];
    __shared__ uint32_t redc[kWaves];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        red[wave][0] = mn0; red[wave][1] = mn1; red[wave][2] = mn2;
        red[wave][3] = mx0; red[wave][4] = mx1; red[wave][5] = mx2;
        redc[wave] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kWaves; w++) {
            mn0 = fminf(mn0, red[w][0]); mn1 = fminf(mn1, red[w][1]); mn2 = fminf(mn2, red[w][2]);
            mx0 = fmaxf(mx0, red[w][3]); mx1 = fmaxf(mx1, red[w][4]); mx2 = fmaxf(mx2, red[w][5]);
            cnt += redc[w];
        }
        // a tile without a finite point: the neutral codes (min = code(+inf), max = code(-inf)) and count 0
        uint4* out = reinterpret_cast<uint4*>(parts + 8u * blockIdx.x);
        out[0] = make_uint4(f2ord(mn0), f2ord(mn1), f2ord(mn2), f2ord(mx0));
        out[1] = make_uint4(f2ord(mx1), f2ord(mx2), cnt, 0u);
    }
}

struct VgGrid {
    int minb[3];
    int mul1, mul2;
    uint32_t total;
    uint32_t pass;
};

__device__ inline VgGrid vg_derive(const uint32_t bmin[3], const uint32_t bmax[3], uint32_t n_valid, float inv) {
    VgGrid g;
    g.pass = 0;
    if (n_valid == 0) {
        g.minb[0] = g.minb[1] = g.minb[2] = 0;
        g.mul1 = g.mul2 = 0;
        g.total = 0;
        return g;
    }
    float mn[3], mx[3];
    for (int a = 0; a < 3; a++) { mn[a] = ord2f(bmin[a]); mx[a] = ord2f(bmax[a]); }
    const long long dx = (long long)((mx[0] - mn[0]) * inv) + 1;
    const long long dy = (long long)((mx[1] - mn[1]) * inv) + 1;
    const long long dz = (long long)((mx[2] - mn[2]) * inv) + 1;
    long long divb[3];
    for (int a = 0; a < 3; a++) {
        g.minb[a] = (int)floorf(mn[a] * inv);
        divb[a] = (long long)((int)floorf(mx[a] * inv)) - g.minb[a] + 1;
    }
    const long long total = divb[0] * divb[1] * divb[2];
    if (dx * dy * dz > 2147483647LL || total > 0xFFFFFFF0LL) g.pass = 1;  // PCL: "leaf size is too small", output = input
    g.mul1 = (int)divb[0];
    g.mul2 = (int)(divb[0] * divb[1]);
    g.total = g.pass ? 0u : (uint32_t)total;
    return g;
}

// voxel index of every point + the histogram of its lowest digit: one workgroup per sort tile, so the first radix pass needs
// no histogram launch of its own (the later passes histogram the re-ordered keys)
template <bool FOLD>
__device__ __forceinline__ void vg_keys_body(const float4* __restrict__ in, uint32_t n, float inv, ScanDev* sd,
                                                           uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t* __restrict__ hist,
                                                           uint32_t nblocks, const uint32_t* __restrict__ parts, uint32_t bx = 0xFFFFFFFFu) {
    if (bx == 0xFFFFFFFFu) bx = blockIdx.x;  // (the tile this call works on: the launch's own block unless the caller maps blocks to tiles itself)
    // the cloud's points first (they do not depend on the box), then the fold of the tiles' box records: every workgroup forms the same box
    const uint32_t base = bx * kTile;
    float4 p[kItems];
#pragma unroll
    for (int r = 0; r < kItems; r++) {
        const uint32_t i = base + r * kThreads + threadIdx.x;
        if (i < n) p[r] = in[i];
    }
    __shared__ uint32_t h[256];
    __shared__ uint32_t sbox[kWaves][8];
    h[threadIdx.x] = 0;
    uint32_t bmin[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, bmax[3] = {0u, 0u, 0u}, n_valid = 0;
    if (!FOLD) {  // the batched chain: vg_bbox_atomic_body left the box in the scan's record
        for (int a = 0; a < 3; a++) { bmin[a] = sd->bbox_min[a]; bmax[a] = sd->bbox_max[a]; }
        n_valid = sd->n_valid;
        __syncthreads();  // (h[] cleared before the atomics below)
    } else {
    for (uint32_t b = threadIdx.x; b < nblocks; b += kThreads) {
        const uint4 r0 = reinterpret_cast<const uint4*>(parts)[2 * b], r1 = reinterpret_cast<const uint4*>(parts)[2 * b + 1];
        if (r1.z) {  // (a tile without finite points carries the codes of +-inf, which no finite point's code beats: skipped anyway)
            bmin[0] = min(bmin[0], r0.x); bmin[1] = min(bmin[1], r0.y); bmin[2] = min(bmin[2], r0.z);
            bmax[0] = max(bmax[0], r0.w); bmax[1] = max(bmax[1], r1.x); bmax[2] = max(bmax[2], r1.y);
            n_valid += r1.z;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int a = 0; a < 3; a++) { bmin[a] = min(bmin[a], (uint32_t)__shfl_xor((int)bmin[a], off)); bmax[a] = max(bmax[a], (uint32_t)__shfl_xor((int)bmax[a], off)); }
        n_valid += (uint32_t)__shfl_xor((int)n_valid, off);
    }
    if ((threadIdx.x & 63) == 0) {
        uint32_t* o = sbox[threadIdx.x >> 6];
        o[0] = bmin[0]; o[1] = bmin[1]; o[2] = bmin[2]; o[3] = bmax[0]; o[4] = bmax[1]; o[5] = bmax[2]; o[6] = n_valid;
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 3; a++) { bmin[a] = 0xFFFFFFFFu; bmax[a] = 0u; }
    n_valid = 0;
#pragma unroll
    for (int w = 0; w < kWaves; w++) {
#pragma unroll
        for (int a = 0; a < 3; a++) { bmin[a] = min(bmin[a], sbox[w][a]); bmax[a] = max(bmax[a], sbox[w][3 + a]); }
        n_valid += sbox[w][6];
    }
    }
    const VgGrid g = vg_derive(bmin, bmax, n_valid, inv);
    if (bx == 0 && threadIdx.x == 0) {
        if (FOLD) {
            for (int a = 0; a < 3; a++) { sd->bbox_min[a] = bmin[a]; sd->bbox_max[a] = bmax[a]; }
            sd->n_valid = n_valid;
        }
        sd->n_ds_prev = sd->cache_n;  // the neighbour cache's size before this scan
        sd->passthrough = g.pass;
        sd->total_cells = g.total;
        sd->nbits = g.total ? (32 - __clz(g.total)) : 0;  // keys 0..total (total = invalid marker) need bits(total)
        sd->minb[0] = g.minb[0]; sd->minb[1] = g.minb[1]; sd->minb[2] = g.minb[2];
        sd->mul1 = g.mul1; sd->mul2 = g.mul2;
    }
#pragma unroll
    for (int r = 0; r < kItems; r++) {
        const uint32_t i = base + r * kThreads + threadIdx.x;
        if (i >= n) continue;
        uint32_t key = g.total;  // invalid marker sorts behind every occupied voxel
        if (g.total && isfinite(p[r].x) && isfinite(p[r].y) && isfinite(p[r].z)) {
            const int i0 = (int)(floorf(p[r].x * inv) - (float)g.minb[0]);
            const int i1 = (int)(floorf(p[r].y * inv) - (float)g.minb[1]);
            const int i2 = (int)(floorf(p[r].z * inv) - (float)g.minb[2]);
            key = (uint32_t)(i0 + i1 * g.mul1 + i2 * g.mul2);
        }
        keys[i] = key;
        vals[i] = i;
        atomicAdd(&h[key & 255u], 1u);
    }
    __syncthreads();
    hist[bx * 256u + threadIdx.x] = h[threadIdx.x];  // [tile][digit]: one coalesced 1-KiB row per workgroup
}

// ---- stable LSD radix sort, 8 bits per pass, ping-pong a -> b -> a ... ------------------------------------
// pass p reads buffer (p & 1 ? b : a); the number of active passes is ceil(nbits / 8), so the sorted data end
// up in (active & 1 ? b : a)
__device__ inline uint32_t active_passes(const ScanDev* sd) { return (sd->nbits + 7u) >> 3; }

__device__ __forceinline__ void radix_hist_body(const uint32_t* __restrict__ ka, const uint32_t* __restrict__ kb, uint32_t n,
                                                              int pass, uint32_t* __restrict__ hist, uint32_t nblocks, const ScanDev* sd, uint32_t bx = 0xFFFFFFFFu) {
    if (bx == 0xFFFFFFFFu) bx = blockIdx.x;  // (the tile this call works on: the launch's own block unless the caller maps blocks to tiles itself)
    // (the pass count is requested with the keys, not before them: one memory round trip instead of two; a pass above the significant bits has
    // loaded a tile it does not use)
    uint32_t nbits_w = sd->nbits;
    const uint32_t* keys = (pass & 1) ? kb : ka;
    const int shift = pass * 8;
    __shared__ uint32_t h[256];
    const uint32_t base = bx * kTile;
    // the tile's keys as kItems UNCONDITIONAL loads at clamped indices, all in flight before the first is used (`if (i < n) ... keys[i]` comes out of
    // the compiler as load + s_waitcnt vmcnt(0) per item: eight memory round trips one after the other -- tools/isa_load_chains.py)
    // (a histogram does not care which thread counts which key: every thread takes 2 x 4 CONSECUTIVE keys as two 16-byte loads -- a quarter of the
    // memory requests of eight strided dword loads)
    uint4 kq[kItems / 4];
#pragma unroll
    for (int r = 0; r < kItems / 4; r++) {
        const uint32_t i = base + 4u * (uint32_t)(r * kThreads + threadIdx.x);
        kq[r] = *reinterpret_cast<const uint4*>(keys + (i + 3u < n ? i : 0u));  // (unconditional; a thread at or beyond the scan's end reads the head and ignores it)
    }
    pin_loaded(nbits_w);
#pragma unroll
    for (int r = 0; r < kItems / 4; r++) pin_loaded(kq[r]);
    if ((uint32_t)pass >= ((nbits_w + 7u) >> 3)) return;
    h[threadIdx.x] = 0;
    __syncthreads();  // (behind the loads: a workgroup barrier waits for every load in flight)
#pragma unroll
    for (int r = 0; r < kItems / 4; r++) {
        const uint32_t i = base + 4u * (uint32_t)(r * kThreads + threadIdx.x);
        if (i + 3u < n) {
            atomicAdd(&h[(kq[r].x >> shift) & 255u], 1u);
            atomicAdd(&h[(kq[r].y >> shift) & 255u], 1u);
            atomicAdd(&h[(kq[r].z >> shift) & 255u], 1u);
            atomicAdd(&h[(kq[r].w >> shift) & 255u], 1u);
        } else {  // the scan's last one to three keys (one thread of one workgroup)
            for (uint32_t e = 0; e < 3u; e++)
                if (i + e < n) atomicAdd(&h[(keys[i + e] >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    hist[bx * 256u + threadIdx.x] = h[threadIdx.x];  // [tile][digit]: one coalesced 1-KiB row per workgroup
}

__device__ inline unsigned long long match_digit(uint32_t d, bool valid) {
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const bool bit = (d >> b) & 1u;
        const unsigned long long m = __ballot(bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}

// (Measured in round 4 and not kept: this pass also building the NEXT pass's per-tile histogram -- one global atomic per key into the row of the
// tile the key lands in -- so that a scan registered alone would need no histogram launch between its scatters: 120 000 scattered atomics cost
// 17 us per pass, 8.5 -> 25.3 us for this kernel, against the 4.8 + 1.5 us of the launch they replace.)
// PREFIXED (the batched chain): radix_prefix_batch has turned the tiles' histogram rows into exclusive prefixes over the tiles (in place) and
// left the digit totals in row `nblocks`: a scatter workgroup reads its own row and the totals -- 2 KB.  Without it every workgroup folds ALL
// rows itself (59 KB for a 120 000-point scan: 223 MB of L2 reads per launch of 64 scans, 4 TB/s -- what the batched scatter spent its 51 us on);
// one scan at a time keeps that form, where a launch costs more than the fold.
template <bool PREFIXED>
__device__ __forceinline__ void radix_scatter_body(uint32_t* __restrict__ ka, uint32_t* __restrict__ va,
                                                                 uint32_t* __restrict__ kb, uint32_t* __restrict__ vb, uint32_t n, int pass,
                                                                 const uint32_t* __restrict__ hist, uint32_t nblocks, const ScanDev* sd, uint32_t bx = 0xFFFFFFFFu) {
    if (bx == 0xFFFFFFFFu) bx = blockIdx.x;  // (the tile this call works on: the launch's own block unless the caller maps blocks to tiles itself)
    uint32_t nbits_w = sd->nbits;  // (requested with the keys: see radix_hist_body)
    const uint32_t* kin = (pass & 1) ? kb : ka;
    const uint32_t* vin = (pass & 1) ? vb : va;
    uint32_t* kout = (pass & 1) ? ka : kb;
    uint32_t* vout = (pass & 1) ? va : vb;
    const int shift = pass * 8;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    __shared__ uint32_t wcnt[kWaves][256];
    __shared__ uint32_t wsum[kWaves];
    // each wave owns a contiguous run of 64*kItems keys so that (round, lane) order == input order
    const uint32_t base = bx * kTile + wave * (64 * kItems);
    uint32_t k[kItems], v[kItems];
    bool ok[kItems];
#pragma unroll
    for (int r = 0; r < kItems; r++) {
        const uint32_t i = base + r * 64 + lane;
        ok[r] = i < n;
        const uint32_t ic = ok[r] ? i : 0u;  // (unconditional loads at clamped indices, in flight together)
        k[r] = kin[ic];
        v[r] = vin[ic];
    }
    pin_loaded(nbits_w);
#pragma unroll
    for (int r = 0; r < kItems; r++) { pin_loaded(k[r]); pin_loaded(v[r]); }
    if ((uint32_t)pass >= ((nbits_w + 7u) >> 3)) return;
#pragma unroll
    for (int r = 0; r < kItems; r++)
        if (!ok[r]) { k[r] = 0u; v[r] = 0u; }
    // this workgroup's global bases, from the raw per-tile histograms: thread d owns digit d.
    //   base[d] = sum_{d' < d} total[d'] + sum_{b' < b} hist[d][b']      (digit-major, then tile order = stable)
    uint32_t tot = 0, pre = 0;
    if (PREFIXED) {
        pre = hist[(size_t)bx * 256u + tid];
        tot = hist[(size_t)nblocks * 256u + tid];
    } else {
        // [tile][digit] layout: for a given tile the 256 threads read one contiguous 1-KiB row (it used to be [digit][tile]: every lane
        // of a load in a different cache line, 64 transactions per instruction)
        const uint32_t* col = hist + tid;
        uint32_t b = 0;
        // sixteen rows in flight (the rows are L2 hits of ~0.5 us each: four at a time made the 59 rows of a 120 000-point scan fifteen dependent
        // round trips -- most of this kernel's 10 us)
        for (; b + 16 <= nblocks; b += 16) {
            uint32_t hv[16];
#pragma unroll
            for (int k = 0; k < 16; k++) hv[k] = col[(size_t)(b + k) * 256u];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                tot += hv[k];
                pre += (b + k < bx) ? hv[k] : 0u;
            }
        }
        for (; b + 4 <= nblocks; b += 4) {
            const uint32_t h0 = col[(size_t)b * 256u], h1 = col[(size_t)(b + 1) * 256u], h2 = col[(size_t)(b + 2) * 256u], h3 = col[(size_t)(b + 3) * 256u];
            tot += (h0 + h1) + (h2 + h3);
            pre += (b < bx ? h0 : 0u) + (b + 1 < bx ? h1 : 0u) + (b + 2 < bx ? h2 : 0u) + (b + 3 < bx ? h3 : 0u);
        }
        for (; b < nblocks; b++) {
            const uint32_t h0 = col[(size_t)b * 256u];
            tot += h0;
            pre += b < bx ? h0 : 0u;
        }
    }
    uint32_t inc = tot;  // inclusive scan of the digit totals across the 256 threads
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    for (int j = tid; j < kWaves * 256; j += kThreads) (&wcnt[0][0])[j] = 0;
    __syncthreads();
    uint32_t gbase = inc - tot + pre;
    for (int w = 0; w < wave; w++) gbase += wsum[w];

    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    // per item: the lanes of this wave holding the same digit (eight ballots), kept as (lanes below me, all of them) for both phases
    uint32_t below[kItems], total[kItems];
#pragma unroll
    for (int r = 0; r < kItems; r++) {
        const uint32_t d = (k[r] >> shift) & 255u;
        const unsigned long long peers = match_digit(d, ok[r]);
        below[r] = (uint32_t)__popcll(peers & lt);
        total[r] = (uint32_t)__popcll(peers);
        if (ok[r] && below[r] == 0) wcnt[wave][d] += total[r];
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    {
        uint32_t g = gbase;
#pragma unroll
        for (int w = 0; w < kWaves; w++) {
            const uint32_t t = wcnt[w][tid];
            wcnt[w][tid] = g;
            g += t;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kItems; r++) {
        const uint32_t d = (k[r] >> shift) & 255u;
        uint32_t pos = 0;
        if (ok[r]) pos = wcnt[wave][d] + below[r];
        __builtin_amdgcn_wave_barrier();
        if (ok[r] && below[r] == 0) wcnt[wave][d] += total[r];
        __builtin_amdgcn_wave_barrier();
        if (ok[r]) { kout[pos] = k[r]; vout[pos] = v[r]; }
    }
}

// ---- voxel heads: occupancy flags -> ballot + prefix-sum compaction -> centroid ------------------------
__device__ __forceinline__ void vg_count_heads_body(const uint32_t* __restrict__ ka, const uint32_t* __restrict__ kb, uint32_t n,
                                                                  const ScanDev* sd, uint32_t* __restrict__ blockcnt, uint32_t bx = 0xFFFFFFFFu) {
    if (bx == 0xFFFFFFFFu) bx = blockIdx.x;  // (the tile this call works on: the launch's own block unless the caller maps blocks to tiles itself)
    const uint32_t* keys = (active_passes(sd) & 1) ? kb : ka;
    __shared__ uint32_t c;
    if (threadIdx.x == 0) c = 0;
    __syncthreads();
    const uint32_t total = sd->total_cells;
    const uint32_t base = bx * kTile;
    uint32_t mine = 0;
    // (2 x kItems unconditional loads at clamped indices, in flight together: see radix_hist_body)
    uint32_t kc[kItems], kp[kItems];
#pragma unroll
    for (int r = 0; r < kItems; r++) {
        const uint32_t i = base + r * kThreads + threadIdx.x;
        const uint32_t ic = i < n ? i : (n ? n - 1u : 0u);
        kc[r] = keys[ic];
        kp[r] = keys[ic ? ic - 1u : 0u];
    }
#pragma unroll
    for (int r = 0; r < kItems; r++) {
        const uint32_t i = base + r * kThreads + threadIdx.x;
        const bool head = i < n && kc[r] < total && (i == 0 || kp[r] != kc[r]);
        mine += __popcll(__ballot(head));
    }
    if ((threadIdx.x & 63) == 0) atomicAdd(&c, mine);
    __syncthreads();
    if (threadIdx.x == 0) blockcnt[bx] = c;
}

// compaction of the voxel heads (ballot + prefix sum, fixed order) and the gather of the points into sorted order.
// (Measured in round 4 and not kept: every load of the tile requested up front, the ballot rounds on registers -- 11.4 -> 13.8 us for one scan,
// 42 -> 79 us for a batched round: the eight gathers through the sort's permutation then leave together and queue behind each other.)
__device__ __forceinline__ void vg_heads_body(const float4* __restrict__ in, const uint32_t* __restrict__ ka,
                                                            const uint32_t* __restrict__ kb, const uint32_t* __restrict__ va,
                                                            const uint32_t* __restrict__ vb, uint32_t n, ScanDev* sd,
                                                            const uint32_t* __restrict__ blockcnt, uint32_t* __restrict__ hpos,
                                                            float4* __restrict__ sorted, float4* __restrict__ out, uint32_t max_ds, uint32_t* __restrict__ host_nds,
                                                            uint32_t launched_passes, uint32_t last_block, uint32_t bx = 0xFFFFFFFFu) {
    if (bx == 0xFFFFFFFFu) bx = blockIdx.x;  // (the tile this call works on: the launch's own block unless the caller maps blocks to tiles itself)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (sd->passthrough) {  // PCL overflow guard: output = input
        if (bx == 0 && tid == 0) host_nds[2] = 0u;
        if (n > max_ds) {
            if (bx == 0 && tid == 0) { sd->err |= 1u; sd->n_ds = 0; host_nds[0] = 0; host_nds[1] = 1u; }
            return;
        }
        const uint32_t base = bx * kTile;
        for (int r = 0; r < kItems; r++) {
            const uint32_t i = base + r * kThreads + tid;
            if (i < n) out[i] = in[i];
        }
        if (bx == 0 && tid == 0) { sd->n_ds = n; host_nds[0] = n; host_nds[1] = 0u; }
        return;
    }
    const bool odd = active_passes(sd) & 1;
    const uint32_t* keys = odd ? kb : ka;
    const uint32_t* vals = odd ? vb : va;
    __shared__ uint32_t red[kWaves];
    // exclusive prefix of the tiles before this one (fixed order -> deterministic output slots)
    uint32_t pre = 0;
    for (uint32_t b = tid; b < bx; b += kThreads) pre += blockcnt[b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) pre += __shfl_xor(pre, off);
    if (lane == 0) red[wave] = pre;
    __syncthreads();
    uint32_t run = 0;
    for (int w = 0; w < kWaves; w++) run += red[w];
    __syncthreads();

    const uint32_t total = sd->total_cells;
    const uint32_t base = bx * kTile;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    // Round 4, second form.  Until then: per item { key, previous key, index, the gather through the index, two workgroup barriers } -- four dependent
    // memory round trips and two barriers eight times over (every load sat inside `if (i < n)`: load, s_waitcnt vmcnt(0), next load;
    // tools/isa_load_chains.py).  Now: the 2 x kItems key loads and the kItems indices in flight together (clamped, unconditional), the head
    // ballots of all items on registers with ONE barrier for the waves' counts, then the gathers four at a time.  (The first attempt at an
    // up-front form held all eight gathered points at once -- 32 more registers -- and was slower; four at a time keeps the occupancy.)
    uint32_t kc[kItems], kp[kItems], vv[kItems];
#pragma unroll
    for (int r = 0; r < kItems; r++) {
        const uint32_t i = base + r * kThreads + tid;
        const uint32_t ic = i < n ? i : (n ? n - 1u : 0u);
        kc[r] = keys[ic];
        kp[r] = keys[ic ? ic - 1u : 0u];
        vv[r] = vals[ic];
    }
    __shared__ uint32_t wcnt[kItems][kWaves];
    unsigned long long hm[kItems];
#pragma unroll
    for (int r = 0; r < kItems; r++) {
        const uint32_t i = base + r * kThreads + tid;
        const bool head = i < n && kc[r] < total && (i == 0 || kp[r] != kc[r]);
        hm[r] = __ballot(head);
        if (lane == 0) wcnt[r][wave] = (uint32_t)__popcll(hm[r]);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kItems; r++) {
        uint32_t woff = 0, rtot = 0;
#pragma unroll
        for (int w = 0; w < kWaves; w++) {
            const uint32_t t = wcnt[r][w];
            if (w < wave) woff += t;
            rtot += t;
        }
        if ((hm[r] >> lane) & 1ull) {
            const uint32_t slot = run + woff + (uint32_t)__popcll(hm[r] & lt);
            if (slot < max_ds) hpos[slot] = base + r * kThreads + tid;
        }
        run += rtot;
    }
#pragma unroll
    for (int r0 = 0; r0 < kItems; r0 += 4) {
        float4 pt[4];
#pragma unroll
        for (int k = 0; k < 4; k++) pt[k] = in[vv[r0 + k]];  // (a thread beyond the scan's end gathers the last point again and drops it)
#pragma unroll
        for (int k = 0; k < 4; k++) pin_loaded(pt[k]);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t i = base + (r0 + k) * kThreads + tid;
            if (i < n) sorted[i] = pt[k];
        }
    }
    if (bx == last_block && tid == 0) {
        uint32_t err = 0;
        if (run > max_ds) { sd->err |= 1u; run = 0; err = 1u; }
        // the host launched as many radix passes as the previous scan needed; if this scan's bounding box needs more, the keys
        // are not fully sorted: say so (bit 2) and let the host run the chain again with all four (rare: the cell count crossed 2^(8k))
        if (active_passes(sd) > launched_passes) { err |= 2u; run = 0; }
        sd->n_ds = run;
        host_nds[0] = run;  // mapped pinned host words: the host reads them after one stream sync, no copy launch
        host_nds[1] = err;
        host_nds[2] = active_passes(sd);
    }
}

// one lane per occupied voxel: centroid of its run in ascending input order (f32 running sums, as PCL does).
// Runs shorter than 32 points are walked by the owning lane, four loads in flight.  Longer runs (the rings
// close to the sensor: a few hundred voxels holding half of the points) are queued for the wave-per-voxel
// kernel below, so that their latency is spread over the chip instead of serialising one wave.
constexpr uint32_t kLongRun = 32;
constexpr uint32_t kCentroidGridCap = 96;  // workgroups per scan of vg_centroid_*: 96 x 256 = 24 576 voxels per sweep of the grid
// ... and the few runs of thousands of points (a wall a metre from the sensor, the ground ring under it) go to a second queue, filled from the
// top of the same array, whose runs are summed one COMPONENT per wave by integer arithmetic inside the running sum's binade (monster_sum below)
constexpr uint32_t kMonsterRun = 2048;
constexpr uint32_t kMonsterBlocks = 16;       // workgroups of the long-run launch that serve the monster queue (one scan)
constexpr uint32_t kMonsterBlocksBatch = 16;  // ... per slot of the batched chain (its own launch there)

__device__ __forceinline__ void vg_centroid_body(const float4* __restrict__ sorted, const uint32_t* __restrict__ hpos,
                                                               ScanDev* __restrict__ sd, float4* __restrict__ out,
                                                               uint32_t* __restrict__ longlist, uint32_t max_ds, uint32_t vb) {
    // vb: the block of kThreads voxels this call works on -- the kernels stride over the scan's blocks (a grid sized for max_ds = 100 000 voxels left
    // 345 of 391 workgroups per slot to start, wait for the scan's size and exit)
    if (sd->passthrough) return;
    const uint32_t nv = sd->n_ds;
    const uint32_t v = vb * kThreads + threadIdx.x;
    if (v >= nv) return;
    // (both run bounds and the scan's valid count requested together: the conditional form was two memory round trips one after the other)
    uint32_t a = hpos[v];
    uint32_t b_next = hpos[v + 1 < nv ? v + 1 : v];
    const uint32_t n_valid = sd->n_valid;
    pin_loaded(a);
    pin_loaded(b_next);
    const uint32_t b = (v + 1 < nv) ? b_next : n_valid;  // invalid (non-finite) points sort behind every voxel
    const bool is_monster = b - a >= kMonsterRun;
    const unsigned long long mm = __ballot(is_monster);
    if (mm) {
        const int lane = threadIdx.x & 63;
        const int leader = __ffsll((long long)mm) - 1;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&sd->n_monster, (uint32_t)__popcll(mm));
        base = __shfl(base, leader);
        if (is_monster) {
            longlist[max_ds - 1u - (base + __popcll(mm & ((1ull << lane) - 1ull)))] = v;
            return;
        }
    }
    // queue the long runs: one atomic per wave (ballot + popcount), not one per voxel
    const bool is_long = b - a >= kLongRun;
    const unsigned long long lm = __ballot(is_long);
    if (lm) {
        const int lane = threadIdx.x & 63;
        const int leader = __ffsll((long long)lm) - 1;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&sd->n_long, (uint32_t)__popcll(lm));
        base = __shfl(base, leader);
        if (is_long) {
            longlist[base + __popcll(lm & ((1ull << lane) - 1ull))] = v;
            return;
        }
    }
    float sx = 0.f, sy = 0.f, sz = 0.f, sw = 0.f;
    // the first eight points of the run in one request (most runs end there: one memory round trip instead of two), then four at a time
    {
        float4 p[8];
#pragma unroll
        for (int k = 0; k < 8; k++) p[k] = sorted[(a + k < b) ? (a + k) : (b - 1)];
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (a + k < b) { sx = sx + p[k].x; sy = sy + p[k].y; sz = sz + p[k].z; sw = sw + p[k].w; }
    }
    for (uint32_t j = a + 8; j < b; j += 4) {
        float4 p[4];
#pragma unroll
        for (int k = 0; k < 4; k++) p[k] = sorted[(j + k < b) ? (j + k) : (b - 1)];
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (j + k < b) { sx = sx + p[k].x; sy = sy + p[k].y; sz = sz + p[k].z; sw = sw + p[k].w; }
    }
    const float c = (float)(b - a);
    out[v] = make_float4(sx / c, sy / c, sz / c, sw / c);
}

// one wave per long run.  The sum itself is PCL's sequential f32 sum -- a dependent chain by definition -- run as four chains side by side (lanes
// 0..3 own one coordinate each) over 64 values parked in LDS by coordinate, ~5 cycles per point.  What the chain should not do is wait for
// memory: FOUR batches of 64 points are in flight ahead of the one being summed (a ring of four registers per lane; a batch beyond the run's
// end is not requested, so a 40-point run costs one load).  Until round 4 one batch was in flight: ~0.15 us of adds behind every ~0.8 us load for
// the voxels that hold thousands of points (a wall a metre from the sensor).  (Also measured in round 4 and not kept: 512 points parked per
// step -- 32 KB of LDS per workgroup and eight loads per lane even for a 40-point run: 31 -> 83 us per batched round.)
// The sequential f32 sum  s <- fl(s + x_i)  of ONE coordinate over a run of thousands of points, exactly, 256 points per step of one wave.
// While the running sum stays inside one binade [2^k, 2^(k+1)) every addition rounds to a multiple of q = 2^(k-23); s = S q with an integer
// S, and fl(s + x) = (S + rne(x / q)) q unless x / q lies exactly half way between two integers (then the tie goes to the even RESULT, which
// depends on S).  Integer additions associate: the 512 addends of a step become integers R_i = rne(x_i / q) (lane l holds points 8l .. 8l + 7
// of the step), a wave prefix sum gives every partial sum T_i = S + R_1 + ... + R_i, and the step is accepted if no x_i / q is a tie, every
// |R_i| < 2^21 (no integer overflow) and every T_i lies strictly inside (2^23, 2^24) with the sign of S -- i.e. every intermediate sum the
// sequential loop would have formed stayed in the binade: then those sums ARE T_i q, bit for bit.  Otherwise (the sum crosses into the next binade
// ~log2(n) times per run; a tie about once per 2^12 points; the first step, from s = 0) the step is redone by the plain loop on one lane.
// tests/test_seqsum_math.py holds the rule against the plain loop on adversarial data on the CPU; tests/test_voxelgrid_monster_gpu.py this code.
// inclusive prefix sum over the 64 lanes of a wave by DPP (row shifts inside the rows of 16, then the row totals broadcast down): ~6 VALU
// instructions instead of six ds_bpermute round trips
__device__ __forceinline__ int wave_inclusive_scan_i32(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);  // row_shr:1 (lanes without a source keep `old` = 0)
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);  // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
    return x;
}

constexpr int kMonsterPer = 8;                       // points per lane and step
constexpr int kMonsterStep = 64 * kMonsterPer;       // 512 points per step of one wave
template <int kRing>
__device__ __forceinline__ float monster_component_sum(const float4* __restrict__ sorted, uint32_t ra, uint32_t rb, int c, float* __restrict__ park /* [kMonsterStep] */,
                                                       int lane) {
    // coordinate c of point i as ONE dword load at a computed address: a float4 load followed by a select on the (run-time) coordinate number made
    // the wave wait for every request on the spot -- nothing was in flight ahead, 2 400 cycles per step instead of ~800
    const float* __restrict__ flat = reinterpret_cast<const float*>(sorted) + c;
    auto comp_at = [flat](uint32_t i) { return flat[(size_t)i * 4u]; };
    float s = 0.f;
    // kRing steps requested ahead of the one being summed.  Three (1536 points, 6 KB of dword loads in flight per wave) where the long-run waves
    // share the launch (one scan at a time: five steps meant 81-94 VGPRs and cost those waves three of their eight per SIMD); eight in the
    // batched chain's own monster launch, where a step waits for its loads (~0.8 us per step of a 30 000-point run) and registers are free.
    constexpr int P = kMonsterPer;
    float nx[kRing][P];
#pragma unroll
    for (int r = 0; r < kRing; r++)
#pragma unroll
        for (int j = 0; j < P; j++) {
            const uint32_t i = ra + (uint32_t)r * kMonsterStep + (uint32_t)j * 64u + lane;
            nx[r][j] = i < rb ? comp_at(i) : 0.f;
        }
    for (uint32_t base = ra; base < rb; base += (uint32_t)kMonsterStep * kRing) {
#pragma unroll
        for (int r = 0; r < kRing; r++) {
            const uint32_t pos = base + (uint32_t)r * kMonsterStep;
            if (pos >= rb) break;  // (uniform over the wave)
            const uint32_t n_here = rb - pos < (uint32_t)kMonsterStep ? rb - pos : (uint32_t)kMonsterStep;
#pragma unroll
            for (int j = 0; j < P; j++) park[j * 64 + lane] = nx[r][j];  // (beyond the run's end: zeros -- adding 0 changes no sum)
            if (pos + (uint32_t)kMonsterStep * kRing < rb) {  // this register set's next step: kRing steps ahead
#pragma unroll
                for (int j = 0; j < P; j++) {
                    const uint32_t i = pos + (uint32_t)kMonsterStep * kRing + (uint32_t)j * 64u + lane;
                    nx[r][j] = i < rb ? comp_at(i) : 0.f;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float xs[P];  // points P lane .. P lane + P - 1 of the step, in order
#pragma unroll
            for (int j = 0; j < P; j += 4) {
                const float4 m4 = *reinterpret_cast<const float4*>(park + lane * P + j);
                xs[j] = m4.x; xs[j + 1] = m4.y; xs[j + 2] = m4.z; xs[j + 3] = m4.w;
            }
            bool fast = false;
            const uint32_t sb = __float_as_uint(s);
            const int be = (int)((sb >> 23) & 0xFFu);
            if (be >= 1 && be <= 254) {  // a normal, non-zero running sum (uniform over the wave)
                // sign-normalised: with s < 0 every addend is negated, so that the integer sums below are those of |s|
                const uint32_t flip = sb & 0x80000000u;
                const int e = be - 127 - 23;
                const int S0 = (int)ldexpf(__uint_as_float(sb & 0x7FFFFFFFu), -e);  // the mantissa: 2^23 <= S0 < 2^24
                float worst = 0.f;   // max |x / q| of this lane's points
                bool tie = false;
                int loc = 0, lmin = 0x7FFFFFFF, lmax = (int)0x80000000;
#pragma unroll
                for (int j = 0; j < P; j++) {  // (the zeros beyond the run's end take part: R = 0, harmless)
                    const float rr = ldexpf(__uint_as_float(__float_as_uint(xs[j]) ^ flip), -e);
                    const float rn = rintf(rr);
                    worst = fmaxf(worst, fabsf(rr));
                    tie |= fabsf(rr - rn) == 0.5f;
                    tie |= !(rr == rr);  // a NaN addend (fmaxf keeps `worst`, the clamp would read it as -2^21): the step takes the plain loop, which yields PCL's NaN
                    loc += (int)fminf(fmaxf(rn, -2097152.f), 2097152.f);  // (clamped: a step with such an addend is rejected below; 512 x 2^21 + 2^24 < 2^31)
                    lmin = min(lmin, loc);
                    lmax = max(lmax, loc);
                }
                const int inc = wave_inclusive_scan_i32(loc);
                const int before = S0 + (inc - loc);
                // every partial sum strictly inside (2^23, 2^24): the smallest and the largest of this lane's are enough
                const bool viol = tie || !(worst < 2097152.f) || !(before + lmin > 8388608) || !(before + lmax < 16777216);
                if (!__ballot(viol)) {
                    const int total = __builtin_amdgcn_readlane(inc, 63);
                    s = __uint_as_float(__float_as_uint(ldexpf((float)(S0 + total), e)) | flip);
                    fast = true;
                }
            }
            if (!fast && (sb << 1) == 0u) {  // s = +-0 (the start of a run; a coordinate that is zero throughout, e.g. an unused intensity):
                bool nz = false;             // 0 + 0 + ... stays what it is -- a step of zeros needs no loop
#pragma unroll
                for (int j = 0; j < P; j++) nz |= (__float_as_uint(xs[j]) << 1) != 0u;
                if (sb == 0u && !__ballot(nz)) fast = true;  // (+0 + -0 = +0 too; a -0 sum is left to the loop)
            }
            if (!fast) {  // the plain loop over the parked step, one lane
                float t = s;
                if (lane == 0) {
                    uint32_t j = 0;
                    for (; j + 4 <= n_here; j += 4) {
                        const float4 a4 = *reinterpret_cast<const float4*>(park + j);
                        t = t + a4.x; t = t + a4.y; t = t + a4.z; t = t + a4.w;
                    }
                    for (; j < n_here; j++) t = t + park[j];
                }
                s = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(t)));  // lane 0's result to the wave
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    return s;
}

// workgroups [0, n_long_blocks) serve the long-run queue one run per wave; the workgroups beyond serve the monster queue one run per
// workgroup, wave c summing coordinate c
// WHICH: 0 = both queues in one launch (one scan at a time: a launch less), 1 = the long-run queue only, 2 = the monster queue only (the batched
// chain: the monster code's registers cost the long-run waves two of their eight waves per SIMD -- 69 -> 85 us per round of 64 when both shared a
// launch -- so there the monster queue gets a small launch of its own)
template <int WHICH>
__device__ __forceinline__ void vg_centroid_long_body(const float4* __restrict__ sorted, const uint32_t* __restrict__ hpos,
                                                                    const ScanDev* __restrict__ sd, float4* __restrict__ out,
                                                                    const uint32_t* __restrict__ longlist, uint32_t max_ds, uint32_t n_long_blocks) {
    if (sd->passthrough) return;
    const uint32_t nv = sd->n_ds, nl = sd->n_long;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __shared__ __attribute__((aligned(16))) float park[kWaves][kMonsterStep > 256 ? kMonsterStep : 256];
    if (WHICH == 2 || (WHICH == 0 && blockIdx.x >= n_long_blocks)) {
        const uint32_t nm = sd->n_monster;
        for (uint32_t m = blockIdx.x - n_long_blocks; m < nm; m += gridDim.x - n_long_blocks) {
            const uint32_t v = longlist[max_ds - 1u - m];
            const uint32_t ra = hpos[v];
            const uint32_t rb = (v + 1 < nv) ? hpos[v + 1] : sd->n_valid;
            const float t = monster_component_sum<WHICH == 2 ? 8 : 3>(sorted, ra, rb, wv, &park[wv][0], lane);
            if (lane == 0) reinterpret_cast<float*>(&out[v])[wv] = t / (float)(rb - ra);
        }
        return;
    }
    if (WHICH == 2) return;
    const uint32_t nwaves = n_long_blocks * kWaves;
    float (*pk)[64] = reinterpret_cast<float (*)[64]>(&park[wv][0]);  // this wave's four coordinate rows of 64
    constexpr int kAhead = 4;
    for (uint32_t w = blockIdx.x * kWaves + wv; w < nl; w += nwaves) {
        const uint32_t v = longlist[w];
        const uint32_t ra = hpos[v];
        const uint32_t rb = (v + 1 < nv) ? hpos[v + 1] : sd->n_valid;
        float t = 0.f;  // lane c < 4: the running sum of coordinate c
        float4 p[kAhead];
#pragma unroll
        for (int k = 0; k < kAhead; k++) {
            const uint32_t c0 = ra + (uint32_t)k * 64u;
            if (c0 < rb) p[k] = sorted[(c0 + lane) < rb ? c0 + lane : rb - 1];
        }
        for (uint32_t c = ra; c < rb; c += 64u * kAhead) {
#pragma unroll
            for (int k = 0; k < kAhead; k++) {
                const uint32_t c0 = c + (uint32_t)k * 64u;
                if (c0 >= rb) break;  // (uniform over the wave)
                pk[0][lane] = p[k].x; pk[1][lane] = p[k].y; pk[2][lane] = p[k].z; pk[3][lane] = p[k].w;
                const uint32_t cn = c0 + 64u * kAhead;  // this register's next batch: four batches ahead
                if (cn < rb) p[k] = sorted[(cn + lane) < rb ? cn + lane : rb - 1];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int m = (rb - c0) < 64u ? (int)(rb - c0) : 64;
                if (lane < 4) {
                    const float* q = pk[lane];
                    int j = 0;
                    for (; j + 4 <= m; j += 4) {
                        const float4 a = *reinterpret_cast<const float4*>(q + j);
                        t = t + a.x; t = t + a.y; t = t + a.z; t = t + a.w;
                    }
                    for (; j < m; j++) t = t + q[j];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        const float cnt = (float)(rb - ra);
        if (lane < 4) reinterpret_cast<float*>(&out[v])[lane] = t / cnt;
    }
}

__global__ void scan_set_nds_kernel(ScanDev* sd, uint32_t n) {
    sd->n_ds_prev = sd->cache_n;
    sd->n_ds = n;
}

// Nearest_Points.resize(feats_down_size) (laserMapping.cpp:1274): entries beyond the new size are destroyed
// ... and re-arm the bbox / counters for the next scan's downsample (saves two memset launches per scan)
// -- but only for a scan that is registered: fastlio_main returns on feats_down_size < 5 (laserMapping.cpp:1250-1254) BEFORE that resize,
// the cache of the previous scan then survives whole (min_ds = 5 on the engines' scans, 0 for a bare lio_scan)
// reset (an independent scan, lio_scan_job.flags == 0): the cache is forgotten first -- Nearest_Points as fastlio_init leaves it
// (laserMapping.cpp:1045-1047) -- i.e. every entry the previous scan left is cleared, not only those beyond the new size (entries
// beyond cache_n are zero by construction; a search that finds nothing in range leaves an entry alone, so a zero count stays zero)
__device__ __forceinline__ void scan_begin_body(ScanDev* sd, int32_t* __restrict__ nn_cnt, uint32_t min_ds, uint32_t reset) {
    const uint32_t n = sd->n_ds, hi = sd->n_ds_prev;
    const uint32_t lo = reset ? 0u : n;
    if (n >= min_ds) {
        for (uint32_t i = lo + blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += gridDim.x * blockDim.x) nn_cnt[i] = 0;
        if (blockIdx.x == 0 && threadIdx.x == 0) sd->cache_n = n;
    }
    if (blockIdx.x == 0 && threadIdx.x < 3) {
        sd->bbox_min[threadIdx.x] = 0xFFFFFFFFu;
        sd->bbox_max[threadIdx.x] = 0u;
        if (threadIdx.x == 0) { sd->n_valid = 0; sd->n_long = 0; sd->n_monster = 0; }
    }
}


// XCD-aware workgroup -> (slot, tile) mapping for the tile kernels of the batched chain (LIO_VG_XCD=0 restores the plain mapping).  The dispatcher deals the
// workgroups of a launch round-robin to the 8 XCDs in dispatch order (x fastest), each XCD with its own L2: with the plain mapping the 59-64 tiles of
// ONE scan are spread over all eight L2s -- the scatter's partial-line writes into a scan's 256 digit regions, and the gather of its cloud, are
// then merged / cached in eight places.  Here workgroup L (= blockIdx.x + gridDim.x * blockIdx.y) takes tile (L / 8) % T of slot
// ((L / 8) / T) * 8 + L % 8: all tiles of a slot on one XCD.  Slots beyond the last whole group of eight keep the plain mapping.
#ifndef LIO_VG_XCD
#define LIO_VG_XCD 1  // (measured, round 5: vg_heads_batch 167 -> 124 us, radix_scatter_batch 79 -> 70 us per launch of 128 scans, chain 810 -> 757 us per round)
#endif
__device__ __forceinline__ void vg_slot_tile(uint32_t n_slots, uint32_t& slot, uint32_t& bx) {
    slot = blockIdx.y;
    bx = blockIdx.x;
#if LIO_VG_XCD
    const uint32_t T = gridDim.x, L = blockIdx.x + T * blockIdx.y, full = n_slots & ~7u;
    if (L < full * T) {
        const uint32_t j = L >> 3;
        slot = (j / T) * 8u + (L & 7u);
        bx = j % T;
    }
#endif
}

// ---- launchable forms: one scan (arguments by value), or the scans of a batch (blockIdx.y = slot, arguments from the slot's
// descriptor in device memory; a workgroup beyond the slot's own tile count, or of an idle slot, exits at once) ----------------
// the tiles' box records live in the head of the `sorted` buffer (free until vg_heads fills it): 32 bytes per tile
__global__ void __launch_bounds__(kThreads) vg_bbox_kernel(const float4* __restrict__ in, uint32_t n, uint32_t* __restrict__ parts) { vg_bbox_body(in, n, parts); }
__global__ void __launch_bounds__(kThreads) vg_bbox_batch(const SlotDesc* __restrict__ slots) {
    const SlotDesc& d = slots[blockIdx.y];
    if (!d.active) return;  // (grid-stride loop over the points: every workgroup of the row takes part)
    vg_bbox_atomic_body(d.raw, d.n_raw, d.sd);
}
__global__ void __launch_bounds__(kThreads) vg_keys_kernel(const float4* __restrict__ in, uint32_t n, float inv, ScanDev* sd,
                                                           uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t* __restrict__ hist,
                                                           uint32_t nblocks, const uint32_t* __restrict__ parts) {
    vg_keys_body<true>(in, n, inv, sd, keys, vals, hist, nblocks, parts);
}
__global__ void __launch_bounds__(kThreads) vg_keys_batch(const SlotDesc* __restrict__ slots, float inv) {
    uint32_t slot, bx;
    vg_slot_tile(gridDim.y, slot, bx);
    const SlotDesc& d = slots[slot];
    if (!d.active || bx >= d.nblocks) return;
    vg_keys_body<false>(d.raw, d.n_raw, inv, d.sd, d.keys_a, d.vals_a, d.hist, d.nblocks, nullptr, bx);
}
__global__ void __launch_bounds__(kThreads) radix_hist_kernel(const uint32_t* __restrict__ ka, const uint32_t* __restrict__ kb, uint32_t n,
                                                              int pass, uint32_t* __restrict__ hist, uint32_t nblocks, const ScanDev* sd) {
    radix_hist_body(ka, kb, n, pass, hist, nblocks, sd);
}
__global__ void __launch_bounds__(kThreads) radix_hist_batch(const SlotDesc* __restrict__ slots, int pass) {
    uint32_t slot, bx;
    vg_slot_tile(gridDim.y, slot, bx);
    const SlotDesc& d = slots[slot];
    if (!d.active || bx >= d.nblocks) return;
    radix_hist_body(d.keys_a, d.keys_b, d.n_raw, pass, d.hist, d.nblocks, d.sd, bx);
}
__global__ void __launch_bounds__(kThreads) radix_scatter_kernel(uint32_t* __restrict__ ka, uint32_t* __restrict__ va,
                                                                 uint32_t* __restrict__ kb, uint32_t* __restrict__ vb, uint32_t n, int pass,
                                                                 const uint32_t* __restrict__ hist, uint32_t nblocks, const ScanDev* sd) {
    radix_scatter_body<false>(ka, va, kb, vb, n, pass, hist, nblocks, sd);
}
__global__ void __launch_bounds__(kThreads) radix_scatter_batch(const SlotDesc* __restrict__ slots, int pass) {
    uint32_t slot, bx;
    vg_slot_tile(gridDim.y, slot, bx);
    const SlotDesc& d = slots[slot];
    if (!d.active || bx >= d.nblocks) return;
    radix_scatter_body<true>(d.keys_a, d.vals_a, d.keys_b, d.vals_b, d.n_raw, pass, d.hist, d.nblocks, d.sd, bx);
}
// one workgroup per scan: digit column d (thread d) of the tiles' histogram rows becomes its exclusive prefix over the tiles, the column's total
// goes to row `nblocks` (sixteen rows in flight; integer sums: the scatter's positions are what the all-rows fold gave)
__global__ void __launch_bounds__(256) radix_prefix_batch(const SlotDesc* __restrict__ slots, int pass) {
    const SlotDesc& d = slots[blockIdx.x];
    if (!d.active) return;
    if ((uint32_t)pass >= active_passes(d.sd)) return;
    const uint32_t nb = d.nblocks;
    uint32_t* col = d.hist + threadIdx.x;
    uint32_t run = 0;
    for (uint32_t b = 0; b < nb; b += 16) {
        uint32_t hv[16];
#pragma unroll
        for (int k = 0; k < 16; k++) hv[k] = col[(size_t)(b + k < nb ? b + k : b) * 256u];
#pragma unroll
        for (int k = 0; k < 16; k++)
            if (b + k < nb) {
                col[(size_t)(b + k) * 256u] = run;
                run += hv[k];
            }
    }
    col[(size_t)nb * 256u] = run;
}
__global__ void __launch_bounds__(kThreads) vg_count_heads_kernel(const uint32_t* __restrict__ ka, const uint32_t* __restrict__ kb, uint32_t n,
                                                                  const ScanDev* sd, uint32_t* __restrict__ blockcnt) {
    vg_count_heads_body(ka, kb, n, sd, blockcnt);
}
__global__ void __launch_bounds__(kThreads) vg_count_heads_batch(const SlotDesc* __restrict__ slots) {
    uint32_t slot, bx;
    vg_slot_tile(gridDim.y, slot, bx);
    const SlotDesc& d = slots[slot];
    if (!d.active || bx >= d.nblocks) return;
    vg_count_heads_body(d.keys_a, d.keys_b, d.n_raw, d.sd, d.blockcnt, bx);
}
__global__ void __launch_bounds__(kThreads) vg_heads_kernel(const float4* __restrict__ in, const uint32_t* __restrict__ ka,
                                                            const uint32_t* __restrict__ kb, const uint32_t* __restrict__ va,
                                                            const uint32_t* __restrict__ vb, uint32_t n, ScanDev* sd,
                                                            const uint32_t* __restrict__ blockcnt, uint32_t* __restrict__ hpos,
                                                            float4* __restrict__ sorted, float4* __restrict__ out, uint32_t max_ds, uint32_t* __restrict__ host_nds,
                                                            uint32_t launched_passes) {
    vg_heads_body(in, ka, kb, va, vb, n, sd, blockcnt, hpos, sorted, out, max_ds, host_nds, launched_passes, gridDim.x - 1);
}
__global__ void __launch_bounds__(kThreads) vg_heads_batch(const SlotDesc* __restrict__ slots, uint32_t launched_passes) {
    uint32_t slot, bx;
    vg_slot_tile(gridDim.y, slot, bx);
    const SlotDesc& d = slots[slot];
    if (!d.active || bx >= d.nblocks) return;
    vg_heads_body(d.raw, d.keys_a, d.keys_b, d.vals_a, d.vals_b, d.n_raw, d.sd, d.blockcnt, d.hpos, d.sorted, d.ds_body, d.max_ds, d.host_nds,
                  launched_passes, d.nblocks - 1, bx);
}
__global__ void __launch_bounds__(kThreads) vg_centroid_kernel(const float4* __restrict__ sorted, const uint32_t* __restrict__ hpos,
                                                               ScanDev* __restrict__ sd, float4* __restrict__ out,
                                                               uint32_t* __restrict__ longlist, uint32_t max_ds) {
    const uint32_t nv = sd->passthrough ? 0u : sd->n_ds;
    for (uint32_t vb = blockIdx.x; vb * kThreads < nv; vb += gridDim.x) vg_centroid_body(sorted, hpos, sd, out, longlist, max_ds, vb);
}
__global__ void __launch_bounds__(kThreads) vg_centroid_batch(const SlotDesc* __restrict__ slots) {
    const SlotDesc& d = slots[blockIdx.y];
    if (!d.active) return;
    const uint32_t nv = d.sd->passthrough ? 0u : d.sd->n_ds;
    for (uint32_t vb = blockIdx.x; vb * kThreads < nv; vb += gridDim.x) vg_centroid_body(d.sorted, d.hpos, d.sd, d.ds_body, d.longlist, d.max_ds, vb);
}
__global__ void __launch_bounds__(kThreads) vg_centroid_long_kernel(const float4* __restrict__ sorted, const uint32_t* __restrict__ hpos,
                                                                    const ScanDev* __restrict__ sd, float4* __restrict__ out,
                                                                    const uint32_t* __restrict__ longlist, uint32_t max_ds, uint32_t n_long_blocks) {
    vg_centroid_long_body<0>(sorted, hpos, sd, out, longlist, max_ds, n_long_blocks);
}
// + the scan-begin duties (Nearest_Points.resize, re-arming the bbox) of the batch: the long-run kernel is the last of the chain
__global__ void __launch_bounds__(kThreads) vg_centroid_long_batch(const SlotDesc* __restrict__ slots) {
    const SlotDesc& d = slots[blockIdx.y];
    if (!d.active) return;
    vg_centroid_long_body<1>(d.sorted, d.hpos, d.sd, d.ds_body, d.longlist, d.max_ds, gridDim.x);
}
__global__ void __launch_bounds__(kThreads) vg_centroid_both_batch(const SlotDesc* __restrict__ slots) {
    const SlotDesc& d = slots[blockIdx.y];
    if (!d.active) return;
    vg_centroid_long_body<0>(d.sorted, d.hpos, d.sd, d.ds_body, d.longlist, d.max_ds, gridDim.x - kMonsterBlocksBatch);
}
__global__ void __launch_bounds__(kThreads) vg_centroid_monster_batch(const SlotDesc* __restrict__ slots) {
    const SlotDesc& d = slots[blockIdx.y];
    if (!d.active || d.sd->n_monster == 0) return;
    vg_centroid_long_body<2>(d.sorted, d.hpos, d.sd, d.ds_body, d.longlist, d.max_ds, 0u);
}
__global__ void scan_begin_kernel(ScanDev* sd, int32_t* __restrict__ nn_cnt, uint32_t min_ds) { scan_begin_body(sd, nn_cnt, min_ds, 0u); }
__global__ void scan_begin_batch(const SlotDesc* __restrict__ slots) {
    const SlotDesc& d = slots[blockIdx.y];
    if (!d.active) return;
    scan_begin_body(d.sd, d.nn_cnt, d.min_ds, d.reset_cache);
}

int vg_downsample(lio_scan* s, float leaf, int passes) {
    const uint32_t n = s->n_raw;
    const float inv = 1.0f / leaf;
    hipStream_t st = s->stream;
    // bbox / n_valid / n_long were re-armed by the previous scan_begin_kernel (or at creation)
    const uint32_t nblocks = (n + kTile - 1) / kTile;
    if (n == 0) {
        hipLaunchKernelGGL(scan_set_nds_kernel, 1, 1, 0, st, s->dev, 0u);
        s->host_nds[0] = 0;
        s->host_nds[1] = 0;
        s->host_nds[2] = 0;
        return LIO_OK;
    }
    uint32_t* parts = reinterpret_cast<uint32_t*>(s->sorted);
    hipLaunchKernelGGL(vg_bbox_kernel, nblocks, kThreads, 0, st, s->raw, n, parts);
    hipLaunchKernelGGL(vg_keys_kernel, nblocks, kThreads, 0, st, s->raw, n, inv, s->dev, s->keys_a, s->vals_a, s->hist, nblocks, parts);
    for (int pass = 0; pass < passes; pass++) {  // kernels of a pass the bounding box does not need return at once
        if (pass > 0) hipLaunchKernelGGL(radix_hist_kernel, nblocks, kThreads, 0, st, s->keys_a, s->keys_b, n, pass, s->hist, nblocks, s->dev);
        hipLaunchKernelGGL(radix_scatter_kernel, nblocks, kThreads, 0, st, s->keys_a, s->vals_a, s->keys_b, s->vals_b, n, pass, s->hist, nblocks,
                           s->dev);
    }
    hipLaunchKernelGGL(vg_count_heads_kernel, nblocks, kThreads, 0, st, s->keys_a, s->keys_b, n, s->dev, s->blockcnt);
    hipLaunchKernelGGL(vg_heads_kernel, nblocks, kThreads, 0, st, s->raw, s->keys_a, s->keys_b, s->vals_a, s->vals_b, n, s->dev, s->blockcnt,
                       s->hpos, s->sorted, s->ds_body, s->max_ds, s->host_nds_dev, (uint32_t)passes);
    const uint32_t vbound = n < s->max_ds ? n : s->max_ds;
    const uint32_t cblocks1 = (vbound + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(vg_centroid_kernel, cblocks1 < 2u * kCentroidGridCap ? cblocks1 : 2u * kCentroidGridCap, kThreads, 0, st, s->sorted, s->hpos, s->dev, s->ds_body,
                       s->longlist, s->max_ds);  // (strides over the scan's blocks)
    hipLaunchKernelGGL(vg_centroid_long_kernel, 256 + kMonsterBlocks, kThreads, 0, st, s->sorted, s->hpos, s->dev, s->ds_body, s->longlist, s->max_ds, 256u);
    LIO_HIP_TRY(hipGetLastError());
    return LIO_OK;
}

// the downsample chain for the scans of a batch: every launch serves all slots (grid.y), nothing is read back
int vg_downsample_batch(hipStream_t st, const SlotDesc* d_slots, int n_slots, uint32_t max_raw, uint32_t max_ds, float leaf, int passes) {
    const float inv = 1.0f / leaf;
    const uint32_t nblocks = (max_raw + kTile - 1) / kTile;  // of the largest scan of the batch
    if (nblocks == 0 || n_slots <= 0) return LIO_OK;
    const uint32_t B = (uint32_t)n_slots;
    hipLaunchKernelGGL(vg_bbox_batch, dim3(nblocks < 48 ? nblocks : 48, B), kThreads, 0, st, d_slots);
    hipLaunchKernelGGL(vg_keys_batch, dim3(nblocks, B), kThreads, 0, st, d_slots, inv);
    for (int pass = 0; pass < passes; pass++) {
        if (pass > 0) hipLaunchKernelGGL(radix_hist_batch, dim3(nblocks, B), kThreads, 0, st, d_slots, pass);
        hipLaunchKernelGGL(radix_prefix_batch, dim3(B), 256, 0, st, d_slots, pass);
        hipLaunchKernelGGL(radix_scatter_batch, dim3(nblocks, B), kThreads, 0, st, d_slots, pass);
    }
    hipLaunchKernelGGL(vg_count_heads_batch, dim3(nblocks, B), kThreads, 0, st, d_slots);
    hipLaunchKernelGGL(vg_heads_batch, dim3(nblocks, B), kThreads, 0, st, d_slots, (uint32_t)passes);
    const uint32_t vbound = max_raw < max_ds ? max_raw : max_ds;
    const uint32_t cblocks = (vbound + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(vg_centroid_batch, dim3(cblocks < kCentroidGridCap ? cblocks : kCentroidGridCap, B), kThreads, 0, st, d_slots);  // (strides over the scan's blocks)
    static const bool split = []() { const char* e = getenv("LIO_VG_MONSTER_SPLIT"); return e && e[0] == '1'; }();
    if (split) {
        hipLaunchKernelGGL(vg_centroid_long_batch, dim3(64, B), kThreads, 0, st, d_slots);
        hipLaunchKernelGGL(vg_centroid_monster_batch, dim3(kMonsterBlocksBatch, B), kThreads, 0, st, d_slots);
    } else {
        hipLaunchKernelGGL(vg_centroid_both_batch, dim3(64 + kMonsterBlocksBatch, B), kThreads, 0, st, d_slots);
    }
    hipLaunchKernelGGL(scan_begin_batch, dim3(16, B), 256, 0, st, d_slots);
    LIO_HIP_TRY(hipGetLastError());
    return LIO_OK;
}

// Nearest_Points.resize / forget for the further sub-map rows of a joint batch (their buffers receive the slot's downsampled cloud from
// share_ds_batch instead of running the chain themselves)
int scan_begin_rows(hipStream_t st, const SlotDesc* d_descs, int n_rows) {
    if (n_rows <= 0) return LIO_OK;
    hipLaunchKernelGGL(scan_begin_batch, dim3(16, (uint32_t)n_rows), 256, 0, st, d_descs);
    LIO_HIP_TRY(hipGetLastError());
    return LIO_OK;
}

int scan_begin(lio_scan* s) {
    hipLaunchKernelGGL(scan_begin_kernel, 64, 256, 0, s->stream, s->dev, s->nn_cnt, s->resize_min);
    LIO_HIP_TRY(hipGetLastError());
    return LIO_OK;
}

int scan_set_nds(lio_scan* s, uint32_t n) {
    hipLaunchKernelGGL(scan_set_nds_kernel, 1, 1, 0, s->stream, s->dev, n);
    LIO_HIP_TRY(hipGetLastError());
    return LIO_OK;
}

}  // namespace lio
