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
//   bbox (wave shuffle + ordered-uint atomics) -> keys -> 4 x {hist, scan, stable scatter} (8-bit LSD radix,
//   passes above the significant key bits degrade to a copy) -> head count -> ballot/prefix-sum compaction
//   fused with the centroid walk.
#include "lio_common.h"

namespace lio {

constexpr int kThreads = 256;
constexpr int kItems = 4;
constexpr int kTile = kThreads * kItems;  // 1024 keys per workgroup

__device__ inline uint32_t f2ord(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u ^ 0x80000000u) : ~u); }

__global__ void __launch_bounds__(kThreads) vg_bbox_kernel(const float4* __restrict__ in, uint32_t n, ScanDev* sd) {
    float mn0 = INFINITY, mn1 = INFINITY, mn2 = INFINITY, mx0 = -INFINITY, mx1 = -INFINITY, mx2 = -INFINITY;
    uint32_t cnt = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 p = in[i];
        if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
            mn0 = fminf(mn0, p.x); mx0 = fmaxf(mx0, p.x);
            mn1 = fminf(mn1, p.y); mx1 = fmaxf(mx1, p.y);
            mn2 = fminf(mn2, p.z); mx2 = fmaxf(mx2, p.z);
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
    if ((threadIdx.x & 63) == 0 && cnt > 0) {
        atomicMin(&sd->bbox_min[0], f2ord(mn0)); atomicMax(&sd->bbox_max[0], f2ord(mx0));
        atomicMin(&sd->bbox_min[1], f2ord(mn1)); atomicMax(&sd->bbox_max[1], f2ord(mx1));
        atomicMin(&sd->bbox_min[2], f2ord(mn2)); atomicMax(&sd->bbox_max[2], f2ord(mx2));
        atomicAdd(&sd->n_valid, cnt);
    }
}

struct VgGrid {
    int minb[3];
    int mul1, mul2;
    uint32_t total;
    uint32_t pass;
};

__device__ inline VgGrid vg_derive(const ScanDev* sd, float inv) {
    VgGrid g;
    g.pass = 0;
    if (sd->n_valid == 0) {
        g.minb[0] = g.minb[1] = g.minb[2] = 0;
        g.mul1 = g.mul2 = 0;
        g.total = 0;
        return g;
    }
    float mn[3], mx[3];
    for (int a = 0; a < 3; a++) { mn[a] = ord2f(sd->bbox_min[a]); mx[a] = ord2f(sd->bbox_max[a]); }
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

__global__ void __launch_bounds__(kThreads) vg_keys_kernel(const float4* __restrict__ in, uint32_t n, float inv, ScanDev* sd,
                                                           uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const VgGrid g = vg_derive(sd, inv);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sd->n_ds_prev = sd->n_ds;  // still the previous scan's size: the centroid kernel runs later
        sd->passthrough = g.pass;
        sd->total_cells = g.total;
        sd->nbits = g.total ? (32 - __clz(g.total)) : 0;  // keys 0..total (total = invalid marker) need bits(total)
        sd->minb[0] = g.minb[0]; sd->minb[1] = g.minb[1]; sd->minb[2] = g.minb[2];
        sd->mul1 = g.mul1; sd->mul2 = g.mul2;
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 p = in[i];
        uint32_t key = g.total;  // invalid marker sorts behind every occupied voxel
        if (g.total && isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
            const int i0 = (int)(floorf(p.x * inv) - (float)g.minb[0]);
            const int i1 = (int)(floorf(p.y * inv) - (float)g.minb[1]);
            const int i2 = (int)(floorf(p.z * inv) - (float)g.minb[2]);
            key = (uint32_t)(i0 + i1 * g.mul1 + i2 * g.mul2);
        }
        keys[i] = key;
        vals[i] = i;
    }
}

// ---- stable LSD radix sort, 8 bits per pass ---------------------------------------------------------
__global__ void __launch_bounds__(kThreads) radix_hist_kernel(const uint32_t* __restrict__ keys, uint32_t n, int shift,
                                                              uint32_t* __restrict__ hist, uint32_t nblocks, const ScanDev* sd) {
    if ((uint32_t)shift >= sd->nbits) return;
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kTile;
#pragma unroll
    for (int r = 0; r < kItems; r++) {
        const uint32_t i = base + r * kThreads + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

__global__ void __launch_bounds__(256) radix_scan_kernel(uint32_t* __restrict__ hist, uint32_t nblocks, int shift, const ScanDev* sd) {
    if ((uint32_t)shift >= sd->nbits) return;
    __shared__ uint32_t s[256];
    const int d = threadIdx.x;
    uint32_t* row = hist + (size_t)d * nblocks;
    uint32_t sum = 0;
    for (uint32_t b = 0; b < nblocks; b++) sum += row[b];
    s[d] = sum;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const uint32_t t = (d >= off) ? s[d - off] : 0u;
        __syncthreads();
        s[d] += t;
        __syncthreads();
    }
    uint32_t run = s[d] - sum;
    for (uint32_t b = 0; b < nblocks; b++) {
        const uint32_t t = row[b];
        row[b] = run;
        run += t;
    }
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

__global__ void __launch_bounds__(kThreads) radix_scatter_kernel(const uint32_t* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                                 uint32_t* __restrict__ kout, uint32_t* __restrict__ vout, uint32_t n,
                                                                 int shift, const uint32_t* __restrict__ hist, uint32_t nblocks,
                                                                 const ScanDev* sd) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if ((uint32_t)shift >= sd->nbits) {  // pass above the significant bits: keep the ping-pong parity, copy through
        const uint32_t base = blockIdx.x * kTile;
#pragma unroll
        for (int r = 0; r < kItems; r++) {
            const uint32_t i = base + r * kThreads + tid;
            if (i < n) { kout[i] = kin[i]; vout[i] = vin[i]; }
        }
        return;
    }
    __shared__ uint32_t wcnt[kThreads / 64][256];
    // each wave owns a contiguous run of 64*kItems keys so that (round, lane) order == input order
    const uint32_t base = blockIdx.x * kTile + wave * (64 * kItems);
    uint32_t k[kItems], v[kItems];
    bool ok[kItems];
#pragma unroll
    for (int r = 0; r < kItems; r++) {
        const uint32_t i = base + r * 64 + lane;
        ok[r] = i < n;
        k[r] = ok[r] ? kin[i] : 0u;
        v[r] = ok[r] ? vin[i] : 0u;
    }
    for (int j = tid; j < (kThreads / 64) * 256; j += kThreads) (&wcnt[0][0])[j] = 0;
    __syncthreads();
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int r = 0; r < kItems; r++) {
        const uint32_t d = (k[r] >> shift) & 255u;
        const unsigned long long peers = match_digit(d, ok[r]);
        if (ok[r] && (peers & lt) == 0) wcnt[wave][d] += __popcll(peers);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    {
        uint32_t g = hist[tid * nblocks + blockIdx.x];
#pragma unroll
        for (int w = 0; w < kThreads / 64; w++) {
            const uint32_t t = wcnt[w][tid];
            wcnt[w][tid] = g;
            g += t;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kItems; r++) {
        const uint32_t d = (k[r] >> shift) & 255u;
        const unsigned long long peers = match_digit(d, ok[r]);
        uint32_t pos = 0;
        if (ok[r]) pos = wcnt[wave][d] + __popcll(peers & lt);
        __builtin_amdgcn_wave_barrier();
        if (ok[r] && (peers & lt) == 0) wcnt[wave][d] += __popcll(peers);
        __builtin_amdgcn_wave_barrier();
        if (ok[r]) { kout[pos] = k[r]; vout[pos] = v[r]; }
    }
}

// ---- voxel heads: occupancy flags -> ballot + prefix-sum compaction -> centroid ------------------------
__global__ void __launch_bounds__(kThreads) vg_count_heads_kernel(const uint32_t* __restrict__ keys, uint32_t n, const ScanDev* sd,
                                                                  uint32_t* __restrict__ blockcnt) {
    __shared__ uint32_t c;
    if (threadIdx.x == 0) c = 0;
    __syncthreads();
    const uint32_t total = sd->total_cells;
    const uint32_t base = blockIdx.x * kTile;
    uint32_t mine = 0;
#pragma unroll
    for (int r = 0; r < kItems; r++) {
        const uint32_t i = base + r * kThreads + threadIdx.x;
        bool head = false;
        if (i < n) {
            const uint32_t key = keys[i];
            head = key < total && (i == 0 || keys[i - 1] != key);
        }
        mine += __popcll(__ballot(head));
    }
    if ((threadIdx.x & 63) == 0) atomicAdd(&c, mine);
    __syncthreads();
    if (threadIdx.x == 0) blockcnt[blockIdx.x] = c;
}

__global__ void __launch_bounds__(kThreads) vg_centroid_kernel(const float4* __restrict__ in, const uint32_t* __restrict__ keys,
                                                               const uint32_t* __restrict__ vals, uint32_t n, ScanDev* sd,
                                                               const uint32_t* __restrict__ blockcnt, float4* __restrict__ out,
                                                               uint32_t max_ds) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (sd->passthrough) {  // PCL overflow guard: output = input
        if (n > max_ds) {
            if (blockIdx.x == 0 && tid == 0) { sd->err |= 1u; sd->n_ds = 0; }
            return;
        }
        const uint32_t base = blockIdx.x * kTile;
        for (int r = 0; r < kItems; r++) {
            const uint32_t i = base + r * kThreads + tid;
            if (i < n) out[i] = in[i];
        }
        if (blockIdx.x == 0 && tid == 0) sd->n_ds = n;
        return;
    }
    __shared__ uint32_t red[kThreads / 64];
    __shared__ uint32_t wtot[kThreads / 64];
    // exclusive prefix of the tiles before this one (fixed order -> deterministic output slots)
    uint32_t pre = 0;
    for (uint32_t b = tid; b < blockIdx.x; b += kThreads) pre += blockcnt[b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) pre += __shfl_xor(pre, off);
    if (lane == 0) red[wave] = pre;
    __syncthreads();
    uint32_t run = 0;
    for (int w = 0; w < kThreads / 64; w++) run += red[w];
    __syncthreads();

    const uint32_t total = sd->total_cells;
    const uint32_t base = blockIdx.x * kTile;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int r = 0; r < kItems; r++) {
        const uint32_t i = base + r * kThreads + tid;
        bool head = false;
        uint32_t key = 0;
        if (i < n) {
            key = keys[i];
            head = key < total && (i == 0 || keys[i - 1] != key);
        }
        const unsigned long long m = __ballot(head);
        if (lane == 0) wtot[wave] = __popcll(m);
        __syncthreads();
        uint32_t woff = 0, rtot = 0;
        for (int w = 0; w < kThreads / 64; w++) {
            const uint32_t t = wtot[w];
            if (w < wave) woff += t;
            rtot += t;
        }
        if (head) {
            const uint32_t slot = run + woff + __popcll(m & lt);
            float sx = 0.f, sy = 0.f, sz = 0.f, sw = 0.f;
            uint32_t j = i;
            do {
                const float4 p = in[vals[j]];
                sx = sx + p.x; sy = sy + p.y; sz = sz + p.z; sw = sw + p.w;
                j++;
            } while (j < n && keys[j] == key);
            const float c = (float)(j - i);
            if (slot < max_ds) out[slot] = make_float4(sx / c, sy / c, sz / c, sw / c);
        }
        run += rtot;
        __syncthreads();
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) {
        if (run > max_ds) { sd->err |= 1u; run = 0; }
        sd->n_ds = run;
    }
}

__global__ void scan_set_nds_kernel(ScanDev* sd, uint32_t n) {
    sd->n_ds_prev = sd->n_ds;
    sd->n_ds = n;
}

// Nearest_Points.resize(feats_down_size) (laserMapping.cpp:1274): entries beyond the new size are destroyed
__global__ void scan_begin_kernel(const ScanDev* sd, int32_t* __restrict__ nn_cnt) {
    const uint32_t lo = sd->n_ds, hi = sd->n_ds_prev;
    for (uint32_t i = lo + blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += gridDim.x * blockDim.x) nn_cnt[i] = 0;
}

int vg_downsample(lio_scan* s, float leaf) {
    const uint32_t n = s->n_raw;
    const float inv = 1.0f / leaf;
    hipStream_t st = s->stream;
    LIO_HIP_TRY(hipMemsetAsync(s->dev->bbox_min, 0xFF, 12, st));
    LIO_HIP_TRY(hipMemsetAsync(s->dev->bbox_max, 0, 16, st));  // bbox_max[3] + n_valid
    const uint32_t nblocks = (n + kTile - 1) / kTile;
    if (n == 0) {
        hipLaunchKernelGGL(scan_set_nds_kernel, 1, 1, 0, st, s->dev, 0u);
        return LIO_OK;
    }
    const uint32_t g1 = nblocks < 1024 ? nblocks : 1024;
    hipLaunchKernelGGL(vg_bbox_kernel, g1, kThreads, 0, st, s->raw, n, s->dev);
    hipLaunchKernelGGL(vg_keys_kernel, nblocks, kThreads, 0, st, s->raw, n, inv, s->dev, s->keys_a, s->vals_a);
    uint32_t *ka = s->keys_a, *kb = s->keys_b, *va = s->vals_a, *vb = s->vals_b;
    for (int pass = 0; pass < 4; pass++) {
        const int shift = pass * 8;
        hipLaunchKernelGGL(radix_hist_kernel, nblocks, kThreads, 0, st, ka, n, shift, s->hist, nblocks, s->dev);
        hipLaunchKernelGGL(radix_scan_kernel, 1, 256, 0, st, s->hist, nblocks, shift, s->dev);
        hipLaunchKernelGGL(radix_scatter_kernel, nblocks, kThreads, 0, st, ka, va, kb, vb, n, shift, s->hist, nblocks, s->dev);
        uint32_t* t = ka; ka = kb; kb = t;
        t = va; va = vb; vb = t;
    }
    hipLaunchKernelGGL(vg_count_heads_kernel, nblocks, kThreads, 0, st, ka, n, s->dev, s->blockcnt);
    hipLaunchKernelGGL(vg_centroid_kernel, nblocks, kThreads, 0, st, s->raw, ka, va, n, s->dev, s->blockcnt, s->ds_body, s->max_ds);
    LIO_HIP_TRY(hipGetLastError());
    return LIO_OK;
}

int scan_begin(lio_scan* s) {
    hipLaunchKernelGGL(scan_begin_kernel, 64, 256, 0, s->stream, s->dev, s->nn_cnt);
    LIO_HIP_TRY(hipGetLastError());
    return LIO_OK;
}

int scan_set_nds(lio_scan* s, uint32_t n) {
    hipLaunchKernelGGL(scan_set_nds_kernel, 1, 1, 0, s->stream, s->dev, n);
    LIO_HIP_TRY(hipGetLastError());
    return LIO_OK;
}

}  // namespace lio
