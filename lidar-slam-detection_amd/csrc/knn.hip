// knn.hip -- stencil kNN on the voxel hash grid: IVox::GetClosestPoint(pt, out, 5, 5.0)
// (/root/reference/slam/mapping/fastlio/include/ivox3d/ivox3d.h:139-171, ivox3d_node.hpp:107-127).
//
// Semantics: the 5 nearest of all points stored in the stencil voxels with d^2 < 5.0 (the per-voxel
// nth_element of the reference is a pruning step that does not change that set), element order = the canonical
// total order (d2, x, y, z) that oracle/lio_oracle.cpp uses; fewer than 5 -> all of them; none -> the output
// is left untouched (stale neighbours survive, ivox3d.h:152-154).
//
// Mapping to the machine: G = 16 lanes (a quarter wave) work on one query, four queries per wave (G = 32 costs the same alone
// and ~8 % more GPU time with several scans in flight: fewer lanes idle on voxels of ~40 points).
//   1. each lane probes stencil cells (one 16-B slot load each; two rounds for NEARBY18, five for the 75-cell start-up
//      stencil); for stencils of at most 2 G cells the hits are compacted into LDS in three buckets of a conservative lower
//      bound of the voxel's distance to the query (ballot + popcount; group sums by DPP row rotation);
//   2. voxel-major sweep, nearest bucket first: four voxel descriptors at a time come back from LDS as ds_read_b128, lane l
//      takes point l (l+G, ...) of each of the four voxels -- four coalesced 16-B loads in flight, addresses are
//      base + lane -- and keeps its own sorted top-5 as (d2 bits, pool index) pairs: branch-free insertion, five
//      independent compares, v_min / v_med3 for the distances and two selects per slot for the indices.  After the first
//      batch the group bounds its fifth-nearest distance from what the lanes hold and skips every voxel whose lower bound is
//      strictly above it (exact: sets, ties and in-range counts are unchanged), stopping at the first bucket whose floor is;
//   3. six rounds of group-wide (d2, index) minima (DPP row rotations, no LDS crossbar) pop the global top-5 and the best loser.
// Exact d2 ties between different points are the only case where (d2, index) order can differ from the
// canonical (d2, x, y, z) order.  They are detected on the sorted top-6 and the query is queued for
// knn_exact_kernel, which redoes it with the full comparison (rare: ~1e-6 per query on float data).
#include "hashgrid.h"
#include "knn_dev.h"
#include "lio_common.h"
#include "refsel.h"

namespace lio {

#ifndef LIO_KNN_G
#define LIO_KNN_G 16
#endif
#ifndef LIO_KNN_U
#define LIO_KNN_U 4
#endif
#ifndef LIO_KNN_WAVES
#define LIO_KNN_WAVES 6  // <= 80 registers, six waves per SIMD, nothing spills (seven: 72 registers with 9-12 spilled -- 4 % slower once the kernel became VALU bound)
#endif
#ifndef LIO_KNN_PRUNE
#define LIO_KNN_PRUNE 1  // distance-ordered sweep with exact pruning of stencil voxels that cannot hold one of the five nearest
#endif
#ifndef LIO_KNN_STAT
#define LIO_KNN_STAT 1  // 0: the timed kernels do not keep the candidate statistic (MapDev::knn_cand word 0; the counting variant always does): 6.82 -> 6.75 us per scan and search, not taken -- bench.py's legs read the statistic of their timed launches
#endif
#ifndef LIO_KNN_WAVES_BATCH
#define LIO_KNN_WAVES_BATCH 7  // the batched / sequence kernels: 72 registers, nothing spills (round 6: the statistic in a scalar register, two cold values formed where they are used)
#endif
#ifdef LIO_KNN_MAXWAVES  // experiment: CAP the kernel's occupancy (leaves registers / issue slots of every SIMD to the kernels of the other rounds in flight)
#define LIO_KNN_OCC __attribute__((amdgpu_waves_per_eu(LIO_KNN_MAXWAVES, LIO_KNN_MAXWAVES)))
#else
#define LIO_KNN_OCC
#endif
constexpr int kG = LIO_KNN_G;          // lanes per query
constexpr int kU = LIO_KNN_U;          // voxels swept together (loads in flight per lane), multiple of 4
constexpr int kGPB = 256 / kG;         // queries per workgroup

// Reductions over the kG lanes of a query group.  With kG = 16 a group is exactly one DPP row: four v_min / v_add with a row-rotate
// modifier (row_ror:8, 4, 2, 1) leave the result in every lane, no LDS crossbar (ds_bpermute) round trips -- the kernel used to issue
// ~100 of those per query, each a dependent ~100-cycle wait.
template <int CTRL>
__device__ inline uint32_t dpp_row(uint32_t v) {
    // old = 0 with bound_ctrl: every lane of a row rotation / quad permutation has a source lane, so `old` is never used -- and in this form the
    // compiler folds the move into the consuming v_min / v_max / v_add (v_min_u32_dpp ...) instead of a v_mov_b32_dpp + s_nop + op
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
__device__ inline uint32_t group_min32(uint32_t v) {
    if constexpr (kG == 8) {  // half a DPP row: two quad permutes and the half-row mirror
        v = min(v, dpp_row<0xB1>(v));
        v = min(v, dpp_row<0x4E>(v));
        v = min(v, dpp_row<0x141>(v));
    } else if constexpr (kG == 16) {
        v = min(v, dpp_row<0x128>(v));
        v = min(v, dpp_row<0x124>(v));
        v = min(v, dpp_row<0x122>(v));
        v = min(v, dpp_row<0x121>(v));
    } else {
#pragma unroll
        for (int off = kG / 2; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, kG));
    }
    return v;
}
// the ballot of a predicate over the kG lanes of this lane's group, as a kG-bit word (bit i = lane i of the group): one select between
// the two halves of the wave-wide ballot and a bit-field extract -- the 64-bit and / popcount of the wave-wide form cost twice that
__device__ inline uint32_t group_ballot(bool p, int lane) {
    const unsigned long long b = __ballot(p);
    if constexpr (kG == 16 || kG == 8) {
        const uint32_t half = (lane & 32) ? (uint32_t)(b >> 32) : (uint32_t)b;
        return (half >> (lane & (32 - kG) & 31)) & ((1u << kG) - 1u);
    } else {
        return (uint32_t)((b >> (lane & ~(kG - 1))) & ((1ull << kG) - 1ull));
    }
}
__device__ inline uint32_t group_max32(uint32_t v) {
    if constexpr (kG == 8) {  // half a DPP row: two quad permutes and the half-row mirror
        v = max(v, dpp_row<0xB1>(v));
        v = max(v, dpp_row<0x4E>(v));
        v = max(v, dpp_row<0x141>(v));
    } else if constexpr (kG == 16) {
        v = max(v, dpp_row<0x128>(v));
        v = max(v, dpp_row<0x124>(v));
        v = max(v, dpp_row<0x122>(v));
        v = max(v, dpp_row<0x121>(v));
    } else {
#pragma unroll
        for (int off = kG / 2; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, kG));
    }
    return v;
}
__device__ inline uint32_t group_sum32(uint32_t v) {
    if constexpr (kG == 8) {
        v += dpp_row<0xB1>(v);
        v += dpp_row<0x4E>(v);
        v += dpp_row<0x141>(v);
    } else if constexpr (kG == 16) {
        v += dpp_row<0x128>(v);
        v += dpp_row<0x124>(v);
        v += dpp_row<0x122>(v);
        v += dpp_row<0x121>(v);
    } else {
#pragma unroll
        for (int off = kG / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kG);
    }
    return v;
}

// the occupied voxels of one query's stencil, compacted: 16-B aligned so that four descriptors come back from one
// ds_read_b128; entries nhit .. nhit+3 are zero-filled (count 0) so that batches of four need no bounds test
#ifndef LIO_KNN_LIST_CAP
#define LIO_KNN_LIST_CAP kMaxStencil
#endif
struct __attribute__((aligned(16))) GroupLds {
    uint32_t v_ptr[LIO_KNN_LIST_CAP + kU + 1];
    uint32_t v_cnt[LIO_KNN_LIST_CAP + kU + 1];
    uint32_t v_dmin[LIO_KNN_LIST_CAP + kU + 1];  // bits of a lower bound of the squared distance from the query to any point of the voxel
};

// probe_stencil for stencils of at most 2 * kG cells (NEARBY6 / 18 / 26) with the hits BUCKETED by that lower bound: bucket 0 below
// (res / 4)^2, bucket 1 below (res / 2)^2, bucket 2 the rest.  On return the list in g is bucket 0, then 1, then 2 (order inside a
// bucket = stencil order), n0 / n01 are the list positions where buckets 1 and 2 start.
template <int KM>
__device__ inline uint32_t probe_stencil_bucketed(const Slot* __restrict__ table, uint32_t mask, const StencilArgs& st, bool active, float4 pw,
                                                  float res, int kx, int ky, int kz, int gl, int lane, unsigned long long gmask, GroupLds& g,
                                                  uint32_t& nhit_out, uint32_t& n0_out, uint32_t& n01_out, uint32_t limit) {
    static_assert(KM * kG <= 32, "bucketed probe: stencils of at most 32 cells");
    uint4 raw[KM];
    BrickProbe bp[KM];
    unsigned long long want[KM];
    uint32_t dmin[KM];
    // Three groups of UNCONDITIONAL loads, each issued together and waited for once: the lane's stencil offsets (a lane-indexed read of the
    // kernel argument: vector memory), then the home slots at clamped indices (a lane without a cell, or whose cell cannot reach `limit`,
    // reads slot 0 and ignores it).  Written as `if (active && s < st.n) { ... raw[k] = table[...]; }` the compiler emitted offset load, wait,
    // hash, slot load, WAIT for every k in turn (round 4, read off the ISA): 2 KM dependent memory round trips per query instead of two.
    int ox[KM], oy[KM], oz[KM];
    bool inb[KM];
#pragma unroll
    for (int k = 0; k < KM; k++) {
        const int s = k * kG + gl;
        inb[k] = active && s < st.n;
        const int sc = s < st.n ? s : 0;
        ox[k] = st.off[sc][0]; oy[k] = st.off[sc][1]; oz[k] = st.off[sc][2];
    }
    uint32_t slot[KM];
#pragma unroll
    for (int k = 0; k < KM; k++) {
        const int cx = kx + ox[k], cy = ky + oy[k], cz = kz + oz[k];
        dmin[k] = inb[k] ? cell_min_d2_bits(pw.x, pw.y, pw.z, cx, cy, cz, res) : 0xFFFFFFFFu;
        const bool go = inb[k] && dmin[k] <= limit;  // `limit`: a known upper bound of the fifth-nearest distance -- a voxel that cannot reach it is not even probed
        want[k] = go ? pack_key(cx, cy, cz) : kEmptyKey;
        bp[k] = brick_probe(cx, cy, cz);
        slot[k] = go ? brick_slot(bp[k], mask) : 0u;
    }
#pragma unroll
    for (int k = 0; k < KM; k++) raw[k] = *reinterpret_cast<const uint4*>(&table[slot[k]]);
#pragma unroll
    for (int k = 0; k < KM; k++) pin_loaded(raw[k]);
    const uint32_t b1 = __float_as_uint(0.0625f * res * res), b2 = __float_as_uint(0.25f * res * res);
    const uint32_t below = (1u << gl) - 1u;
    uint32_t ptr[KM], cnt[KM], total = 0;
    int bucket[KM];
    uint32_t mb[KM][3];
#pragma unroll
    for (int k = 0; k < KM; k++) {
        ptr[k] = 0; cnt[k] = 0;
        if (want[k] != kEmptyKey) {
            uint4 r = raw[k];
            for (uint32_t probe = 0; probe <= (mask >> 6); probe++) {
                const unsigned long long kk = ((unsigned long long)r.y << 32) | r.x;
                if (kk == want[k]) { ptr[k] = r.z; cnt[k] = r.w; break; }
                if (kk == kEmptyKey) break;
                brick_next(bp[k]);
                r = *reinterpret_cast<const uint4*>(&table[brick_slot(bp[k], mask)]);
            }
        }
        const bool hit = cnt[k] > 0;
        bucket[k] = dmin[k] < b1 ? 0 : (dmin[k] < b2 ? 1 : 2);
#pragma unroll
        for (int b = 0; b < 3; b++) mb[k][b] = group_ballot(hit && bucket[k] == b, lane);
        total += cnt[k];  // (this LANE's share of the candidate statistic: the lanes' shares are summed once per workgroup at the end -- a group sum here was eight DPP adds per four queries for a diagnostic counter)
    }
    uint32_t nb[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < KM; k++)
#pragma unroll
        for (int b = 0; b < 3; b++) nb[b] += __popc(mb[k][b]);
    const uint32_t base[3] = {0, nb[0], nb[0] + nb[1]};
    const uint32_t nhit = nb[0] + nb[1] + nb[2];
#pragma unroll
    for (int k = 0; k < KM; k++) {
        if (cnt[k] > 0) {
            uint32_t at = base[bucket[k]];
#pragma unroll
            for (int kk = 0; kk < k; kk++) at += __popc(mb[kk][bucket[k]]);
            at += __popc(mb[k][bucket[k]] & below);
            g.v_ptr[at] = ptr[k];
            g.v_cnt[at] = cnt[k];
            g.v_dmin[at] = dmin[k];
        }
    }
    if (gl < kU) { g.v_ptr[nhit + gl] = 0; g.v_cnt[nhit + gl] = 0; g.v_dmin[nhit + gl] = 0xFFFFFFFFu; }
    nhit_out = nhit;
    n0_out = base[1];
    n01_out = base[2];
    return total;
}

// probe the stencil of the group's query; on return the hit voxels are compacted in g (ptr, cnt) and
// this lane's share of the candidate count is returned.  All lanes of the wave must call this (ballots inside).
template <int KM>
__device__ inline uint32_t probe_stencil(const Slot* __restrict__ table, uint32_t mask, const StencilArgs& st, bool active, int kx, int ky,
                                         int kz, int gl, int lane, unsigned long long gmask, GroupLds& g, uint32_t& nhit_out) {
    uint4 raw[KM];
    BrickProbe bp[KM];
    unsigned long long want[KM];
    // (unconditional loads at clamped indices, issued together: see probe_stencil_bucketed)
    int ox[KM], oy[KM], oz[KM];
    bool inb[KM];
#pragma unroll
    for (int k = 0; k < KM; k++) {
        const int s = k * kG + gl;
        inb[k] = active && s < st.n;
        const int sc = s < st.n ? s : 0;
        ox[k] = st.off[sc][0]; oy[k] = st.off[sc][1]; oz[k] = st.off[sc][2];
    }
    uint32_t slot[KM];
#pragma unroll
    for (int k = 0; k < KM; k++) {
        const int cx = kx + ox[k], cy = ky + oy[k], cz = kz + oz[k];
        want[k] = inb[k] ? pack_key(cx, cy, cz) : kEmptyKey;
        bp[k] = brick_probe(cx, cy, cz);
        slot[k] = inb[k] ? brick_slot(bp[k], mask) : 0u;
    }
#pragma unroll
    for (int k = 0; k < KM; k++) raw[k] = *reinterpret_cast<const uint4*>(&table[slot[k]]);
#pragma unroll
    for (int k = 0; k < KM; k++) pin_loaded(raw[k]);
    uint32_t nhit = 0, total = 0;
    const uint32_t below = (1u << gl) - 1u;
#pragma unroll
    for (int k = 0; k < KM; k++) {
        uint32_t ptr = 0, cnt = 0;
        if (want[k] != kEmptyKey) {
            uint4 r = raw[k];
            for (uint32_t probe = 0; probe <= (mask >> 6); probe++) {  // double hashing by window; load factor <= 0.5
                const unsigned long long kk = ((unsigned long long)r.y << 32) | r.x;
                if (kk == want[k]) { ptr = r.z; cnt = r.w; break; }
                if (kk == kEmptyKey) break;
                brick_next(bp[k]);
                r = *reinterpret_cast<const uint4*>(&table[brick_slot(bp[k], mask)]);
            }
        }
        const bool hit = cnt > 0;
        const uint32_t m = group_ballot(hit, lane);
        if (hit) {
            const uint32_t at = nhit + __popc(m & below);
            g.v_ptr[at] = ptr;
            g.v_cnt[at] = cnt;
        }
        nhit += __popc(m);
        total += cnt;  // (this lane's share: see probe_stencil_bucketed)
    }
    if (gl < kU) { g.v_ptr[nhit + gl] = 0; g.v_cnt[nhit + gl] = 0; }
    nhit_out = nhit;
    return total;
}

// The LDS staging area of a query group is written and read by lanes of ONE wave only: ordering within the wave is all that
// is needed (LDS operations of a wave complete in order), no workgroup barrier -- the four waves of a workgroup never wait
// for each other.
__device__ inline void group_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ inline uint32_t med3_u32(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// An upper bound of the squared distance of the query's fifth nearest candidate, from what the lanes hold so far: any lane's own fifth
// (its five entries are five candidates at most that far), and the fifth smallest of the lanes' nearest (five candidates in five
// lanes; equal values are counted once, which can only loosen the bound).  0xFFFFFFFF while fewer than five candidates are known.
#ifndef LIO_KNN_BOUND
#define LIO_KNN_BOUND 5
#endif
__device__ inline uint32_t fifth_bound(uint32_t d0, uint32_t d2, uint32_t d4) {
    uint32_t t = group_min32(d4);
#if LIO_KNN_BOUND == 2
    // a cheaper (looser, equally valid) form: two lanes whose THIRD entries are both <= X hold six candidates <= X -- the second smallest of the
    // lanes' thirds (equal values counted once) -- three group reductions instead of six
    (void)d0;
    const uint32_t m1 = group_min32(d2);
    return min(t, group_min32(d2 == m1 ? 0xFFFFFFFFu : d2));
#endif
    (void)d2;
    uint32_t v = d0, m = 0xFFFFFFFFu;
#pragma unroll
    for (int r = 0; r < 5; r++) {
        m = group_min32(v);
        if (v == m) v = 0xFFFFFFFFu;
    }
    return min(t, m);
}

struct Cand {
    float d2;
    uint32_t id;
};

// strict total order (d2, x, y, z); the coordinate comparison only runs on exact d2 ties
__device__ __noinline__ bool cand_tie_less(const Cand& a, const Cand& b, const float4* __restrict__ pool) {
    if (a.id == b.id) return false;
    if (a.id == kNoIdx || b.id == kNoIdx) return a.id < b.id;
    const float4 pa = pool[a.id], pb = pool[b.id];
    if (pa.x != pb.x) return pa.x < pb.x;
    if (pa.y != pb.y) return pa.y < pb.y;
    if (pa.z != pb.z) return pa.z < pb.z;
    return a.id < b.id;
}
__device__ inline bool cand_less(const Cand& a, const Cand& b, const float4* __restrict__ pool) {
    if (a.d2 != b.d2) return a.d2 < b.d2;
    return cand_tie_less(a, b, pool);
}


// MODE 0: queries are body-frame ds points of a scan (transformed here, world point stored);
// MODE 1: queries are world-frame points (diagnostic lio_map_knn).
// COUNT: a diagnostic build of the same kernel that also counts the candidate points the sweep really loads (`touched`, the second word
// of a shard of MapDev::knn_cand) beside the stencil's residents -- bench.py's roofline leg wants both; never the timed variant.
// The body -> world transform's fourteen doubles live in LDS (pose_s), not in scalar registers: held there for the whole kernel they were 28 of its
// 106 SGPRs -- 19 spilled into lanes of a VGPR, two VGPRs to scratch, re-read in every query's epilogue (round 5's -Rpass-analysis) -- and they are
// used by one wave in four, once per 64 queries.  `fill(pose_s)` is called by every thread once, before the first barrier.
template <int KM, int MODE, bool COUNT = false, class PoseFill>
__device__ __forceinline__ void knn_body(const Slot* __restrict__ table, uint32_t mask, const float4* __restrict__ pool,
                                         float inv_res, const StencilArgs& st, PoseFill fill, const float4* __restrict__ queries,
                                         uint32_t n_host, const ScanDev* __restrict__ sd, float4* __restrict__ world_out,
                                         float4* __restrict__ nn_pts, uint32_t nn_stride, int32_t* __restrict__ nn_cnt,
                                         MapDev* md, uint32_t* __restrict__ n_tie, uint32_t* __restrict__ tie_list) {
    __shared__ GroupLds lds[kGPB];
    const int tid = threadIdx.x;
    const int grp = tid / kG, gl = tid % kG;
    const int lane = tid & 63;
    GroupLds& g = lds[grp];
    const uint32_t n = sd ? sd->n_ds : n_host;
    const unsigned long long gmask = ((1ull << kG) - 1ull) << (lane - gl);
    __shared__ uint32_t stat_s[256 / 64];  // the waves' shares of the candidate statistic (a few hundred queries' stencils: far below 2^32)
    if (tid < 256 / 64) stat_s[tid] = 0;
    uint32_t touched = 0;  // COUNT only: candidate points whose 16 bytes the sweep asked for
    uint32_t fresh = 0;    // COUNT only: ... of those, the ones no query of this launch had asked for before (distinct points: MapDev::touch_bits)
    uint32_t* const bits = COUNT ? md->touch_bits : nullptr;
    const float res = 1.0f / inv_res;
    const uint32_t b1_bits = __float_as_uint(0.0625f * res * res), b2_bits = __float_as_uint(0.25f * res * res);
    (void)b1_bits; (void)b2_bits;

    // XCD-aware workgroup -> query mapping: the dispatcher deals workgroups round-robin to the 8 XCDs (b % 8), each
    // with its own L2.  Consecutive queries are spatial neighbours that share most of their candidate voxels, so
    // every XCD gets one CONTIGUOUS eighth of the queries (gridDim.x is a multiple of 8): the shared voxels are then
    // fetched into one L2 instead of up to eight.  Placement only affects speed, never results.
    // The eighths are eighths of the QUERIES, not of the grid: a launch sized for the largest scan it may meet (the batched form) must
    // not leave the XCDs with the high block numbers idle.
    const uint32_t nblk = (n + kGPB - 1) / kGPB;       // query blocks of this scan
    const uint32_t per_xcd = (nblk + 7u) >> 3;         // ... per XCD, contiguous
    const uint32_t wg_per_xcd = gridDim.x >> 3;        // workgroups of this launch on one XCD: they stride over its share
    // The body -> world transform (two quaternion rotations in f64, ~140 issue slots) is the same for the sixteen lanes of a query: the
    // first wave does it once, one lane per query, for the next four query blocks of this workgroup (64 queries), and leaves the world
    // points in LDS (and in world_out, coalesced).
    constexpr uint32_t kAhead = 64 / kGPB;
    __shared__ float4 pw_s[64];
    __shared__ PoseArgs pose_s;
    if constexpr (MODE == 0) fill(pose_s);
    uint32_t ahead = kAhead;  // workgroup-uniform: query blocks already taken from the staged chunk
    for (uint32_t j = blockIdx.x >> 3; j < per_xcd; j += wg_per_xcd) {
        const uint32_t q0 = ((blockIdx.x & 7u) * per_xcd + j) * kGPB;
        const uint32_t q = q0 + grp;
        const bool active = q < n;
        float4 pw = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (MODE == 0) {
            if (ahead == kAhead) {
                __syncthreads();
                if (tid < 64) {
                    const uint32_t j2 = j + (uint32_t)(tid / kGPB) * wg_per_xcd;
                    const uint32_t q2 = ((blockIdx.x & 7u) * per_xcd + j2) * kGPB + (uint32_t)(tid % kGPB);
                    if (j2 < per_xcd && q2 < n) {
                        float4 w2;
                        body_to_world(pose_s, queries[q2], w2);
                        world_out[q2] = w2;
                        uint32_t t2 = (uint32_t)tid;
                        asm volatile("" : "+v"(t2));  // (the LDS address is formed here, once per 64 queries: hoisted out of the query loop it was one of the two values spilled to scratch at seven waves per SIMD)
                        pw_s[t2] = w2;
                    }
                }
                __syncthreads();
                ahead = 0;
            }
            if (active) pw = pw_s[ahead * kGPB + grp];
            ahead++;
        } else {
            if (active) pw = queries[q];
        }
        int kx = 0, ky = 0, kz = 0;
        pos2grid(pw.x, pw.y, pw.z, inv_res, kx, ky, kz);
        uint32_t nhit = 0;
        constexpr bool kPrune = LIO_KNN_PRUNE && KM * kG <= 32;
        uint32_t n0 = 0, n01 = 0, total;
        const uint32_t limit = 0xFFFFFFFFu;  // (an upper bound of the fifth-nearest distance known before the probe: none)
        if constexpr (kPrune) total = probe_stencil_bucketed<KM>(table, mask, st, active, pw, res, kx, ky, kz, gl, lane, gmask, g, nhit, n0, n01, limit);
        else total = probe_stencil<KM>(table, mask, st, active, kx, ky, kz, gl, lane, gmask, g, nhit);
        if constexpr (LIO_KNN_STAT || COUNT) {   // the candidate statistic, summed over the wave at once and kept in LDS: a per-lane accumulator across the query loop (and its
            // 64-bit reduction behind it) was the register that stood between this kernel and seven waves per SIMD without a spill
            const uint32_t gs = group_sum32(total);
            const uint32_t ws = (uint32_t)__builtin_amdgcn_readlane((int)gs, 0) + (uint32_t)__builtin_amdgcn_readlane((int)gs, 16) +
                                (uint32_t)__builtin_amdgcn_readlane((int)gs, 32) + (uint32_t)__builtin_amdgcn_readlane((int)gs, 48);
            if (lane == 0) stat_s[tid >> 6] += ws;  // (this wave's own word: no other wave touches it)
        }
        group_lds_sync();
        // every lane keeps its own ascending top-5 as (d2 bits, pool index) pairs, ordered by d2 alone: candidates with an
        // equal d2 keep their arrival order -- any such pair that reaches the global top-6 is an exact tie and the query
        // is redone by knn_exact_kernel anyway.  Insertion is the branch-free parallel form: five independent compares
        // c_i = d2 < e_i, then e_i' = c_{i-1} ? e_{i-1} : (c_i ? new : e_i); for the distances that is min / med3.
        uint32_t d0 = 0xFFFFFFFFu, d1 = 0xFFFFFFFFu, dd2 = 0xFFFFFFFFu, d3 = 0xFFFFFFFFu, d4 = 0xFFFFFFFFu;
        uint32_t i0d = 0xFFFFFFFFu, i1d = 0xFFFFFFFFu, i2d = 0xFFFFFFFFu, i3d = 0xFFFFFFFFu, i4d = 0xFFFFFFFFu;
        uint32_t inrange = 0;
        bool drop_tie = false;
        // voxel-major sweep: kU voxel descriptors per batch come back from LDS as ds_read_b128 pairs, lane l takes point l
        // (l+32, ...) of each -- addresses are base + lane (no per-candidate search), kU 16-B loads in flight per lane
        uint32_t bound5 = 0xFFFFFFFFu;
        const uint32_t ptr_first = g.v_ptr[0];  // (0 when nothing is listed: the pool's first record -- the sweep does not run then)
        bool need_bound = true;  // group-uniform: a lane of the group has taken a candidate since the bound was last computed
#ifndef LIO_KNN_HOME_FIRST
#define LIO_KNN_HOME_FIRST 0
#endif
        // LIO_KNN_HOME_FIRST (experiment): the first listed voxel -- the query's own, when it exists -- is swept ALONE, the bound is formed from it, and
        // the other three of the first batch go through the pruning test like every later batch: the first batch of four voxels is otherwise
        // swept unpruned (~156 of the ~162 candidates a query touches on the metric map)
        int phase = (kPrune && LIO_KNN_HOME_FIRST) ? 0 : 2;  // 0: the first voxel alone; 1: the rest of batch 0; 2: batches as listed
        for (uint32_t s0 = 0; s0 < nhit; s0 += (phase >= 1 ? kU : 0u), phase = (phase < 2 ? phase + 1 : 2)) {
            uint32_t ptr4[kU], cnt4[kU];
            uint32_t cmax = 0;
            bool ins = false;
            if constexpr (kPrune) {
                // exact pruning: a voxel whose nearest possible point is farther than five candidates already seen cannot change the
                // five nearest (nor tie with the fifth: the comparison is strict); the list is in distance buckets, so once the bound
                // is below the floor of the bucket the next batch starts in, nothing that follows can matter either -- and if none of
                // the voxels still listed can reach the bound the sweep ends here (the usual case after the first batch: one bound
                // computation per query instead of one per batch of four listed voxels)
                if (s0 > 0 || phase == 1) {
                    if (need_bound) bound5 = fifth_bound(d0, dd2, d4);
                    const uint32_t floor_bits = s0 < n0 ? 0u : (s0 < n01 ? b1_bits : b2_bits);
                    if (bound5 < floor_bits) break;
                    const uint32_t r0 = s0 + gl, r1 = r0 + kG, r2 = r1 + kG, r3 = r2 + kG;  // a pruned stencil has at most 32 cells
                    const bool mine = (r0 < nhit && g.v_dmin[r0] <= bound5) || (r1 < nhit && g.v_dmin[r1] <= bound5) ||
                                      (kG < 16 && ((r2 < nhit && g.v_dmin[r2] <= bound5) || (r3 < nhit && g.v_dmin[r3] <= bound5)));
                    if (!group_ballot(mine, lane)) break;
                }
            }
#pragma unroll
            for (int u = 0; u < kU; u += 4) {
                const uint4 vp = *reinterpret_cast<const uint4*>(&g.v_ptr[s0 + u]);
                uint4 vc = *reinterpret_cast<const uint4*>(&g.v_cnt[s0 + u]);
                if constexpr (kPrune) {
                    const uint4 vd = *reinterpret_cast<const uint4*>(&g.v_dmin[s0 + u]);
                    if (vd.x > bound5) vc.x = 0;
                    if (vd.y > bound5) vc.y = 0;
                    if (vd.z > bound5) vc.z = 0;
                    if (vd.w > bound5) vc.w = 0;
                    if (LIO_KNN_HOME_FIRST && u == 0) {
                        if (phase == 0) { vc.y = 0; vc.z = 0; vc.w = 0; }
                        else if (phase == 1) vc.x = 0;
                    }
                }
                ptr4[u] = vp.x; ptr4[u + 1] = vp.y; ptr4[u + 2] = vp.z; ptr4[u + 3] = vp.w;
                cnt4[u] = vc.x; cnt4[u + 1] = vc.y; cnt4[u + 2] = vc.z; cnt4[u + 3] = vc.w;
                cmax = max(cmax, max(max(vc.x, vc.y), max(vc.z, vc.w)));
                if constexpr (COUNT) { if (gl == 0) touched += (vc.x + vc.y) + (vc.z + vc.w); }
            }
            for (uint32_t i0 = gl; i0 < cmax + gl; i0 += kG) {
                // UNCONDITIONAL loads at clamped addresses (a lane beyond a voxel's count re-reads the first listed voxel's first point -- a line the
                // group holds anyway -- and ignores it): a load inside `if (i0 < cnt)` comes out of the compiler as load + s_waitcnt vmcnt(0) + moves
                // per voxel (round 4, read off the ISA: the kU loads "in flight" were four memory round trips one after the other)
                float4 p[kU];
#pragma unroll
                for (int u = 0; u < kU; u++) p[u] = pool[i0 < cnt4[u] ? ptr4[u] + i0 : ptr_first];
#pragma unroll
                for (int u = 0; u < kU; u++) {
                    // branch-lean form: a lane without a candidate (beyond the voxel's count, or out of range) offers d2 = +inf bits, which every
                    // comparison below rejects; ONE wave-uniform branch (no lane of the wave takes its candidate) skips the insertion network --
                    // instead of three nested divergent branches per candidate (each an s_and_saveexec / s_cbranch_execz / s_or triple)
                    const bool have = i0 < cnt4[u];
                    if constexpr (COUNT) {
                        if (have && bits) {
                            const uint32_t id = ptr4[u] + i0, bit = 1u << (id & 31u);
                            if (!(atomicOr(&bits[id >> 5], bit) & bit)) fresh++;
                        }
                    }
                    const float dx = p[u].x - pw.x, dy = p[u].y - pw.y, dz = p[u].z - pw.z;
                    const float d2 = dx * dx + (dy * dy + dz * dz);  // ivox3d_node.hpp:12-15: Vector3f::squaredNorm() = Eigen's unrolled tree x0 + (x1 + x2)
                    const bool in = have && d2 < 5.0f;
                    inrange += in ? 1u : 0u;
                    const uint32_t kd = in ? __float_as_uint(d2) : 0xFFFFFFFFu;  // d2 >= 0: float order == unsigned order of the bits
                    const bool take = kd < d4;
                    drop_tie |= in && !take && kd == d4;  // a candidate as far as this lane's fifth is not kept: see the merge
                    if (__ballot(take)) {
                        ins |= take;
                        // ... nor is the entry this insertion pushes OUT of a full list, if it is exactly as far as the entry that becomes the lane's fifth
                        // (d3 == d4 before: two equally distant candidates, the later one leaves) -- the other way a tie at the fifth place can go unseen.
                        // (Remembering the last value pushed out in a register and comparing once after the sweep is one instruction less here and two
                        // registers spilled at seven waves per SIMD.)
                        // (until round 6 only candidates refused on arrival were remembered; found by tools/experiments/lru_tie_stress.py: one query in
                        // ~3 000 on sparse lattice stencils kept the lower pool index instead of the reference's choice)
                        drop_tie |= take && d3 == d4 && d4 != 0xFFFFFFFFu;
                        const uint32_t id = ptr4[u] + i0;
                        const bool c0 = kd < d0, c1 = kd < d1, c2 = kd < dd2, c3 = kd < d3;
                        i4d = c3 ? i3d : (take ? id : i4d);
                        i3d = c2 ? i2d : (c3 ? id : i3d);
                        i2d = c1 ? i1d : (c2 ? id : i2d);
                        i1d = c0 ? i0d : (c1 ? id : i1d);
                        i0d = c0 ? id : i0d;
                        d4 = med3_u32(d3, d4, kd);  // the list is sorted, so slot i becomes the median of (e_{i-1}, e_i, new); kd >= d4 leaves it as it is
                        d3 = med3_u32(dd2, d3, kd);
                        dd2 = med3_u32(d1, dd2, kd);
                        d1 = med3_u32(d0, d1, kd);
                        d0 = min(d0, kd);
                    }
                }
            }
            if constexpr (kPrune) need_bound = group_ballot(ins, lane) != 0;
        }
        inrange = group_sum32(inrange);
        // merge: six rounds pop the group's smallest (d2, index) head -- the global top-5 (lane r keeps winner r) and the best loser
        uint32_t win = 0xFFFFFFFFu, prev_d = 0xFFFFFFFFu;
        bool tie = false;
        int pops = 0;
#pragma unroll
        for (int r = 0; r < 6; r++) {
            const uint32_t bd = group_min32(d0);
            const uint32_t bi = group_min32(d0 == bd ? i0d : 0xFFFFFFFFu);
            const bool some = bi != 0xFFFFFFFFu;  // an empty head is (0xFFFFFFFF, 0xFFFFFFFF); a real d2 < 5 is never all ones
            if (r > 0 && some && bd == prev_d) tie = true;  // equal d2, different point
            prev_d = some ? bd : 0xFFFFFFFFu;
            if (gl == r) win = bi;
            if (some && d0 == bd && i0d == bi) {
                d0 = d1; d1 = dd2; dd2 = d3; d3 = d4; d4 = 0xFFFFFFFFu;
                i0d = i1d; i1d = i2d; i2d = i3d; i3d = i4d; i4d = 0xFFFFFFFFu;
                pops++;
            }
        }
        // The per-lane lists cannot lose a member of the top-5 (whatever a lane drops is no nearer than its own fifth), but
        // a dropped candidate exactly as far as that fifth would be an undetected tie if all five of the lane's entries
        // are among the six best: redo such a query exactly.
        if (group_ballot(pops >= 5 && drop_tie, lane)) tie = true;
        // results.  No in-range candidate at all: GetClosestPoint returns before touching the output
        // (ivox3d.h:152-154), the cached neighbours of an earlier scan survive.
        if (active && inrange > 0) {
            // (32-bit index: five planes of max_ds points; the 64-bit per-lane plane address, hoisted out of the query loop, was the one value the batched kernel spilled to scratch)
            if (gl < 5) {
                uint32_t g2 = (uint32_t)gl;
                asm volatile("" : "+v"(g2));  // (... and the plane offset gl * nn_stride the other: formed where it is used)
                nn_pts[g2 * nn_stride + q] = (win != 0xFFFFFFFFu) ? pool[win] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (gl == 0) {
                nn_cnt[q] = inrange < 5 ? (int32_t)inrange : 5;
                if (tie) tie_list[atomicAdd(n_tie, 1u)] = q;
            }
        }
        group_lds_sync();
    }
    // statistics: one atomic per workgroup, spread over 64 counters that each own a 128-B line (same-line
    // atomics serialise in one L2 channel at ~10 ns apiece -- 9k of them used to cost more than the kernel)
    __shared__ unsigned long long vred[256 / 64];
    if constexpr (LIO_KNN_STAT || COUNT) {
        __syncthreads();
        uint32_t te = threadIdx.x;
        asm volatile("" : "+v"(te));  // (the index is formed here: kept alive from the kernel's head to this tail it was spilled)
        if (te < 256 / 64) vred[te] = stat_s[te];
        __syncthreads();
        if (tid == 0) {
            const unsigned long long v = (vred[0] + vred[1]) + (vred[2] + vred[3]);
            if (v) atomicAdd(&md->knn_cand[(blockIdx.x & 63) * 16], v);
        }
    }
    if constexpr (COUNT) {
        __syncthreads();
        unsigned long long tsum = touched;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) tsum += __shfl_xor(tsum, off);
        if (lane == 0) vred[tid >> 6] = tsum;
        __syncthreads();
        if (tid == 0) {
            const unsigned long long v = (vred[0] + vred[1]) + (vred[2] + vred[3]);
            if (v) atomicAdd(&md->knn_cand[(blockIdx.x & 63) * 16 + 1], v);
        }
        __syncthreads();
        unsigned long long fsum = fresh;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) fsum += __shfl_xor(fsum, off);
        if (lane == 0) vred[tid >> 6] = fsum;
        __syncthreads();
        if (tid == 0) {
            const unsigned long long v = (vred[0] + vred[1]) + (vred[2] + vred[3]);
            if (v) atomicAdd(&md->knn_cand[(blockIdx.x & 63) * 16 + 2], v);
        }
    }
}

template <int KM, int MODE>
__global__ void __launch_bounds__(256, LIO_KNN_WAVES) knn_kernel(const Slot* __restrict__ table, uint32_t mask, const float4* __restrict__ pool,
                                                  float inv_res, StencilArgs st, PoseArgs pose, const float4* __restrict__ queries,
                                                  uint32_t n_host, const ScanDev* __restrict__ sd, float4* __restrict__ world_out,
                                                  float4* __restrict__ nn_pts, uint32_t nn_stride, int32_t* __restrict__ nn_cnt,
                                                  MapDev* md, uint32_t* __restrict__ n_tie, uint32_t* __restrict__ tie_list) {
    knn_body<KM, MODE>(table, mask, pool, inv_res, st, [&](PoseArgs& P) { if (threadIdx.x == 0) P = pose; }, queries, n_host, sd, world_out, nn_pts, nn_stride, nn_cnt, md,
                       n_tie, tie_list);
}
// the scans of a batch (lio_batch_*): blockIdx.y = slot; pose from the slot's device-resident filter; a slot whose update has finished,
// or whose filter did not ask for a neighbour search this pass, exits at once
template <int KM, bool COUNT>
__global__ void LIO_KNN_OCC __launch_bounds__(256, COUNT ? LIO_KNN_WAVES : LIO_KNN_WAVES_BATCH) knn_batch_kernel(const Slot* __restrict__ table, uint32_t mask, const float4* __restrict__ pool,
                                                                       float inv_res, StencilArgs st, const SlotDesc* __restrict__ slots, MapDev* md) {
    const SlotDesc& d = slots[blockIdx.y];
    if (!d.active) return;
    const SlotGateLite sg = slot_gate_lite(d);
    if ((sg.status != EK_RUNNING) | (sg.converge == 0) | (sg.n_ds < d.min_ds)) return;
    const double* __restrict__ x = d.ctrl->x;
    knn_body<KM, 0, COUNT>(table, mask, pool, inv_res, st, [x](PoseArgs& P) { pose_fill_from_state(P, x); }, d.ds_body, sg.n_ds, nullptr, d.ds_world, d.nn_pts,
                                  d.max_ds, d.nn_cnt, md, &d.sd->n_tie, d.tie_list);
}

// ---- exact redo of the queries whose top-6 contained an exact d2 tie ---------------------------------------
// Two steps.  (1) The canonical list: the five smallest in the strict total order (d2, x, y, z), sixteen lanes per query.  It is the answer
// unless the fifth and the sixth of that order are equally far -- then WHICH of the equally distant candidates belong to the five is the
// reference's to say (ivox3d_node.hpp:107-127, ivox3d.h:156-164: std::nth_element on the distance alone, whatever it leaves in front).
// (2) Such a query is redone by the whole workgroup the way the reference does it: stencil voxels in nearby_grids_ order, a voxel's in-range
// points in push_back order (the pool keeps arrival order; MapDev's sequence numbers restore the push_back order), each voxel cut to five,
// the list cut to five -- refsel.h restates libstdc++'s introselect, one lane runs it on the list in LDS.  The survivors are written in the
// canonical order.  About one query in a million on sensor data; a map of lattice points sends every query here (tests).
constexpr int kSelCap = 2560;  // points of one RUN of stencil voxels the staging area holds (a run = as many consecutive voxels as fit; one voxel must)
struct SelLds {
    uint32_t a_d[kSelCap], a_id[kSelCap], a_seq[kSelCap];  // the in-range candidates in stencil order, inside a voxel in pool order
    refsel::Rec list[kSelCap];                              // ... inside a voxel in push_back order: the reference's candidate list before any cut
    refsel::Rec fin[5 * kMaxStencil + 8];                   // ... after every voxel's cut to five
    uint32_t v_ptr[kMaxStencil], v_cnt[kMaxStencil];        // the stencil's voxels in nearby_grids_ order
    uint32_t v_first[kMaxStencil + 1];                      // exclusive prefix of v_cnt (flat index of a voxel's first point)
    uint32_t seg_m[kMaxStencil], seg_base[kMaxStencil + 1], seg_keep[kMaxStencil];  // in-range candidates per voxel, their prefix, what the cut keeps
    uint32_t wsum[4];
    uint32_t out_d[8], out_id[8];
    int size;
};

// One query, the whole workgroup.  (1) thread t < stencil size looks its voxel up; (2) the points of all stencil voxels, flattened in stencil / pool
// order, are loaded 256 at a time, the in-range ones compacted in that order (ballot + prefix sums) with their sequence numbers; (3) every candidate
// ranks itself inside its voxel's segment by sequence number: push_back order; (4) lane v cuts voxel v's segment to five (ivox3d_node.hpp:119-124) --
// the voxels' selections are independent, so they run side by side; (5) one lane strings the kept candidates together and makes the two
// nth_element calls of ivox3d.h:156-162.  The serial parts are a few hundred dependent LDS operations: ~20 us for a query of the metric map.
template <int MODE>
__device__ __forceinline__ void refsel_query(const Slot* __restrict__ table, uint32_t mask, const float4* __restrict__ pool, const uint32_t* __restrict__ seq,
                                          float inv_res, const StencilArgs& st, const PoseArgs& pose, const float4* __restrict__ queries, uint32_t q,
                                          float4* __restrict__ nn_pts, uint32_t nn_stride, MapDev* md, SelLds& S, bool keep_order) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float4 pw;
    {
        const float4 pq = queries[q];
        if (MODE == 0) body_to_world(pose, pq, pw);
        else pw = pq;
    }
    int kx = 0, ky = 0, kz = 0;
    pos2grid(pw.x, pw.y, pw.z, inv_res, kx, ky, kz);
    if (tid < st.n) {
        uint32_t ptr = 0, cnt = 0;
        grid_find(table, mask, kx + st.off[tid][0], ky + st.off[tid][1], kz + st.off[tid][2], ptr, cnt);
        S.v_ptr[tid] = ptr;
        S.v_cnt[tid] = cnt;
        S.seg_m[tid] = 0;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t acc = 0;
        for (int v = 0; v < st.n; v++) { S.v_first[v] = acc; acc += S.v_cnt[v]; }
        S.v_first[st.n] = acc;
    }
    __syncthreads();
    if (tid == 0) S.size = 0;
    // the stencil's voxels in runs that fit the staging area (all of them at once unless the map is dense or the stencil is the 75-cell one); a single
    // voxel with more points than the area holds cannot be staged: the query stays unresolved (counted)
    bool overflow = false;
    int g0 = 0;
    while (g0 < st.n) {
        int g1 = g0;
        uint32_t total = 0;
        while (g1 < st.n && total + S.v_cnt[g1] <= (uint32_t)kSelCap) { total += S.v_cnt[g1]; g1++; }
        if (g1 == g0) { overflow = true; break; }
        const uint32_t f0 = S.v_first[g0];
        uint32_t m_all = 0;  // in-range candidates of this run so far (workgroup-uniform)
        for (uint32_t base = 0; base < total; base += 256) {  // ivox3d_node.hpp:111-116 for every voxel of the run, stencil order, pool order
            const uint32_t f = f0 + base + (uint32_t)tid;
            const bool have = base + (uint32_t)tid < total;
            int v = g0;
            if (have)
                while (v + 1 < g1 && S.v_first[v + 1] <= f) v++;
            const uint32_t idx = have ? S.v_ptr[v] + (f - S.v_first[v]) : 0u;
            const float4 p = pool[idx];
            const float dx = p.x - pw.x, dy = p.y - pw.y, dz = p.z - pw.z;
            const float d2 = dx * dx + (dy * dy + dz * dz);
            const bool in = have && d2 < 5.0f;
            const unsigned long long bal = __ballot(in);
            if (lane == 0) S.wsum[wave] = (uint32_t)__popcll(bal);
            __syncthreads();
            uint32_t at = m_all + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            for (int w = 0; w < wave; w++) at += S.wsum[w];
            const uint32_t tot = (S.wsum[0] + S.wsum[1]) + (S.wsum[2] + S.wsum[3]);
            if (in) {
                S.a_d[at] = __float_as_uint(d2);
                S.a_id[at] = idx;
                S.a_seq[at] = seq[idx];
                atomicAdd(&S.seg_m[v], 1u);
            }
            m_all += tot;
            __syncthreads();
        }
        if (tid == 0) {
            uint32_t acc = 0;
            for (int v = g0; v < g1; v++) { S.seg_base[v] = acc; acc += S.seg_m[v]; }
            S.seg_base[g1] = acc;
        }
        __syncthreads();
        // push_back order inside every voxel: rank by sequence number (signed difference: the counter wraps); equal numbers -- never inside one voxel
        // short of a wrap -- by pool position
        for (uint32_t e = (uint32_t)tid; e < m_all; e += 256) {
            int v = g0;
            while (v + 1 < g1 && S.seg_base[v + 1] <= e) v++;
            const uint32_t b0 = S.seg_base[v], b1 = S.seg_base[v + 1];
            const uint32_t se = S.a_seq[e];
            uint32_t r = 0;
            for (uint32_t f = b0; f < b1; f++) {
                const uint32_t sf = S.a_seq[f];
                r += ((int32_t)(sf - se) < 0 || (sf == se && f < e)) ? 1u : 0u;
            }
            refsel::Rec rec;
            rec.d = S.a_d[e];
            rec.id = S.a_id[e];
            S.list[b0 + r] = rec;
        }
        __syncthreads();
        if (tid < g1 - g0) {  // ivox3d_node.hpp:119-124, every voxel of the run on its own lane
            const int v = g0 + tid;
            const int b0 = (int)S.seg_base[v], m = (int)S.seg_m[v];
            S.seg_keep[v] = (uint32_t)(refsel::voxel_cut(S.list, b0, b0 + m, 5) - b0);
        }
        __syncthreads();
        if (tid == 0) {
            int size = S.size;
            for (int v = g0; v < g1; v++) {
                const int b0 = (int)S.seg_base[v], keep = (int)S.seg_keep[v];
                for (int k = 0; k < keep; k++) S.fin[size++] = S.list[b0 + k];
            }
            S.size = size;
        }
        __syncthreads();
        g0 = g1;
    }
    if (tid == 0) {
        if (overflow) {
            atomicAdd(&md->n_tie_unresolved, 1ull);  // (the canonical list written by step 1 stays)
        } else {
            const int size = S.size;
            const int n = size ? refsel::final_cut(S.fin, size, 5) : 0;  // ivox3d.h:156-162
            for (int k = 0; k < n; k++) {  // the survivors in the canonical order -- or, tie mode 2, as the reference returns them (the list stays in LDS: no private array)
                const Cand x = {__uint_as_float(S.fin[k].d), S.fin[k].id};
                int at = k;
                while (!keep_order && at > 0) {
                    const Cand y = {__uint_as_float(S.out_d[at - 1]), S.out_id[at - 1]};
                    if (!cand_less(x, y, pool)) break;
                    S.out_d[at] = S.out_d[at - 1];
                    S.out_id[at] = S.out_id[at - 1];
                    at--;
                }
                S.out_d[at] = S.fin[k].d;
                S.out_id[at] = x.id;
            }
            for (int k = 0; k < n; k++) nn_pts[(size_t)k * nn_stride + q] = pool[S.out_id[k]];
        }
        if (!keep_order) atomicAdd(&md->n_tie_boundary, 1ull);
    }
    __syncthreads();
}

template <int KM, int MODE>
__device__ __forceinline__ void knn_exact_body(const Slot* __restrict__ table, uint32_t mask, const float4* __restrict__ pool,
                                               float inv_res, const StencilArgs& st, const PoseArgs& pose, const float4* __restrict__ queries,
                                               float4* __restrict__ nn_pts, uint32_t nn_stride, const uint32_t* __restrict__ n_tie,
                                               uint32_t* __restrict__ tie_list, int tie_mode) {
    __shared__ GroupLds lds[kGPB];
    const int tid = threadIdx.x;
    const int grp = tid / kG, gl = tid % kG;
    const int lane = tid & 63;
    GroupLds& g = lds[grp];
    const uint32_t n = *n_tie;
    const unsigned long long gmask = ((1ull << kG) - 1ull) << (lane - gl);
    for (uint32_t w0 = blockIdx.x * kGPB; w0 < n; w0 += gridDim.x * kGPB) {
        const uint32_t w = w0 + grp;
        const bool active = w < n;
        const uint32_t q = active ? (tie_list[w] & 0x7FFFFFFFu) : 0u;
        float4 pw = make_float4(0.f, 0.f, 0.f, 0.f);
        if (active) {
            const float4 pq = queries[q];
            if (MODE == 0) body_to_world(pose, pq, pw);
            else pw = pq;
        }
        int kx = 0, ky = 0, kz = 0;
        pos2grid(pw.x, pw.y, pw.z, inv_res, kx, ky, kz);
        uint32_t nhit = 0;
        const uint32_t total = probe_stencil<KM>(table, mask, st, active, kx, ky, kz, gl, lane, gmask, g, nhit);
        __syncthreads();
        // every lane keeps its SIX smallest: the sixth of the whole query decides whether the fifth place is contested
        Cand e[6];
        for (int k = 0; k < 6; k++) e[k] = {INFINITY, kNoIdx};
        (void)total;
        // two voxels x four points per lane and step, the eight loads requested together (clamped, unconditional; the list is zero-filled behind
        // its end).  One load per step made a tied query ~24 memory round trips one after the other, and this kernel sits on the round's critical
        // path five times (12 us per launch for a handful of queries).  The lists are kept in the strict total order: the order of arrival
        // does not matter.
        const uint32_t ptr_first = g.v_ptr[0];
        for (uint32_t sv = 0; sv < nhit; sv += 2) {
            const uint32_t vptr[2] = {g.v_ptr[sv], g.v_ptr[sv + 1]}, vcnt[2] = {g.v_cnt[sv], g.v_cnt[sv + 1]};
            const uint32_t cmax = max(vcnt[0], vcnt[1]);
            for (uint32_t i0 = 0; i0 < cmax; i0 += 4 * kG) {
                float4 p[2][4];
#pragma unroll
                for (int v = 0; v < 2; v++)
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t i = i0 + (uint32_t)k * kG + (uint32_t)gl;
                        p[v][k] = pool[i < vcnt[v] ? vptr[v] + i : ptr_first];
                    }
#pragma unroll
                for (int v = 0; v < 2; v++)
#pragma unroll
                    for (int k = 0; k < 4; k++) pin_loaded(p[v][k]);
#pragma unroll
                for (int v = 0; v < 2; v++)
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t i = i0 + (uint32_t)k * kG + (uint32_t)gl;
                        if (i >= vcnt[v]) continue;
                        const uint32_t id = vptr[v] + i;
                        const float dx = p[v][k].x - pw.x, dy = p[v][k].y - pw.y, dz = p[v][k].z - pw.z;
                        const float d2 = dx * dx + (dy * dy + dz * dz);
                        if (d2 < 5.0f) {
                            Cand cd = {d2, id};
                            if (cand_less(cd, e[5], pool)) {
                                e[5] = cd;
                                for (int kk = 5; kk > 0; kk--)
                                    if (cand_less(e[kk], e[kk - 1], pool)) { const Cand t = e[kk - 1]; e[kk - 1] = e[kk]; e[kk] = t; }
                            }
                        }
                    }
            }
        }
        uint32_t win = kNoIdx;
        float d_fifth = -1.f, d_sixth = -2.f;
        for (int r = 0; r < 6; r++) {
            Cand best = e[0];
            for (int off = kG / 2; off > 0; off >>= 1) {
                Cand o;
                o.d2 = __shfl_xor(best.d2, off, kG);
                o.id = __shfl_xor(best.id, off, kG);
                if (cand_less(o, best, pool)) best = o;
            }
            if (r < 5 && gl == r) win = best.id;
            if (best.id != kNoIdx) {
                if (r == 4) d_fifth = best.d2;
                if (r == 5) d_sixth = best.d2;
            }
            if (best.id != kNoIdx && e[0].id == best.id) {
                for (int k = 0; k < 5; k++) e[k] = e[k + 1];
                e[5] = {INFINITY, kNoIdx};
            }
        }
        if (active && gl < 5 && win != kNoIdx) nn_pts[(size_t)gl * nn_stride + q] = pool[win];
        // the fifth place is contested: the entry is marked (top bit) for the selection kernel that follows (knn_refsel_*)
        if (tie_mode != 0 && active && gl == 0 && d_fifth == d_sixth) tie_list[w] = q | 0x80000000u;
        __syncthreads();
    }
}

template <int KM, int MODE>
__global__ void __launch_bounds__(256) knn_exact_kernel(const Slot* __restrict__ table, uint32_t mask, const float4* __restrict__ pool,
                                                        float inv_res, StencilArgs st, PoseArgs pose, const float4* __restrict__ queries,
                                                        float4* __restrict__ nn_pts, uint32_t nn_stride, const uint32_t* __restrict__ n_tie,
                                                        uint32_t* __restrict__ tie_list, int tie_mode) {
    knn_exact_body<KM, MODE>(table, mask, pool, inv_res, st, pose, queries, nn_pts, nn_stride, n_tie, tie_list, tie_mode);
}
// step 2: the marked entries of the tie list (tie mode 2: every query, q = 0 .. n_all - 1), one at a time by the whole workgroup
template <int MODE>
__device__ __forceinline__ void knn_refsel_body(const Slot* __restrict__ table, uint32_t mask, const float4* __restrict__ pool, const uint32_t* __restrict__ seq,
                                                float inv_res, const StencilArgs& st, const PoseArgs& pose, const float4* __restrict__ queries,
                                                float4* __restrict__ nn_pts, uint32_t nn_stride, const uint32_t* __restrict__ n_tie,
                                                const uint32_t* __restrict__ tie_list, MapDev* md, int tie_mode, uint32_t n_all) {
    __shared__ SelLds sel;
    const bool all = tie_mode == 2;
    const uint32_t n = all ? n_all : *n_tie;
    for (uint32_t w = blockIdx.x; w < n; w += gridDim.x) {
        const uint32_t ent = all ? (w | 0x80000000u) : tie_list[w];
        if (!(ent >> 31)) continue;  // (workgroup-uniform)
        refsel_query<MODE>(table, mask, pool, seq, inv_res, st, pose, queries, ent & 0x7FFFFFFFu, nn_pts, nn_stride, md, sel, all);
    }
}
template <int MODE>
__global__ void __launch_bounds__(256) knn_refsel_kernel(const Slot* __restrict__ table, uint32_t mask, const float4* __restrict__ pool, const uint32_t* __restrict__ seq,
                                                         float inv_res, StencilArgs st, PoseArgs pose, const float4* __restrict__ queries, float4* __restrict__ nn_pts,
                                                         uint32_t nn_stride, const uint32_t* __restrict__ n_tie, const uint32_t* __restrict__ tie_list, MapDev* md,
                                                         int tie_mode, uint32_t n_all) {
    knn_refsel_body<MODE>(table, mask, pool, seq, inv_res, st, pose, queries, nn_pts, nn_stride, n_tie, tie_list, md, tie_mode, n_all);
}
__global__ void __launch_bounds__(256) knn_refsel_batch_kernel(const Slot* __restrict__ table, uint32_t mask, const float4* __restrict__ pool,
                                                               const uint32_t* __restrict__ seq, float inv_res, StencilArgs st, const SlotDesc* __restrict__ slots,
                                                               MapDev* md, int tie_mode) {
    const SlotDesc& d = slots[blockIdx.y];
    if (!d.active || (d.sd->n_tie == 0 && tie_mode != 2)) return;
    const SlotGate sg = slot_gate(d);
    if ((sg.status != EK_RUNNING) | (sg.converge == 0) | (sg.n_ds < d.min_ds)) return;
    knn_refsel_body<0>(table, mask, pool, seq, inv_res, st, sg.pose, d.ds_body, d.nn_pts, d.max_ds, &d.sd->n_tie, d.tie_list, md, tie_mode, sg.n_ds);
}
__global__ void __launch_bounds__(256) knn_refsel_seq_kernel(const MapRef* __restrict__ maps, StencilArgs st, int stencil_id, const SlotDesc* __restrict__ slots) {
    const SlotDesc& d = slots[blockIdx.y];
    if (!d.active) return;
    const MapRef& r = maps[blockIdx.y];
    if (r.stencil_id != stencil_id || r.tie_mode == 0 || !r.pool_seq || (d.sd->n_tie == 0 && r.tie_mode != 2)) return;
    const SlotGate sg = slot_gate(d);
    if ((sg.status != EK_RUNNING) | (sg.converge == 0) | (sg.n_ds < d.min_ds)) return;
    knn_refsel_body<0>(r.table, r.mask, r.pool, r.pool_seq, r.inv_res, st, sg.pose, d.ds_body, d.nn_pts, d.max_ds, &d.sd->n_tie, d.tie_list, r.md, r.tie_mode, sg.n_ds);
}
// batch form: the queries the search of this pass queued (usually none: the kernel then ends at once); the queue is re-armed by the
// filter-pass kernel that follows the linearisation
template <int KM>
__global__ void __launch_bounds__(256) knn_exact_batch_kernel(const Slot* __restrict__ table, uint32_t mask, const float4* __restrict__ pool,
                                                              float inv_res, StencilArgs st, const SlotDesc* __restrict__ slots, int tie_mode) {
    const SlotDesc& d = slots[blockIdx.y];
    if (!d.active || d.sd->n_tie == 0) return;
    const SlotGate sg = slot_gate(d);
    if ((sg.status != EK_RUNNING) | (sg.converge == 0) | (sg.n_ds < d.min_ds)) return;
    const PoseArgs& pose = sg.pose;
    knn_exact_body<KM, 0>(table, mask, pool, inv_res, st, pose, d.ds_body, d.nn_pts, d.max_ds, &d.sd->n_tie, d.tie_list, tie_mode);
}

// Tied queries are queued and redone by a second (usually empty) launch.  (Measured and dropped in round 3: redoing them in place -- one
// launch less per pass at 14 more registers, slower; the re-search of a later pass from the previous neighbours -- exact, no gain; one lane
// per query with a flattened sweep -- 2-4 x slower.  tools/experiments/README.md has the numbers.)
int knn_batch_launch(lio_map* m, hipStream_t st, const SlotDesc* d_slots, int n_slots, uint32_t grid_x, int count_touched) {
    // (may run inside a stream capture: the callers settle a pending insert of the map first -- run_update; batched maps are static)
    // Workgroups per slot: enough of them over all slots to fill the machine a few times (256 CUs x 7 resident workgroups), not more -- a
    // workgroup that takes several query blocks in turn pays the kernel's prologue and epilogue (~11 % of a single block's instructions)
    // once.  One slot alone (the single-scan engine) keeps the full 2048; 24 slots get 512 each (measured: 16.9 -> 14.9 us per scan and
    // search; 256 and fewer lose to the tail).  LIO_KNN_GRID overrides.
    static const uint32_t forced = [] { const char* e = getenv("LIO_KNN_GRID"); const int v = e ? atoi(e) : 0; return v >= 8 ? (uint32_t)v : 0u; }();
    uint32_t cap = forced ? forced : (12288u / (uint32_t)(n_slots > 0 ? n_slots : 1));
    if (cap > 2048u) cap = 2048u;
    if (cap < 128u) cap = 128u;
    if (grid_x > cap) grid_x = cap;  // grid-stride loop inside: 16 queries per workgroup and round
    if (grid_x == 0) grid_x = 8;
    const dim3 grid((grid_x + 7u) & ~7u, (uint32_t)n_slots);
    const dim3 gridx(n_slots > 8 ? 8 : 64, (uint32_t)n_slots);  // the tie queue of a scan holds a handful of queries at most: a few workgroups per slot (grid-stride inside)
    const int tmode = m->pool_seq ? m->tie_mode : 0;
    const int km = (m->stencil.n + kG - 1) / kG;
    // (the counting variant also counts DISTINCT points per launch: its bitmap starts empty)
    if (count_touched && m->touch_bits) LIO_HIP_TRY(hipMemsetAsync(m->touch_bits, 0, (size_t)((m->pool_cap + 31) / 32) * 4, st));
#define KNNB_LAUNCH(KM)                                                                                                                              \
    do {                                                                                                                                             \
        if (count_touched) hipLaunchKernelGGL((knn_batch_kernel<KM, true>), grid, 256, 0, st, m->table, m->table_mask, m->pool, m->inv_res, m->stencil, d_slots, m->dev); \
        else hipLaunchKernelGGL((knn_batch_kernel<KM, false>), grid, 256, 0, st, m->table, m->table_mask, m->pool, m->inv_res, m->stencil, d_slots, m->dev); \
        hipLaunchKernelGGL((knn_exact_batch_kernel<KM>), gridx, 256, 0, st, m->table, m->table_mask, m->pool, m->inv_res, m->stencil, d_slots, tmode); \
    } while (0)
    if (km <= 1) KNNB_LAUNCH(1);
    else if (km <= 2) KNNB_LAUNCH(2);
    else if (km <= 3) KNNB_LAUNCH(3);
    else KNNB_LAUNCH((kMaxStencil + kG - 1) / kG);
#undef KNNB_LAUNCH
    // step 2 of the exact redo (the reference's selection for the queries whose fifth place is contested): usually finds nothing marked and ends at once
    if (tmode != 0)
        hipLaunchKernelGGL(knn_refsel_batch_kernel, dim3(tmode == 2 ? 256 : 4, (uint32_t)n_slots), 256, 0, st, m->table, m->table_mask, m->pool, m->pool_seq, m->inv_res,
                           m->stencil, d_slots, m->dev, tmode);
    LIO_HIP_TRY(hipGetLastError());
    return LIO_OK;
}

// sequence mode (lio_batch_create_sequences): every slot searches ITS OWN map -- table, pool and counters come from the slot's MapRef instead of
// the kernel arguments.  The stencil travels by value, so one launch serves the slots whose map uses that stencil (normally all of them: 19).
template <int KM>
__global__ void __launch_bounds__(256, KM <= 3 ? LIO_KNN_WAVES_BATCH : LIO_KNN_WAVES) knn_seq_kernel(const MapRef* __restrict__ maps, StencilArgs st, int stencil_id,
                                                                     const SlotDesc* __restrict__ slots) {
    const SlotDesc& d = slots[blockIdx.y];
    if (!d.active) return;
    const MapRef& r = maps[blockIdx.y];
    if (r.stencil_id != stencil_id) return;
    const SlotGateLite sg = slot_gate_lite(d);
    if ((sg.status != EK_RUNNING) | (sg.converge == 0) | (sg.n_ds < d.min_ds)) return;
    const double* __restrict__ x = d.ctrl->x;
    knn_body<KM, 0, false>(r.table, r.mask, r.pool, r.inv_res, st, [x](PoseArgs& P) { pose_fill_from_state(P, x); }, d.ds_body, sg.n_ds, nullptr, d.ds_world, d.nn_pts,
                                  d.max_ds, d.nn_cnt, r.md, &d.sd->n_tie, d.tie_list);
}
template <int KM>
__global__ void __launch_bounds__(256) knn_exact_seq_kernel(const MapRef* __restrict__ maps, StencilArgs st, int stencil_id, const SlotDesc* __restrict__ slots) {
    const SlotDesc& d = slots[blockIdx.y];
    if (!d.active) return;
    const MapRef& r = maps[blockIdx.y];
    if (r.stencil_id != stencil_id || d.sd->n_tie == 0) return;
    const SlotGate sg = slot_gate(d);
    if ((sg.status != EK_RUNNING) | (sg.converge == 0) | (sg.n_ds < d.min_ds)) return;
    const PoseArgs& pose = sg.pose;
    knn_exact_body<KM, 0>(r.table, r.mask, r.pool, r.inv_res, st, pose, d.ds_body, d.nn_pts, d.max_ds, &d.sd->n_tie, d.tie_list, r.pool_seq ? r.tie_mode : 0);
}

int knn_seq_launch(hipStream_t st, const MapRef* d_maps, const SlotDesc* d_slots, int n_slots, uint32_t grid_x, const StencilArgs* stencils, const int* stencil_ids,
                   int n_stencils) {
    static const uint32_t forced = [] { const char* e = getenv("LIO_KNN_GRID"); const int v = e ? atoi(e) : 0; return v >= 8 ? (uint32_t)v : 0u; }();
    uint32_t cap = forced ? forced : (12288u / (uint32_t)(n_slots > 0 ? n_slots : 1));
    if (cap > 2048u) cap = 2048u;
    if (cap < 128u) cap = 128u;
    if (grid_x > cap) grid_x = cap;
    if (grid_x == 0) grid_x = 8;
    const dim3 grid((grid_x + 7u) & ~7u, (uint32_t)n_slots);
    const dim3 gridx(n_slots > 8 ? 8 : 64, (uint32_t)n_slots);
    for (int k = 0; k < n_stencils; k++) {
        const StencilArgs& sa = stencils[k];
        const int km = (sa.n + kG - 1) / kG;
#define KNNS_LAUNCH(KM)                                                                                                  \
    do {                                                                                                                 \
        hipLaunchKernelGGL((knn_seq_kernel<KM>), grid, 256, 0, st, d_maps, sa, stencil_ids[k], d_slots);                 \
        hipLaunchKernelGGL((knn_exact_seq_kernel<KM>), gridx, 256, 0, st, d_maps, sa, stencil_ids[k], d_slots);          \
    } while (0)
        if (km <= 1) KNNS_LAUNCH(1);
        else if (km <= 2) KNNS_LAUNCH(2);
        else if (km <= 3) KNNS_LAUNCH(3);
        else KNNS_LAUNCH((kMaxStencil + kG - 1) / kG);
#undef KNNS_LAUNCH
        hipLaunchKernelGGL(knn_refsel_seq_kernel, dim3(4, (uint32_t)n_slots), 256, 0, st, d_maps, sa, stencil_ids[k], d_slots);
    }
    LIO_HIP_TRY(hipGetLastError());
    return LIO_OK;
}

template <int MODE>
static int launch_knn(lio_map* m, hipStream_t st, const PoseArgs& pose, const float4* q, uint32_t n_host, const ScanDev* sd,
                      float4* world_out, float4* nn_pts, uint32_t nn_stride, int32_t* nn_cnt, uint32_t n_bound, uint32_t* n_tie,
                      uint32_t* tie_list) {
    uint32_t blocks = (n_bound + kGPB - 1) / kGPB;
    if (blocks > 16384) blocks = 16384;
    if (blocks == 0) return LIO_OK;
    blocks = (blocks + 7u) & ~7u;  // multiple of 8: one contiguous slice of the queries per XCD
#define KNN_LAUNCH(KM)                                                                                                                  \
    hipLaunchKernelGGL((knn_kernel<KM, MODE>), blocks, 256, 0, st, m->table, m->table_mask, m->pool, m->inv_res, m->stencil, pose, q, n_host, \
                       sd, world_out, nn_pts, nn_stride, nn_cnt, m->dev, n_tie, tie_list)
    const int km = (m->stencil.n + kG - 1) / kG;  // stencil cells per lane
    if (km <= 1) KNN_LAUNCH(1);
    else if (km <= 2) KNN_LAUNCH(2);
    else if (km <= 3) KNN_LAUNCH(3);
    else KNN_LAUNCH((kMaxStencil + kG - 1) / kG);
#undef KNN_LAUNCH
    LIO_HIP_TRY(hipGetLastError());
    return LIO_OK;
}

template <int MODE>
static int launch_knn_exact(lio_map* m, hipStream_t st, const PoseArgs& pose, const float4* q, float4* nn_pts, uint32_t nn_stride,
                            uint32_t n_tie_host, const uint32_t* n_tie, uint32_t* tie_list, uint32_t n_all = 0) {
    const int tmode = m->pool_seq ? m->tie_mode : 0;
    if (tmode == 2 && n_all) {  // every query through the reference's selection (the canonical lists of the search stay where a voxel overflows the staging area)
        hipLaunchKernelGGL((knn_refsel_kernel<MODE>), n_all < 4096u ? n_all : 4096u, 256, 0, st, m->table, m->table_mask, m->pool, m->pool_seq, m->inv_res, m->stencil, pose,
                           q, nn_pts, nn_stride, n_tie, tie_list, m->dev, tmode, n_all);
        LIO_HIP_TRY(hipGetLastError());
        return LIO_OK;
    }
    uint32_t blocks = (n_tie_host + kGPB - 1) / kGPB;
    if (blocks == 0) return LIO_OK;
    if (blocks > 4096) blocks = 4096;
#define KNNX_LAUNCH(KM)                                                                                                                      \
    hipLaunchKernelGGL((knn_exact_kernel<KM, MODE>), blocks, 256, 0, st, m->table, m->table_mask, m->pool, m->inv_res, m->stencil, pose, q, nn_pts, \
                       nn_stride, n_tie, tie_list, tmode)
    const int km = (m->stencil.n + kG - 1) / kG;
    if (km <= 1) KNNX_LAUNCH(1);
    else if (km <= 2) KNNX_LAUNCH(2);
    else if (km <= 3) KNNX_LAUNCH(3);
    else KNNX_LAUNCH((kMaxStencil + kG - 1) / kG);
#undef KNNX_LAUNCH
    if (tmode != 0)
        hipLaunchKernelGGL((knn_refsel_kernel<MODE>), n_tie_host < 1024u ? n_tie_host : 1024u, 256, 0, st, m->table, m->table_mask, m->pool, m->pool_seq, m->inv_res, m->stencil,
                           pose, q, nn_pts, nn_stride, n_tie, tie_list, m->dev, tmode, 0u);
    LIO_HIP_TRY(hipGetLastError());
    return LIO_OK;
}

int map_knn_plane(lio_map* m, lio_scan* s, const PoseArgs& pose, int redo_knn) {
    { const int rc_settle = map_settle(m); if (rc_settle != LIO_OK) return rc_settle; }  // an insert still running on the map's stream
    (void)redo_knn;
    const uint32_t bound = s->have_ds > 0 ? (uint32_t)s->have_ds : (s->n_raw && s->n_raw < s->max_ds ? s->n_raw : s->max_ds);
    kt_begin(s, 0);
    // the query count comes from the host whenever it knows it (the usual case: the downsample was waited for): one dependent
    // global load less at the head of every wave
    const bool known = s->have_ds > 0;
    const int rc = launch_knn<0>(m, s->stream, pose, s->ds_body, known ? (uint32_t)s->have_ds : 0u, known ? nullptr : s->dev, s->ds_world, s->nn_pts,
                                 s->max_ds, s->nn_cnt, bound,
                                 &s->dev->n_tie, s->tie_list);
    kt_end(s, 0);
    return rc;
}

// redo the queued tie queries of the last map_knn_plane exactly (n_tie_host = count read back by the caller)
int map_knn_exact(lio_map* m, lio_scan* s, const PoseArgs& pose, uint32_t n_tie_host) {
    { const int rc_settle = map_settle(m); if (rc_settle != LIO_OK) return rc_settle; }  // an insert still running on the map's stream
    // (tie mode 2: n_tie_host is the number of queries of the scan -- all of them are redone)
    return launch_knn_exact<0>(m, s->stream, pose, s->ds_body, s->nn_pts, s->max_ds, n_tie_host, &s->dev->n_tie_done, s->tie_list, m->tie_mode == 2 ? n_tie_host : 0u);
}

int knn_batch(lio_map* m, const float4* d_q, uint32_t n, float4* d_out, int32_t* d_cnt, uint32_t* d_tie /* [0] = count, [1..] = list */) {
    PoseArgs pose;
    memset(&pose, 0, sizeof(pose));
    int rc = launch_knn<1>(m, m->stream, pose, d_q, n, nullptr, nullptr, d_out, n, d_cnt, n, d_tie, d_tie + 1);
    if (rc != LIO_OK) return rc;
    uint32_t nt = 0;
    LIO_HIP_TRY(hipMemcpyAsync(&nt, d_tie, 4, hipMemcpyDeviceToHost, m->stream));
    LIO_HIP_TRY(hipStreamSynchronize(m->stream));
    if (m->tie_mode == 2 && m->pool_seq && n) rc = launch_knn_exact<1>(m, m->stream, pose, d_q, d_out, n, n, d_tie, d_tie + 1, n);
    else if (nt) rc = launch_knn_exact<1>(m, m->stream, pose, d_q, d_out, n, nt, d_tie, d_tie + 1);
    return rc;
}

}  // namespace lio
