// knn_q.hip -- stencil kNN, throughput form: G = 4 lanes per query (16 queries per wave), no LDS.
//
// Same semantics as knn.hip (IVox::GetClosestPoint(pt, out, 5, 5.0), /root/reference/slam/mapping/fastlio/include/ivox3d/
// ivox3d.h:139-171 + ivox3d_node.hpp:107-127: the 5 nearest of all points stored in the stencil voxels with d^2 < 5.0, canonical
// order (d2, x, y, z), fewer than five -> all, none -> output untouched), a different mapping to the machine.  knn.hip gives a
// query 16 lanes: the right shape when ONE scan has to fill the chip, but every per-query step (f64 transform, hashing, list
// compaction, merge, output) is then executed by 16 lanes for one query -- ~300 wave instructions per query, and the kernel is
// bound by instruction issue and by the dependent chain of its few resident waves, not by bytes.  When scans are registered in
// batches (lio_batch_*: one launch serves B scans) there are enough queries to fill the chip with four lanes each:
//   * each lane hashes and probes ceil(S / 4) stencil cells (five independent slot loads in flight for NEARBY18) and KEEPS its
//     hits in registers -- no LDS list, no bucket compaction;
//   * the group walks the cells in stencil order (home cell, the six face neighbours, the edge neighbours: already near to far):
//     the owner lane broadcasts (ptr, cnt, lower bound of the voxel's distance) with a DPP quad broadcast, a voxel that cannot
//     beat the five candidates known so far is skipped (exact: strict comparison, same sets / ties / counts as the unpruned
//     sweep), the others are read 16 points per step -- lane l takes points l, l+4, l+8, l+12: four 16-B loads in flight per
//     lane, 64 contiguous bytes per group and load;
//   * per-lane sorted top-5 (v_med3 insertion), six rounds of quad minima pop the global top-5 and the best loser;
//   * an exact d2 tie among the six best (the only case where (d2, pool index) order can differ from the canonical order) is
//     redone in place by the same four lanes with the full comparison -- no tie queue, no second launch.
// Used by the batched engine (batch.hip) and, for the tests' sake, selectable for lio_map_knn (LIO_KNN_Q=1).
#include "hashgrid.h"
#include "knn_dev.h"
#include "lio_common.h"

namespace lio {

template <int G>
__device__ inline uint32_t qmin32(uint32_t v) {
    if constexpr (G == 4) {
        v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));  // quad_perm [1,0,3,2]
        v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));  // quad_perm [2,3,0,1]
    } else {
#pragma unroll
        for (int off = G / 2; off > 0; off >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, off, G));
    }
    return v;
}
template <int G>
__device__ inline uint32_t qsum32(uint32_t v) {
    if constexpr (G == 4) {
        v += (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false);
        v += (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false);
    } else {
#pragma unroll
        for (int off = G / 2; off > 0; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off, G);
    }
    return v;
}
template <int G>
__device__ inline bool qany(bool p) { return qsum32<G>(p ? 1u : 0u) != 0u; }
// value of lane `o` of the group in every lane of the group (o is uniform)
template <int G>
__device__ inline uint32_t qbcast(uint32_t v, int o) {
    if constexpr (G == 4) {
        switch (o) {
            case 0: return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x00, 0xF, 0xF, false);
            case 1: return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x55, 0xF, 0xF, false);
            case 2: return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xAA, 0xF, 0xF, false);
            default: return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xFF, 0xF, 0xF, false);
        }
    } else {
        return (uint32_t)__shfl((int)v, o, G);
    }
}

__device__ inline uint32_t q_med3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// sorted per-lane top-5 of (d2 bits, pool index), ordered by d2 alone (arrival order on equal d2: any such pair that reaches the
// global top-6 is an exact tie and the query is redone exactly)
struct Top5 {
    uint32_t d0, d1, d2, d3, d4;
    uint32_t i0, i1, i2, i3, i4;
    __device__ inline void clear() { d0 = d1 = d2 = d3 = d4 = 0xFFFFFFFFu; i0 = i1 = i2 = i3 = i4 = 0xFFFFFFFFu; }
    __device__ inline void insert(uint32_t kd, uint32_t id) {  // requires kd < d4
        const bool c0 = kd < d0, c1 = kd < d1, c2 = kd < d2, c3 = kd < d3;
        i4 = c3 ? i3 : id;
        i3 = c2 ? i2 : (c3 ? id : i3);
        i2 = c1 ? i1 : (c2 ? id : i2);
        i1 = c0 ? i0 : (c1 ? id : i1);
        i0 = c0 ? id : i0;
        d4 = q_med3(d3, d4, kd);
        d3 = q_med3(d2, d3, kd);
        d2 = q_med3(d1, d2, kd);
        d1 = q_med3(d0, d1, kd);
        d0 = min(d0, kd);
    }
    __device__ inline void pop() {
        d0 = d1; d1 = d2; d2 = d3; d3 = d4; d4 = 0xFFFFFFFFu;
        i0 = i1; i1 = i2; i2 = i3; i3 = i4; i4 = 0xFFFFFFFFu;
    }
};

// upper bound of the squared distance of the query's fifth nearest candidate from what the lanes hold (see knn.hip fifth_bound)
template <int G>
__device__ inline uint32_t q_fifth_bound(uint32_t d0, uint32_t d4) {
    const uint32_t t = qmin32<G>(d4);
    uint32_t v = d0, m = 0xFFFFFFFFu;
#pragma unroll
    for (int r = 0; r < 5; r++) {
        m = qmin32<G>(v);
        if (v == m) v = 0xFFFFFFFFu;
    }
    return min(t, m);
}

struct QCand {
    float d2;
    uint32_t id;
};
__device__ __noinline__ bool q_tie_less(const QCand& a, const QCand& b, const float4* __restrict__ pool) {
    if (a.id == b.id) return false;
    if (a.id == kNoIdx || b.id == kNoIdx) return a.id < b.id;
    const float4 pa = pool[a.id], pb = pool[b.id];
    if (pa.x != pb.x) return pa.x < pb.x;
    if (pa.y != pb.y) return pa.y < pb.y;
    if (pa.z != pb.z) return pa.z < pb.z;
    return a.id < b.id;
}
__device__ inline bool q_less(const QCand& a, const QCand& b, const float4* __restrict__ pool) {
    if (a.d2 != b.d2) return a.d2 < b.d2;
    return q_tie_less(a, b, pool);
}

// exact redo of one query by its G lanes: every stencil voxel, strict total order (d2, x, y, z).  `win[r]` of lane r % G ... the
// winners come back as: winner r in lane r for r < G, the remaining ones (r >= G) in lane r - G's second slot.
template <int G, int KM>
__device__ __noinline__ void q_exact_redo(const float4* __restrict__ pool, const float4 pw, const uint32_t (&vptr)[KM], const uint32_t (&vcnt)[KM],
                                          int gl, uint32_t (&win)[2]) {
    QCand e[5];
#pragma unroll
    for (int k = 0; k < 5; k++) e[k] = {INFINITY, kNoIdx};
    for (int k = 0; k < KM; k++)
        for (int o = 0; o < G; o++) {
            const uint32_t p = qbcast<G>(vptr[k], o), c = qbcast<G>(vcnt[k], o);
            for (uint32_t i = gl; i < c; i += G) {
                const uint32_t id = p + i;
                const float4 q = pool[id];
                const float dx = q.x - pw.x, dy = q.y - pw.y, dz = q.z - pw.z;
                const float d2 = dx * dx + (dy * dy + dz * dz);
                if (d2 < 5.0f) {
                    QCand cd = {d2, id};
                    if (q_less(cd, e[4], pool)) {
                        e[4] = cd;
                        for (int j = 4; j > 0; j--)
                            if (q_less(e[j], e[j - 1], pool)) { const QCand t = e[j - 1]; e[j - 1] = e[j]; e[j] = t; }
                    }
                }
            }
        }
    win[0] = kNoIdx;
    win[1] = kNoIdx;
    for (int r = 0; r < 5; r++) {
        QCand best = e[0];
        for (int off = G / 2; off > 0; off >>= 1) {
            QCand o;
            o.d2 = __shfl_xor(best.d2, off, G);
            o.id = (uint32_t)__shfl_xor((int)best.id, off, G);
            if (q_less(o, best, pool)) best = o;
        }
        if (gl == r % G) win[r / G] = best.id;
        if (best.id != kNoIdx && e[0].id == best.id) {
            for (int k = 0; k < 4; k++) e[k] = e[k + 1];
            e[4] = {INFINITY, kNoIdx};
        }
    }
}

// One query by the G lanes of its group.  All lanes of the group call this together (`active` is group-uniform).
// Returns through nn_pts / nn_cnt / world_out exactly what knn.hip's kernel + its exact redo write.
template <int G, int KM>
__device__ inline uint32_t knn_query(const Slot* __restrict__ table, uint32_t mask, const float4* __restrict__ pool, float inv_res, float res,
                                     const StencilArgs& st, bool active, const float4 pw, int gl, uint32_t q, float4* __restrict__ nn_pts,
                                     uint32_t nn_stride, int32_t* __restrict__ nn_cnt) {
    int kx = 0, ky = 0, kz = 0;
    pos2grid(pw.x, pw.y, pw.z, inv_res, kx, ky, kz);
    // 1. probe: lane gl owns stencil cells gl, gl + G, ...; all home-slot loads first (independent, in flight together)
    uint4 raw[KM];
    BrickProbe bp[KM];
    unsigned long long want[KM];
    uint32_t vdmin[KM];
#pragma unroll
    for (int k = 0; k < KM; k++) {
        const int s = k * G + gl;
        want[k] = kEmptyKey;
        raw[k] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u);
        bp[k] = BrickProbe{0u, 1u, 0u};
        vdmin[k] = 0xFFFFFFFFu;
        if (active && s < st.n) {
            const int cx = kx + st.off[s][0], cy = ky + st.off[s][1], cz = kz + st.off[s][2];
            want[k] = pack_key(cx, cy, cz);
            bp[k] = brick_probe(cx, cy, cz);
            raw[k] = *reinterpret_cast<const uint4*>(&table[brick_slot(bp[k], mask)]);
            vdmin[k] = cell_min_d2_bits(pw.x, pw.y, pw.z, cx, cy, cz, res);
        }
    }
    uint32_t vptr[KM], vcnt[KM], total = 0;
#pragma unroll
    for (int k = 0; k < KM; k++) {
        vptr[k] = 0; vcnt[k] = 0;
        if (want[k] != kEmptyKey) {
            uint4 r = raw[k];
            for (uint32_t probe = 0; probe <= (mask >> 6); probe++) {  // double hashing by window; load factor <= 0.5
                const unsigned long long kk = ((unsigned long long)r.y << 32) | r.x;
                if (kk == want[k]) { vptr[k] = r.z; vcnt[k] = r.w; break; }
                if (kk == kEmptyKey) break;
                brick_next(bp[k]);
                r = *reinterpret_cast<const uint4*>(&table[brick_slot(bp[k], mask)]);
            }
        }
        total += vcnt[k];
    }
    total = qsum32<G>(total);
    // 2. sweep in stencil order with exact pruning
    Top5 t;
    t.clear();
    uint32_t inrange = 0, bound5 = 0xFFFFFFFFu;
    bool drop_tie = false;
#pragma unroll
    for (int k = 0; k < KM; k++) {
#pragma unroll
        for (int o = 0; o < G; o++) {
            if (k * G + o >= kMaxStencil) continue;
            const uint32_t c = qbcast<G>(vcnt[k], o);
            if (c == 0) continue;
            const uint32_t dm = qbcast<G>(vdmin[k], o);
            if (dm > bound5) continue;  // cannot hold one of the five nearest (nor tie with the fifth: strict)
            const uint32_t p = qbcast<G>(vptr[k], o);
            for (uint32_t i0 = gl; i0 < c + gl; i0 += 4 * G) {  // (c + gl: every lane of the group runs the same number of steps)
                float4 pt[4];
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (i0 + u * G < c) pt[u] = pool[p + i0 + u * G];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (i0 + u * G >= c) continue;
                    const float dx = pt[u].x - pw.x, dy = pt[u].y - pw.y, dz = pt[u].z - pw.z;
                    const float d2 = dx * dx + (dy * dy + dz * dz);  // ivox3d_node.hpp:12-15: Eigen's unrolled tree x0 + (x1 + x2)
                    if (d2 < 5.0f) {
                        inrange++;
                        const uint32_t kd = __float_as_uint(d2);
                        if (kd < t.d4) t.insert(kd, p + i0 + u * G);
                        else drop_tie |= kd == t.d4;
                    }
                }
            }
            bound5 = q_fifth_bound<G>(t.d0, t.d4);
        }
    }
    inrange = qsum32<G>(inrange);
    // 3. merge: six rounds pop the group's smallest (d2, index) head -- the global top-5 and the best loser
    uint32_t win[2] = {0xFFFFFFFFu, 0xFFFFFFFFu}, prev_d = 0xFFFFFFFFu;
    bool tie = false;
    int pops = 0;
#pragma unroll
    for (int r = 0; r < 6; r++) {
        const uint32_t bd = qmin32<G>(t.d0);
        const uint32_t bi = qmin32<G>(t.d0 == bd ? t.i0 : 0xFFFFFFFFu);
        const bool some = bi != 0xFFFFFFFFu;
        if (r > 0 && some && bd == prev_d) tie = true;
        prev_d = some ? bd : 0xFFFFFFFFu;
        if (r < 5 && gl == r % G) win[r / G] = bi;
        if (some && t.d0 == bd && t.i0 == bi) { t.pop(); pops++; }
    }
    if (qany<G>(pops >= 5 && drop_tie)) tie = true;
    if (tie && active && inrange > 0) {
        uint32_t vp2[KM], vc2[KM];  // copies: the redo indexes them at run time, the originals stay in registers
#pragma unroll
        for (int k = 0; k < KM; k++) { vp2[k] = vptr[k]; vc2[k] = vcnt[k]; }
        q_exact_redo<G, KM>(pool, pw, vp2, vc2, gl, win);
    }
    // 4. results.  No in-range candidate at all: GetClosestPoint returns before touching the output (ivox3d.h:152-154)
    if (active && inrange > 0) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int r = j * G + gl;
            if (r < 5) nn_pts[(size_t)r * nn_stride + q] = (win[j] != 0xFFFFFFFFu) ? pool[win[j]] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (gl == 0) nn_cnt[q] = inrange < 5 ? (int32_t)inrange : 5;
    }
    return total;
}

constexpr int kQThreads = 256;

template <int G, int KM, int MODE>
__device__ __forceinline__ void knn_q_body(const Slot* __restrict__ table, uint32_t mask, const float4* __restrict__ pool, float inv_res,
                                           const StencilArgs& st, const PoseArgs& pose, const float4* __restrict__ queries, uint32_t n,
                                           float4* __restrict__ world_out, float4* __restrict__ nn_pts, uint32_t nn_stride,
                                           int32_t* __restrict__ nn_cnt, MapDev* md) {
    constexpr int QPB = kQThreads / G;
    const int tid = threadIdx.x;
    const int grp = tid / G, gl = tid % G;
    const int lane = tid & 63;
    const float res = 1.0f / inv_res;
    unsigned long long visited = 0;
    // XCD-aware workgroup -> query mapping (see knn.hip): gridDim.x is a multiple of 8, XCD b % 8 gets one contiguous eighth
    const uint32_t per_xcd = gridDim.x >> 3;
    const uint32_t vb = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    for (uint32_t q0 = vb * QPB; q0 < n; q0 += gridDim.x * QPB) {
        const uint32_t q = q0 + grp;
        const bool active = q < n;
        float4 pw = make_float4(0.f, 0.f, 0.f, 0.f);
        if (active) {
            const float4 pq = queries[q];
            if (MODE == 0) {
                body_to_world(pose, pq, pw);
                if (gl == 0) world_out[q] = pw;
            } else {
                pw = pq;
            }
        }
        const uint32_t total = knn_query<G, KM>(table, mask, pool, inv_res, res, st, active, pw, gl, q, nn_pts, nn_stride, nn_cnt);
        if (gl == 0) visited += total;
    }
    if (md) {
        __shared__ unsigned long long vred[kQThreads / 64];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) visited += __shfl_xor(visited, off);
        if (lane == 0) vred[tid >> 6] = visited;
        __syncthreads();
        if (tid == 0) {
            const unsigned long long v = (vred[0] + vred[1]) + (vred[2] + vred[3]);
            if (v) atomicAdd(&md->knn_cand[(blockIdx.x & 63) * 16], v);
        }
    }
}

// world-frame queries (diagnostic lio_map_knn with LIO_KNN_Q=1: the adversarial tests of the sweep run against this kernel too)
template <int G, int KM>
__global__ void __launch_bounds__(kQThreads) knn_q_world_kernel(const Slot* __restrict__ table, uint32_t mask, const float4* __restrict__ pool,
                                                                float inv_res, StencilArgs st, const float4* __restrict__ queries, uint32_t n,
                                                                float4* __restrict__ nn_pts, uint32_t nn_stride, int32_t* __restrict__ nn_cnt,
                                                                MapDev* md) {
    PoseArgs pose;
    knn_q_body<G, KM, 1>(table, mask, pool, inv_res, st, pose, queries, n, nullptr, nn_pts, nn_stride, nn_cnt, md);
}

// the scans of a batch: blockIdx.y = slot; body-frame queries, pose from the slot's device-resident filter state; a slot whose
// update has finished, or whose filter did not ask for a neighbour search this pass, exits at once
template <int G, int KM>
__global__ void __launch_bounds__(kQThreads) knn_q_batch_kernel(const Slot* __restrict__ table, uint32_t mask, const float4* __restrict__ pool,
                                                                float inv_res, StencilArgs st, const SlotDesc* __restrict__ slots, MapDev* md) {
    const SlotDesc& d = slots[blockIdx.y];
    if (!d.active) return;
    const EskfDev* c = d.ctrl;
    if (c->status != EK_RUNNING || !c->converge || d.sd->n_ds < d.min_ds) return;
    const PoseArgs pose = pose_from_state(c->x);
    knn_q_body<G, KM, 0>(table, mask, pool, inv_res, st, pose, d.ds_body, d.sd->n_ds, d.ds_world, d.nn_pts, d.max_ds, d.nn_cnt, md);
}

int knn_q_batch(lio_map* m, hipStream_t st, const SlotDesc* d_slots, int n_slots, uint32_t grid_x) {
    const dim3 grid((grid_x + 7u) & ~7u, (uint32_t)n_slots);
    constexpr int G = 4;
#define KNNQ_LAUNCH(KM) \
    hipLaunchKernelGGL((knn_q_batch_kernel<G, KM>), grid, kQThreads, 0, st, m->table, m->table_mask, m->pool, m->inv_res, m->stencil, d_slots, m->dev)
    const int km = (m->stencil.n + G - 1) / G;
    if (km <= 1) KNNQ_LAUNCH(1);
    else if (km <= 2) KNNQ_LAUNCH(2);
    else if (km <= 5) KNNQ_LAUNCH(5);
    else if (km <= 7) KNNQ_LAUNCH(7);
    else KNNQ_LAUNCH((kMaxStencil + G - 1) / G);
#undef KNNQ_LAUNCH
    LIO_HIP_TRY(hipGetLastError());
    return LIO_OK;
}

int knn_q_world(lio_map* m, const float4* d_q, uint32_t n, float4* d_out, int32_t* d_cnt) {
    constexpr int G = 4;
    uint32_t blocks = (n + kQThreads / G - 1) / (kQThreads / G);
    if (blocks == 0) return LIO_OK;
    if (blocks > 8192) blocks = 8192;
    blocks = (blocks + 7u) & ~7u;
#define KNNQ_LAUNCH(KM) \
    hipLaunchKernelGGL((knn_q_world_kernel<G, KM>), blocks, kQThreads, 0, m->stream, m->table, m->table_mask, m->pool, m->inv_res, m->stencil, d_q, n, d_out, n, d_cnt, m->dev)
    const int km = (m->stencil.n + G - 1) / G;
    if (km <= 1) KNNQ_LAUNCH(1);
    else if (km <= 2) KNNQ_LAUNCH(2);
    else if (km <= 5) KNNQ_LAUNCH(5);
    else if (km <= 7) KNNQ_LAUNCH(7);
    else KNNQ_LAUNCH((kMaxStencil + G - 1) / G);
#undef KNNQ_LAUNCH
    LIO_HIP_TRY(hipGetLastError());
    return LIO_OK;
}

}  // namespace lio
