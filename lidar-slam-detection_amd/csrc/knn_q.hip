// knn_q.hip -- stencil kNN, throughput form: ONE lane per query (64 queries per wave).
//
// Same semantics as knn.hip (IVox::GetClosestPoint(pt, out, 5, 5.0), /root/reference/slam/mapping/fastlio/include/ivox3d/
// ivox3d.h:139-171 + ivox3d_node.hpp:107-127: the 5 nearest of all points stored in the stencil voxels with d^2 < 5.0, canonical
// order (d2, x, y, z), fewer than five -> all, none -> output untouched), a different mapping to the machine.  knn.hip gives a
// query 16 lanes: the right shape when ONE scan has to fill the chip, but every per-query step (f64 transform, hashing, list
// compaction, merge, output) is then executed by 16 lanes for one query (~300 wave instructions per query) and the kernel is bound
// by instruction issue and by the dependent chain of its waves, not by bytes.  When scans are registered in batches (lio_batch_*:
// one launch serves B scans, ~18 000 queries each) there are enough queries to give each its own lane:
//   * the lane probes its stencil cells five at a time (five independent 16-B slot loads in flight) and appends the occupied ones
//     to a private list in LDS (column layout [entry][lane]: conflict free), in stencil order -- home cell, the six face
//     neighbours, the edge neighbours: already near to far;
//   * ONE flattened loop walks list and points together: four consecutive points (64 contiguous bytes) per step, the next step's
//     loads issued before the current points are processed; a voxel whose lower distance bound exceeds the lane's current fifth
//     nearest is skipped (exact: strict comparison, same sets / ties / counts as the unpruned sweep).  Lanes of a wave advance
//     through their own lists independently, so the wave's time is the longest list, not the sum of per-step maxima (a first
//     version with four lanes per query walking the stencil in lock step was 4x SLOWER than knn.hip for exactly that reason);
//   * sorted top-5 per lane (v_med3 insertion), no cross-lane traffic at all;
//   * an exact d2 tie among the kept five or with a dropped candidate (the only case where (d2, pool index) order can differ from
//     the canonical order) is redone in place with the full comparison -- no tie queue, no second launch.
// Used by the batched engine (batch.hip) and, for the tests' sake, selectable for lio_map_knn (LIO_KNN_Q=1).
#include "hashgrid.h"
#include "knn_dev.h"
#include "lio_common.h"

namespace lio {

__device__ inline uint32_t q_med3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// sorted top-5 of (d2 bits, pool index), ordered by d2 alone (arrival order on equal d2: such a query is redone exactly)
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
};

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

constexpr int kQWave = 64;   // one wave per workgroup: the LDS lists are private to lanes, no barrier anywhere
constexpr int kQChunk = 5;   // stencil cells probed together

// the lane's list of occupied stencil voxels: entry e of lane l at [e][l]
template <int KMAX>
struct __attribute__((aligned(16))) QList {
    uint32_t ptr[KMAX][kQWave];
    uint32_t cnt[KMAX][kQWave];
    uint32_t dmin[KMAX][kQWave];
};

// exact redo: every listed voxel, strict total order (d2, x, y, z); result = pool indices in canonical order
template <int KMAX>
__device__ __noinline__ void q_exact_redo(const float4* __restrict__ pool, const float4 pw, const QList<KMAX>& L, int lane, uint32_t nh, uint32_t* win) {
    QCand e[5];
#pragma unroll
    for (int k = 0; k < 5; k++) e[k] = {INFINITY, kNoIdx};
    for (uint32_t j = 0; j < nh; j++) {
        const uint32_t p = L.ptr[j][lane], c = L.cnt[j][lane];
        for (uint32_t i = 0; i < c; i++) {
            const uint32_t id = p + i;
            const float4 q = pool[id];
            const float dx = q.x - pw.x, dy = q.y - pw.y, dz = q.z - pw.z;
            const float d2 = dx * dx + (dy * dy + dz * dz);
            if (d2 < 5.0f) {
                QCand cd = {d2, id};
                if (q_less(cd, e[4], pool)) {
                    e[4] = cd;
                    for (int k = 4; k > 0; k--)
                        if (q_less(e[k], e[k - 1], pool)) { const QCand t = e[k - 1]; e[k - 1] = e[k]; e[k] = t; }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 5; k++) win[k] = e[k].id;
}

// One query per lane.  `active` false: the lane idles through (no loads, no stores).
template <int KMAX>
__device__ inline uint32_t knn_query(const Slot* __restrict__ table, uint32_t mask, const float4* __restrict__ pool, float inv_res, float res,
                                     const StencilArgs& st, bool active, const float4 pw, int lane, uint32_t q, QList<KMAX>& L,
                                     float4* __restrict__ nn_pts, uint32_t nn_stride, int32_t* __restrict__ nn_cnt) {
    int kx = 0, ky = 0, kz = 0;
    pos2grid(pw.x, pw.y, pw.z, inv_res, kx, ky, kz);
    // 1. probe the stencil, kQChunk cells at a time (independent home-slot loads in flight together), append the occupied ones
    uint32_t nh = 0, total = 0;
    const int ncell = active ? st.n : 0;
    for (int s0 = 0; s0 < ncell; s0 += kQChunk) {
        uint4 raw[kQChunk];
        BrickProbe bp[kQChunk];
        unsigned long long want[kQChunk];
        uint32_t dm[kQChunk];
#pragma unroll
        for (int k = 0; k < kQChunk; k++) {
            const int s = s0 + k;
            want[k] = kEmptyKey;
            raw[k] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u);
            bp[k] = BrickProbe{0u, 1u, 0u};
            dm[k] = 0xFFFFFFFFu;
            if (s < ncell) {
                const int cx = kx + st.off[s][0], cy = ky + st.off[s][1], cz = kz + st.off[s][2];
                want[k] = pack_key(cx, cy, cz);
                bp[k] = brick_probe(cx, cy, cz);
                raw[k] = *reinterpret_cast<const uint4*>(&table[brick_slot(bp[k], mask)]);
                dm[k] = cell_min_d2_bits(pw.x, pw.y, pw.z, cx, cy, cz, res);
            }
        }
#pragma unroll
        for (int k = 0; k < kQChunk; k++) {
            if (want[k] == kEmptyKey) continue;
            uint4 r = raw[k];
            uint32_t ptr = 0, cnt = 0;
            for (uint32_t probe = 0; probe <= (mask >> 6); probe++) {  // double hashing by window; load factor <= 0.5
                const unsigned long long kk = ((unsigned long long)r.y << 32) | r.x;
                if (kk == want[k]) { ptr = r.z; cnt = r.w; break; }
                if (kk == kEmptyKey) break;
                brick_next(bp[k]);
                r = *reinterpret_cast<const uint4*>(&table[brick_slot(bp[k], mask)]);
            }
            if (cnt) {
                L.ptr[nh][lane] = ptr;
                L.cnt[nh][lane] = cnt;
                L.dmin[nh][lane] = dm[k];
                nh++;
                total += cnt;
            }
        }
    }
    // 2. one flattened loop over (voxel, four points): the lane holds its position in its list and the loads of the next step
    Top5 t;
    t.clear();
    uint32_t inrange = 0;
    bool drop_tie = false;
    uint32_t j = 0, i = 0, cptr = 0, ccnt = 0;
    // the next voxel that the lane's current fifth nearest lets through (d4 is all ones while fewer than five are known): exact pruning
    auto next_voxel = [&]() -> bool {
        while (j < nh) {
            const uint32_t dmn = L.dmin[j][lane];
            if (dmn <= t.d4) {
                cptr = L.ptr[j][lane];
                ccnt = L.cnt[j][lane];
                i = 0;
                j++;
                return true;
            }
            j++;
        }
        return false;
    };
    bool have = next_voxel();
    float4 cur[4];
    uint32_t cur_base = 0, cur_n = 0;
    if (have) {
        cur_base = cptr;
        cur_n = ccnt < 4u ? ccnt : 4u;
#pragma unroll
        for (int u = 0; u < 4; u++)
            if ((uint32_t)u < cur_n) cur[u] = pool[cptr + u];
        i = 4;
    }
    while (have) {
        // issue the next step's loads first: the rest of this voxel, or the head of the next one that the CURRENT bound lets through
        // (the bound can only tighten while the current points are processed; a voxel admitted here and no longer needed afterwards
        // costs four loads, never a wrong answer -- pruning is an optimisation, the comparisons below decide)
        float4 nxt[4];
        uint32_t nxt_base = 0, nxt_n = 0;
        bool more = true;
        if (i >= ccnt) more = next_voxel();
        if (more) {
            nxt_base = cptr + i;
            const uint32_t left = ccnt - i;
            nxt_n = left < 4u ? left : 4u;
#pragma unroll
            for (int u = 0; u < 4; u++)
                if ((uint32_t)u < nxt_n) nxt[u] = pool[nxt_base + u];
            i += 4;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if ((uint32_t)u >= cur_n) continue;
            const float dx = cur[u].x - pw.x, dy = cur[u].y - pw.y, dz = cur[u].z - pw.z;
            const float d2 = dx * dx + (dy * dy + dz * dz);  // ivox3d_node.hpp:12-15: Eigen's unrolled tree x0 + (x1 + x2)
            if (d2 < 5.0f) {
                inrange++;
                const uint32_t kd = __float_as_uint(d2);
                if (kd < t.d4) {
                    const uint32_t out = t.d4;  // the entry this insertion pushes off the list (all ones: the list was not full)
                    t.insert(kd, cur_base + u);
                    drop_tie |= out != 0xFFFFFFFFu && out == t.d4;  // pushed out, yet exactly as far as the new fifth
                } else {
                    drop_tie |= kd == t.d4;  // not kept, yet exactly as far as the fifth
                }
            }
        }
        have = more;
#pragma unroll
        for (int u = 0; u < 4; u++) cur[u] = nxt[u];
        cur_base = nxt_base;
        cur_n = nxt_n;
    }
    // 3. results.  No in-range candidate at all: GetClosestPoint returns before touching the output (ivox3d.h:152-154)
    if (active && inrange > 0) {
        uint32_t win[5] = {t.i0, t.i1, t.i2, t.i3, t.i4};
        // equal d2 among the five kept, or a dropped candidate as far as the fifth: (d2, arrival) order may differ from (d2, x, y, z)
        const bool tie = drop_tie || (t.d1 != 0xFFFFFFFFu && t.d0 == t.d1) || (t.d2 != 0xFFFFFFFFu && t.d1 == t.d2) ||
                         (t.d3 != 0xFFFFFFFFu && t.d2 == t.d3) || (t.d4 != 0xFFFFFFFFu && t.d3 == t.d4);
        if (tie) {
            uint32_t w2[5];
            q_exact_redo<KMAX>(pool, pw, L, lane, nh, w2);
#pragma unroll
            for (int r = 0; r < 5; r++) win[r] = w2[r];
        }
#pragma unroll
        for (int r = 0; r < 5; r++) nn_pts[(size_t)r * nn_stride + q] = (win[r] != 0xFFFFFFFFu) ? pool[win[r]] : make_float4(0.f, 0.f, 0.f, 0.f);
        nn_cnt[q] = inrange < 5 ? (int32_t)inrange : 5;
    }
    return total;
}

template <int KMAX, int MODE>
__device__ __forceinline__ void knn_q_body(const Slot* __restrict__ table, uint32_t mask, const float4* __restrict__ pool, float inv_res,
                                           const StencilArgs& st, const PoseArgs& pose, const float4* __restrict__ queries, uint32_t n,
                                           float4* __restrict__ world_out, float4* __restrict__ nn_pts, uint32_t nn_stride,
                                           int32_t* __restrict__ nn_cnt, MapDev* md) {
    __shared__ QList<KMAX> L;
    const int lane = threadIdx.x;
    const float res = 1.0f / inv_res;
    unsigned long long visited = 0;
    // XCD-aware workgroup -> query mapping (see knn.hip): gridDim.x is a multiple of 8, XCD b % 8 gets one contiguous eighth
    // (eighths of the queries, not of the grid: the launch is sized for the largest scan it may meet)
    const uint32_t nblk = (n + kQWave - 1) / kQWave;
    const uint32_t per_xcd = (nblk + 7u) >> 3;
    const uint32_t wg_per_xcd = gridDim.x >> 3;
    for (uint32_t j = blockIdx.x >> 3; j < per_xcd; j += wg_per_xcd) {
        const uint32_t q0 = ((blockIdx.x & 7u) * per_xcd + j) * kQWave;
        const uint32_t q = q0 + lane;
        const bool active = q < n;
        float4 pw = make_float4(0.f, 0.f, 0.f, 0.f);
        if (active) {
            const float4 pq = queries[q];
            if (MODE == 0) {
                body_to_world(pose, pq, pw);
                world_out[q] = pw;
            } else {
                pw = pq;
            }
        }
        visited += knn_query<KMAX>(table, mask, pool, inv_res, res, st, active, pw, lane, q, L, nn_pts, nn_stride, nn_cnt);
    }
    if (md) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) visited += __shfl_xor(visited, off);
        if (lane == 0 && visited) atomicAdd(&md->knn_cand[(blockIdx.x & 63) * 16], visited);
    }
}

// world-frame queries (diagnostic lio_map_knn with LIO_KNN_Q=1: the adversarial tests of the sweep run against this kernel too)
template <int KMAX>
__global__ void __launch_bounds__(kQWave) knn_q_world_kernel(const Slot* __restrict__ table, uint32_t mask, const float4* __restrict__ pool,
                                                             float inv_res, StencilArgs st, const float4* __restrict__ queries, uint32_t n,
                                                             float4* __restrict__ nn_pts, uint32_t nn_stride, int32_t* __restrict__ nn_cnt,
                                                             MapDev* md) {
    PoseArgs pose;
    knn_q_body<KMAX, 1>(table, mask, pool, inv_res, st, pose, queries, n, nullptr, nn_pts, nn_stride, nn_cnt, md);
}

// the scans of a batch: blockIdx.y = slot; body-frame queries, pose from the slot's device-resident filter state; a slot whose
// update has finished, or whose filter did not ask for a neighbour search this pass, exits at once
template <int KMAX>
__global__ void __launch_bounds__(kQWave) knn_q_batch_kernel(const Slot* __restrict__ table, uint32_t mask, const float4* __restrict__ pool,
                                                             float inv_res, StencilArgs st, const SlotDesc* __restrict__ slots, MapDev* md) {
    const SlotDesc& d = slots[blockIdx.y];
    if (!d.active) return;
    const EskfDev* c = d.ctrl;
    if (c->status != EK_RUNNING || !c->converge || d.sd->n_ds < d.min_ds) return;
    const PoseArgs pose = pose_from_state(c->x);
    knn_q_body<KMAX, 0>(table, mask, pool, inv_res, st, pose, d.ds_body, d.sd->n_ds, d.ds_world, d.nn_pts, d.max_ds, d.nn_cnt, md);
}

int knn_q_batch(lio_map* m, hipStream_t st, const SlotDesc* d_slots, int n_slots, uint32_t grid_x) {
    const dim3 grid((grid_x + 7u) & ~7u, (uint32_t)n_slots);
    if (m->stencil.n <= 19)
        hipLaunchKernelGGL((knn_q_batch_kernel<19>), grid, kQWave, 0, st, m->table, m->table_mask, m->pool, m->inv_res, m->stencil, d_slots, m->dev);
    else
        hipLaunchKernelGGL((knn_q_batch_kernel<kMaxStencil>), grid, kQWave, 0, st, m->table, m->table_mask, m->pool, m->inv_res, m->stencil, d_slots, m->dev);
    LIO_HIP_TRY(hipGetLastError());
    return LIO_OK;
}

int knn_q_world(lio_map* m, const float4* d_q, uint32_t n, float4* d_out, int32_t* d_cnt) {
    uint32_t blocks = (n + kQWave - 1) / kQWave;
    if (blocks == 0) return LIO_OK;
    if (blocks > 16384) blocks = 16384;
    blocks = (blocks + 7u) & ~7u;
    if (m->stencil.n <= 19)
        hipLaunchKernelGGL((knn_q_world_kernel<19>), blocks, kQWave, 0, m->stream, m->table, m->table_mask, m->pool, m->inv_res, m->stencil, d_q, n, d_out, n,
                           d_cnt, m->dev);
    else
        hipLaunchKernelGGL((knn_q_world_kernel<kMaxStencil>), blocks, kQWave, 0, m->stream, m->table, m->table_mask, m->pool, m->inv_res, m->stencil, d_q, n,
                           d_out, n, d_cnt, m->dev);
    LIO_HIP_TRY(hipGetLastError());
    return LIO_OK;
}

}  // namespace lio
