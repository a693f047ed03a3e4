// hashmap.hip -- the map: an iVox-equivalent voxel hash grid resident in HBM, and its stencil kNN.
//
// Replaces faster_lio::IVox (reference: /root/reference/slam/mapping/fastlio/include/ivox3d/ivox3d.h,
// ivox3d_node.hpp).  The reference keeps unordered_map<Vec3i, list::iterator> + list<pair<key, vector<Point>>>;
// here:
//   * one open-addressing table of 16-byte slots {key(63 bit), ptr, cnt}; a probe is one 16-B load;
//   * "brick-coherent" hashing: a 4x4x4 brick of voxels hashes to 64 consecutive slots (1 KiB), so the
//     19/75 probes of one stencil land in a handful of 128-B lines of the XCD's L2 instead of 19 random ones;
//   * every voxel's points are contiguous in one float4 pool (bump allocated; a voxel that outgrows its
//     region is moved to a twice larger one), so candidates stream in as coalesced 16-B lanes.
// kNN semantics (ivox3d.h:139-171 + ivox3d_node.hpp:107-127): the 5 nearest of all points stored in the
// stencil voxels with d^2 < 5.0 -- the per-voxel nth_element there is a pruning step that does not change
// that set.  Ties are broken by the canonical total order (d2, x, y, z) that oracle/lio_oracle.cpp uses.
#include "lio_common.h"

namespace lio {

__device__ __host__ inline unsigned long long pack_key(int x, int y, int z) {
    return ((unsigned long long)((uint32_t)x & 0x1FFFFFu)) | ((unsigned long long)((uint32_t)y & 0x1FFFFFu) << 21) |
           ((unsigned long long)((uint32_t)z & 0x1FFFFFu) << 42);
}

__device__ inline uint32_t brick_hash(int x, int y, int z, uint32_t mask) {
    unsigned long long b = pack_key(x >> 2, y >> 2, z >> 2);
    b *= 0x9E3779B97F4A7C15ull;
    b ^= b >> 29;
    b *= 0xBF58476D1CE4E5B9ull;
    b ^= b >> 32;
    const uint32_t local = (uint32_t)(x & 3) | ((uint32_t)(y & 3) << 2) | ((uint32_t)(z & 3) << 4);
    return (((uint32_t)b << 6) | local) & mask;
}

// ivox3d.h:258-261: Pos2Grid = round(p * inv_res) per axis (std::round: half away from zero), in f32
__device__ inline void pos2grid(float x, float y, float z, float inv_res, int& kx, int& ky, int& kz) {
    kx = (int)roundf(x * inv_res);
    ky = (int)roundf(y * inv_res);
    kz = (int)roundf(z * inv_res);
}

__device__ inline bool slot_lookup(const Slot* __restrict__ table, uint32_t mask, int x, int y, int z, uint32_t& ptr, uint32_t& cnt) {
    const unsigned long long key = pack_key(x, y, z);
    uint32_t h = brick_hash(x, y, z, mask);
    for (uint32_t probe = 0; probe <= mask; probe++) {
        const uint4 raw = *reinterpret_cast<const uint4*>(&table[h]);
        const unsigned long long k = ((unsigned long long)raw.y << 32) | raw.x;
        if (k == key) { ptr = raw.z; cnt = raw.w; return true; }
        if (k == kEmptyKey) return false;
        h = (h + 1) & mask;
    }
    return false;
}

// ---------------------------------------------------------------------------------------------------
// batch insert = IVox::AddPoints (ivox3d.h:231-256) without the LRU list
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) map_insert_claim_kernel(Slot* table, uint32_t mask, uint32_t* __restrict__ pending,
                                                               float* __restrict__ created, const float4* __restrict__ pts,
                                                               unsigned long long n_host, const uint32_t* __restrict__ n_dev,
                                                               float inv_res, float travel, uint32_t max_voxels, MapDev* md,
                                                               uint32_t* __restrict__ slot_of_point) {
    const unsigned long long n = n_dev ? (unsigned long long)*n_dev : n_host;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const float4 p = pts[i];
        int kx, ky, kz;
        pos2grid(p.x, p.y, p.z, inv_res, kx, ky, kz);
        const unsigned long long key = pack_key(kx, ky, kz);
        uint32_t h = brick_hash(kx, ky, kz, mask);
        uint32_t found = kNoIdx;
        for (uint32_t probe = 0; probe <= mask; probe++) {
            unsigned long long k = *reinterpret_cast<volatile unsigned long long*>(&table[h].key);
            if (k == kEmptyKey) {
                k = atomicCAS(&table[h].key, kEmptyKey, key);
                if (k == kEmptyKey) {  // this thread created the voxel
                    created[h] = travel;
                    const uint32_t nv = atomicAdd(&md->n_voxels, 1u) + 1u;
                    if (nv > max_voxels) atomicOr(&md->err, 4u);
                    k = key;
                }
            }
            if (k == key) { found = h; break; }
            h = (h + 1) & mask;
        }
        if (found == kNoIdx) {
            atomicOr(&md->err, 1u);
            slot_of_point[i] = kNoIdx;
            continue;
        }
        const uint32_t before = atomicAdd(&pending[found], 1u);
        // the first arriver of a voxel in this batch is its leader (bit 31) and sizes the region in pass 2
        slot_of_point[i] = found | (before == 0 ? 0x80000000u : 0u);
    }
}

__global__ void __launch_bounds__(256) map_insert_grow_kernel(Slot* table, uint32_t* __restrict__ cap, uint32_t* __restrict__ pending,
                                                              float4* pool, unsigned long long pool_cap, unsigned long long n_host,
                                                              const uint32_t* __restrict__ n_dev, MapDev* md,
                                                              const uint32_t* __restrict__ slot_of_point, int tight) {
    const unsigned long long n = n_dev ? (unsigned long long)*n_dev : n_host;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint32_t sp = slot_of_point[i];
        if (sp == kNoIdx || !(sp & 0x80000000u)) continue;
        const uint32_t h = sp & 0x7FFFFFFFu;
        const uint32_t have = table[h].cnt, add = pending[h], need = have + add;
        pending[h] = 0;
        atomicAdd(&md->n_points, (unsigned long long)add);
        if (need <= cap[h]) continue;
        uint32_t ncap = need;
        if (!tight) {  // leave room: the voxel is on the sensor's path and will be appended to again
            ncap = 8;
            while (ncap < need) ncap <<= 1;
        }
        const unsigned long long at = atomicAdd(&md->pool_top, (unsigned long long)ncap);
        if (at + ncap > pool_cap) {
            atomicOr(&md->err, 2u);
            continue;
        }
        const uint32_t old = table[h].ptr;
        for (uint32_t j = 0; j < have; j++) pool[at + j] = pool[old + j];
        table[h].ptr = (uint32_t)at;
        cap[h] = ncap;
    }
}

__global__ void __launch_bounds__(256) map_insert_write_kernel(Slot* table, const uint32_t* __restrict__ cap, float4* pool,
                                                               const float4* __restrict__ pts, unsigned long long n_host,
                                                               const uint32_t* __restrict__ n_dev,
                                                               const uint32_t* __restrict__ slot_of_point) {
    const unsigned long long n = n_dev ? (unsigned long long)*n_dev : n_host;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint32_t sp = slot_of_point[i];
        if (sp == kNoIdx) continue;
        const uint32_t h = sp & 0x7FFFFFFFu;
        const uint32_t idx = atomicAdd(&table[h].cnt, 1u);
        if (idx < cap[h]) pool[(unsigned long long)table[h].ptr + idx] = pts[i];
    }
}

int map_insert_dev(lio_map* m, hipStream_t stream, const float4* d_pts, uint64_t n, const uint32_t* d_n, double travel) {
    if (n == 0) return LIO_OK;
    if (n > m->slot_of_point_cap) {
        set_error("map insert batch of %llu exceeds scratch %llu", (unsigned long long)n, (unsigned long long)m->slot_of_point_cap);
        return LIO_E_CAPACITY;
    }
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    // the first batch into an empty map is a prebuilt-map load: size every voxel exactly (CSR-tight)
    const int tight = m->n_batches == 0 ? 1 : 0;
    m->n_batches++;
    hipLaunchKernelGGL(map_insert_claim_kernel, (uint32_t)blocks, 256, 0, stream, m->table, m->table_mask, m->pending, m->created,
                       d_pts, (unsigned long long)n, d_n, m->inv_res, (float)travel, (uint32_t)m->max_voxels, m->dev, m->slot_of_point);
    hipLaunchKernelGGL(map_insert_grow_kernel, (uint32_t)blocks, 256, 0, stream, m->table, m->cap, m->pending, m->pool,
                       (unsigned long long)m->pool_cap, (unsigned long long)n, d_n, m->dev, m->slot_of_point, tight);
    hipLaunchKernelGGL(map_insert_write_kernel, (uint32_t)blocks, 256, 0, stream, m->table, m->cap, m->pool, d_pts,
                       (unsigned long long)n, d_n, m->slot_of_point);
    LIO_HIP_TRY(hipGetLastError());
    return LIO_OK;
}

// ---------------------------------------------------------------------------------------------------
// stencil kNN: G lanes cooperate on one query
// ---------------------------------------------------------------------------------------------------
struct Cand {
    float d2;
    uint32_t id;  // pool index
};

// strict total order (d2, x, y, z); the coordinate comparison only runs on exact d2 ties
__device__ inline bool cand_less(const Cand& a, const Cand& b, const float4* __restrict__ pool) {
    if (a.d2 != b.d2) return a.d2 < b.d2;
    if (a.id == b.id) return false;
    if (a.id == kNoIdx || b.id == kNoIdx) return a.id < b.id;
    const float4 pa = pool[a.id], pb = pool[b.id];
    if (pa.x != pb.x) return pa.x < pb.x;
    if (pa.y != pb.y) return pa.y < pb.y;
    if (pa.z != pb.z) return pa.z < pb.z;
    return a.id < b.id;
}

__device__ inline void body_to_world(const PoseArgs& P, const float4 pb, float4& pw) {
    // laserMapping.cpp:831-836: p_global = rot * (offset_R_L_I * p_body + offset_T_L_I) + pos, in double, stored float.
    // Quaternion * vector as Eigen's _transformVector: uv = 2 (q.vec x v); v + w uv + q.vec x uv
    const double vx = (double)pb.x, vy = (double)pb.y, vz = (double)pb.z;
    double ux = P.ql[1] * vz - P.ql[2] * vy, uy = P.ql[2] * vx - P.ql[0] * vz, uz = P.ql[0] * vy - P.ql[1] * vx;
    ux += ux; uy += uy; uz += uz;
    double cx = P.ql[1] * uz - P.ql[2] * uy, cy = P.ql[2] * ux - P.ql[0] * uz, cz = P.ql[0] * uy - P.ql[1] * ux;
    const double ix = ((vx + P.ql[3] * ux) + cx) + P.tl[0];
    const double iy = ((vy + P.ql[3] * uy) + cy) + P.tl[1];
    const double iz = ((vz + P.ql[3] * uz) + cz) + P.tl[2];
    ux = P.qw[1] * iz - P.qw[2] * iy; uy = P.qw[2] * ix - P.qw[0] * iz; uz = P.qw[0] * iy - P.qw[1] * ix;
    ux += ux; uy += uy; uz += uz;
    cx = P.qw[1] * uz - P.qw[2] * uy; cy = P.qw[2] * ux - P.qw[0] * uz; cz = P.qw[0] * uy - P.qw[1] * ux;
    pw.x = (float)(((ix + P.qw[3] * ux) + cx) + P.tw[0]);
    pw.y = (float)(((iy + P.qw[3] * uy) + cy) + P.tw[1]);
    pw.z = (float)(((iz + P.qw[3] * uz) + cz) + P.tw[2]);
    pw.w = pb.w;
}

template <int G>
__device__ inline Cand group_min(Cand c, const float4* __restrict__ pool) {
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) {
        Cand o;
        o.d2 = __shfl_xor(c.d2, off);
        o.id = __shfl_xor(c.id, off);
        if (cand_less(o, c, pool)) c = o;
    }
    return c;
}

// MODE 0: queries are body-frame ds points of a scan (transformed here, world point stored);
// MODE 1: queries are world-frame points (diagnostic lio_map_knn).
template <int G, int MODE>
__global__ void __launch_bounds__(256) knn_kernel(const Slot* __restrict__ table, uint32_t mask, const float4* __restrict__ pool,
                                                  float inv_res, StencilArgs st, PoseArgs pose, const float4* __restrict__ queries,
                                                  uint32_t n_host, const ScanDev* __restrict__ sd, float4* __restrict__ world_out,
                                                  float4* __restrict__ nn_pts, uint32_t nn_stride, int32_t* __restrict__ nn_cnt,
                                                  MapDev* md) {
    constexpr int GPB = 256 / G;  // query groups per block
    __shared__ uint32_t v_ptr[GPB][kMaxStencil];
    __shared__ uint32_t v_end[GPB][kMaxStencil + 1];  // exclusive prefix of counts, v_end[g][0] = 0
    const int tid = threadIdx.x;
    const int grp = tid / G, gl = tid % G;
    const int lane = tid & 63;
    const uint32_t n = sd ? sd->n_ds : n_host;
    const unsigned long long gmask = (G == 64) ? ~0ull : (((1ull << G) - 1ull) << (lane - gl));
    unsigned long long visited = 0;

    for (uint32_t q0 = blockIdx.x * GPB; q0 < n; q0 += gridDim.x * GPB) {
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
        int kx = 0, ky = 0, kz = 0;
        pos2grid(pw.x, pw.y, pw.z, inv_res, kx, ky, kz);
        // 1. probe the stencil, compact the occupied voxels into LDS.  A lane owns cells gl, gl+G, ...; the
        //    home-slot loads of all its cells are issued back to back, collisions (rare at load <= 0.5) loop after.
        uint32_t nhit = 0;
        {
            constexpr int KM = (kMaxStencil + G - 1) / G;
            uint4 raw[KM];
            uint32_t hh[KM];
            unsigned long long want[KM];
#pragma unroll
            for (int k = 0; k < KM; k++) {
                const int s = k * G + gl;
                want[k] = kEmptyKey;
                raw[k] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u);
                if (active && s < st.n) {
                    const int cx = kx + st.off[s][0], cy = ky + st.off[s][1], cz = kz + st.off[s][2];
                    want[k] = pack_key(cx, cy, cz);
                    hh[k] = brick_hash(cx, cy, cz, mask);
                    raw[k] = *reinterpret_cast<const uint4*>(&table[hh[k]]);
                }
            }
#pragma unroll
            for (int k = 0; k < KM; k++) {
                if (k * G >= st.n) break;  // uniform
                bool hit = false;
                uint32_t ptr = 0, cnt = 0;
                if (want[k] != kEmptyKey) {
                    uint4 r = raw[k];
                    uint32_t h = hh[k];
                    for (uint32_t probe = 0; probe <= mask; probe++) {
                        const unsigned long long kk = ((unsigned long long)r.y << 32) | r.x;
                        if (kk == want[k]) { hit = r.w > 0; ptr = r.z; cnt = r.w; break; }
                        if (kk == kEmptyKey) break;
                        h = (h + 1) & mask;
                        r = *reinterpret_cast<const uint4*>(&table[h]);
                    }
                }
                const unsigned long long m = __ballot(hit) & gmask;
                if (hit) {
                    const uint32_t at = nhit + __popcll(m & ((1ull << lane) - 1ull));
                    v_ptr[grp][at] = ptr;
                    v_end[grp][at + 1] = cnt;
                }
                nhit += __popcll(m);
            }
        }
        __syncthreads();
        // 2. exclusive prefix over the voxel counts (<= 75 entries; one lane, the group is in lockstep)
        if (gl == 0) {
            uint32_t run = 0;
            v_end[grp][0] = 0;
            for (uint32_t j = 0; j < nhit; j++) {
                run += v_end[grp][j + 1];
                v_end[grp][j + 1] = run;
            }
        }
        __syncthreads();
        const uint32_t total = active ? v_end[grp][nhit] : 0;
        // 3. every lane keeps its own sorted top-5 over candidates gl, gl+G, ...; four candidate loads in flight
        Cand e0 = {INFINITY, kNoIdx}, e1 = e0, e2 = e0, e3 = e0, e4 = e0;
        uint32_t inrange = 0;
        uint32_t j = 0;
        constexpr int U = 4;
        for (uint32_t c0 = gl; c0 < total; c0 += U * G) {
            uint32_t id[U];
            float4 p[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t c = c0 + u * G;
                id[u] = kNoIdx;
                if (c < total) {
                    while (c >= v_end[grp][j + 1]) j++;
                    id[u] = v_ptr[grp][j] + (c - v_end[grp][j]);
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++)
                if (id[u] != kNoIdx) p[u] = pool[id[u]];
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (id[u] == kNoIdx) continue;
                const float dx = p[u].x - pw.x, dy = p[u].y - pw.y, dz = p[u].z - pw.z;
                const float d2 = (dx * dx + dy * dy) + dz * dz;  // ivox3d_node.hpp:12-15 (f32 squaredNorm)
                if (d2 < 5.0f) {
                    inrange++;
                    Cand cd = {d2, id[u]};
                    if (cand_less(cd, e4, pool)) {
                        e4 = cd;
                        if (cand_less(e4, e3, pool)) { Cand t = e3; e3 = e4; e4 = t; }
                        if (cand_less(e3, e2, pool)) { Cand t = e2; e2 = e3; e3 = t; }
                        if (cand_less(e2, e1, pool)) { Cand t = e1; e1 = e2; e2 = t; }
                        if (cand_less(e1, e0, pool)) { Cand t = e0; e0 = e1; e1 = t; }
                    }
                }
            }
        }
        visited += total > gl ? (total - gl + G - 1) / G : 0;
#pragma unroll
        for (int off = G / 2; off > 0; off >>= 1) inrange += __shfl_xor(inrange, off);
        // 4. merge: five rounds of group-wide argmin over the lanes' heads
        uint32_t win = kNoIdx;
#pragma unroll
        for (int r = 0; r < 5; r++) {
            const Cand best = group_min<G>(e0, pool);
            if (gl == r) win = best.id;
            if (best.id != kNoIdx && e0.id == best.id) { e0 = e1; e1 = e2; e2 = e3; e3 = e4; e4 = {INFINITY, kNoIdx}; }
        }
        // 5. results.  No in-range candidate at all: GetClosestPoint returns before touching the output
        //    (ivox3d.h:152-154), the cached neighbours of an earlier scan survive.
        if (active && inrange > 0) {
            if (gl < 5) nn_pts[(size_t)gl * nn_stride + q] = (win != kNoIdx) ? pool[win] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (gl == 0) nn_cnt[q] = inrange < 5 ? (int32_t)inrange : 5;
        }
        __syncthreads();
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) visited += __shfl_xor(visited, off);
    if (lane == 0 && visited) atomicAdd(&md->knn_cand[blockIdx.x & 63], visited);
}

template <int MODE>
static int launch_knn(lio_map* m, hipStream_t st, const PoseArgs& pose, const float4* q, uint32_t n_host, const ScanDev* sd,
                      float4* world_out, float4* nn_pts, uint32_t nn_stride, int32_t* nn_cnt, uint32_t n_bound) {
    constexpr int G = 16;
    constexpr int GPB = 256 / G;
    uint32_t blocks = (n_bound + GPB - 1) / GPB;
    if (blocks > 8192) blocks = 8192;
    if (blocks == 0) return LIO_OK;
    hipLaunchKernelGGL((knn_kernel<G, MODE>), blocks, 256, 0, st, m->table, m->table_mask, m->pool, m->inv_res, m->stencil, pose, q,
                       n_host, sd, world_out, nn_pts, nn_stride, nn_cnt, m->dev);
    LIO_HIP_TRY(hipGetLastError());
    return LIO_OK;
}

int map_knn_plane(lio_map* m, lio_scan* s, const PoseArgs& pose, int redo_knn) {
    (void)redo_knn;
    kt_begin(s, 0);
    const int rc = launch_knn<0>(m, s->stream, pose, s->ds_body, 0, s->dev, s->ds_world, s->nn_pts, s->max_ds, s->nn_cnt, s->max_ds < s->n_raw || !s->n_raw ? s->max_ds : s->n_raw);
    kt_end(s, 0);
    return rc;
}

int knn_batch(lio_map* m, const float4* d_q, uint32_t n, float4* d_out, int32_t* d_cnt) {
    PoseArgs pose;
    memset(&pose, 0, sizeof(pose));
    return launch_knn<1>(m, m->stream, pose, d_q, n, nullptr, nullptr, d_out, n, d_cnt, n);
}

}  // namespace lio
