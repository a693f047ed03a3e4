// hashmap.hip -- the map: an iVox-equivalent voxel hash grid resident in HBM (layout + batch insert; the
// stencil kNN that reads it is in knn.hip).
//
// Replaces faster_lio::IVox (reference: /root/reference/slam/mapping/fastlio/include/ivox3d/ivox3d.h,
// ivox3d_node.hpp).  The reference keeps unordered_map<Vec3i, list::iterator> + list<pair<key, vector<Point>>>;
// here:
//   * one open-addressing table of 16-byte slots {key(63 bit), ptr, cnt}; a probe is one 16-B load;
//   * "brick-coherent" hashing: a 4x4x4 brick of voxels hashes to 64 consecutive slots (1 KiB), so the
//     19/75 probes of one stencil land in a handful of 128-B lines of the XCD's L2 instead of 19 random ones;
//   * every voxel's points are contiguous in one float4 pool, so candidates stream in as coalesced 16-B lanes.
//     A prebuilt map (first batch into an empty map) is laid out by an exclusive scan over the table in slot
//     order -- exact sizes, and the voxels of a brick end up adjacent in the pool (same pages, same L2 lines
//     for neighbouring queries).  Later batches append in place; a voxel that outgrows its region moves to a
//     region twice as large taken from a bump allocator.
// kNN semantics (ivox3d.h:139-171 + ivox3d_node.hpp:107-127): the 5 nearest of all points stored in the
// stencil voxels with d^2 < 5.0 -- the per-voxel nth_element there is a pruning step that does not change
// that set.  Ties are broken by the canonical total order (d2, x, y, z) that oracle/lio_oracle.cpp uses.
#include "hashgrid.h"
#include "lio_common.h"

namespace lio {

// ---------------------------------------------------------------------------------------------------
// batch insert = IVox::AddPoints (ivox3d.h:231-256) without the LRU list
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) map_insert_claim_kernel(Slot* table, uint32_t mask, uint32_t* __restrict__ pending,
                                                               float* __restrict__ created, const float4* __restrict__ pts,
                                                               unsigned long long n_host, const uint32_t* __restrict__ n_dev,
                                                               float inv_res, float res, int key_mode, float travel, uint32_t max_voxels,
                                                               MapDev* md, uint32_t* __restrict__ slot_of_point) {
    const unsigned long long n = n_dev ? (unsigned long long)*n_dev : n_host;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const float4 p = pts[i];
        int kx, ky, kz;
        if (key_mode == 0) pos2grid(p.x, p.y, p.z, inv_res, kx, ky, kz);
        else pos2grid_ndt(p.x, p.y, p.z, res, kx, ky, kz);
        const unsigned long long key = pack_key(kx, ky, kz);
        BrickProbe bp = brick_probe(kx, ky, kz);
        uint32_t found = kNoIdx;
        for (uint32_t probe = 0; probe <= (mask >> 6); probe++) {
            const uint32_t h = brick_slot(bp, mask);
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
            brick_next(bp);
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
                                                              const uint32_t* __restrict__ slot_of_point) {
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
        uint32_t ncap = 8;  // leave room: the voxel is on the sensor's path and will be appended to again
        while (ncap < need) ncap <<= 1;
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

// prebuilt-map layout: exclusive scan of the per-slot point counts in table order (three launches)
constexpr int kScanItems = 8;
constexpr int kScanTile = 256 * kScanItems;

__global__ void __launch_bounds__(256) map_layout_sums_kernel(const uint32_t* __restrict__ pending, uint32_t table_cap,
                                                              unsigned long long* __restrict__ tile_sum) {
    __shared__ unsigned long long red[4];
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    unsigned long long s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; k++) s += (base + k < table_cap) ? pending[base + k] : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(1024) map_layout_scan_kernel(unsigned long long* __restrict__ tile_sum, uint32_t ntiles, MapDev* md,
                                                               unsigned long long pool_cap) {
    // one workgroup, sequential chunks of 1024 tiles (a 2^30-slot table has 2^19 tiles: 512 chunks at most)
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned long long carry_s;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid == 0) carry_s = md->pool_top;
    __syncthreads();
    for (uint32_t c0 = 0; c0 < ntiles; c0 += 1024) {
        const uint32_t t = c0 + tid;
        const unsigned long long val = t < ntiles ? tile_sum[t] : 0ull;
        unsigned long long inc = val;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long u = __shfl_up(inc, off);
            if (lane >= off) inc += u;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        unsigned long long base = carry_s;
        for (int w = 0; w < wave; w++) base += wsum[w];
        if (t < ntiles) tile_sum[t] = base + inc - val;
        __syncthreads();
        if (tid == 1023) carry_s = base + inc;
        __syncthreads();
    }
    if (tid == 0) {
        const unsigned long long total = carry_s - md->pool_top;
        if (carry_s > pool_cap) md->err |= 2u;
        md->pool_top = carry_s;
        md->n_points += total;
    }
}

__global__ void __launch_bounds__(256) map_layout_assign_kernel(Slot* table, uint32_t* __restrict__ cap, uint32_t* __restrict__ pending,
                                                                uint32_t table_cap, const unsigned long long* __restrict__ tile_base) {
    __shared__ uint32_t wsum[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t base = blockIdx.x * kScanTile + tid * kScanItems;
    uint32_t c[kScanItems], s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; k++) {
        c[k] = (base + k < table_cap) ? pending[base + k] : 0u;
        s += c[k];
    }
    uint32_t inc = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(inc, off);
        if (lane >= off) inc += u;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned long long at = tile_base[blockIdx.x] + (inc - s);
    for (int w = 0; w < wave; w++) at += wsum[w];
#pragma unroll
    for (int k = 0; k < kScanItems; k++) {
        if (c[k]) {
            table[base + k].ptr = (uint32_t)at;
            cap[base + k] = c[k];
            pending[base + k] = 0;
            at += c[k];
        }
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
    // the first batch into an empty map is a prebuilt-map load: exact sizes, brick-coherent pool order
    const bool layout = m->n_batches == 0;
    m->n_batches++;
    hipLaunchKernelGGL(map_insert_claim_kernel, (uint32_t)blocks, 256, 0, stream, m->table, m->table_mask, m->pending, m->created,
                       d_pts, (unsigned long long)n, d_n, m->inv_res, m->res, m->key_mode, (float)travel, (uint32_t)m->max_voxels, m->dev,
                       m->slot_of_point);
    if (layout) {
        const uint32_t ntiles = (m->table_cap + kScanTile - 1) / kScanTile;
        hipLaunchKernelGGL(map_layout_sums_kernel, ntiles, 256, 0, stream, m->pending, m->table_cap, m->tile_sum);
        hipLaunchKernelGGL(map_layout_scan_kernel, 1, 1024, 0, stream, m->tile_sum, ntiles, m->dev, (unsigned long long)m->pool_cap);
        hipLaunchKernelGGL(map_layout_assign_kernel, ntiles, 256, 0, stream, m->table, m->cap, m->pending, m->table_cap, m->tile_sum);
    } else {
        hipLaunchKernelGGL(map_insert_grow_kernel, (uint32_t)blocks, 256, 0, stream, m->table, m->cap, m->pending, m->pool,
                           (unsigned long long)m->pool_cap, (unsigned long long)n, d_n, m->dev, m->slot_of_point);
    }
    hipLaunchKernelGGL(map_insert_write_kernel, (uint32_t)blocks, 256, 0, stream, m->table, m->cap, m->pool, d_pts,
                       (unsigned long long)n, d_n, m->slot_of_point);
    LIO_HIP_TRY(hipGetLastError());
    return LIO_OK;
}

}  // namespace lio
