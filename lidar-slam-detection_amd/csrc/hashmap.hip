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
#include <utility>

#include "hashgrid.h"
#include "lio_common.h"

namespace lio {

// ---------------------------------------------------------------------------------------------------
// batch insert = IVox::AddPoints (ivox3d.h:231-256); the LRU list is further down
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void map_insert_claim_body(Slot* table, uint32_t mask, uint32_t* __restrict__ pending,
                                                               float* __restrict__ created, const float4* __restrict__ pts,
                                                               unsigned long long n_host, const uint32_t* __restrict__ n_dev,
                                                               float inv_res, float res, int key_mode, float travel, uint32_t max_voxels,
                                                               MapDev* md, uint32_t* __restrict__ slot_of_point,
                                                               unsigned long long* __restrict__ touch, unsigned long long* __restrict__ prev_touch,
                                                               unsigned long long stamp_base, unsigned long long* __restrict__ first_touch = nullptr) {
    const unsigned long long n = n_dev ? (unsigned long long)*n_dev : n_host;
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // the batch's sequence numbers (read by the write kernel, a later launch): seq_acc .. seq_acc + n
        const uint32_t base = md->seq_acc;
        md->seq_cur = base;
        md->seq_acc = base + (uint32_t)n;
    }
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const float4 p = pts[i];
        int kx, ky, kz;
        if (key_mode == 0) pos2grid(p.x, p.y, p.z, inv_res, kx, ky, kz);
        else if (key_mode == 2) pos2grid_vgicp(p.x, p.y, p.z, (double)res, kx, ky, kz);
        else pos2grid_ndt(p.x, p.y, p.z, res, kx, ky, kz);
        const unsigned long long key = pack_key(kx, ky, kz);
        BrickProbe bp = brick_probe(kx, ky, kz);
        uint32_t found = kNoIdx;
        bool made = false;
        for (uint32_t probe = 0; probe <= (mask >> 6); probe++) {
            const uint32_t h = brick_slot(bp, mask);
            unsigned long long k = *reinterpret_cast<volatile unsigned long long*>(&table[h].key);
            if (k == kEmptyKey) {
                k = atomicCAS(&table[h].key, kEmptyKey, key);
                if (k == kEmptyKey) {  // this thread created the voxel
                    created[h] = travel;
                    made = true;
                    k = key;
                }
            }
            if (k == key) { found = h; break; }
            brick_next(bp);
        }
        {   // the voxel count: one atomic per wave for the voxels its lanes created (one each serialised a few hundred same-address atomics per scan)
            const unsigned long long mm = __ballot(made);
            if (mm) {
                const int lane = threadIdx.x & 63, leader = __ffsll((long long)mm) - 1;
                uint32_t base = 0;
                if (lane == leader) base = atomicAdd(&md->n_voxels, (uint32_t)__popcll(mm));
                base = __shfl(base, leader);
                if (made && base + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull)) + 1u > max_voxels) atomicOr(&md->err, 4u);
            }
        }
        if (found == kNoIdx) {
            atomicOr(&md->err, 1u);
            slot_of_point[i] = kNoIdx;
            continue;
        }
        if (touch) {  // the LAST point of the batch that lands in a voxel decides its LRU position
            const unsigned long long old = atomicMax(&touch[found], stamp_base + i);
            if (old < stamp_base) prev_touch[found] = old;  // exactly one thread per voxel and batch sees the stamp of an earlier batch
            // ... and the FIRST point of the batch in the voxel (a later batch beats an earlier one, inside a batch the smaller index wins): when the
            // reference's point-by-point order would have dropped the voxel before that point, it re-creates it there (lru_exact_*)
            if (first_touch) atomicMax(&first_touch[found], stamp_base + (kStampIdxMask - i));
        }
        const uint32_t before = atomicAdd(&pending[found], 1u);
        // the first arriver of a voxel in this batch is its leader (bit 31) and sizes the region in pass 2
        slot_of_point[i] = found | (before == 0 ? 0x80000000u : 0u);
    }
}

__device__ __forceinline__ void map_insert_grow_body(Slot* table, uint32_t* __restrict__ cap, uint32_t* __restrict__ pending,
                                                              float4* pool, uint32_t* seq, unsigned long long pool_cap, unsigned long long n_host,
                                                              const uint32_t* __restrict__ n_dev, MapDev* md,
                                                              const uint32_t* __restrict__ slot_of_point, const uint32_t* __restrict__ free_items,
                                                              uint32_t free_cap, uint32_t* __restrict__ free_in) {
    const unsigned long long n = n_dev ? (unsigned long long)*n_dev : n_host;
    // the map's point count: one atomic per wave (the leaders' additions summed by shuffles first) -- one per touched voxel serialised ~1 000 same-address
    // atomics per scan at ~12 ns each, most of this kernel's 14 us (round 4)
    unsigned long long added = 0;
    for (unsigned long long i0 = blockIdx.x * (unsigned long long)blockDim.x; i0 < n; i0 += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned long long i = i0 + threadIdx.x;
        if (i >= n) continue;
        const uint32_t sp = slot_of_point[i];
        if (sp == kNoIdx || !(sp & 0x80000000u)) continue;
        const uint32_t h = sp & 0x7FFFFFFFu;
        const uint32_t have = table[h].cnt, add = pending[h], need = have + add;
        pending[h] = 0;
        added += add;
        if (need <= cap[h]) continue;
        uint32_t ncap = 8;  // leave room: the voxel is on the sensor's path and will be appended to again
        while (ncap < need) ncap <<= 1;
        unsigned long long at = ~0ull;
        if (free_items) {  // a region an evicted or outgrown voxel gave back (this kernel only pops these lists; what it frees itself
                           // goes to the incoming lists below, which lru_evict_kernel folds in after this kernel has ended)
            const int c = 31 - __clz(ncap);
            // look before popping: the counters share cache lines and same-line atomics serialise (~10 ns each) -- with empty
            // lists (no eviction yet) two atomics per growing voxel cost 50 us per scan
            if (*reinterpret_cast<volatile int*>(&md->free_top[c]) > 0) {
                const int t = atomicSub(&md->free_top[c], 1);
                if (t > 0) at = free_items[(size_t)c * free_cap + (uint32_t)(t - 1)];
                else atomicAdd(&md->free_top[c], 1);
            }
        }
        if (at == ~0ull) {
            at = atomicAdd(&md->pool_top, (unsigned long long)ncap);
            if (at + ncap > pool_cap) {
                // pool exhausted: the voxel keeps its region; the write kernel stores what still fits and clamps cnt to cap, the
                // points that do not fit are taken out of the count again (the error bit is sticky, the table stays consistent)
                atomicOr(&md->err, 2u);
                const uint32_t room = cap[h] > have ? cap[h] - have : 0u;
                if (add > room) atomicAdd(&md->n_points, ~(unsigned long long)(add - room) + 1ull);
                continue;
            }
        }
        const uint32_t old = table[h].ptr, old_cap = cap[h];
        for (uint32_t j = 0; j < have; j++) pool[at + j] = pool[old + j];
        if (seq) for (uint32_t j = 0; j < have; j++) seq[at + j] = seq[old + j];
        table[h].ptr = (uint32_t)at;
        cap[h] = ncap;
        if (free_in && old_cap >= 8) {  // the outgrown region is recycled (without this an LRU map leaks ~one slot per inserted point)
            const int oc = 31 - __clz(old_cap);
            const int pos = atomicAdd(&md->free_in_top[oc], 1);
            if (pos >= 0 && (uint32_t)pos < free_cap) free_in[(size_t)oc * free_cap + (uint32_t)pos] = old;
            else atomicSub(&md->free_in_top[oc], 1);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) added += __shfl_xor(added, off);
    if ((threadIdx.x & 63) == 0 && added) atomicAdd(&md->n_points, added);
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

__device__ __forceinline__ void map_insert_write_body(Slot* table, const uint32_t* __restrict__ cap, float4* pool, uint32_t* __restrict__ seq,
                                                               const MapDev* __restrict__ md, const float4* __restrict__ pts, unsigned long long n_host,
                                                               const uint32_t* __restrict__ n_dev,
                                                               const uint32_t* __restrict__ slot_of_point) {
    const unsigned long long n = n_dev ? (unsigned long long)*n_dev : n_host;
    // A point's place inside its voxel's region is its arrival rank at the slot's counter -- any order; its push_back rank (the batch's first
    // sequence number + its index in the batch) is written beside it
    const uint32_t seq0 = seq ? md->seq_cur : 0u;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint32_t sp = slot_of_point[i];
        if (sp == kNoIdx) continue;
        const uint32_t h = sp & 0x7FFFFFFFu;
        const uint32_t idx = atomicAdd(&table[h].cnt, 1u);
        if (idx < cap[h]) {
            const unsigned long long at = (unsigned long long)table[h].ptr + idx;
            pool[at] = pts[i];
            if (seq) seq[at] = seq0 + (uint32_t)i;
        } else atomicSub(&table[h].cnt, 1u);  // only after a pool overflow (err bit 2): cnt never exceeds cap, readers stay in bounds
    }
}

// ---------------------------------------------------------------------------------------------------
// LRU eviction = the grids_cache_ list of IVox::AddPoints (ivox3d.h:231-256).  The reference touches voxels one point at a
// time (splice to the list front) and, after every point, drops the list's back if the map holds more than `capacity`
// voxels and that voxel was created more than max_distance of travel ago.  Here:
//   * every slot carries the stamp (batch << 26 | point index) of its last touch (atomicMax in the claim kernel);
//   * lru_append_kernel appends one log entry per voxel touched by the batch, in point order (the point that holds a
//     voxel's final stamp emits it), so the log is sorted by stamp: an entry is live iff its stamp is still the slot's;
//   * lru_evict_kernel walks the log from its tail: live entries ARE the list from the back.  The number of evictions the
//     point-by-point process performs is order independent, E = clamp(n_voxels - capacity, 0, n_points_in_batch), cut short
//     at the first candidate younger than max_distance (the reference then keeps testing that same back voxel).
// Evicted slots become tombstones (probes walk on), their pool regions go to per-size free lists that the grow kernel
// pops; the table is rebuilt into its twin when tombstones pile up (map_rebuild).  A voxel near the back that the SAME
// batch touches after its turn to go is dropped and re-created by the reference's point-by-point order: lru_exact_* below
// replays the batch's pops in order and does the same (round 6; counted only until then).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lru_append_body(const uint32_t* __restrict__ slot_of_point, const unsigned long long* __restrict__ touch,
                                                          unsigned long long n_host, const uint32_t* __restrict__ n_dev, unsigned long long stamp_base,
                                                          LruEntry* __restrict__ log, unsigned long long log_mask, MapDev* md) {
    const unsigned long long n = n_dev ? (unsigned long long)*n_dev : n_host;
    __shared__ uint32_t wsum[16];
    __shared__ unsigned long long head_s;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid == 0) { head_s = md->log_head; md->log_head_prev = md->log_head; }
    __syncthreads();
    constexpr int kI = 4;  // consecutive points per lane: 4096 points per round of the one workgroup
    for (unsigned long long base = 0; base < n; base += 1024ull * kI) {
        uint32_t hs[kI];
        bool fl[kI];
        uint32_t mine = 0;
#pragma unroll
        for (int j = 0; j < kI; j++) {
            const unsigned long long i = base + (unsigned long long)tid * kI + j;
            hs[j] = 0;
            fl[j] = false;
            if (i < n) {
                const uint32_t sp = slot_of_point[i];
                if (sp != kNoIdx) {
                    hs[j] = sp & 0x7FFFFFFFu;
                    fl[j] = touch[hs[j]] == stamp_base + i;
                }
            }
            mine += fl[j] ? 1u : 0u;
        }
        uint32_t inc = mine;  // inclusive scan of the per-lane counts: wave shuffles, then the 16 wave totals
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(inc, off);
            if (lane >= off) inc += t;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t off = inc - mine, total = 0;
        for (int w = 0; w < 16; w++) {
            if (w < wave) off += wsum[w];
            total += wsum[w];
        }
#pragma unroll
        for (int j = 0; j < kI; j++) {
            if (!fl[j]) continue;
            LruEntry e;
            e.stamp = stamp_base + base + (unsigned long long)tid * kI + j;
            e.slot = hs[j];
            e.pad = 0;
            log[(head_s + off) & log_mask] = e;
            off++;
        }
        __syncthreads();
        if (tid == 0) head_s += total;
        __syncthreads();
    }
    if (tid == 0) {
        md->log_head = head_s;
        if (head_s - md->log_tail > log_mask + 1ull) atomicOr(&md->err, 8u);  // log overrun: more live + stale entries than the ring holds
    }
}

// ---- the reference's point-by-point order inside ONE batch (ivox3d.h:231-256) -------------------------------------------------------------
// AddPoints handles a batch point by point: the point's voxel is created at the front of the list or moved there, and then, if the map holds more
// voxels than its capacity and the voxel at the BACK is old enough, that voxel goes.  A voxel near the back that the batch touches only AFTER the
// point at which it went is therefore dropped with all it held and created again, holding the batch's points alone, stamped with the distance of
// now -- while the walk above, which sees the batch as a whole, skips every voxel the batch touched.  (The voxels that are dropped for good are the
// same either way: every re-creation is one more creation, i.e. one more pop, and that pop takes the voxel the walk takes instead.)
// Which touched voxels are dropped is decided by replaying the pops in order: pops happen at the batch's voxel-creating points (g, in point order;
// the first capacity - size of them fill the map up and pop nothing) and at the re-creations they cause; the pop at point t takes the oldest voxel
// of the list that has not been touched by a point before t.  One lane replays that over the list's old entries staged in LDS; the voxels found are
// cut down to the batch's own points afterwards (the pool keeps arrival order, the sequence numbers tell the batch's points).  The replay gives up --
// the batch is handled as before and counted in n_lru_inexact -- where pops do not coincide with creations: the map above its capacity before the
// batch, or a voxel younger than max_distance at the back.
struct LruExact {
    unsigned long long* first_touch;
    uint32_t* g;
    uint32_t* rec;
    const uint32_t* slot_of_point;
    float4* pool;
    uint32_t* seq;
};
constexpr int kLruRq = 8192;  // pending re-creations the replay holds (a binary min-heap in LDS)

// returns the number of voxels to re-create (their slots in ex.rec); *w_pops = the untouched voxels the batch drops for good (what the walk below
// evicts), or 0xFFFFFFFF when the replay gave up and the walk applies its own rule
__device__ __forceinline__ uint32_t lru_exact_replay(const Slot* table, const float* __restrict__ created, const unsigned long long* __restrict__ touch,
                                                     const unsigned long long* __restrict__ prev_touch, unsigned long long stamp_base,
                                                     const LruEntry* __restrict__ log, unsigned long long log_mask, unsigned long long n_add, uint32_t capacity,
                                                     float travel, float max_distance, MapDev* md, const LruExact& ex, uint32_t* w_pops) {
    __shared__ uint32_t xw[4], xsp[4];
    constexpr uint32_t kGLds = 4096;    // the first creation points are kept in LDS as well: the replaying lane reads one per pop, and a global load is a microsecond to it
    __shared__ uint32_t x_g[kGLds];
    __shared__ uint32_t x_ng, x_gi, x_rn, x_recn, x_wp, x_excess, x_deficit;
    __shared__ long long x_t;
    __shared__ int x_state, x_have;  // state: 0 replaying, 1 done, 2 given up
    __shared__ uint32_t x_rq[kLruRq];   // pending re-creation points: a binary min-heap, the next one at [0]
    constexpr int kChunk = 1024;        // list entries staged per turn: four per lane, their loads requested together
    __shared__ uint8_t x_cls[kChunk];   // 0 stale, 1 untouched, 2 touched by this batch; bit 7: younger than max_distance
    __shared__ uint32_t x_f[kChunk], x_slot[kChunk];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // (1) the batch's voxel-creating points in point order: point i creates a voxel iff the voxel had no stamp before this batch and i is its first point.
    // Four consecutive points per lane, their loads requested together (this kernel is one workgroup: it pays for every dependent round trip in full).
    if (tid == 0) x_ng = 0;
    __syncthreads();
    constexpr int kI = 4;
    for (unsigned long long base = 0; base < n_add; base += 256ull * kI) {
        uint32_t hs[kI];
        bool ok[kI], fl[kI];
#pragma unroll
        for (int j = 0; j < kI; j++) {
            const unsigned long long i = base + (unsigned long long)tid * kI + j;
            const uint32_t sp = i < n_add ? ex.slot_of_point[i] : kNoIdx;
            ok[j] = sp != kNoIdx;
            hs[j] = ok[j] ? (sp & 0x7FFFFFFFu) : 0u;
        }
        unsigned long long ft[kI], pv[kI];
#pragma unroll
        for (int j = 0; j < kI; j++) { ft[j] = ex.first_touch[hs[j]]; pv[j] = prev_touch[hs[j]]; }
        uint32_t mine = 0;
#pragma unroll
        for (int j = 0; j < kI; j++) {
            const unsigned long long i = base + (unsigned long long)tid * kI + j;
            fl[j] = ok[j] && pv[j] == 0ull && ft[j] >= stamp_base && (kStampIdxMask - (ft[j] - stamp_base)) == i;
            mine += fl[j] ? 1u : 0u;
        }
        uint32_t inc = mine;  // inclusive scan of the per-lane counts: wave shuffles, then the four wave totals
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(inc, off);
            if (lane >= off) inc += t;
        }
        if (lane == 63) xw[wave] = inc;
        __syncthreads();
        uint32_t off = x_ng + inc - mine;
        for (int w = 0; w < wave; w++) off += xw[w];
#pragma unroll
        for (int j = 0; j < kI; j++)
            if (fl[j]) {
                const uint32_t gv = (uint32_t)(base + (unsigned long long)tid * kI + j);
                ex.g[off] = gv;
                if (off < kGLds) x_g[off] = gv;
                off++;
            }
        __syncthreads();
        if (tid == 0) x_ng += (xw[0] + xw[1]) + (xw[2] + xw[3]);
        __syncthreads();
    }
    const uint32_t n_g = x_ng, n_vox = md->n_voxels;
    const uint32_t s0 = n_vox - n_g;  // voxels before the batch
    if (tid == 0) {
        x_state = 0;
        x_have = 0;
        x_rn = 0;
        x_recn = 0;
        x_wp = 0;
        x_gi = 0;
        x_t = -1;
        x_excess = s0 > capacity ? s0 - capacity : 0u;   // the map may already hold more than its capacity (nothing was old enough to go): a pop at every point
        x_deficit = s0 < capacity ? capacity - s0 : 0u;  // ... or less: the first creations fill it up and pop nothing
        if (x_excess == 0 && n_g <= x_deficit) x_state = 1;  // the batch ends at or below the capacity: no pop at all
    }
    __syncthreads();
    // (2) the replay over the list's old entries, oldest first.  A pop is attempted after EVERY point while the map holds more voxels than its capacity
    // (ivox3d.h:251): at the creation that takes it over the capacity, and at every following point until it is back.
    const unsigned long long limit = md->log_head_prev;
    unsigned long long tail = md->log_tail;
    for (; x_state == 0 && tail < limit; tail += kChunk) {
        LruEntry en[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned long long idx = tail + (unsigned long long)tid * 4ull + j;  // (four consecutive entries per lane: list order = lane order)
            en[j].stamp = 0; en[j].slot = 0; en[j].pad = 0;
            if (idx < limit) en[j] = log[idx & log_mask];
        }
        unsigned long long nowv[4], pvv[4], ftv[4];
        float crv[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {  // (unconditional: slot 0 for an empty entry -- four words per entry in flight instead of a chain of dependent loads)
            nowv[j] = touch[en[j].slot];
            pvv[j] = prev_touch[en[j].slot];
            ftv[j] = ex.first_touch[en[j].slot];
            crv[j] = created[en[j].slot];
        }
        // the entries that are a voxel's place in the list -- a few among thousands of stale ones (a voxel in view leaves an entry behind with every
        // batch) -- are compacted in list order: the one lane that replays the pops walks over those alone
        uint8_t cl[4];
        uint32_t fv[4], mine = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            cl[j] = 0;
            fv[j] = 0xFFFFFFFFu;
            if (en[j].stamp != 0) {
                if (nowv[j] == en[j].stamp) cl[j] = 1;
                else if (nowv[j] >= stamp_base && pvv[j] == en[j].stamp) {
                    cl[j] = 2;
                    fv[j] = (uint32_t)(kStampIdxMask - (ftv[j] - stamp_base));
                }
                if (cl[j] && !((travel - crv[j]) > max_distance)) cl[j] |= 0x80;
            }
            mine += cl[j] ? 1u : 0u;
        }
        // (anything but an untouched voxel old enough to go -- the usual content of the list's back -- makes the chunk one for the lane-by-lane replay)
        const bool special = (cl[0] > 1) | (cl[1] > 1) | (cl[2] > 1) | (cl[3] > 1);
        const unsigned long long spm = __ballot(special);
        uint32_t inc = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(inc, off);
            if (lane >= off) inc += t;
        }
        if (lane == 63) xw[wave] = inc;
        if (lane == 0) xsp[wave] = spm != 0ull ? 1u : 0u;
        __syncthreads();
        uint32_t at = inc - mine;
        for (int w = 0; w < wave; w++) at += xw[w];
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (cl[j]) {
                x_cls[at] = cl[j];
                x_f[at] = fv[j];
                x_slot[at] = en[j].slot;
                at++;
            }
        __syncthreads();
        if (tid == 0) {
            const uint32_t len = (xw[0] + xw[1]) + (xw[2] + xw[3]);  // the chunk's entries that count
            uint32_t gi = x_gi, rn = x_rn, recn = x_recn, wp = x_wp, excess = x_excess, deficit = x_deficit;
            long long t = x_t;
            int state = 0, have = x_have;
            // the next voxel-creating point (a new voxel's first point, or the first point of a voxel dropped earlier in the batch), 2^32 - 1 when none is left
            // a chunk of untouched voxels old enough to go, the map exactly at its capacity, nothing pending: every creation pops the next of them -- no
            // decision to replay (the usual case on a drive that does not come back: ~1 000 pops per batch, which the one lane would take ~0.1 us each)
            if (!((xsp[0] | xsp[1]) | (xsp[2] | xsp[3])) && excess == 0 && rn == 0 && !have) {
                if (deficit) {  // (the creations that fill the map up to its capacity pop nothing)
                    const uint32_t fill = deficit < n_g - gi ? deficit : n_g - gi;
                    gi += fill;
                    deficit -= fill;
                    x_deficit = deficit;
                }
                const uint32_t left = n_g - gi, take = len < left ? len : left;
                if (take) {
                    gi += take;
                    wp += take;
                    t = (long long)(gi - 1 < kGLds ? x_g[gi - 1] : ex.g[gi - 1]);
                }
                if (gi >= n_g) state = 1;
                x_gi = gi; x_wp = wp; x_t = t;
                if (state) x_state = state;
            } else {
            auto next_creation = [&]() -> uint32_t {
                const uint32_t tg = gi < n_g ? (gi < kGLds ? x_g[gi] : ex.g[gi]) : 0xFFFFFFFFu, tr = rn > 0 ? x_rq[0] : 0xFFFFFFFFu;
                return tg < tr ? tg : tr;
            };
            auto take_creation = [&]() {
                const uint32_t tg = gi < n_g ? (gi < kGLds ? x_g[gi] : ex.g[gi]) : 0xFFFFFFFFu, tr = rn > 0 ? x_rq[0] : 0xFFFFFFFFu;
                if (tg < tr) {
                    gi++;
                } else {  // pop the heap's root: the last leaf sinks from the top
                    const uint32_t v = x_rq[--rn];
                    uint32_t at = 0;
                    for (;;) {
                        uint32_t ch = 2 * at + 1;
                        if (ch >= rn) break;
                        if (ch + 1 < rn && x_rq[ch + 1] < x_rq[ch]) ch++;
                        if (!(x_rq[ch] < v)) break;
                        x_rq[at] = x_rq[ch];
                        at = ch;
                    }
                    if (rn) x_rq[at] = v;
                }
                if (deficit > 0) deficit--; else excess++;
            };
            for (uint32_t k = 0; k < len && state == 0; k++) {
                const uint8_t c = x_cls[k];
                for (;;) {
                    if (!have) {  // the point after which the next pop is attempted
                        if (excess == 0) {
                            const uint32_t tc = next_creation();
                            if (tc == 0xFFFFFFFFu) { state = 1; break; }
                            take_creation();
                            if (excess == 0) continue;  // (it filled the map up)
                            t = (long long)tc;
                        } else {
                            t = t + 1;
                            if (t >= (long long)n_add) { state = 1; break; }
                        }
                        while (next_creation() != 0xFFFFFFFFu && (long long)next_creation() <= t) take_creation();
                        have = 1;
                    }
                    if ((c & 0x7F) == 2 && (long long)x_f[k] <= t) break;  // touched by a point up to t: at the front of the list by then -- the next entry is the back
                    if (c & 0x80) {  // at the back and too young to go: no pop
                        if ((c & 0x7F) == 2) {  // ... until the batch touches it (its first point f): the map keeps growing meanwhile
                            t = (long long)x_f[k] - 1;
                            have = 0;
                            continue;
                        }
                        state = 1;  // for the rest of the batch
                        break;
                    }
                    excess--;  // the pop after point t takes this voxel
                    have = 0;
                    if ((c & 0x7F) == 2) {  // ... and a later point of the batch creates it again: one more creation
                        if (recn >= kLruRecCap || rn >= (uint32_t)kLruRq) { state = 2; break; }
                        ex.rec[recn++] = x_slot[k];
                        const uint32_t fk = x_f[k];
                        uint32_t at = rn++;  // heap push: the new leaf rises
                        while (at > 0 && fk < x_rq[(at - 1) / 2]) { x_rq[at] = x_rq[(at - 1) / 2]; at = (at - 1) / 2; }
                        x_rq[at] = fk;
                    } else {
                        wp++;
                    }
                    // nothing left that could make the map exceed its capacity again: the replay ends here (not at the next entry that counts, which
                    // may lie thousands of stale entries further on)
                    if (excess == 0 && next_creation() == 0xFFFFFFFFu) state = 1;
                    break;
                }
            }
            x_gi = gi; x_rn = rn; x_recn = recn; x_wp = wp; x_excess = excess; x_deficit = deficit; x_t = t; x_have = have;
            if (state) x_state = state;
            }
        }
        __syncthreads();
    }
    __syncthreads();
    // the list's old entries used up with pops still due: the back would be a voxel of this very batch -- not followed
    if (tid == 0 && x_state == 0) {
        bool due = x_excess > 0 && x_t + 1 < (long long)n_add;
        if (x_excess == 0) {
            uint32_t left = (n_g - x_gi) + x_rn;
            due = left > x_deficit;
        }
        x_state = due ? 2 : 1;
    }
    __syncthreads();
    const int state = x_state;
    const uint32_t recn = x_recn;
    if (tid == 0 && state == 2) md->n_lru_inexact++;
    *w_pops = state == 2 ? 0xFFFFFFFFu : x_wp;
    return state == 2 ? 0u : recn;
}

// the voxels the replay found: what they held before the batch goes, the batch's own points move to the front of the region in the order they have,
// the voxel is as old as the batch (NodeType(distance), ivox3d.h:240).  One wave per voxel.
__device__ __forceinline__ void lru_exact_recreate(Slot* table, float* __restrict__ created, uint32_t recn, unsigned long long n_add, float travel, MapDev* md,
                                                   const LruExact& ex) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t seq0 = md->seq_cur;
    for (uint32_t v = (uint32_t)wave; v < recn; v += 4u) {
        const uint32_t h = ex.rec[v];
        const uint32_t ptr = table[h].ptr, cnt = table[h].cnt;
        uint32_t kept = 0;
        for (uint32_t base = 0; base < cnt; base += 64u) {
            const uint32_t j = base + (uint32_t)lane;
            float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
            uint32_t sq = 0;
            bool keep = false;
            if (j < cnt) {
                p = ex.pool[ptr + j];
                sq = ex.seq[ptr + j];
                keep = (unsigned long long)(uint32_t)(sq - seq0) < n_add;  // one of this batch's points
            }
            const unsigned long long m = __ballot(keep);
            const uint32_t pos = kept + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            __builtin_amdgcn_s_waitcnt(0);  // (every lane's load has arrived before any lane stores: the stores land at or before the places read)
            if (keep) {
                ex.pool[ptr + pos] = p;
                ex.seq[ptr + pos] = sq;
            }
            kept += (uint32_t)__popcll(m);
        }
        if (lane == 0) {
            table[h].cnt = kept;
            created[h] = travel;
            atomicAdd(&md->n_points, (unsigned long long)0 - (unsigned long long)(cnt - kept));
            atomicAdd(&md->n_lru_recreated, 1ull);
        }
    }
}

__device__ __forceinline__ void lru_evict_body(Slot* table, uint32_t* __restrict__ cap, float* __restrict__ created,
                                                        unsigned long long* __restrict__ touch, const unsigned long long* __restrict__ prev_touch,
                                                        unsigned long long stamp_base, const LruEntry* __restrict__ log,
                                                        unsigned long long log_mask, uint32_t* __restrict__ free_items, uint32_t free_cap,
                                                        unsigned long long n_host, const uint32_t* __restrict__ n_dev, uint32_t capacity,
                                                        float travel, float max_distance, MapDev* md, const uint32_t* __restrict__ free_in, const LruExact ex) {
    const unsigned long long n_add = n_dev ? (unsigned long long)*n_dev : n_host;
    // which of the voxels this batch touched the reference dropped on the way (before the walk below changes the stamps the replay reads)
    uint32_t w_exact = 0xFFFFFFFFu;
    const uint32_t n_rec = ex.first_touch ? lru_exact_replay(table, created, touch, prev_touch, stamp_base, log, log_mask, n_add, capacity, travel, max_distance, md, ex, &w_exact) : 0u;
    // regions the grow kernel of this batch freed (voxels that moved to a larger one): fold them into the free lists it pops from.  The 21 size
    // classes' counters are read in one go (they used to be read class by class between two barriers: 21 dependent round trips, ~10 us of this
    // kernel's 16), the classes that received something -- usually two or three -- are then copied in turn
    __shared__ int fin_s[24], ftop_s[24];
    if (threadIdx.x < 24) {
        fin_s[threadIdx.x] = threadIdx.x >= 3 ? md->free_in_top[threadIdx.x] : 0;
        ftop_s[threadIdx.x] = threadIdx.x >= 3 ? md->free_top[threadIdx.x] : 0;
    }
    __syncthreads();
    for (int c = 3; c < 24; c++) {
        const int n_in = fin_s[c], top = ftop_s[c];
        if (n_in == 0) continue;  // (uniform: every thread reads the same shared words)
        const int room = (int)free_cap - top, take = n_in < room ? n_in : room;
        for (int j = threadIdx.x; j < take; j += 256) free_items[(size_t)c * free_cap + (uint32_t)(top + j)] = free_in[(size_t)c * free_cap + (uint32_t)j];
        if (threadIdx.x == 0) { md->free_top[c] = top + (take > 0 ? take : 0); md->free_in_top[c] = 0; }
    }
    __syncthreads();
    __shared__ uint32_t wsum[4];
    __shared__ unsigned long long first_young_s, tail_s;
    __shared__ uint32_t want_s, pts_s;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t n_vox = md->n_voxels;
    const unsigned long long limit = md->log_head_prev;  // never evict what this very batch touched
    if (tid == 0) {
        unsigned long long e = n_vox > capacity ? (unsigned long long)(n_vox - capacity) : 0ull;
        want_s = (uint32_t)(e < n_add ? e : n_add);
        if (w_exact != 0xFFFFFFFFu) want_s = w_exact;  // the replay's count: pops are one per point, and a voxel that is dropped and re-created takes one
        tail_s = md->log_tail;
        pts_s = 0;
    }
    __syncthreads();
    uint32_t done = 0;
    bool stop = false;
    while (!stop) {
        const uint32_t want = want_s - done;
        const unsigned long long tail = tail_s;
        if (want == 0 || tail >= limit) break;
        const unsigned long long idx = tail + tid;
        LruEntry e;
        e.stamp = 0; e.slot = 0; e.pad = 0;
        bool live = false, young = false;
        if (idx < limit) {
            e = log[idx & log_mask];
            const unsigned long long now = touch[e.slot];
            live = e.stamp != 0 && now == e.stamp;
            young = live && !((travel - created[e.slot]) > max_distance);
            // this voxel sat at the back of the list until the current batch touched it: the reference may have dropped and
            // re-created it in between (point-by-point order); counted, not reproduced
            if (!live && e.stamp != 0 && now >= stamp_base && prev_touch[e.slot] == e.stamp) atomicAdd(&md->n_lru_interleaved, 1ull);
        }
        if (tid == 0) first_young_s = ~0ull;
        __syncthreads();
        if (young) atomicMin(&first_young_s, idx);
        const unsigned long long m = __ballot(live);
        const uint32_t rank_w = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(m);
        __syncthreads();
        uint32_t rank = rank_w;
        for (int w = 0; w < wave; w++) rank += wsum[w];
        const unsigned long long fy = first_young_s;
        const bool evict = live && idx < fy && rank < want;
        if (evict) {
            const uint32_t h = e.slot;
            const uint32_t c = cap[h], cnt = table[h].cnt, ptr = table[h].ptr;
            table[h].key = kTombKey;
            table[h].cnt = 0;
            cap[h] = 0;
            touch[h] = 0;
            atomicAdd(&pts_s, cnt);
            if (c >= 8 && free_items) {
                const int cls = 31 - __clz(c);
                const int pos = atomicAdd(&md->free_top[cls], 1);
                if (pos >= 0 && (uint32_t)pos < free_cap) free_items[(size_t)cls * free_cap + (uint32_t)pos] = ptr;
                else atomicSub(&md->free_top[cls], 1);
            }
        }
        // where the walk resumes: at the first young voxel (the reference keeps testing it), right after the last eviction when
        // the quota is used up, else after this chunk
        const unsigned long long ev_mask = __ballot(evict);
        __syncthreads();
        if (lane == 0) wsum[wave] = __popcll(ev_mask);
        __syncthreads();
        const uint32_t n_ev = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
        unsigned long long next = tail + 256;
        if (next > limit) next = limit;
        if (done + n_ev >= want_s) {
            // the quota is used up: the last evicted entry is the (want)-th live one of the chunk, everything up to it is consumed -- and nothing
            // beyond it, a young voxel further on or not (until round 6 a young voxel anywhere in the chunk moved the tail to ITS place: live voxels
            // between the last eviction and it fell off the list without being dropped, and the next batch started evicting behind them)
            __shared__ unsigned long long last_s;
            if (tid == 0) last_s = 0;
            __syncthreads();
            if (evict) atomicMax(&last_s, idx + 1);
            __syncthreads();
            next = last_s > tail ? last_s : tail;
            stop = true;
        } else if (fy != ~0ull) {
            next = fy;
            stop = true;
        }
        done += n_ev;
        __syncthreads();
        if (tid == 0) tail_s = next;
        __syncthreads();
    }
    if (tid == 0) {
        md->log_tail = tail_s;
        if (done) {
            md->n_voxels = n_vox - done;
            md->n_points -= pts_s;
            md->n_tombs += done;
            md->n_evicted += done;
        }
    }
    __syncthreads();
    if (n_rec) lru_exact_recreate(table, created, n_rec, n_add, travel, md, ex);
}

// ---- launchable forms: one map (arguments by value), or the maps of a sequence batch (blockIdx.y = slot, arguments from the slot's MapRef in
// device memory; a slot whose scan does not enter its map this round -- SeqDev::go == 0 -- exits at once) -----------------------------------------
__global__ void __launch_bounds__(256) map_insert_claim_kernel(Slot* table, uint32_t mask, uint32_t* __restrict__ pending, float* __restrict__ created,
                                                               const float4* __restrict__ pts, unsigned long long n_host, const uint32_t* __restrict__ n_dev,
                                                               float inv_res, float res, int key_mode, float travel, uint32_t max_voxels, MapDev* md,
                                                               uint32_t* __restrict__ slot_of_point, unsigned long long* __restrict__ touch,
                                                               unsigned long long* __restrict__ prev_touch, unsigned long long stamp_base,
                                                               unsigned long long* __restrict__ first_touch) {
    map_insert_claim_body(table, mask, pending, created, pts, n_host, n_dev, inv_res, res, key_mode, travel, max_voxels, md, slot_of_point, touch, prev_touch,
                          stamp_base, first_touch);
}
__global__ void __launch_bounds__(256) map_insert_claim_seq(const MapRef* __restrict__ maps, const SeqDev* __restrict__ seq) {
    if (!seq[blockIdx.y].go) return;
    const MapRef& r = maps[blockIdx.y];
    map_insert_claim_body(r.table, r.mask, r.pending, r.created, r.stage, 0ull, &r.md->n_add, r.inv_res, r.res, r.key_mode, (float)seq[blockIdx.y].travel,
                          r.max_voxels, r.md, r.slot_of_point, r.lru_capacity ? r.touch : nullptr, r.prev_touch, r.stamp_base, r.lru_capacity ? r.first_touch : nullptr);
}
__global__ void __launch_bounds__(256) map_insert_grow_kernel(Slot* table, uint32_t* __restrict__ cap, uint32_t* __restrict__ pending, float4* pool, uint32_t* seq,
                                                              unsigned long long pool_cap, unsigned long long n_host, const uint32_t* __restrict__ n_dev,
                                                              MapDev* md, const uint32_t* __restrict__ slot_of_point, const uint32_t* __restrict__ free_items,
                                                              uint32_t free_cap, uint32_t* __restrict__ free_in) {
    map_insert_grow_body(table, cap, pending, pool, seq, pool_cap, n_host, n_dev, md, slot_of_point, free_items, free_cap, free_in);
}
__global__ void __launch_bounds__(256) map_insert_grow_seq(const MapRef* __restrict__ maps, const SeqDev* __restrict__ seq) {
    if (!seq[blockIdx.y].go) return;
    const MapRef& r = maps[blockIdx.y];
    map_insert_grow_body(r.table, r.cap, r.pending, r.pool, r.pool_seq, r.pool_cap, 0ull, &r.md->n_add, r.md, r.slot_of_point, r.lru_capacity ? r.free_items : nullptr,
                         r.free_cap, r.lru_capacity ? r.free_in : nullptr);
}
__global__ void __launch_bounds__(256) map_insert_write_kernel(Slot* table, const uint32_t* __restrict__ cap, float4* pool, uint32_t* __restrict__ seq,
                                                               const MapDev* __restrict__ md, const float4* __restrict__ pts,
                                                               unsigned long long n_host, const uint32_t* __restrict__ n_dev,
                                                               const uint32_t* __restrict__ slot_of_point) {
    map_insert_write_body(table, cap, pool, seq, md, pts, n_host, n_dev, slot_of_point);
}
__global__ void __launch_bounds__(256) map_insert_write_seq(const MapRef* __restrict__ maps, const SeqDev* __restrict__ seq) {
    if (!seq[blockIdx.y].go) return;
    const MapRef& r = maps[blockIdx.y];
    map_insert_write_body(r.table, r.cap, r.pool, r.pool_seq, r.md, r.stage, 0ull, &r.md->n_add, r.slot_of_point);
}
__global__ void __launch_bounds__(1024) lru_append_kernel(const uint32_t* __restrict__ slot_of_point, const unsigned long long* __restrict__ touch,
                                                          unsigned long long n_host, const uint32_t* __restrict__ n_dev, unsigned long long stamp_base,
                                                          LruEntry* __restrict__ log, unsigned long long log_mask, MapDev* md) {
    lru_append_body(slot_of_point, touch, n_host, n_dev, stamp_base, log, log_mask, md);
}
__global__ void __launch_bounds__(1024) lru_append_seq(const MapRef* __restrict__ maps, const SeqDev* __restrict__ seq) {
    const MapRef& r = maps[blockIdx.y];
    if (!seq[blockIdx.y].go || !r.lru_capacity) return;
    lru_append_body(r.slot_of_point, r.touch, 0ull, &r.md->n_add, r.stamp_base, r.lru_log, r.log_mask, r.md);
}
__global__ void __launch_bounds__(256) lru_evict_kernel(Slot* table, uint32_t* __restrict__ cap, float* __restrict__ created,
                                                        unsigned long long* __restrict__ touch, const unsigned long long* __restrict__ prev_touch,
                                                        unsigned long long stamp_base, const LruEntry* __restrict__ log, unsigned long long log_mask,
                                                        uint32_t* __restrict__ free_items, uint32_t free_cap, unsigned long long n_host,
                                                        const uint32_t* __restrict__ n_dev, uint32_t capacity, float travel, float max_distance, MapDev* md,
                                                        const uint32_t* __restrict__ free_in, const LruExact ex) {
    lru_evict_body(table, cap, created, touch, prev_touch, stamp_base, log, log_mask, free_items, free_cap, n_host, n_dev, capacity, travel, max_distance, md,
                   free_in, ex);
}
__global__ void __launch_bounds__(256) lru_evict_seq(const MapRef* __restrict__ maps, const SeqDev* __restrict__ seq) {
    const MapRef& r = maps[blockIdx.y];
    if (!seq[blockIdx.y].go || !r.lru_capacity) return;
    lru_evict_body(r.table, r.cap, r.created, r.touch, r.prev_touch, r.stamp_base, r.lru_log, r.log_mask, r.free_items, r.free_cap, 0ull, &r.md->n_add,
                   r.lru_capacity, (float)seq[blockIdx.y].travel, r.lru_max_distance, r.md, r.free_in,
                   LruExact{r.first_touch, r.lru_g, r.lru_rec, r.slot_of_point, r.pool, r.pool_seq});
}

// table rebuild: live slots are re-inserted into the twin table (no tombstones), the touch log is re-pointed
__global__ void __launch_bounds__(256) rebuild_insert_kernel(const Slot* __restrict__ told, const uint32_t* __restrict__ cap_old,
                                                             const float* __restrict__ created_old, const unsigned long long* __restrict__ touch_old,
                                                             const unsigned long long* __restrict__ prev_old, Slot* tnew, uint32_t* __restrict__ cap_new,
                                                             float* __restrict__ created_new, unsigned long long* __restrict__ touch_new,
                                                             unsigned long long* __restrict__ prev_new, uint32_t table_cap, uint32_t mask,
                                                             uint32_t* __restrict__ remap, MapDev* md) {
    const uint32_t h = blockIdx.x * 256u + threadIdx.x;
    if (h >= table_cap) return;
    const Slot s = told[h];
    remap[h] = kNoIdx;
    if (s.key == kEmptyKey || s.key == kTombKey) return;
    const int kx = ((int)((uint32_t)(s.key & 0x1FFFFFu) << 11)) >> 11, ky = ((int)((uint32_t)((s.key >> 21) & 0x1FFFFFu) << 11)) >> 11,
              kz = ((int)((uint32_t)((s.key >> 42) & 0x1FFFFFu) << 11)) >> 11;
    BrickProbe bp = brick_probe(kx, ky, kz);
    for (uint32_t probe = 0; probe <= (mask >> 6); probe++) {
        const uint32_t g = brick_slot(bp, mask);
        if (atomicCAS(&tnew[g].key, kEmptyKey, s.key) == kEmptyKey) {
            tnew[g].ptr = s.ptr;
            tnew[g].cnt = s.cnt;
            cap_new[g] = cap_old[h];
            created_new[g] = created_old[h];
            touch_new[g] = touch_old[h];
            prev_new[g] = prev_old[h];
            remap[h] = g;
            return;
        }
        brick_next(bp);
    }
    atomicOr(&md->err, 1u);
}

__global__ void __launch_bounds__(256) rebuild_log_kernel(LruEntry* __restrict__ log, unsigned long long log_mask, const unsigned long long* __restrict__ touch_old,
                                                          const uint32_t* __restrict__ remap, MapDev* md) {
    const unsigned long long tail = md->log_tail, head = md->log_head;
    for (unsigned long long i = tail + blockIdx.x * 256ull + threadIdx.x; i < head; i += (unsigned long long)gridDim.x * 256ull) {
        LruEntry e = log[i & log_mask];
        if (e.stamp != 0 && touch_old[e.slot] == e.stamp && remap[e.slot] != kNoIdx) e.slot = remap[e.slot];
        else e.stamp = 0;  // stale: superseded or evicted
        log[i & log_mask] = e;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) md->n_tombs = 0;
}

int map_rebuild(lio_map* m, hipStream_t stream) {
    LIO_HIP_TRY(hipMemsetAsync(m->table2, 0xFF, (size_t)m->table_cap * sizeof(Slot), stream));
    LIO_HIP_TRY(hipMemset2DAsync(reinterpret_cast<char*>(m->table2) + 8, sizeof(Slot), 0, 8, m->table_cap, stream));
    LIO_HIP_TRY(hipMemsetAsync(m->cap2, 0, (size_t)m->table_cap * 4, stream));
    LIO_HIP_TRY(hipMemsetAsync(m->pending2, 0, (size_t)m->table_cap * 4, stream));
    LIO_HIP_TRY(hipMemsetAsync(m->touch2, 0, (size_t)m->table_cap * 8, stream));
    hipLaunchKernelGGL(rebuild_insert_kernel, (m->table_cap + 255) / 256, 256, 0, stream, m->table, m->cap, m->created, m->touch, m->prev_touch, m->table2,
                       m->cap2, m->created2, m->touch2, m->prev_touch2, m->table_cap, m->table_mask, m->remap, m->dev);
    hipLaunchKernelGGL(rebuild_log_kernel, 256, 256, 0, stream, m->lru_log, (unsigned long long)(m->lru_log_cap - 1), m->touch, m->remap, m->dev);
    std::swap(m->table, m->table2);
    std::swap(m->cap, m->cap2);
    std::swap(m->pending, m->pending2);
    std::swap(m->created, m->created2);
    std::swap(m->touch, m->touch2);
    std::swap(m->prev_touch, m->prev_touch2);
    m->tomb_bound = 0;
    LIO_HIP_TRY(hipGetLastError());
    return LIO_OK;
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
    const bool lru = m->lru_capacity != 0;
    const unsigned long long stamp_base = (unsigned long long)m->n_batches << kStampIdxBits;
    if (lru && n >= (1ull << kStampIdxBits)) { set_error("map insert batch of %llu points is too large for the LRU stamps", (unsigned long long)n); return LIO_E_CAPACITY; }
    if (lru && m->tomb_bound > m->table_cap / 4) {  // tombstones of evicted voxels may fill a quarter of the table: rebuild it first
        const int rc = map_rebuild(m, stream);
        if (rc != LIO_OK) return rc;
    }
    hipLaunchKernelGGL(map_insert_claim_kernel, (uint32_t)blocks, 256, 0, stream, m->table, m->table_mask, m->pending, m->created,
                       d_pts, (unsigned long long)n, d_n, m->inv_res, m->res, m->key_mode, (float)travel, (uint32_t)m->max_voxels, m->dev,
                       m->slot_of_point, lru ? m->touch : nullptr, m->prev_touch, stamp_base, lru ? m->first_touch : nullptr);
    if (layout) {
        const uint32_t ntiles = (m->table_cap + kScanTile - 1) / kScanTile;
        hipLaunchKernelGGL(map_layout_sums_kernel, ntiles, 256, 0, stream, m->pending, m->table_cap, m->tile_sum);
        hipLaunchKernelGGL(map_layout_scan_kernel, 1, 1024, 0, stream, m->tile_sum, ntiles, m->dev, (unsigned long long)m->pool_cap);
        hipLaunchKernelGGL(map_layout_assign_kernel, ntiles, 256, 0, stream, m->table, m->cap, m->pending, m->table_cap, m->tile_sum);
    } else {
        hipLaunchKernelGGL(map_insert_grow_kernel, (uint32_t)blocks, 256, 0, stream, m->table, m->cap, m->pending, m->pool, m->pool_seq,
                           (unsigned long long)m->pool_cap, (unsigned long long)n, d_n, m->dev, m->slot_of_point, lru ? m->free_items : nullptr,
                           m->free_cap, lru ? m->free_in : nullptr);
    }
    hipLaunchKernelGGL(map_insert_write_kernel, (uint32_t)blocks, 256, 0, stream, m->table, m->cap, m->pool, m->pool_seq, m->dev, d_pts,
                       (unsigned long long)n, d_n, m->slot_of_point);
    if (lru) {
        hipLaunchKernelGGL(lru_append_kernel, 1, 1024, 0, stream, m->slot_of_point, m->touch, (unsigned long long)n, d_n, stamp_base, m->lru_log,
                           (unsigned long long)(m->lru_log_cap - 1), m->dev);
        hipLaunchKernelGGL(lru_evict_kernel, 1, 256, 0, stream, m->table, m->cap, m->created, m->touch, m->prev_touch, stamp_base, m->lru_log,
                           (unsigned long long)(m->lru_log_cap - 1), m->free_items, m->free_cap, (unsigned long long)n, d_n, (uint32_t)m->lru_capacity,
                           (float)travel, m->lru_max_distance, m->dev, m->free_in,
                           LruExact{m->first_touch, m->lru_g, m->lru_rec, m->slot_of_point, m->pool, m->pool_seq});
        m->tomb_bound += n;
    }
    LIO_HIP_TRY(hipGetLastError());
    return LIO_OK;
}

// IVox::AddPoints for the maps of a sequence batch: the staged points of every slot whose scan enters its map this round (SeqDev::go; the count
// is the map's own MapDev::n_add, left by the batched classify), one launch per step of map_insert_dev's chain with blockIdx.y = slot.  Never the
// prebuilt-map layout: a session's first batch (the seed) goes through its engine's own path.  `bound` = upper bound of a slot's staged points.
int map_insert_seq(hipStream_t st, const MapRef* d_maps, const SeqDev* d_seq, int n_slots, uint32_t bound, int any_lru) {
    if (bound == 0 || n_slots <= 0) return LIO_OK;
    uint32_t blocks = (bound + 255u) / 256u;
    // the points a scan adds are a fraction of its downsampled cloud (~2 000 of 7-12 k): a grid-stride loop over a fraction of the bound
    if (blocks > 64u) blocks = 64u;
    const dim3 grid(blocks, (uint32_t)n_slots);
    hipLaunchKernelGGL(map_insert_claim_seq, grid, 256, 0, st, d_maps, d_seq);
    hipLaunchKernelGGL(map_insert_grow_seq, grid, 256, 0, st, d_maps, d_seq);
    hipLaunchKernelGGL(map_insert_write_seq, grid, 256, 0, st, d_maps, d_seq);
    if (any_lru) {
        hipLaunchKernelGGL(lru_append_seq, dim3(1, (uint32_t)n_slots), 1024, 0, st, d_maps, d_seq);
        hipLaunchKernelGGL(lru_evict_seq, dim3(1, (uint32_t)n_slots), 256, 0, st, d_maps, d_seq);
    }
    LIO_HIP_TRY(hipGetLastError());
    return LIO_OK;
}

}  // namespace lio
