// capi.hip -- handle management and the kernel-level entry points of include/lio_hip.h.
#include <stdarg.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "eskf.h"
#include <sched.h>

#include "lio_common.h"

namespace lio {

static thread_local char g_err[512] = "";

static thread_local char g_warn[512] = "";
void set_warning(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_warn, sizeof(g_warn), fmt, ap);
    va_end(ap);
}

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

template <typename T>
static bool dev_alloc(T** p, uint64_t count, uint64_t* bytes) {
    const uint64_t b = count * sizeof(T);
    if (hipMalloc(reinterpret_cast<void**>(p), b ? b : 16) != hipSuccess) {
        set_error("hipMalloc of %llu bytes failed", (unsigned long long)b);
        *p = nullptr;
        return false;
    }
    *bytes += b;
    return true;
}

static int fill_stencil(StencilArgs& st, int id) {
    // ivox3d.h:179-210 (GenerateNearbyGrids)
    static const signed char n18[19][3] = {{0, 0, 0},  {-1, 0, 0}, {1, 0, 0},  {0, 1, 0},   {0, -1, 0}, {0, 0, -1}, {0, 0, 1},
                                           {1, 1, 0},  {-1, 1, 0}, {1, -1, 0}, {-1, -1, 0}, {1, 0, 1},  {-1, 0, 1}, {1, 0, -1},
                                           {-1, 0, -1}, {0, 1, 1}, {0, -1, 1}, {0, 1, -1},  {0, -1, -1}};
    static const signed char n26x[8][3] = {{1, 1, 1}, {-1, 1, 1}, {1, -1, 1}, {1, 1, -1}, {-1, -1, 1}, {-1, 1, -1}, {1, -1, -1}, {-1, -1, -1}};
    st.n = 0;
    auto push = [&](int a, int b, int c) {
        st.off[st.n][0] = (signed char)a; st.off[st.n][1] = (signed char)b; st.off[st.n][2] = (signed char)c;
        st.n++;
    };
    if (id == 1) push(0, 0, 0);
    else if (id == 7) for (int i = 0; i < 7; i++) push(n18[i][0], n18[i][1], n18[i][2]);
    else if (id == 19) for (int i = 0; i < 19; i++) push(n18[i][0], n18[i][1], n18[i][2]);
    else if (id == 27) {
        for (int i = 0; i < 19; i++) push(n18[i][0], n18[i][1], n18[i][2]);
        for (int i = 0; i < 8; i++) push(n26x[i][0], n26x[i][1], n26x[i][2]);
    } else if (id == 75) {  // "NEARBY74": 5 x 5 x 3 including the centre
        for (int i = -2; i <= 2; i++)
            for (int j = -2; j <= 2; j++)
                for (int k = -1; k <= 1; k++) push(i, j, k);
    } else return LIO_E_INVALID;
    return LIO_OK;
}

static int map_check(lio_map* m, hipStream_t st) {
    if (hipMemcpyAsync(m->host_dev, m->dev, sizeof(MapDev), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        set_error("map status read-back failed: %s", hipGetErrorString(hipGetLastError()));
        return LIO_E_DEVICE;
    }
    if (m->host_dev->err) {
        set_error("map capacity exceeded (err bits 0x%x: 1 table full, 2 point pool full, 4 more than max_voxels voxels, 8 LRU log overrun)", m->host_dev->err);
        return LIO_E_CAPACITY;
    }
    return LIO_OK;
}

// live kernel timing: a recycled pool of event pairs per kernel class, resolved lazily
struct KernelTimer {
    static constexpr int kPool = 1024;
    int on = 0;  // bit mask of timed kernel classes: 1 kNN, 2 linearize(+report), 4 unused
    hipEvent_t ev[3][kPool][2];
    int used[3] = {0, 0, 0};
    double us[3] = {0, 0, 0};
    uint32_t launches[3] = {0, 0, 0};
    bool created = false;
    void resolve(hipStream_t st) {
        hipStreamSynchronize(st);
        for (int w = 0; w < 3; w++) {
            for (int i = 0; i < used[w]; i++) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, ev[w][i][0], ev[w][i][1]) == hipSuccess) { us[w] += (double)ms * 1000.0; launches[w]++; }
            }
            used[w] = 0;
        }
    }
};

void kt_begin(lio_scan* s, int which) {
    KernelTimer* k = s->kt;
    if (!k || !(k->on & (1 << which))) return;
    if (k->used[which] >= KernelTimer::kPool) k->resolve(s->stream);
    hipEventRecord(k->ev[which][k->used[which]][0], s->stream);
}
void kt_end(lio_scan* s, int which) {
    KernelTimer* k = s->kt;
    if (!k || !(k->on & (1 << which))) return;
    hipEventRecord(k->ev[which][k->used[which]][1], s->stream);
    k->used[which]++;
}

// wait for the record of the last linearize launch: spin on the sequence word in mapped host memory (a few us
// cheaper than hipStreamSynchronize per filter pass); falls back to a stream sync if the kernel never reports
int wait_report(lio_scan* s) {
    volatile uint32_t* seq = &s->h_result->seq;
    const uint32_t want = s->seq_expected;
    for (uint64_t spin = 0; *seq != want; spin++) {
        __builtin_ia32_pause();
        if (spin > 4000 && (spin & 63) == 0) sched_yield();  // a pass normally reports within ~30 us; beyond that let other engines' threads run
        if (spin > 20000000ull) {  // >= 100 ms: something went wrong on the device
            LIO_HIP_TRY(hipStreamSynchronize(s->stream));
            if (*seq != want) { set_error("linearize_kernel did not report (seq %u, expected %u)", *seq, want); return LIO_E_DEVICE; }
            break;
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return LIO_OK;
}

}  // namespace lio

using namespace lio;

// key_mode 0: an iVox map (with the per-point insertion sequence numbers the tie-exact neighbour redo reads); 1 / 2: the Gaussian-voxel grids of
// the NDT / VGICP matchers, which have no neighbour lists to break ties in
lio_map* lio::map_create_mode(int device, float resolution, uint64_t max_points, uint64_t max_voxels, int stencil, int key_mode) {
    if (!(resolution > 0.f) || max_points == 0 || max_voxels == 0) { set_error("lio_map_create: bad argument"); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { set_error("lio_map_create: no HIP device %d (this library has no CPU fallback)", device); return nullptr; }
    lio_map* m = new lio_map();
    memset(m, 0, sizeof(*m));
    m->device = device;
    m->res = resolution;
    m->inv_res = 1.0f / resolution;
    m->max_points = max_points;
    m->max_voxels = max_voxels;
    m->key_mode = key_mode;
    m->tie_mode = 1;
    if (key_mode == 0) {  // LIO_TIE_MODE=0|1|2: the default of lio_map_set_tie_mode for maps made from now on (A/B runs of unchanged callers)
        const char* tm = getenv("LIO_TIE_MODE");
        if (tm && tm[0] >= '0' && tm[0] <= '2' && !tm[1]) m->tie_mode = tm[0] - '0';
    }
    if (fill_stencil(m->stencil, stencil) != LIO_OK) { set_error("lio_map_create: stencil must be 1, 7, 19, 27 or 75"); delete m; return nullptr; }
    m->stencil_id = stencil;
    uint64_t cap = 1024;
    while (cap < max_voxels * 2) cap <<= 1;  // load factor <= 0.5
    if (cap > 0x40000000ull) { set_error("lio_map_create: max_voxels too large"); delete m; return nullptr; }
    m->table_cap = (uint32_t)cap;
    m->table_mask = (uint32_t)cap - 1;
    // voxel regions double when they fill up and the old region is not recycled: <= 2 x (2 x points + 8 x voxels)
    m->pool_cap = 4 * max_points + 16 * max_voxels;
    if (m->pool_cap > 0xFFFFFFF0ull) m->pool_cap = 0xFFFFFFF0ull;
    m->slot_of_point_cap = max_points;
    m->stage_cap = 1u << 20;
    bool ok = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) == hipSuccess;
    m->own_stream = ok;
    ok = ok && dev_alloc(&m->table, cap, &m->bytes) && dev_alloc(&m->cap, cap, &m->bytes) && dev_alloc(&m->pending, cap, &m->bytes) &&
         dev_alloc(&m->created, cap, &m->bytes) && dev_alloc(&m->pool, m->pool_cap, &m->bytes) && dev_alloc(&m->dev, 1, &m->bytes) &&
         dev_alloc(&m->slot_of_point, m->slot_of_point_cap, &m->bytes) && dev_alloc(&m->tile_sum, cap / 2048 + 1, &m->bytes) && dev_alloc(&m->stage, m->stage_cap, &m->bytes);
    if (key_mode == 0) ok = ok && dev_alloc(&m->pool_seq, m->pool_cap, &m->bytes);
    ok = ok && hipHostMalloc(reinterpret_cast<void**>(&m->host_dev), sizeof(MapDev)) == hipSuccess;
    if (ok) {
        ok = hipMemsetAsync(m->table, 0xFF, cap * sizeof(Slot), m->stream) == hipSuccess &&  // key = empty; ptr/cnt fixed below
             hipMemsetAsync(m->cap, 0, cap * 4, m->stream) == hipSuccess && hipMemsetAsync(m->pending, 0, cap * 4, m->stream) == hipSuccess &&
             hipMemsetAsync(m->dev, 0, sizeof(MapDev), m->stream) == hipSuccess;
    }
    if (ok) {
        // cnt must start at 0: clear the (ptr, cnt) halves with a strided 2D memset
        ok = hipMemset2DAsync(reinterpret_cast<char*>(m->table) + 8, sizeof(Slot), 0, 8, cap, m->stream) == hipSuccess &&
             hipStreamSynchronize(m->stream) == hipSuccess;
    }
    if (!ok) {
        if (!g_err[0]) set_error("lio_map_create: device setup failed: %s", hipGetErrorString(hipGetLastError()));
        lio_map_destroy(m);
        return nullptr;
    }
    return m;
}


// the bitmap behind lio_map_knn_unique (one bit per pool entry), made when the counting variant of the kNN kernel is first asked for
int lio::map_enable_touch_bits(lio_map* m) {
    if (m->touch_bits) return LIO_OK;
    hipSetDevice(m->device);
    if (!dev_alloc(&m->touch_bits, (m->pool_cap + 31) / 32, &m->bytes)) return LIO_E_DEVICE;
    LIO_HIP_TRY(hipMemcpy(&m->dev->touch_bits, &m->touch_bits, sizeof(uint32_t*), hipMemcpyHostToDevice));
    return LIO_OK;
}

extern "C" {

const char* lio_last_error(void) { return g_err; }
const char* lio_last_warning(void) { return g_warn; }

int lio_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int lio_abi_version(void) { return LIO_ABI_VERSION; }

void* lio_pinned_alloc(uint64_t bytes) {
    void* p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) { set_error("lio_pinned_alloc: %llu bytes of page-locked memory not available", (unsigned long long)bytes); return nullptr; }
    return p;
}

void lio_pinned_free(void* p) {
    if (p) (void)hipHostFree(p);
}

uint64_t lio_map_bytes(const lio_map* m) { return m ? m->bytes : 0; }

lio_map* lio_map_create(int device, float resolution, uint64_t max_points, uint64_t max_voxels, int stencil) {
    return map_create_mode(device, resolution, max_points, max_voxels, stencil, 0);
}

// back to an empty map without giving the memory back (a matcher's setInputTarget replaces its target: hipFree + hipMalloc of the pool and the
// table cost milliseconds, five memsets microseconds); asynchronous on the map's stream.  Not for maps with the LRU list on.
int map_clear(lio_map* m) {
    if (!m) return LIO_E_INVALID;
    map_settle(m);
    hipStream_t st = m->stream;
    LIO_HIP_TRY(hipMemsetAsync(m->table, 0xFF, (size_t)m->table_cap * sizeof(Slot), st));
    LIO_HIP_TRY(hipMemset2DAsync(reinterpret_cast<char*>(m->table) + 8, sizeof(Slot), 0, 8, m->table_cap, st));
    LIO_HIP_TRY(hipMemsetAsync(m->cap, 0, (size_t)m->table_cap * 4, st));
    LIO_HIP_TRY(hipMemsetAsync(m->pending, 0, (size_t)m->table_cap * 4, st));
    uint32_t* keep_bits = m->touch_bits;
    LIO_HIP_TRY(hipMemsetAsync(m->dev, 0, sizeof(MapDev), st));
    if (keep_bits) LIO_HIP_TRY(hipMemcpyAsync(&m->dev->touch_bits, &m->touch_bits, sizeof(uint32_t*), hipMemcpyHostToDevice, st));
    if (m->lru_capacity) {  // the LRU list starts over too: no stamps, an empty touch log, empty free lists (their counters live in MapDev)
        LIO_HIP_TRY(hipMemsetAsync(m->touch, 0, (size_t)m->table_cap * 8, st));
        LIO_HIP_TRY(hipMemsetAsync(m->prev_touch, 0, (size_t)m->table_cap * 8, st));
        LIO_HIP_TRY(hipMemsetAsync(m->lru_log, 0, (size_t)m->lru_log_cap * sizeof(LruEntry), st));
        if (m->first_touch) LIO_HIP_TRY(hipMemsetAsync(m->first_touch, 0, (size_t)m->table_cap * 8, st));  // (the batch numbers start over: an old stamp would beat every new one)
        m->tomb_bound = 0;
    }
    m->n_batches = 0;
    return LIO_OK;
}

void lio_map_destroy(lio_map* m) {
    if (!m) return;
    hipSetDevice(m->device);
    if (m->stream) hipStreamSynchronize(m->stream);
    hipFree(m->table); hipFree(m->cap); hipFree(m->pending); hipFree(m->created); hipFree(m->pool); hipFree(m->pool_seq); hipFree(m->touch_bits); hipFree(m->dev);
    hipFree(m->slot_of_point); hipFree(m->tile_sum); hipFree(m->stage);
    hipFree(m->touch); hipFree(m->prev_touch); hipFree(m->touch2); hipFree(m->prev_touch2); hipFree(m->lru_log); hipFree(m->free_items); hipFree(m->free_in);
    hipFree(m->first_touch); hipFree(m->lru_g); hipFree(m->lru_rec);
    hipFree(m->table2); hipFree(m->cap2); hipFree(m->pending2); hipFree(m->created2); hipFree(m->remap);
    if (m->host_dev) hipHostFree(m->host_dev);
    if (m->ev_classified) hipEventDestroy(m->ev_classified);
    if (m->ev_inserted) hipEventDestroy(m->ev_inserted);
    if (m->stream && m->own_stream) hipStreamDestroy(m->stream);
    delete m;
}

// IVox::Options capacity_ / max_distance_ (ivox3d.h:46-52, set at laserMapping.cpp:1060-1064): enable the LRU list
int lio_map_set_lru(lio_map* m, uint64_t capacity_voxels, double max_distance) {
    if (!m || capacity_voxels == 0 || !(max_distance >= 0)) return LIO_E_INVALID;
    if (m->key_mode != 0) { set_error("lio_map_set_lru: only iVox maps evict"); return LIO_E_STATE; }
    if (m->n_batches != 0) { set_error("lio_map_set_lru: call before the first insert"); return LIO_E_STATE; }
    if (capacity_voxels >= m->max_voxels) { set_error("lio_map_set_lru: capacity %llu must be below max_voxels %llu (the map may exceed its capacity while no voxel is old enough to go)", (unsigned long long)capacity_voxels, (unsigned long long)m->max_voxels); return LIO_E_INVALID; }
    hipSetDevice(m->device);
    const size_t cap = m->table_cap;
    uint64_t lc = 1024;
    while (lc < 8 * (uint64_t)m->max_voxels) lc <<= 1;  // live entries <= voxels; stale ones are skipped as the tail passes them
    m->lru_log_cap = lc;
    m->free_cap = (uint32_t)m->max_voxels;
    bool ok = dev_alloc(&m->touch, cap, &m->bytes) && dev_alloc(&m->prev_touch, cap, &m->bytes) && dev_alloc(&m->touch2, cap, &m->bytes) &&
              dev_alloc(&m->prev_touch2, cap, &m->bytes) && dev_alloc(&m->lru_log, lc, &m->bytes) && dev_alloc(&m->free_items, (uint64_t)24 * m->free_cap, &m->bytes) && dev_alloc(&m->free_in, (uint64_t)24 * m->free_cap, &m->bytes) &&
              dev_alloc(&m->table2, cap, &m->bytes) && dev_alloc(&m->cap2, cap, &m->bytes) && dev_alloc(&m->pending2, cap, &m->bytes) &&
              dev_alloc(&m->created2, cap, &m->bytes) && dev_alloc(&m->remap, cap, &m->bytes);
    {   // the reference's point-by-point order inside a batch (hashmap.hip lru_exact_*): LIO_LRU_EXACT=0 leaves it counted, not followed
        const char* ex = getenv("LIO_LRU_EXACT");
        if (!(ex && ex[0] == '0'))
            ok = ok && dev_alloc(&m->first_touch, cap, &m->bytes) && dev_alloc(&m->lru_g, m->slot_of_point_cap, &m->bytes) && dev_alloc(&m->lru_rec, (uint64_t)kLruRecCap, &m->bytes) &&
                 hipMemsetAsync(m->first_touch, 0, cap * 8, m->stream) == hipSuccess;
    }
    ok = ok && hipMemsetAsync(m->touch, 0, cap * 8, m->stream) == hipSuccess && hipMemsetAsync(m->prev_touch, 0, cap * 8, m->stream) == hipSuccess &&
         hipMemsetAsync(m->prev_touch2, 0, cap * 8, m->stream) == hipSuccess && hipMemsetAsync(m->lru_log, 0, lc * sizeof(LruEntry), m->stream) == hipSuccess &&
         hipStreamSynchronize(m->stream) == hipSuccess;
    if (!ok) { if (!g_err[0]) set_error("lio_map_set_lru: allocation failed"); return LIO_E_DEVICE; }
    m->lru_capacity = capacity_voxels;
    m->lru_max_distance = (float)max_distance;
    return LIO_OK;
}

int lio_map_lru_stats(lio_map* m, uint64_t* n_evicted, uint64_t* n_interleaved) {
    if (!m) return LIO_E_INVALID;
    hipSetDevice(m->device);
    const int rc = map_check(m, m->stream);
    if (n_evicted) *n_evicted = m->host_dev->n_evicted;
    if (n_interleaved) *n_interleaved = m->host_dev->n_lru_interleaved;
    return rc;
}

int lio_map_lru_exact_stats(lio_map* m, uint64_t* n_recreated, uint64_t* n_batches_not_followed) {
    if (!m) return LIO_E_INVALID;
    hipSetDevice(m->device);
    const int rc = map_check(m, m->stream);
    if (n_recreated) *n_recreated = m->host_dev->n_lru_recreated;
    if (n_batches_not_followed) *n_batches_not_followed = m->host_dev->n_lru_inexact;
    return rc;
}

int lio_map_clear(lio_map* m) {
    if (!m) return LIO_E_INVALID;
    hipSetDevice(m->device);
    const int rc = map_clear(m);
    if (rc != LIO_OK) return rc;
    LIO_HIP_TRY(hipStreamSynchronize(m->stream));
    return LIO_OK;
}

int lio_map_set_tie_mode(lio_map* m, int mode) {
    if (!m || mode < 0 || mode > 2) return LIO_E_INVALID;
    if (mode != 0 && !m->pool_seq) { set_error("lio_map_set_tie_mode: not an iVox map"); return LIO_E_STATE; }
    m->tie_mode = mode;
    return LIO_OK;
}

int lio_map_tie_stats(lio_map* m, uint64_t* n_boundary, uint64_t* n_unresolved) {
    if (!m) return LIO_E_INVALID;
    hipSetDevice(m->device);
    const int rc = map_check(m, m->stream);
    if (n_boundary) *n_boundary = m->host_dev->n_tie_boundary;
    if (n_unresolved) *n_unresolved = m->host_dev->n_tie_unresolved;
    return rc;
}

int lio_map_set_stencil(lio_map* m, int stencil) {
    if (!m) return LIO_E_INVALID;
    StencilArgs st;
    if (fill_stencil(st, stencil) != LIO_OK) { set_error("stencil must be 1, 7, 19, 27 or 75"); return LIO_E_INVALID; }
    m->stencil = st;
    m->stencil_id = stencil;
    return LIO_OK;
}

int lio_map_insert_device(lio_map* m, const void* d_pts, uint64_t n, double travel) {
    if (!m || (!d_pts && n)) return LIO_E_INVALID;
    hipSetDevice(m->device);
    const uint64_t chunk = m->slot_of_point_cap;
    for (uint64_t a = 0; a < n; a += chunk) {
        const uint64_t c = (n - a) < chunk ? (n - a) : chunk;
        const int rc = map_insert_dev(m, m->stream, reinterpret_cast<const float4*>(d_pts) + a, c, nullptr, travel);
        if (rc != LIO_OK) return rc;
    }
    return map_check(m, m->stream);
}

int lio_map_insert(lio_map* m, const float* pts, uint64_t n, double travel) {
    if (!m || (!pts && n)) return LIO_E_INVALID;
    if (n == 0) return LIO_OK;
    hipSetDevice(m->device);
    float4* tmp = nullptr;
    LIO_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&tmp), n * sizeof(float4)));
    int rc = LIO_OK;
    if (hipMemcpyAsync(tmp, pts, n * sizeof(float4), hipMemcpyHostToDevice, m->stream) != hipSuccess) {
        set_error("lio_map_insert: upload failed");
        rc = LIO_E_DEVICE;
    } else {
        rc = lio_map_insert_device(m, tmp, n, travel);
    }
    hipStreamSynchronize(m->stream);
    hipFree(tmp);
    return rc;
}

int lio_map_stats(lio_map* m, uint64_t* n_points, uint64_t* n_voxels) {
    if (!m) return LIO_E_INVALID;
    hipSetDevice(m->device);
    const int rc = map_check(m, m->stream);
    if (n_points) *n_points = m->host_dev->n_points;
    if (n_voxels) *n_voxels = m->host_dev->n_voxels;
    return rc;
}


int lio_map_pool_stats(lio_map* m, uint64_t* pool_top, uint64_t* pool_cap) {
    if (!m) return LIO_E_INVALID;
    hipSetDevice(m->device);
    const int rc = map_check(m, m->stream);
    if (pool_top) *pool_top = m->host_dev->pool_top;
    if (pool_cap) *pool_cap = m->pool_cap;
    return rc;
}

uint64_t lio_map_knn_candidates(lio_map* m) {
    if (!m) return 0;
    hipSetDevice(m->device);
    MapDev tmp;
    if (hipMemcpy(&tmp, m->dev, sizeof(MapDev), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    uint64_t s = 0;
    for (int k = 0; k < 64; k++) s += tmp.knn_cand[k * 16];
    return s;
}

uint64_t lio_map_knn_touched(lio_map* m) {
    if (!m) return 0;
    hipSetDevice(m->device);
    MapDev tmp;
    if (hipMemcpy(&tmp, m->dev, sizeof(MapDev), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    uint64_t s = 0;
    for (int k = 0; k < 64; k++) s += tmp.knn_cand[k * 16 + 1];
    return s;
}

uint64_t lio_map_knn_unique(lio_map* m) {
    if (!m) return 0;
    hipSetDevice(m->device);
    MapDev tmp;
    if (hipMemcpy(&tmp, m->dev, sizeof(MapDev), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    uint64_t s = 0;
    for (int k = 0; k < 64; k++) s += tmp.knn_cand[k * 16 + 2];
    return s;
}

int64_t lio_map_dump(lio_map* m, float* out, uint64_t cap_points) {
    if (!m) return LIO_E_INVALID;
    hipSetDevice(m->device);
    if (map_check(m, m->stream) == LIO_E_DEVICE) return LIO_E_DEVICE;
    const uint64_t n = m->host_dev->n_points;
    if (n > cap_points || !out) return -(int64_t)n;
    std::vector<Slot> tab(m->table_cap);
    if (hipMemcpy(tab.data(), m->table, sizeof(Slot) * m->table_cap, hipMemcpyDeviceToHost) != hipSuccess) return LIO_E_DEVICE;
    uint64_t k = 0;
    for (uint32_t h = 0; h < m->table_cap; h++) {
        if (tab[h].key == kEmptyKey || tab[h].key == kTombKey || tab[h].cnt == 0) continue;
        if (k + tab[h].cnt > cap_points) return LIO_E_CAPACITY;
        if (hipMemcpy(out + k * 4, m->pool + tab[h].ptr, sizeof(float4) * tab[h].cnt, hipMemcpyDeviceToHost) != hipSuccess) return LIO_E_DEVICE;
        k += tab[h].cnt;
    }
    return (int64_t)k;
}

int lio_map_knn(lio_map* m, const float* q, uint32_t n, float* out_pts, int32_t* out_cnt) {
    if (!m || !q || !out_pts || !out_cnt) return LIO_E_INVALID;
    if (n == 0) return LIO_OK;
    hipSetDevice(m->device);
    float4 *dq = nullptr, *dout = nullptr;
    int32_t* dcnt = nullptr;
    uint32_t* dtie = nullptr;
    int rc = LIO_OK;
    if (hipMalloc(reinterpret_cast<void**>(&dq), n * sizeof(float4)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&dtie), ((size_t)n + 1) * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&dout), (size_t)n * 5 * sizeof(float4)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&dcnt), n * sizeof(int32_t)) != hipSuccess) {
        set_error("lio_map_knn: hipMalloc failed");
        rc = LIO_E_DEVICE;
    }
    if (rc == LIO_OK) {
        hipMemcpyAsync(dq, q, n * sizeof(float4), hipMemcpyHostToDevice, m->stream);
        hipMemsetAsync(dout, 0, (size_t)n * 5 * sizeof(float4), m->stream);
        hipMemsetAsync(dcnt, 0, n * sizeof(int32_t), m->stream);
        hipMemsetAsync(dtie, 0, sizeof(uint32_t), m->stream);
        rc = knn_batch(m, dq, n, dout, dcnt, dtie);
    }
    if (rc == LIO_OK) {
        std::vector<float4> soa((size_t)n * 5);
        if (hipMemcpyAsync(soa.data(), dout, soa.size() * sizeof(float4), hipMemcpyDeviceToHost, m->stream) != hipSuccess ||
            hipMemcpyAsync(out_cnt, dcnt, n * sizeof(int32_t), hipMemcpyDeviceToHost, m->stream) != hipSuccess ||
            hipStreamSynchronize(m->stream) != hipSuccess) {
            set_error("lio_map_knn: %s", hipGetErrorString(hipGetLastError()));
            rc = LIO_E_DEVICE;
        } else {
            for (uint32_t i = 0; i < n; i++)
                for (int k = 0; k < 5; k++) memcpy(out_pts + ((size_t)i * 5 + k) * 4, &soa[(size_t)k * n + i], sizeof(float4));
        }
    }
    hipFree(dq); hipFree(dout); hipFree(dcnt); hipFree(dtie);
    return rc;
}

// ---------------------------------------------------------------------------------------------------
lio_scan* lio_scan_create(int device, uint32_t max_raw, uint32_t max_ds) {
    if (max_raw == 0 || max_ds == 0) { set_error("lio_scan_create: bad argument"); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { set_error("lio_scan_create: no HIP device %d (this library has no CPU fallback)", device); return nullptr; }
    lio_scan* s = new lio_scan();
    memset(s, 0, sizeof(*s));
    s->device = device;
    s->max_raw = max_raw;
    s->max_ds = max_ds;
    const uint32_t nblocks = (max_raw + 1023) / 1024;
    s->partial_blocks = (max_ds + kLinThreads - 1) / kLinThreads;
    bool ok = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && dev_alloc(&s->raw_own, max_raw, &s->bytes) && dev_alloc(&s->ds_body, max_ds, &s->bytes) && dev_alloc(&s->ds_world, max_ds, &s->bytes) &&
         dev_alloc(&s->nn_pts, (uint64_t)max_ds * 5, &s->bytes) && dev_alloc(&s->nn_cnt, max_ds, &s->bytes) && dev_alloc(&s->selected, max_ds, &s->bytes) &&
         dev_alloc(&s->normvec, (uint64_t)max_ds + (max_ds + 3) / 4, &s->bytes) && dev_alloc(&s->keys_a, max_raw, &s->bytes) && dev_alloc(&s->keys_b, max_raw, &s->bytes) &&
         dev_alloc(&s->vals_a, max_raw, &s->bytes) && dev_alloc(&s->vals_b, max_raw, &s->bytes) && dev_alloc(&s->hist, std::max<uint64_t>((uint64_t)256 * (nblocks + 1) /* + the digit totals' row of the batched chain */, 2 * (((uint64_t)max_ds + 255) / 256) + 2), &s->bytes) &&
         dev_alloc(&s->blockcnt, nblocks, &s->bytes) && dev_alloc(&s->hpos, (uint64_t)max_ds + 1, &s->bytes) && dev_alloc(&s->longlist, max_ds, &s->bytes) && dev_alloc(&s->tie_list, max_ds, &s->bytes) && dev_alloc(&s->sorted, max_raw, &s->bytes) && dev_alloc(&s->partial, (uint64_t)s->partial_blocks * kAcc, &s->bytes) &&
         dev_alloc(&s->dev, 1, &s->bytes) && dev_alloc(&s->d_result, 1, &s->bytes);
    ok = ok && hipHostMalloc(reinterpret_cast<void**>(&s->host_dev), sizeof(ScanDev)) == hipSuccess &&
         hipHostMalloc(reinterpret_cast<void**>(&s->h_result), sizeof(lio_normal_eq), hipHostMallocMapped) == hipSuccess &&
         hipHostMalloc(reinterpret_cast<void**>(&s->host_nds), 4 * sizeof(uint32_t), hipHostMallocMapped) == hipSuccess &&
         hipHostGetDevicePointer(reinterpret_cast<void**>(&s->host_nds_dev), s->host_nds, 0) == hipSuccess &&
         hipHostGetDevicePointer(reinterpret_cast<void**>(&s->h_result_dev), s->h_result, 0) == hipSuccess;
    if (ok) {
        // point_selected_surf starts all-true (laserMapping.cpp:1046), Nearest_Points empty
        ok = hipMemsetAsync(s->selected, 1, max_ds, s->stream) == hipSuccess && hipMemsetAsync(s->nn_cnt, 0, (size_t)max_ds * 4, s->stream) == hipSuccess &&
             hipMemsetAsync(s->nn_pts, 0, (size_t)max_ds * 5 * sizeof(float4), s->stream) == hipSuccess &&
             hipMemsetAsync(s->normvec, 0, ((size_t)max_ds + (max_ds + 3) / 4) * sizeof(float4), s->stream) == hipSuccess &&
             hipMemsetAsync(s->dev, 0, sizeof(ScanDev), s->stream) == hipSuccess &&
             hipMemsetAsync(s->dev->bbox_min, 0xFF, 12, s->stream) == hipSuccess && hipStreamSynchronize(s->stream) == hipSuccess;
    }
    if (!ok) {
        if (!g_err[0]) set_error("lio_scan_create: device setup failed: %s", hipGetErrorString(hipGetLastError()));
        lio_scan_destroy(s);
        return nullptr;
    }
    memset(s->h_result, 0, sizeof(lio_normal_eq));
    s->host_nds[0] = s->host_nds[1] = s->host_nds[2] = s->host_nds[3] = 0;
    s->pred_passes = 4;
    s->raw = s->raw_own;
    return s;
}

void lio_scan_destroy(lio_scan* s) {
    if (!s) return;
    hipSetDevice(s->device);
    if (s->stream) hipStreamSynchronize(s->stream);
    hipFree(s->raw_own); hipFree(s->ds_body); hipFree(s->ds_world); hipFree(s->nn_pts); hipFree(s->nn_cnt); hipFree(s->selected);
    hipFree(s->normvec); hipFree(s->keys_a); hipFree(s->keys_b); hipFree(s->vals_a); hipFree(s->vals_b); hipFree(s->hist);
    hipFree(s->blockcnt); hipFree(s->hpos); hipFree(s->longlist); hipFree(s->tie_list); hipFree(s->sorted); hipFree(s->partial); hipFree(s->dev); hipFree(s->d_result);
    if (s->host_dev) hipHostFree(s->host_dev);
    if (s->h_result) hipHostFree(s->h_result);
    if (s->host_nds) hipHostFree(s->host_nds);
    if (s->d_batch_desc) hipFree(s->d_batch_desc);
    if (s->h_batch_desc) hipHostFree(s->h_batch_desc);
    if (s->kt) {
        if (s->kt->created)
            for (int w = 0; w < 3; w++)
                for (int i = 0; i < KernelTimer::kPool; i++) { hipEventDestroy(s->kt->ev[w][i][0]); hipEventDestroy(s->kt->ev[w][i][1]); }
        delete s->kt;
    }
    if (s->stream) hipStreamDestroy(s->stream);
    delete s;
}

int lio_scan_reset(lio_scan* s) {
    if (!s) return LIO_E_INVALID;
    hipSetDevice(s->device);
    LIO_HIP_TRY(hipMemsetAsync(s->selected, 1, s->max_ds, s->stream));
    LIO_HIP_TRY(hipMemsetAsync(s->nn_cnt, 0, (size_t)s->max_ds * 4, s->stream));
    LIO_HIP_TRY(hipMemsetAsync(s->nn_pts, 0, (size_t)s->max_ds * 5 * sizeof(float4), s->stream));
    // (the planes a searching pass left for the passes that do not search go with the neighbours they were fitted to: linearize_body)
    LIO_HIP_TRY(hipMemsetAsync(s->normvec, 0, ((size_t)s->max_ds + (s->max_ds + 3) / 4) * sizeof(float4), s->stream));
    LIO_HIP_TRY(hipMemsetAsync(&s->dev->cache_n, 0, sizeof(uint32_t), s->stream));
    LIO_HIP_TRY(hipStreamSynchronize(s->stream));
    return LIO_OK;
}

// the neighbour cache alone, asynchronously on the scan's stream: what an independent job of lio_engines_process_batch starts from
int scan_forget_cache(lio_scan* s) {
    if (!s) return LIO_E_INVALID;
    LIO_HIP_TRY(hipMemsetAsync(s->nn_cnt, 0, (size_t)s->max_ds * 4, s->stream));
    LIO_HIP_TRY(hipMemsetAsync(&s->dev->cache_n, 0, sizeof(uint32_t), s->stream));
    return LIO_OK;
}

int lio_scan_enable_kernel_timing(lio_scan* s, int on) {
    if (!s) return LIO_E_INVALID;
    hipSetDevice(s->device);
    if (!s->kt) s->kt = new KernelTimer();
    if (on && !s->kt->created) {
        for (int w = 0; w < 3; w++)
            for (int i = 0; i < KernelTimer::kPool; i++) {
                LIO_HIP_TRY(hipEventCreate(&s->kt->ev[w][i][0]));
                LIO_HIP_TRY(hipEventCreate(&s->kt->ev[w][i][1]));
            }
        s->kt->created = true;
    }
    s->kt->on = on & 7;
    return LIO_OK;
}

int lio_scan_kernel_times(lio_scan* s, lio_kernel_times* out, int reset) {
    if (!s || !out) return LIO_E_INVALID;
    memset(out, 0, sizeof(*out));
    if (!s->kt) return LIO_OK;
    hipSetDevice(s->device);
    s->kt->resolve(s->stream);
    out->knn_us = s->kt->us[0]; out->linearize_us = s->kt->us[1]; out->finalize_us = s->kt->us[2];
    out->knn_launches = s->kt->launches[0]; out->linearize_launches = s->kt->launches[1]; out->finalize_launches = s->kt->launches[2];
    if (reset)
        for (int w = 0; w < 3; w++) { s->kt->us[w] = 0; s->kt->launches[w] = 0; }
    return LIO_OK;
}

int lio_scan_upload(lio_scan* s, const float* body, uint32_t n_raw) {
    if (!s || (!body && n_raw)) return LIO_E_INVALID;
    if (n_raw > s->max_raw) { set_error("scan of %u points exceeds max_raw %u", n_raw, s->max_raw); return LIO_E_CAPACITY; }
    hipSetDevice(s->device);
    if (n_raw) LIO_HIP_TRY(hipMemcpyAsync(s->raw_own, body, (size_t)n_raw * sizeof(float4), hipMemcpyHostToDevice, s->stream));
    s->raw = s->raw_own;
    s->n_raw = n_raw;
    return LIO_OK;
}

int lio_scan_set_device(lio_scan* s, const void* d_body, uint32_t n_raw) {
    if (!s || (!d_body && n_raw)) return LIO_E_INVALID;
    if (n_raw > s->max_raw) { set_error("scan of %u points exceeds max_raw %u", n_raw, s->max_raw); return LIO_E_CAPACITY; }
    s->raw = reinterpret_cast<const float4*>(d_body);
    s->n_raw = n_raw;
    return LIO_OK;
}

// ---- constant-velocity motion compensation of the localisation mode (slam_utils.cpp:163-191) ----
// Host part, once per scan, in f32 exactly as Eigen evaluates it there: Quaternionf(delta.block<3,3>(0,0)) (Shoemake, Quaternion.h
// quaternionbase_assign_impl<Other,3,3>), AngleAxisf(quaternion) (AngleAxis.h:170-190, with stableNorm below epsilon).
static void delta_pose_args(const float D[16], double scan_period, DeltaArgs& A) {
    const float m[3][3] = {{D[0], D[1], D[2]}, {D[4], D[5], D[6]}, {D[8], D[9], D[10]}};
    float q[4];  // x y z w
    float t = m[0][0] + (m[1][1] + m[2][2]);
    if (t > 0.f) {
        t = sqrtf(t + 1.0f);
        q[3] = 0.5f * t;
        t = 0.5f / t;
        q[0] = (m[2][1] - m[1][2]) * t;
        q[1] = (m[0][2] - m[2][0]) * t;
        q[2] = (m[1][0] - m[0][1]) * t;
    } else {
        int i = 0;
        if (m[1][1] > m[0][0]) i = 1;
        if (m[2][2] > m[i][i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrtf(m[i][i] - m[j][j] - m[k][k] + 1.0f);
        q[i] = 0.5f * t;
        t = 0.5f / t;
        q[3] = (m[k][j] - m[j][k]) * t;
        q[j] = (m[j][i] + m[i][j]) * t;
        q[k] = (m[k][i] + m[i][k]) * t;
    }
    float n = sqrtf(q[0] * q[0] + (q[1] * q[1] + q[2] * q[2]));
    if (n < 1.1920929e-07f) {  // stableNorm (StableNorm.h:18-50 on one 3-segment): scaled by the largest |coefficient|, summed in sequence
        const float mx = fmaxf(fabsf(q[0]), fmaxf(fabsf(q[1]), fabsf(q[2])));
        float scale = 0.f, inv = 1.f, ssq = 0.f;
        if (mx > scale) {
            const float tmp = 1.f / mx;
            if (tmp > 3.4028235e+38f) { inv = 3.4028235e+38f; scale = 1.f / inv; }
            else { scale = mx; inv = tmp; }
        }
        if (scale > 0.f) {
            const float a = q[0] * inv, b = q[1] * inv, c2 = q[2] * inv;
            ssq = (a * a + b * b) + c2 * c2;
        }
        n = scale * sqrtf(ssq);
    }
    float angle = 0.f, axis[3] = {1.f, 0.f, 0.f};
    if (n != 0.f) {
        angle = 2.f * atan2f(n, fabsf(q[3]));
        if (q[3] < 0.f) n = -n;
        for (int i = 0; i < 3; i++) axis[i] = q[i] / n;
    }
    for (int i = 0; i < 3; i++) {
        A.t[i] = D[4 * i + 3];
        A.aa[i] = angle * axis[i];
    }
    A.scan_period = scan_period;
}

int lio_scan_undistort_delta(lio_scan* s, const uint32_t* stamp_us, int stamps_on_device, const float delta_pose[16], double scan_period) {
    if (!s || !delta_pose || (!stamp_us && s->n_raw)) return LIO_E_INVALID;
    if (s->n_raw == 0) return LIO_OK;
    hipSetDevice(s->device);
    const uint32_t* d_stamp = stamp_us;
    if (!stamps_on_device) {  // the key buffer of the downsample is free until lio_scan_voxel_downsample runs (same stream)
        LIO_HIP_TRY(hipMemcpyAsync(s->keys_b, stamp_us, (size_t)s->n_raw * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
        d_stamp = s->keys_b;
    }
    DeltaArgs A;
    delta_pose_args(delta_pose, scan_period, A);
    const int rc = undistort_delta_launch(s->stream, s->raw, d_stamp, s->n_raw, s->raw_own, A);  // in place when the scan owns its cloud
    if (rc != LIO_OK) return rc;
    s->raw = s->raw_own;
    if (!stamps_on_device) LIO_HIP_TRY(hipStreamSynchronize(s->stream));  // the caller's stamp array may go away
    return LIO_OK;
}

int lio_scan_undistort_poses(lio_scan* s, const uint32_t* stamp_us, int stamps_on_device, uint64_t header_stamp_us, const uint64_t* pose_stamp_us,
                             const double* pose_T, uint32_t n_poses) {
    if (!s || !pose_stamp_us || !pose_T || (!stamp_us && s->n_raw)) return LIO_E_INVALID;
    if (n_poses > (uint32_t)kMaxPoseList) { set_error("undistort: %u poses (at most %d)", n_poses, kMaxPoseList); return LIO_E_CAPACITY; }
    if (s->n_raw == 0 || n_poses < 2) return LIO_OK;  // no interval: the reference's loop body never runs
    PoseListArgs A;
    memset(&A, 0, sizeof(A));
    A.n_poses = (int)n_poses;
    for (uint32_t i = 1; i < n_poses; i++) {
        A.limit[i] = pose_stamp_us[i] - header_stamp_us;  // unsigned, as there: a pose before the header stamp "never ends"
        if (i > 1 && A.limit[i] < A.limit[i - 1]) {
            set_error("undistort: the pose intervals must not end earlier than their predecessors (pose %u)", i);
            return LIO_E_INVALID;
        }
        float D[16];
        for (int k = 0; k < 16; k++) D[k] = (float)pose_T[16 * (size_t)i + k];  // poses[i].T.cast<float>()
        delta_pose_args(D, (double)(pose_stamp_us[i] - pose_stamp_us[0]) / 1000000.0, A.d[i]);
    }
    hipSetDevice(s->device);
    const uint32_t* d_stamp = stamp_us;
    if (!stamps_on_device) {
        LIO_HIP_TRY(hipMemcpyAsync(s->keys_b, stamp_us, (size_t)s->n_raw * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
        d_stamp = s->keys_b;
    }
    const int rc = undistort_poses_launch(s->stream, s->raw, d_stamp, s->n_raw, s->raw_own, A, s->keys_a);  // keys_a: free until the downsample
    if (rc != LIO_OK) return rc;
    s->raw = s->raw_own;
    if (!stamps_on_device) LIO_HIP_TRY(hipStreamSynchronize(s->stream));
    return LIO_OK;
}

int lio_scan_download_raw(lio_scan* s, float* out_xyzi, uint32_t cap) {
    if (!s || !out_xyzi) return LIO_E_INVALID;
    if (s->n_raw > cap) { set_error("raw cloud of %u points exceeds cap %u", s->n_raw, cap); return LIO_E_CAPACITY; }
    hipSetDevice(s->device);
    if (s->n_raw) LIO_HIP_TRY(hipMemcpyAsync(out_xyzi, s->raw, (size_t)s->n_raw * sizeof(float4), hipMemcpyDeviceToHost, s->stream));
    LIO_HIP_TRY(hipStreamSynchronize(s->stream));
    return (int)s->n_raw;
}

static int scan_sync_dev(lio_scan* s) {
    LIO_HIP_TRY(hipMemcpyAsync(s->host_dev, s->dev, sizeof(ScanDev), hipMemcpyDeviceToHost, s->stream));
    LIO_HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->host_dev->err & 1u) {
        set_error("downsampled scan exceeds max_ds %u", s->max_ds);
        hipMemsetAsync(&s->dev->err, 0, 4, s->stream);
        return LIO_E_CAPACITY;
    }
    return LIO_OK;
}

int lio_scan_voxel_downsample(lio_scan* s, float leaf, int sync, uint32_t* n_ds) {
    if (!s || !(leaf > 0.f)) return LIO_E_INVALID;
    hipSetDevice(s->device);
    // The sort needs ceil(log2(cells of the bounding box) / 8) radix passes, known only on the device.  A caller that waits for the
    // result lets the next scan launch just as many as this one needed (two launches fewer for the usual 17..24-bit keys); the device
    // checks, and a scan whose box needs more is run again with all four.  Without a wait there is no check: all four are launched.
    int passes = sync ? s->pred_passes : 4;
    for (;;) {
        int rc = vg_downsample(s, leaf, passes);
        if (rc != LIO_OK) return rc;
        rc = scan_begin(s);
        if (rc != LIO_OK) return rc;
        s->have_ds = -1;
        if (!sync) return LIO_OK;
        LIO_HIP_TRY(hipStreamSynchronize(s->stream));
        if ((s->host_nds[1] & 2u) && passes < 4) {  // under-launched sort: nothing downstream has consumed the result yet
            hipMemsetAsync(&s->dev->err, 0, 4, s->stream);
            passes = 4;
            continue;
        }
        break;
    }
    if (s->host_nds[1] & 1u) {
        set_error("downsampled scan exceeds max_ds %u", s->max_ds);
        hipMemsetAsync(&s->dev->err, 0, 4, s->stream);
        return LIO_E_CAPACITY;
    }
    const int needed = (int)s->host_nds[2];
    s->pred_passes = needed >= 1 && needed <= 4 ? needed : 4;
    s->have_ds = (int)s->host_nds[0];
    if (n_ds) *n_ds = s->host_nds[0];
    return LIO_OK;
}

// pcl::VoxelGrid of n scans with ONE set of launches: the batched chain of the throughput engine (blockIdx.y = scan), for callers that
// hold many clouds at once -- the candidate alignments of the map-merge tools, relocalisation, offline re-registration
// (lio_ndt_align_batch takes the scans as they leave here).  Same result per scan as lio_scan_voxel_downsample.
int lio_scan_voxel_downsample_batch(lio_scan** scans, int n, float leaf, uint32_t* n_ds) {
    if (!scans || n <= 0 || n > 4096 || !(leaf > 0.f)) return LIO_E_INVALID;
    for (int i = 0; i < n; i++) {
        if (!scans[i] || scans[i]->device != scans[0]->device) { set_error("lio_scan_voxel_downsample_batch: the scans of a call live on one device"); return LIO_E_INVALID; }
        for (int j = 0; j < i; j++)
            if (scans[j] == scans[i]) { set_error("lio_scan_voxel_downsample_batch: a scan appears twice"); return LIO_E_INVALID; }
    }
    lio_scan* s0 = scans[0];
    hipSetDevice(s0->device);
    if (s0->batch_desc_cap < (uint32_t)n) {
        if (s0->d_batch_desc) hipFree(s0->d_batch_desc);
        if (s0->h_batch_desc) hipHostFree(s0->h_batch_desc);
        s0->d_batch_desc = nullptr; s0->h_batch_desc = nullptr; s0->batch_desc_cap = 0;
        const uint32_t cap = (uint32_t)n < 64u ? 64u : (uint32_t)n;
        LIO_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s0->d_batch_desc), sizeof(SlotDesc) * cap));
        LIO_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s0->h_batch_desc), sizeof(SlotDesc) * cap, hipHostMallocDefault));
        s0->batch_desc_cap = cap;
    }
    uint32_t max_raw = 0, max_ds = 0;
    int passes = 1;
    for (int i = 0; i < n; i++) {
        lio_scan* s = scans[i];
        if (s != s0) LIO_HIP_TRY(hipStreamSynchronize(s->stream));  // its cloud has arrived, nothing of an earlier call still uses its buffers
        SlotDesc& d = s0->h_batch_desc[i];
        memset(&d, 0, sizeof(d));
        d.raw = s->raw;
        d.n_raw = s->n_raw;
        d.nblocks = (s->n_raw + 2047u) / 2048u;
        d.active = s->n_raw ? 1u : 0u;
        d.max_ds = s->max_ds;
        d.partial_blocks = s->partial_blocks;
        d.min_ds = s->resize_min;
        d.reset_cache = 0;
        d.sd = s->dev;
        d.keys_a = s->keys_a; d.keys_b = s->keys_b; d.vals_a = s->vals_a; d.vals_b = s->vals_b;
        d.hist = s->hist; d.blockcnt = s->blockcnt; d.hpos = s->hpos; d.longlist = s->longlist; d.tie_list = s->tie_list;
        d.sorted = s->sorted; d.ds_body = s->ds_body; d.ds_world = s->ds_world; d.nn_pts = s->nn_pts; d.normvec = s->normvec;
        d.nn_cnt = s->nn_cnt; d.selected = s->selected; d.partial = s->partial;
        d.host_nds = s->host_nds_dev;
        if (s->n_raw > max_raw) max_raw = s->n_raw;
        if (s->max_ds > max_ds) max_ds = s->max_ds;
        if (s->pred_passes > passes) passes = s->pred_passes;
        s->have_ds = -1;
        s->host_nds[0] = 0; s->host_nds[1] = 0; s->host_nds[2] = 0;
    }
    hipStream_t st = s0->stream;
    for (;;) {
        LIO_HIP_TRY(hipMemcpyAsync(s0->d_batch_desc, s0->h_batch_desc, sizeof(SlotDesc) * n, hipMemcpyHostToDevice, st));
        const int rc = vg_downsample_batch(st, s0->d_batch_desc, n, max_raw, max_ds, leaf, passes);
        if (rc != LIO_OK) return rc;
        LIO_HIP_TRY(hipStreamSynchronize(st));
        bool again = false;
        for (int i = 0; i < n; i++)
            if (scans[i]->n_raw && (scans[i]->host_nds[1] & 2u) && passes < 4) again = true;  // an under-launched sort: nothing has consumed it yet
        if (!again) break;
        for (int i = 0; i < n; i++) hipMemsetAsync(&scans[i]->dev->err, 0, 4, st);
        passes = 4;
    }
    int first_err = LIO_OK;
    for (int i = 0; i < n; i++) {
        lio_scan* s = scans[i];
        if (!s->n_raw) { s->have_ds = 0; if (n_ds) n_ds[i] = 0; continue; }
        if (s->host_nds[1] & 1u) {
            set_error("downsampled scan %d exceeds max_ds %u", i, s->max_ds);
            hipMemsetAsync(&s->dev->err, 0, 4, st);
            if (first_err == LIO_OK) first_err = LIO_E_CAPACITY;
            if (n_ds) n_ds[i] = 0;
            continue;
        }
        const int needed = (int)s->host_nds[2];
        s->pred_passes = needed >= 1 && needed <= 4 ? needed : 4;
        s->have_ds = (int)s->host_nds[0];
        if (n_ds) n_ds[i] = s->host_nds[0];
    }
    if (first_err != LIO_OK) LIO_HIP_TRY(hipStreamSynchronize(st));
    return first_err;
}

int lio_scan_set_ds(lio_scan* s, const float* ds, uint32_t n) {
    if (!s || (!ds && n)) return LIO_E_INVALID;
    if (n > s->max_ds) { set_error("%u downsampled points exceed max_ds %u", n, s->max_ds); return LIO_E_CAPACITY; }
    hipSetDevice(s->device);
    if (n) LIO_HIP_TRY(hipMemcpyAsync(s->ds_body, ds, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, s->stream));
    int rc = scan_set_nds(s, n);
    if (rc != LIO_OK) return rc;
    rc = scan_begin(s);
    if (rc != LIO_OK) return rc;
    LIO_HIP_TRY(hipStreamSynchronize(s->stream));  // the host buffer may be released by the caller
    s->have_ds = (int)n;
    return LIO_OK;
}

// the downsampled cloud of another scan on the same device (a joint registration downsamples the cloud once and hands it to the scan
// buffers of the other sub-maps): what lio_scan_set_ds does, from device memory
int scan_share_ds(lio_scan* dst, lio_scan* src, uint32_t n) {
    if (!dst || !src || dst->device != src->device) return LIO_E_INVALID;
    if (n > dst->max_ds) { set_error("%u downsampled points exceed max_ds %u", n, dst->max_ds); return LIO_E_CAPACITY; }
    hipSetDevice(dst->device);
    if (n) LIO_HIP_TRY(hipMemcpyAsync(dst->ds_body, src->ds_body, (size_t)n * sizeof(float4), hipMemcpyDeviceToDevice, dst->stream));
    int rc = scan_set_nds(dst, n);
    if (rc != LIO_OK) return rc;
    rc = scan_begin(dst);
    if (rc != LIO_OK) return rc;
    dst->have_ds = (int)n;
    dst->n_raw = src->n_raw;
    return LIO_OK;
}

int lio_scan_num_ds(lio_scan* s) {
    if (!s) return LIO_E_INVALID;
    hipSetDevice(s->device);
    const int rc = scan_sync_dev(s);
    if (rc != LIO_OK) return rc;
    s->have_ds = (int)s->host_dev->n_ds;
    return s->have_ds;
}

static int download_f4(lio_scan* s, const float4* src, float* out, uint32_t cap) {
    const int n = lio_scan_num_ds(s);
    if (n < 0) return n;
    if ((uint32_t)n > cap) return LIO_E_CAPACITY;
    if (n) {
        LIO_HIP_TRY(hipMemcpyAsync(out, src, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost, s->stream));
        LIO_HIP_TRY(hipStreamSynchronize(s->stream));
    }
    return n;
}

int lio_scan_download_ds(lio_scan* s, float* out, uint32_t cap) { return (!s || !out) ? LIO_E_INVALID : download_f4(s, s->ds_body, out, cap); }
int lio_scan_download_world(lio_scan* s, float* out, uint32_t cap) { return (!s || !out) ? LIO_E_INVALID : download_f4(s, s->ds_world, out, cap); }

int lio_scan_download_match(lio_scan* s, uint8_t* selected, float* normvec, int32_t* nn_cnt, float* nn_pts) {
    if (!s) return LIO_E_INVALID;
    const int n = lio_scan_num_ds(s);
    if (n <= 0) return n;
    if (selected) LIO_HIP_TRY(hipMemcpyAsync(selected, s->selected, n, hipMemcpyDeviceToHost, s->stream));
    if (normvec) LIO_HIP_TRY(hipMemcpyAsync(normvec, s->normvec, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost, s->stream));
    if (nn_cnt) LIO_HIP_TRY(hipMemcpyAsync(nn_cnt, s->nn_cnt, (size_t)n * 4, hipMemcpyDeviceToHost, s->stream));
    std::vector<float4> soa;
    if (nn_pts) {
        soa.resize((size_t)n * 5);
        for (int k = 0; k < 5; k++)
            LIO_HIP_TRY(hipMemcpyAsync(soa.data() + (size_t)k * n, s->nn_pts + (size_t)k * s->max_ds, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost, s->stream));
    }
    LIO_HIP_TRY(hipStreamSynchronize(s->stream));
    if (nn_pts)
        for (int i = 0; i < n; i++)
            for (int k = 0; k < 5; k++) memcpy(nn_pts + ((size_t)i * 5 + k) * 4, &soa[(size_t)k * n + i], sizeof(float4));
    return n;
}

// the launches of one linearisation (neighbour search if asked, linearise + report) without waiting for the record: several scans -- the
// sub-maps of a joint registration -- are put in flight on their own streams before any of them is waited for
int p2plane_linearize_begin(lio_map* m, lio_scan* s, const double pose_wi[7], const double ext_il[7], int redo_knn) {
    if (!m || !s || !pose_wi || !ext_il) return LIO_E_INVALID;
    if (m->device != s->device) { set_error("map and scan live on different devices"); return LIO_E_INVALID; }
    hipSetDevice(s->device);
    const PoseArgs pose = make_pose(pose_wi, ext_il);
    int rc = LIO_OK;
    if (redo_knn) {
        rc = map_knn_plane(m, s, pose, redo_knn);
        if (rc != LIO_OK) return rc;
    }
    return p2plane_reduce(m, s, pose, redo_knn);
}

int lio_p2plane_linearize(lio_map* m, lio_scan* s, const double pose_wi[7], const double ext_il[7], int redo_knn, lio_normal_eq* out) {
    if (!out) return LIO_E_INVALID;
    const int rc = p2plane_linearize_begin(m, s, pose_wi, ext_il, redo_knn);
    if (rc != LIO_OK) return rc;
    return p2plane_linearize_end(m, s, pose_wi, ext_il, redo_knn, out);
}

int p2plane_linearize_end(lio_map* m, lio_scan* s, const double pose_wi[7], const double ext_il[7], int redo_knn, lio_normal_eq* out) {
    if (!m || !s || !pose_wi || !ext_il || !out) return LIO_E_INVALID;
    hipSetDevice(s->device);
    const PoseArgs pose = make_pose(pose_wi, ext_il);
    // the reporting workgroup stores the record straight into mapped pinned host memory: no copy launch, no sync call
    int rc = wait_report(s);
    if (rc != LIO_OK) return rc;
    if (s->h_result->n_tie || (redo_knn && m->tie_mode == 2 && s->h_result->n_ds)) {  // exact d2 ties among some top-6: redo those queries exactly (tie mode 2: all of them)
        const uint32_t nt = m->tie_mode == 2 ? s->h_result->n_ds : s->h_result->n_tie;
        rc = map_knn_exact(m, s, pose, nt);
        if (rc != LIO_OK) return rc;
        rc = p2plane_reduce(m, s, pose, redo_knn);
        if (rc != LIO_OK) return rc;
        rc = wait_report(s);
        if (rc != LIO_OK) return rc;
        s->h_result->n_tie = nt;
    }
    *out = *s->h_result;
    s->have_ds = (int)out->n_ds;
    // degeneracy detection (laserMapping.cpp:934-964).  The eigen-decomposition of sum n n^T runs here on the
    // host.  A direction i is declared degenerate only if contri_i < 250 (and strong_i < 50), where
    // contri_i = sum of a_j = |n^_j . v_i| over rows with a_j > 0.1736.  Since a_j <= 1, a_j >= a_j^2 and
    //   contri_i >= sum_j a_j^2 - sum_{a_j <= 0.1736} a_j^2 >= lambda_i - 0.1736^2 N_eff.
    // When that bound is >= 250 for all three eigenvalues the test cannot fire and the per-point pass is skipped
    // (contri/strong are then reported as +inf = "not evaluated, provably not degenerate").
    eig3_sym(out->nnT, out->eigval, out->eigvec);
    bool need = s->force_degeneracy == 1;
    for (int i = 0; i < 3; i++)
        if (!(out->eigval[i] * (1.0 - 1e-5) - 0.030138 * (double)out->n_eff >= 250.0 + 1e-3)) need = true;
    if (s->force_degeneracy == 2) need = false;
    if (!need || out->n_eff == 0) {
        for (int i = 0; i < 3; i++) { out->contri[i] = INFINITY; out->strong[i] = INFINITY; }
        return LIO_OK;
    }
    LIO_HIP_TRY(hipMemcpyAsync(s->d_result->eigvec, out->eigvec, sizeof(double) * 9, hipMemcpyHostToDevice, s->stream));
    LIO_HIP_TRY(hipMemsetAsync(s->d_result->contri, 0, sizeof(double) * 6, s->stream));
    rc = p2plane_degeneracy(s);
    if (rc != LIO_OK) return rc;
    LIO_HIP_TRY(hipMemcpyAsync(s->h_result->contri, s->d_result->contri, sizeof(double) * 6, hipMemcpyDeviceToHost, s->stream));
    LIO_HIP_TRY(hipStreamSynchronize(s->stream));
    for (int i = 0; i < 3; i++) { out->contri[i] = s->h_result->contri[i]; out->strong[i] = s->h_result->strong[i]; }
    return LIO_OK;
}

int lio_scan_set_degeneracy_mode(lio_scan* s, int mode) {
    if (!s || mode < 0 || mode > 2) return LIO_E_INVALID;
    s->force_degeneracy = mode;
    return LIO_OK;
}

int lio_p2plane_degeneracy(lio_scan* s, const double V[9], double contri[3], double strong[3]) {
    if (!s || !V || !contri || !strong) return LIO_E_INVALID;
    hipSetDevice(s->device);
    LIO_HIP_TRY(hipMemcpyAsync(s->d_result->eigvec, V, sizeof(double) * 9, hipMemcpyHostToDevice, s->stream));
    LIO_HIP_TRY(hipMemsetAsync(s->d_result->contri, 0, sizeof(double) * 6, s->stream));
    const int rc = p2plane_degeneracy(s);
    if (rc != LIO_OK) return rc;
    LIO_HIP_TRY(hipMemcpyAsync(s->h_result->contri, s->d_result->contri, sizeof(double) * 6, hipMemcpyDeviceToHost, s->stream));
    LIO_HIP_TRY(hipStreamSynchronize(s->stream));
    for (int i = 0; i < 3; i++) { contri[i] = s->h_result->contri[i]; strong[i] = s->h_result->strong[i]; }
    return LIO_OK;
}

int lio_p2plane_rows(lio_scan* s, const double pose_wi[7], const double ext_il[7], double* h_x6, double* h, uint32_t cap_rows) {
    if (!s || !h_x6 || !h) return LIO_E_INVALID;
    const int n = lio_scan_num_ds(s);
    if (n < 0) return n;
    std::vector<uint8_t> sel(n);
    std::vector<float4> nv(n), body(n);
    if (n) {
        LIO_HIP_TRY(hipMemcpyAsync(sel.data(), s->selected, n, hipMemcpyDeviceToHost, s->stream));
        LIO_HIP_TRY(hipMemcpyAsync(nv.data(), s->normvec, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost, s->stream));
        LIO_HIP_TRY(hipMemcpyAsync(body.data(), s->ds_body, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost, s->stream));
        LIO_HIP_TRY(hipStreamSynchronize(s->stream));
    }
    const double* qw = pose_wi + 3;
    const double* ql = ext_il + 3;
    auto qrot = [](const double q[4], const double v[3], double o[3]) {
        double u[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
        u[0] += u[0]; u[1] += u[1]; u[2] += u[2];
        const double c[3] = {q[1] * u[2] - q[2] * u[1], q[2] * u[0] - q[0] * u[2], q[0] * u[1] - q[1] * u[0]};
        for (int i = 0; i < 3; i++) o[i] = (v[i] + q[3] * u[i]) + c[i];
    };
    uint32_t r = 0;
    for (int i = 0; i < n; i++) {
        if (!sel[i]) continue;
        if (r >= cap_rows) return LIO_E_CAPACITY;
        const double pb[3] = {body[i].x, body[i].y, body[i].z};
        double pi[3];
        qrot(ql, pb, pi);
        for (int k = 0; k < 3; k++) pi[k] += ext_il[k];
        const double nn[3] = {nv[i].x, nv[i].y, nv[i].z};
        const double qc[4] = {-qw[0], -qw[1], -qw[2], qw[3]};
        double C[3];
        qrot(qc, nn, C);
        double* row = h_x6 + (size_t)r * 6;
        row[0] = nn[0]; row[1] = nn[1]; row[2] = nn[2];
        row[3] = pi[1] * C[2] - pi[2] * C[1];
        row[4] = pi[2] * C[0] - pi[0] * C[2];
        row[5] = pi[0] * C[1] - pi[1] * C[0];
        h[r] = -(double)nv[i].w;
        r++;
    }
    return (int)r;
}

static int incremental_common(lio_map* m, lio_scan* s, const double pose_wi[7], const double ext_il[7], float map_leaf, int ekf_inited,
                              int seed_all, double travel) {
    if (!m || !s || !pose_wi || !ext_il) return LIO_E_INVALID;
    if (m->device != s->device) { set_error("map and scan live on different devices"); return LIO_E_INVALID; }
    hipSetDevice(s->device);
    const PoseArgs pose = make_pose(pose_wi, ext_il);
    int rc = map_settle(m);
    if (rc != LIO_OK) return rc;
    rc = incremental_classify(m, s, pose, map_leaf, ekf_inited, seed_all);
    if (rc != LIO_OK) return rc;
    const uint32_t bound = s->have_ds > 0 ? (uint32_t)s->have_ds : (s->n_raw && s->n_raw < s->max_ds ? s->n_raw : s->max_ds);
    rc = map_insert_dev(m, s->stream, m->stage, bound, &m->dev->n_add, travel);
    if (rc != LIO_OK) return rc;
    rc = map_check(m, s->stream);
    if (rc != LIO_OK) return rc;
    return (int)m->host_dev->n_add;
}

// The outcome of an insert that was enqueued and not waited for.  A failed wait leaves the insert pending (the next caller tries again); an
// overflow found here belongs to the scan that enqueued the insert, not to the caller: it is handed to exactly one caller -- whoever looks
// first: the next process_scan (before it launches anything), lio_engine_timings or lio_engine_flush -- with a message that says so.
int map_settle(lio_map* m) {
    if (!m || !m->insert_pending) return LIO_OK;
    if (hipEventSynchronize(m->ev_inserted) != hipSuccess) { set_error("map_incremental (deferred): %s", hipGetErrorString(hipGetLastError())); return LIO_E_DEVICE; }
    m->insert_pending = false;
    m->settled_n_add = m->host_dev->n_add;
    if (m->host_dev->err) {
        set_error("map capacity exceeded by the map_incremental of the PREVIOUS scan, whose call had already returned (err bits 0x%x: 1 table full, 2 point pool "
                  "full, 4 more than max_voxels voxels, 8 LRU log overrun)", m->host_dev->err);
        return LIO_E_CAPACITY;
    }
    return LIO_OK;
}

// the same without waiting: LIO_OK while the insert is still running
int map_settle_if_done(lio_map* m) {
    if (!m || !m->insert_pending) return LIO_OK;
    if (hipEventQuery(m->ev_inserted) != hipSuccess) { (void)hipGetLastError(); return LIO_OK; }
    return map_settle(m);
}

int map_incremental_async(lio_map* m, lio_scan* s, const double pose_wi[7], const double ext_il[7], float map_leaf, int ekf_inited, double travel) {
    if (!m || !s || !pose_wi || !ext_il) return LIO_E_INVALID;
    if (m->device != s->device) { set_error("map and scan live on different devices"); return LIO_E_INVALID; }
    hipSetDevice(s->device);
    int rc = map_settle(m);
    if (rc != LIO_OK) return rc;
    if (!m->ev_classified) {
        LIO_HIP_TRY(hipEventCreateWithFlags(&m->ev_classified, hipEventDisableTiming));
        LIO_HIP_TRY(hipEventCreateWithFlags(&m->ev_inserted, hipEventDisableTiming));
    }
    const PoseArgs pose = make_pose(pose_wi, ext_il);
    rc = incremental_classify(m, s, pose, map_leaf, ekf_inited, 0);
    if (rc != LIO_OK) return rc;
    const uint32_t bound = s->have_ds > 0 ? (uint32_t)s->have_ds : (s->n_raw && s->n_raw < s->max_ds ? s->n_raw : s->max_ds);
    LIO_HIP_TRY(hipEventRecord(m->ev_classified, s->stream));
    LIO_HIP_TRY(hipStreamWaitEvent(m->stream, m->ev_classified, 0));
    rc = map_insert_dev(m, m->stream, m->stage, bound, &m->dev->n_add, travel);
    if (rc != LIO_OK) return rc;
    LIO_HIP_TRY(hipMemcpyAsync(m->host_dev, m->dev, sizeof(MapDev), hipMemcpyDeviceToHost, m->stream));
    LIO_HIP_TRY(hipEventRecord(m->ev_inserted, m->stream));
    m->insert_pending = true;
    return LIO_OK;
}

int lio_map_incremental(lio_map* m, lio_scan* s, const double pose_wi[7], const double ext_il[7], float map_leaf, int ekf_inited, double travel) {
    return incremental_common(m, s, pose_wi, ext_il, map_leaf, ekf_inited, 0, travel);
}

int lio_map_seed(lio_map* m, lio_scan* s, const double pose_wi[7], const double ext_il[7], double travel) {
    return incremental_common(m, s, pose_wi, ext_il, 0.5f, 0, 1, travel);
}

}  // extern "C"
