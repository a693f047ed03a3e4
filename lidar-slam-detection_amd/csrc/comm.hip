// comm.hip -- multi-GPU joint registration over RCCL, natively in the C ABI (lio_comm_* / lio_allgather_normal_eq / lio_engine_set_joint /
// the joint mode of lio_batch).
//
// BASELINE.json config 5 (multi-map merge): sub-maps live on different GPUs (one process per GPU); every rank linearises the SAME scan
// against ITS sub-map(s), the per-rank records of 32 doubles -- J^T J upper triangle (21), J^T r (6), sum |r|, N_eff, padding -- are
// all-gathered over xGMI and summed in fixed rank order on the device, and every rank runs the same 23-DoF update on the same numbers:
// bitwise identical states on all ranks, no broadcast.  The payload is 256 bytes per rank, scan and pass: latency-bound, not bandwidth-bound
//  -- which is why the batched engine gathers the records of ALL scans of a round in one collective per pass.
// The reference has no multi-GPU code; what must hold is that the summed normal equations equal the single-process ones (tests/test_dist.py).
//
// librccl is loaded on first use (dlopen), not linked: a single-GPU host without RCCL loads liblio_hip.so and runs everything but a
// communicator of more than one rank.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <mutex>
#include <vector>

#include "eskf.h"
#include "lio_common.h"

using namespace lio;

struct lio_comm {
    ncclComm_t nccl = nullptr;
    int device = 0, rank = 0, world = 1;
    hipStream_t stream = nullptr;
    double* d_local = nullptr;      // 32
    double* d_gathered = nullptr;   // world x 32
    double* d_sum = nullptr;        // 32
    double* h_pin = nullptr;        // pinned 64: [0..32) send, [32..64) receive
    double coll_us = 0;             // host-observed time of the collectives (send upload .. sum download)
    uint64_t n_coll = 0;
    uint64_t n_coll_device = 0;     // collectives enqueued on a caller's stream (the batched engine: no host round trip, not timed)
};

namespace {

struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    char why[256] = "symbols missing";  // dlerror() of the last dlopen attempt, captured once (a second dlerror() call returns NULL)
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.so) break;
            const char* e = dlerror();
            if (e) snprintf(r.why, sizeof(r.why), "%s", e);
        }
        if (!r.so) return;
        snprintf(r.why, sizeof(r.why), "symbols missing");
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.so, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.so, "ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.so, "ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.so, "ncclAllGather"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.so, "ncclGetErrorString"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.GetErrorString;
    });
    return r;
}

bool need_rccl() {
    if (rccl().ok) return true;
    set_error("librccl could not be loaded (%s): a communicator of more than one rank needs RCCL", rccl().why);
    return false;
}

// fixed rank order: every rank forms the identical sum
__global__ void sum_ranks_kernel(const double* __restrict__ gathered, int world, double* __restrict__ sum) {
    const int i = threadIdx.x;
    if (i >= 32) return;
    double s = gathered[i];
    for (int r = 1; r < world; r++) s += gathered[(size_t)r * 32 + i];
    sum[i] = s;
}

#define LIO_NCCL_TRY(expr)                                                                       \
    do {                                                                                         \
        ncclResult_t r__ = (expr);                                                               \
        if (r__ != ncclSuccess) {                                                                \
            set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, rccl().GetErrorString(r__)); \
            return LIO_E_DEVICE;                                                                 \
        }                                                                                        \
    } while (0)

}  // namespace

extern "C" {

int lio_comm_unique_id(uint8_t id[128]) {
    if (!id) return LIO_E_INVALID;
    if (!need_rccl()) return LIO_E_DEVICE;
    ncclUniqueId u;
    static_assert(sizeof(u) == 128, "ncclUniqueId is 128 bytes");
    LIO_NCCL_TRY(rccl().GetUniqueId(&u));
    memcpy(id, &u, 128);
    return LIO_OK;
}

lio_comm* lio_comm_init(int device, int rank, int world, const uint8_t id[128]) {
    if (rank < 0 || world < 1 || rank >= world || (world > 1 && !id)) { set_error("lio_comm_init: bad argument"); return nullptr; }
    if (world > 1 && !need_rccl()) return nullptr;
    if (hipSetDevice(device) != hipSuccess) { set_error("lio_comm_init: no HIP device %d", device); return nullptr; }
    lio_comm* c = new lio_comm();
    c->device = device;
    c->rank = rank;
    c->world = world;
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&c->d_local), 32 * sizeof(double)) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&c->d_gathered), (size_t)world * 32 * sizeof(double)) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&c->d_sum), 32 * sizeof(double)) == hipSuccess &&
              hipHostMalloc(reinterpret_cast<void**>(&c->h_pin), 64 * sizeof(double), hipHostMallocDefault) == hipSuccess;
    if (ok && world > 1) {  // a world of one needs no communicator: the "gather" is a copy
        ncclUniqueId u;
        memcpy(&u, id, 128);
        const ncclResult_t r = rccl().CommInitRank(&c->nccl, world, u, rank);
        if (r != ncclSuccess) { set_error("ncclCommInitRank: %s", rccl().GetErrorString(r)); ok = false; }
    }
    if (!ok) {
        if (!lio_last_error()[0]) set_error("lio_comm_init: device setup failed");
        lio_comm_destroy(c);
        return nullptr;
    }
    return c;
}

void lio_comm_destroy(lio_comm* c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->nccl) rccl().CommDestroy(c->nccl);
    if (c->d_local) hipFree(c->d_local);
    if (c->d_gathered) hipFree(c->d_gathered);
    if (c->d_sum) hipFree(c->d_sum);
    if (c->h_pin) hipHostFree(c->h_pin);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

int lio_comm_rank(const lio_comm* c) { return c ? c->rank : LIO_E_INVALID; }
int lio_comm_world(const lio_comm* c) { return c ? c->world : LIO_E_INVALID; }

int lio_allgather_records(lio_comm* c, const double* d_local, double* d_gathered, uint32_t n_records, void* stream) {
    if (!c || !d_local || !d_gathered || n_records == 0) return LIO_E_INVALID;
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : c->stream;
    const size_t count = (size_t)n_records * 32;
    if (c->world > 1) LIO_NCCL_TRY(rccl().AllGather(d_local, d_gathered, count, ncclDouble, c->nccl, st));
    else if (d_gathered != d_local) LIO_HIP_TRY(hipMemcpyAsync(d_gathered, d_local, count * sizeof(double), hipMemcpyDeviceToDevice, st));
    c->n_coll_device++;
    return LIO_OK;
}

int lio_allgather_normal_eq(lio_comm* c, const double* d_local32, double* d_gathered, double* d_sum32, void* stream) {
    if (!c || !d_local32 || !d_gathered) return LIO_E_INVALID;
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : c->stream;
    const int rc = lio_allgather_records(c, d_local32, d_gathered, 1, st);
    if (rc != LIO_OK) return rc;
    if (d_sum32) {
        hipLaunchKernelGGL(sum_ranks_kernel, 1, 32, 0, st, d_gathered, c->world, d_sum32);
        LIO_HIP_TRY(hipGetLastError());
    }
    return LIO_OK;
}

int lio_comm_stats(lio_comm* c, uint64_t* n_collectives, double* total_us) {
    if (!c) return LIO_E_INVALID;
    if (n_collectives) *n_collectives = c->n_coll;
    if (total_us) *total_us = c->coll_us;
    return LIO_OK;
}

}  // extern "C"

namespace lio {
int lio_allgather_records_internal(::lio_comm* c, const double* d_local, double* d_gathered, uint32_t n_records, hipStream_t st) {
    return lio_allgather_records(c, d_local, d_gathered, n_records, st);
}
}  // namespace lio

// host records (the host-driven filter loop of a joint registration has its sums on the host): n <= 32 doubles in, the fixed-order sum
// over ranks out.  A rank that failed locally still takes part (the caller hands in NaNs: every rank then sees NaN sums and gives the
// registration up in the same pass -- nobody is left waiting in a collective).
int comm_reduce_host(lio_comm* c, double* buf, int n) {
    if (!c || n > 32) return LIO_E_INVALID;
    hipSetDevice(c->device);
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 32; i++) c->h_pin[i] = i < n ? buf[i] : 0.0;
    LIO_HIP_TRY(hipMemcpyAsync(c->d_local, c->h_pin, 32 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const int rc = lio_allgather_normal_eq(c, c->d_local, c->d_gathered, c->d_sum, c->stream);
    if (rc != LIO_OK) return rc;
    LIO_HIP_TRY(hipMemcpyAsync(c->h_pin + 32, c->d_sum, 32 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    LIO_HIP_TRY(hipStreamSynchronize(c->stream));
    for (int i = 0; i < n; i++) buf[i] = c->h_pin[32 + i];
    c->coll_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    c->n_coll++;
    return LIO_OK;
}
