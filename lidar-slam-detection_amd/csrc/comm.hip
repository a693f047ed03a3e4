// comm.hip -- multi-GPU joint registration over RCCL, natively in the C ABI (lio_comm_* / lio_allgather_normal_eq / lio_engine_set_joint).
//
// BASELINE.json config 5 (multi-map merge): sub-maps live on different GPUs (one process per GPU); every rank linearises the SAME scan
// against ITS sub-map(s), the per-rank records of 32 doubles -- J^T J upper triangle (21), J^T r (6), sum |r|, N_eff, padding -- are
// all-gathered over xGMI and summed in fixed rank order on the device, and every rank runs the same 23-DoF update on the same numbers:
// bitwise identical states on all ranks, no broadcast.  The payload is 256 bytes per rank and pass: latency-bound, not bandwidth-bound
// (SURVEY.md section 8e).  The reference has no multi-GPU code; what must hold is that the summed normal equations equal the
// single-process ones (tests/test_dist.py).
#include <rccl/rccl.h>

#include <chrono>
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
};

namespace {

// fixed rank order: every rank forms the identical sum
__global__ void sum_ranks_kernel(const double* __restrict__ gathered, int world, double* __restrict__ sum) {
    const int i = threadIdx.x;
    if (i >= 32) return;
    double s = gathered[i];
    for (int r = 1; r < world; r++) s += gathered[(size_t)r * 32 + i];
    sum[i] = s;
}

#define LIO_NCCL_TRY(expr)                                                                 \
    do {                                                                                   \
        ncclResult_t r__ = (expr);                                                         \
        if (r__ != ncclSuccess) {                                                          \
            set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, ncclGetErrorString(r__)); \
            return LIO_E_DEVICE;                                                           \
        }                                                                                  \
    } while (0)

}  // namespace

extern "C" {

int lio_comm_unique_id(uint8_t id[128]) {
    if (!id) return LIO_E_INVALID;
    ncclUniqueId u;
    static_assert(sizeof(u) == 128, "ncclUniqueId is 128 bytes");
    LIO_NCCL_TRY(ncclGetUniqueId(&u));
    memcpy(id, &u, 128);
    return LIO_OK;
}

lio_comm* lio_comm_init(int device, int rank, int world, const uint8_t id[128]) {
    if (rank < 0 || world < 1 || rank >= world || (world > 1 && !id)) { set_error("lio_comm_init: bad argument"); return nullptr; }
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
        const ncclResult_t r = ncclCommInitRank(&c->nccl, world, u, rank);
        if (r != ncclSuccess) { set_error("ncclCommInitRank: %s", ncclGetErrorString(r)); ok = false; }
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
    if (c->nccl) ncclCommDestroy(c->nccl);
    if (c->d_local) hipFree(c->d_local);
    if (c->d_gathered) hipFree(c->d_gathered);
    if (c->d_sum) hipFree(c->d_sum);
    if (c->h_pin) hipHostFree(c->h_pin);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

int lio_comm_rank(const lio_comm* c) { return c ? c->rank : LIO_E_INVALID; }
int lio_comm_world(const lio_comm* c) { return c ? c->world : LIO_E_INVALID; }

int lio_allgather_normal_eq(lio_comm* c, const double* d_local32, double* d_gathered, double* d_sum32, void* stream) {
    if (!c || !d_local32 || !d_gathered) return LIO_E_INVALID;
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : c->stream;
    if (c->world > 1) LIO_NCCL_TRY(ncclAllGather(d_local32, d_gathered, 32, ncclDouble, c->nccl, st));
    else LIO_HIP_TRY(hipMemcpyAsync(d_gathered, d_local32, 32 * sizeof(double), hipMemcpyDeviceToDevice, st));
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

// host records (the host-driven filter loop of a joint registration has its sums on the host): n <= 32 doubles in, the fixed-order sum
// over ranks out
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
