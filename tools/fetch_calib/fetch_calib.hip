// tools/fetch_calib/fetch_calib.hip -- what rocprofv3's FETCH_SIZE reports for the kNN sweep's access SHAPE, on a byte count known in advance.
// (a measurement tool, like tools/valu_peak: not product, not oracle)
//
// /opt/skills/guides/MI355X_MICROARCH.md calibrates FETCH_SIZE on gfx950 for wide coalesced streams only (it reports half of their bytes) and says:
// calibrate other shapes yourself.  The kNN kernel's candidate loads are 16 lanes x 16 B = one 256-byte run per (query, voxel batch) at a 16-byte
// aligned, otherwise arbitrary address; four such runs per wave instruction, the four anywhere in a 640 MB pool.  Three kernels, each reading
// every byte of its share of an 8 GiB buffer exactly once (nothing can hit in L2 or the 256 MiB Infinity Cache):
//   calib_stream            64 lanes x 16 B consecutive (the guide's case: expected FETCH_SIZE = bytes / 2)
//   calib_gather16_aligned  16-lane groups, each a 256-byte run at a 256-byte aligned pseudo-random place
//   calib_gather16          ... at a 16-byte aligned pseudo-random place (the kNN shape): a run touches 2 or 3 128-byte lines, 4 or 5 64-byte sectors
// The host prints the bytes requested and the bytes of the distinct 64-byte sectors / 128-byte lines the runs cover; tools/fetch_calib/run.sh divides
// the counter by them.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) calib_stream(const float4* __restrict__ buf, uint64_t n_vec, float* __restrict__ sink) {
    float acc = 0.f;
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n_vec; i += (uint64_t)gridDim.x * 256ull) {
        const float4 v = buf[i];
        acc += v.x + v.w;
    }
    if (acc == 12345.678f) sink[0] = acc;
}

// run r (of n_runs) starts at byte  perm(r) * 256 + jitter(r) * 16  with perm a bijection of [0, n_runs) and jitter in [0, 16) (0 when aligned)
__device__ __host__ inline uint64_t run_start(uint64_t r, uint64_t n_runs, int aligned) {
    const uint64_t p = (r * 0x9E3779B97F4A7C15ull) % n_runs;  // odd multiplier, n_runs a power of two: a bijection
    const uint64_t j = aligned ? 0ull : ((r * 2654435761ull) >> 7) & 15ull;
    return p * 256ull + j * 16ull;
}
template <int ALIGNED>
__global__ void __launch_bounds__(256) calib_gather16(const char* __restrict__ buf, uint64_t n_runs, uint64_t runs_read, float* __restrict__ sink) {
    const int gl = threadIdx.x & 15;
    float acc = 0.f;
    // four runs per wave instruction (sixteen lanes each), four instructions in flight per lane: the sweep's shape (kU = 4)
    for (uint64_t g = (blockIdx.x * 256ull + threadIdx.x) >> 4; g * 4 + 3 < runs_read; g += ((uint64_t)gridDim.x * 256ull) >> 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const float4*>(buf + run_start(g * 4 + u, n_runs, ALIGNED) + (uint64_t)gl * 16ull);
#pragma unroll
        for (int u = 0; u < 4; u++) acc += v[u].x + v[u].w;
    }
    if (acc == 12345.678f) sink[0] = acc;
}

int main(int argc, char** argv) {
    const uint64_t bytes = 8ull << 30;            // the buffer (one 256-byte slot per run, plus a tail for the jitter)
    const uint64_t n_runs = bytes / 256;          // 2^25 slots
    const uint64_t runs_read = n_runs / 4;        // 2 GiB requested per gather kernel: 8 M runs
    char* buf = nullptr;
    float* sink = nullptr;
    CK(hipMalloc(reinterpret_cast<void**>(&buf), bytes + 4096));
    CK(hipMalloc(reinterpret_cast<void**>(&sink), 64));
    CK(hipMemset(buf, 0, bytes + 4096));
    CK(hipDeviceSynchronize());
    const int reps = argc > 1 ? atoi(argv[1]) : 3;
    for (int r = 0; r < reps; r++) {
        hipLaunchKernelGGL(calib_stream, 8192, 256, 0, 0, reinterpret_cast<const float4*>(buf), (2ull << 30) / 16, sink);
        hipLaunchKernelGGL(calib_gather16<1>, 8192, 256, 0, 0, buf, n_runs, runs_read, sink);
        hipLaunchKernelGGL(calib_gather16<0>, 8192, 256, 0, 0, buf, n_runs, runs_read, sink);
        CK(hipDeviceSynchronize());
    }
    // what the runs cover: distinct 64-byte sectors and 128-byte lines (a slot holds one run and the jitter never leaves slot + 240 bytes, so runs
    // of different slots share at most their first / last line with a neighbour slot's run -- counted exactly below with a bitmap)
    for (int aligned = 1; aligned >= 0; aligned--) {
        const uint64_t n64 = (bytes + 4096) / 64, n128 = (bytes + 4096) / 128;
        uint8_t* b64 = static_cast<uint8_t*>(calloc(n64 / 8 + 1, 1));
        uint8_t* b128 = static_cast<uint8_t*>(calloc(n128 / 8 + 1, 1));
        uint64_t c64 = 0, c128 = 0;
        for (uint64_t r = 0; r + 3 < runs_read; r++) {
            const uint64_t a = run_start(r, n_runs, aligned);
            for (uint64_t s = a / 64; s <= (a + 255) / 64; s++)
                if (!(b64[s >> 3] & (1u << (s & 7)))) { b64[s >> 3] |= (uint8_t)(1u << (s & 7)); c64++; }
            for (uint64_t s = a / 128; s <= (a + 255) / 128; s++)
                if (!(b128[s >> 3] & (1u << (s & 7)))) { b128[s >> 3] |= (uint8_t)(1u << (s & 7)); c128++; }
        }
        printf("{\"kernel\": \"calib_gather16<%d>\", \"requested_bytes\": %llu, \"distinct_64B_sector_bytes\": %llu, \"distinct_128B_line_bytes\": %llu}\n", aligned,
               (unsigned long long)((runs_read / 4) * 4 * 256), (unsigned long long)(c64 * 64), (unsigned long long)(c128 * 128));
        free(b64);
        free(b128);
    }
    printf("{\"kernel\": \"calib_stream\", \"requested_bytes\": %llu, \"distinct_64B_sector_bytes\": %llu, \"distinct_128B_line_bytes\": %llu}\n",
           (unsigned long long)(2ull << 30), (unsigned long long)(2ull << 30), (unsigned long long)(2ull << 30));
    return 0;
}
