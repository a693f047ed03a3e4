// Measured VALU issue rate of the chip: wave-instructions per second with every SIMD holding `waves_per_simd` waves that do nothing
// but independent 32-bit VALU work.  bench.py's roofline.frac_valu divides the kNN kernel's SQ_INSTS_VALU per launch by THIS rate
// (the line assumed 4 cycles per wave64 instruction, the micro-architecture guide says 2 -- measure it).
// A measurement tool, not part of the product library:  hipcc --offload-arch=gfx950 -O3 -shared -fPIC valu_peak.hip -o libvalu_peak.so
//
// mix 0: v_add_u32 only            (the plainest full-rate op)
// mix 1: v_add_u32 / v_min_u32 / v_add_u32 / v_add_u32_dpp row_ror:1   (the kNN merge's diet: integer compares + DPP row rotations)
// mix 2: v_fma_f32 only            (the guide's 2-cycle row)
// mix 3: v_add_f64 only            (the transforms' type)
// mix 4..16: ONE instruction form each (valu_one below): add_dpp row_ror, min, min_dpp, cndmask, mov_dpp quad_perm, med3, mov, cmp, mul_f32,
//            add_dpp row_shr / quad_perm / row_bcast, add3 -- what the kNN merge network's instructions cost one by one
// Eight independent register chains per wave, 256 instructions per loop body (the loop's scalar bookkeeping is < 2 % of the issue slots).
#include <hip/hip_runtime.h>
#include <stdint.h>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP32(x) REP16(x) REP16(x)

// one instruction form repeated over eight independent chains: T(d, o) is the text with d = the chain's register, o = another chain's, %8 = a constant
#define CHAINS8(OPA, OPB, OPC, OPD, OPE, OPF, OPG, OPH) OPA "\n" OPB "\n" OPC "\n" OPD "\n" OPE "\n" OPF "\n" OPG "\n" OPH "\n"
template <int KIND>
__global__ __launch_bounds__(256) void valu_one(uint32_t* sink, uint32_t iters, uint32_t seed) {
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
    const uint32_t k = seed | 1u;
#define BODY(T) REP32(asm volatile(CHAINS8(T("%0", "%4"), T("%1", "%5"), T("%2", "%6"), T("%3", "%7"), T("%4", "%0"), T("%5", "%1"), T("%6", "%2"), T("%7", "%3")) \
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k) : "vcc", "s20", "s21");)
#define T4(d, o) "v_add_u32_dpp " d ", " d ", " o " row_ror:1 row_mask:0xf bank_mask:0xf"
#define T5(d, o) "v_min_u32 " d ", " d ", " o
#define T6(d, o) "v_min_u32_dpp " d ", " d ", " o " row_ror:1 row_mask:0xf bank_mask:0xf"
#define T7(d, o) "v_cndmask_b32 " d ", " d ", " o ", vcc"
#define T8(d, o) "v_mov_b32_dpp " d ", " o " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define T9(d, o) "v_med3_u32 " d ", " d ", " o ", %8"
#define T10(d, o) "v_mov_b32 " d ", " o
#define T11(d, o) "v_cmp_lt_u32 vcc, " d ", " o
#define T12(d, o) "v_mul_f32 " d ", " d ", " o
#define T13(d, o) "v_add_u32_dpp " d ", " d ", " o " row_shr:1 row_mask:0xf bank_mask:0xf"
#define T14(d, o) "v_add_u32_dpp " d ", " d ", " o " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define T15(d, o) "v_add_u32_dpp " d ", " d ", " o " row_bcast:15 row_mask:0xa bank_mask:0xf"
#define T16(d, o) "v_add3_u32 " d ", " d ", " o ", %8"
#define T17(d, o) "v_sub_f32 " d ", " d ", " o
#define T18(d, o) "v_add_f32 " d ", " d ", " o
#define T19(d, o) "v_max_f32 " d ", " d ", " o
#define T20(d, o) "v_min_f32 " d ", " d ", " o
#define T21(d, o) "v_and_b32 " d ", " d ", " o
#define T22(d, o) "v_xor_b32 " d ", " d ", " o
#define T23(d, o) "v_lshlrev_b32 " d ", 1, " o
#define T24(d, o) "v_cndmask_b32 " d ", " d ", " o ", s[20:21]"
#define T25(d, o) "v_cmp_lt_u32 s[20:21], " d ", " o
#define T26(d, o) "v_sub_u32 " d ", " d ", " o
#define T27(d, o) "v_max_u32 " d ", " d ", " o
#define T28(d, o) "v_bfe_u32 " d ", " o ", 3, 7"
#define T29(d, o) "v_lshl_add_u32 " d ", " d ", 1, " o
#define T30(d, o) "v_mad_u32_u24 " d ", " d ", " o ", %8"
#define T31(d, o) "v_fma_f32 " d ", " d ", " o ", %8"
#define T32(d, o) "v_fmac_f32 " d ", " o ", %8"
#define T33(d, o) "v_min_i32 " d ", " d ", " o
#define T34(d, o) "v_or_b32 " d ", " d ", " o
#define T35(d, o) "v_cmp_lt_f32 vcc, " d ", " o
#define T36(d, o) "v_med3_f32 " d ", " d ", " o ", %8"
#define T37(d, o) "v_bcnt_u32_b32 " d ", " o ", " d
    for (uint32_t i = 0; i < iters; i++) {
        if (KIND == 4) { BODY(T4) }
        if (KIND == 5) { BODY(T5) }
        if (KIND == 6) { BODY(T6) }
        if (KIND == 7) { BODY(T7) }
        if (KIND == 8) { BODY(T8) }
        if (KIND == 9) { BODY(T9) }
        if (KIND == 10) { BODY(T10) }
        if (KIND == 11) { BODY(T11) }
        if (KIND == 12) { BODY(T12) }
        if (KIND == 13) { BODY(T13) }
        if (KIND == 14) { BODY(T14) }
        if (KIND == 15) { BODY(T15) }
        if (KIND == 16) { BODY(T16) }
        if (KIND == 17) { BODY(T17) }
        if (KIND == 18) { BODY(T18) }
        if (KIND == 19) { BODY(T19) }
        if (KIND == 20) { BODY(T20) }
        if (KIND == 21) { BODY(T21) }
        if (KIND == 22) { BODY(T22) }
        if (KIND == 23) { BODY(T23) }
        if (KIND == 24) { BODY(T24) }
        if (KIND == 25) { BODY(T25) }
        if (KIND == 26) { BODY(T26) }
        if (KIND == 27) { BODY(T27) }
        if (KIND == 28) { BODY(T28) }
        if (KIND == 29) { BODY(T29) }
        if (KIND == 30) { BODY(T30) }
        if (KIND == 31) { BODY(T31) }
        if (KIND == 32) { BODY(T32) }
        if (KIND == 33) { BODY(T33) }
        if (KIND == 34) { BODY(T34) }
        if (KIND == 35) { BODY(T35) }
        if (KIND == 36) { BODY(T36) }
        if (KIND == 37) { BODY(T37) }
    }
    a0 += a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (a0 == 0x12345678u) sink[0] = a0;
}

template <int MIX>
__global__ __launch_bounds__(256) void valu_spin(uint32_t* sink, uint32_t iters, uint32_t seed) {
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
    const uint32_t k = seed | 1u;
    if (MIX == 3) {
        double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7, dk = 1.0 + 1e-9 * k;
        for (uint32_t i = 0; i < iters; i++) {
            REP32(asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                               "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dk));)
        }
        a0 = (uint32_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    } else if (MIX == 2) {
        float f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = a4, f5 = a5, f6 = a6, f7 = a7, fk = 1.0f + 1e-7f * k;
        for (uint32_t i = 0; i < iters; i++) {
            REP32(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                               "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                               : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fk));)
        }
        a0 = (uint32_t)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
    } else if (MIX == 1) {
        for (uint32_t i = 0; i < iters; i++) {
            REP32(asm volatile("v_add_u32 %0, %0, %8\n v_min_u32 %1, %1, %4\n v_add_u32 %2, %2, %8\n v_add_u32_dpp %3, %3, %7 row_ror:1 row_mask:0xf bank_mask:0xf\n"
                               "v_add_u32 %4, %4, %8\n v_min_u32 %5, %5, %0\n v_add_u32 %6, %6, %8\n v_add_u32_dpp %7, %7, %3 row_ror:1 row_mask:0xf bank_mask:0xf\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));)
        }
        a0 += a1 + a2 + a3 + a4 + a5 + a6 + a7;
    } else {
        for (uint32_t i = 0; i < iters; i++) {
            REP32(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                               "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));)
        }
        a0 += a1 + a2 + a3 + a4 + a5 + a6 + a7;
    }
    if (a0 == 0x12345678u) sink[0] = a0;  // (never true in practice: keeps the chains alive)
}

extern "C" {

// wave-instructions per second of the whole chip for instruction mix `mix` with `waves_per_simd` waves resident per SIMD; < 0 on error.
// cus_out / clock_mhz_out (may be NULL): the device's CU count and its reported peak engine clock.
double valu_peak_wave_insts_per_s(int device, int waves_per_simd, int mix, int* cus_out, int* clock_mhz_out) {
    if (hipSetDevice(device) != hipSuccess) return -1.0;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) return -1.0;
    const int cus = p.multiProcessorCount;
    if (cus_out) *cus_out = cus;
    if (clock_mhz_out) *clock_mhz_out = p.clockRate / 1000;
    if (waves_per_simd < 1) waves_per_simd = 1;
    if (waves_per_simd > 8) waves_per_simd = 8;
    // a 256-lane workgroup = 4 waves = one wave per SIMD of a CU: `waves_per_simd` workgroups per CU
    const int blocks = cus * waves_per_simd;
    uint32_t* sink = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&sink), 64) != hipSuccess) return -1.0;
    hipStream_t st;
    (void)hipStreamCreate(&st);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    auto launch = [&](uint32_t iters) {
        switch (mix) {
            case 1: valu_spin<1><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 2: valu_spin<2><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 3: valu_spin<3><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 4: valu_one<4><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 5: valu_one<5><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 6: valu_one<6><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 7: valu_one<7><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 8: valu_one<8><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 9: valu_one<9><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 10: valu_one<10><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 11: valu_one<11><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 12: valu_one<12><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 13: valu_one<13><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 14: valu_one<14><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 15: valu_one<15><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 16: valu_one<16><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 17: valu_one<17><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 18: valu_one<18><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 19: valu_one<19><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 20: valu_one<20><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 21: valu_one<21><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 22: valu_one<22><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 23: valu_one<23><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 24: valu_one<24><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 25: valu_one<25><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 26: valu_one<26><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 27: valu_one<27><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 28: valu_one<28><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 29: valu_one<29><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 30: valu_one<30><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 31: valu_one<31><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 32: valu_one<32><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 33: valu_one<33><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 34: valu_one<34><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 35: valu_one<35><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 36: valu_one<36><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            case 37: valu_one<37><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
            default: valu_spin<0><<<blocks, 256, 0, st>>>(sink, iters, 12345u); break;
        }
    };
    launch(2000);  // warm: clocks up, code resident
    (void)hipStreamSynchronize(st);
    const uint32_t iters = 20000;  // 20000 x 256 = 5.1e6 instructions per wave: a few ms
    (void)hipEventRecord(e0, st);
    launch(iters);
    (void)hipEventRecord(e1, st);
    (void)hipStreamSynchronize(st);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const bool ok = hipGetLastError() == hipSuccess && ms > 0.f;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipStreamDestroy(st);
    (void)hipFree(sink);
    if (!ok) return -1.0;
    const double waves = (double)blocks * 4.0;
    return waves * (double)iters * 256.0 / (ms * 1e-3);
}

}
