#!/usr/bin/env python
"""Print the chip's measured VALU issue rate (tools/valu_peak/valu_peak.hip) for every instruction mix at 1..8 waves per SIMD, as JSON."""
import ctypes as C
import json
import os

L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvalu_peak.so"))
L.valu_peak_wave_insts_per_s.restype = C.c_double
L.valu_peak_wave_insts_per_s.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
cus, mhz = C.c_int(0), C.c_int(0)
names = {0: "v_add_u32", 1: "add/min/add/add_dpp", 2: "v_fma_f32", 3: "v_add_f64", 4: "v_add_u32_dpp row_ror", 5: "v_min_u32", 6: "v_min_u32_dpp row_ror",
         7: "v_cndmask_b32 vcc", 8: "v_mov_b32_dpp quad_perm", 9: "v_med3_u32", 10: "v_mov_b32", 11: "v_cmp_lt_u32 vcc", 12: "v_mul_f32", 13: "v_add_u32_dpp row_shr",
         14: "v_add_u32_dpp quad_perm", 15: "v_add_u32_dpp row_bcast15", 16: "v_add3_u32", 17: "v_sub_f32", 18: "v_add_f32", 19: "v_max_f32", 20: "v_min_f32",
         21: "v_and_b32", 22: "v_xor_b32", 23: "v_lshlrev_b32", 24: "v_cndmask_b32 sgpr pair", 25: "v_cmp_lt_u32 sgpr pair", 26: "v_sub_u32", 27: "v_max_u32",
         28: "v_bfe_u32", 29: "v_lshl_add_u32", 30: "v_mad_u32_u24", 31: "v_fma_f32 (3 src)", 32: "v_fmac_f32", 33: "v_min_i32", 34: "v_or_b32", 35: "v_cmp_lt_f32 vcc",
         36: "v_med3_f32", 37: "v_bcnt_u32_b32"}
out = {}
for mix in sorted(names):
    row = {}
    for w in ((1, 2, 4, 6, 8) if mix < 4 else (1, 6)):
        r = max(L.valu_peak_wave_insts_per_s(0, w, mix, C.byref(cus), C.byref(mhz)) for _ in range(2))
        row[str(w)] = {"wave_insts_per_s": r, "cycles_per_wave_inst_per_simd_at_reported_clock": cus.value * 4 * mhz.value * 1e6 / r if r > 0 else None}
    out[names[mix]] = row
print(json.dumps({"cus": cus.value, "clock_mhz": mhz.value, "rates": out}, indent=1))
