"""The C-ABI library loads and exports every symbol include/lio_hip.h declares; struct layouts agree between the
header (compiled by gcc) and the ctypes mirror.  No compute calls: runs without a GPU."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lio_hip.h")


def _declared():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lio_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported():
    from lsd_amd import capi

    L = capi.lib()
    names = _declared()
    assert len(names) >= 45
    for n in names:
        assert hasattr(L, n), f"{n} declared in lio_hip.h but not exported by liblio_hip.so"
    assert sorted(capi.SYMBOLS) == names, set(names) ^ set(capi.SYMBOLS)


def test_abi_revision_of_the_library_is_the_headers():
    """revisions only append (include/lio_hip.h: LIO_ABI_VERSION); a library older than the header a caller was built with must be detectable"""
    from lsd_amd import capi

    hdr = int(re.search(r"#define LIO_ABI_VERSION (\d+)", open(HEADER).read()).group(1))
    assert capi.lib().lio_abi_version() == hdr >= 5
    known = int(re.search(r"#define LIO_JOB_FLAGS_KNOWN (\d+)u", open(HEADER).read()).group(1))
    assert known == 1 | 2 | 4  # KEEP_CACHE | IDLE | HOST_RAW: every bit of the mask is a documented flag


def test_struct_layouts_match_header():
    from lsd_amd import capi

    src = """
    #include <stdio.h>
    #include <stddef.h>
    #include "lio_hip.h"
    int main(void) {
        printf("%zu %zu %zu %zu %zu\\n", sizeof(lio_normal_eq), sizeof(lio_pass_log), sizeof(lio_timings), sizeof(lio_kernel_times),
               offsetof(lio_normal_eq, n_eff));
        return 0;
    }"""
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "t")
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = subprocess.check_output([exe]).decode().split()
    sizes = [int(x) for x in out]
    assert sizes[0] == C.sizeof(capi.NormalEq)
    assert sizes[1] == C.sizeof(capi.PassLog)
    assert sizes[2] == C.sizeof(capi.Timings)
    assert sizes[3] == C.sizeof(capi.KernelTimes)
    assert sizes[4] == capi.NormalEq.n_eff.offset


def test_no_cpu_fallback_without_a_device():
    """on a box without a GPU every handle constructor must fail loudly (never a silent CPU path)"""
    from lsd_amd import capi, lio

    if capi.lib().lio_device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(capi.LioError):
        lio.Map()
    with pytest.raises(capi.LioError):
        lio.Scan()
    with pytest.raises(capi.LioError):
        lio.Engine()


def test_product_never_imports_the_oracle():
    """the oracle is test infrastructure: nothing under the package may import, link or call it"""
    pkg = os.path.join(ROOT, "lidar-slam-detection_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".cpp", ".h", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                bad = re.findall(r'#include\s*[<"][^>"]*oracle|^\s*(?:import|from)\s+oracle|liblio_oracle|\borc_[a-z]+\s*\(|libref_harness', txt, re.M)
                assert not bad, (os.path.join(dp, f), bad)


def test_header_is_plain_c():
    """the boundary is a C ABI: the header must compile as C99 on its own (no C++, no HIP, no torch types)"""
    import subprocess
    import tempfile

    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "h.c")
        with open(src, "w") as f:
            f.write('#include "lio_hip.h"\nint main(void) { lio_normal_eq ne; lio_ndt_params p; (void)ne; (void)p; return LIO_OK; }\n')
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), src])


def test_host_only_entry_points_reject_bad_arguments():
    """entry points that never touch the device answer invalid arguments with LIO_E_INVALID (and NULL handles from *_create), no crash:
    the C ABI's error contract (negative codes, nothing thrown) can be exercised without a GPU on these"""
    import ctypes as C

    import numpy as np
    from lsd_amd import capi

    L = capi.lib()
    f32p, f64p = C.POINTER(C.c_float), C.POINTER(C.c_double)
    assert not L.lio_pose_estimator_create(None, 0, None, None, 1.0)
    ext, pos, q = np.eye(4, dtype=np.float32).reshape(-1), np.zeros(3, np.float32), np.array([1, 0, 0, 0], np.float32)
    pe = L.lio_pose_estimator_create(ext.ctypes.data_as(f32p), 0, pos.ctypes.data_as(f32p), q.ctypes.data_as(f32p), 0.0)
    assert pe
    acc = np.zeros(3, np.float32)
    assert L.lio_pose_estimator_predict(None, 1, None, None) == capi.LIO_E_INVALID
    assert L.lio_pose_estimator_predict(pe, 1, acc.ctypes.data_as(f32p), None) == capi.LIO_E_INVALID  # acc without gyro
    assert L.lio_pose_estimator_correct(pe, 1, None) == capi.LIO_E_INVALID
    assert L.lio_pose_estimator_matrix(pe, None) == capi.LIO_E_INVALID
    assert L.lio_pose_estimator_guess(pe, None, None) == capi.LIO_E_INVALID
    assert L.lio_pose_estimator_observe(pe, None, None, 1, None, None, None) == capi.LIO_E_INVALID
    assert L.lio_pose_estimator_get_timed_pose(pe, 5, None, None, None) == capi.LIO_E_INVALID
    assert L.lio_pose_estimator_predict_nostate(pe, 5, None) == capi.LIO_E_INVALID
    assert L.lio_pose_estimator_match(pe, None, None, None, None, None) == capi.LIO_E_INVALID
    obs = np.zeros(7, np.float32)
    assert L.lio_pose_estimator_match_gps_only(pe, None, obs.ctypes.data_as(f32p), None) == 0  # nothing to fuse: the filter's pose, "false"
    assert np.allclose(obs, [0, 0, 0, 1, 0, 0, 0])
    L.lio_pose_estimator_destroy(pe)
    L.lio_pose_estimator_destroy(None)
    # null handles on the device-side API: invalid argument, not a crash
    for name in ("lio_map_destroy", "lio_scan_destroy", "lio_engine_destroy", "lio_ndt_destroy", "lio_localmap_destroy"):
        getattr(L, name)(None)
    assert L.lio_scan_upload(None, None, 0) == capi.LIO_E_INVALID
    assert L.lio_scan_undistort_delta(None, None, 0, None, 0.1) == capi.LIO_E_INVALID
    assert L.lio_engine_process_scan(None, None, 0, 0.0) < 0
    assert L.lio_fastlio_main(None) < 0
    assert L.lio_ndt_align(None, None, None, None, None, None, None) == capi.LIO_E_INVALID
    s26, d23, out = np.zeros(26), np.zeros(23), np.zeros(26)
    s26[6] = s26[10] = 1.0
    s26[23:26] = [0, 0, -9.809]
    L.lio_state_boxplus(s26.ctypes.data_as(f64p), d23.ctypes.data_as(f64p), out.ctypes.data_as(f64p))
    assert np.allclose(out, s26)
