"""Guards the round-4 finding about loads in flight (tools/isa_load_chains.py): the kernels below were written with groups of
independent loads, and the compiler had turned every one of those groups into load / s_waitcnt vmcnt(0) / next load wherever the loads sat inside
`if (lane has work)`.  The sources now issue them unconditionally at clamped addresses and pin them (lio_common.h: pin_loaded); this test compiles
the device code of the translation units for gfx950 (hipcc cross-compiles without a GPU, ~10 s per file) and checks in the assembly that the groups
are still groups -- an edit or a compiler update that serialises them again shows up here, not as a silent 15 % on the GPU box."""
import importlib.util
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lidar-slam-detection_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

spec = importlib.util.spec_from_file_location("isa_load_chains", os.path.join(ROOT, "tools", "isa_load_chains.py"))
isa = importlib.util.module_from_spec(spec)
spec.loader.exec_module(isa)


def device_asm(tmp_path_factory, name):
    out = tmp_path_factory.mktemp("isa") / (name + ".s")
    cmd = [HIPCC, "--offload-arch=gfx950", "--cuda-device-only", "-S", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-w",
           os.path.join(CSRC, name + ".hip"), "-o", str(out)]
    subprocess.check_call(cmd)
    return str(out)


def groups_of(path, kernel_substring):
    found = {}
    for name, body in isa.kernels(path):
        if kernel_substring in name:
            loads, waits0, serial, runs = isa.analyse(body)
            found[name] = (loads, runs)
    assert found, f"no kernel matching {kernel_substring} in {path}"
    return found


pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")


@pytest.fixture(scope="module")
def knn_s(tmp_path_factory):
    return device_asm(tmp_path_factory, "knn")


@pytest.fixture(scope="module")
def vg_s(tmp_path_factory):
    return device_asm(tmp_path_factory, "voxelgrid")


@pytest.fixture(scope="module")
def p2p_s(tmp_path_factory):
    return device_asm(tmp_path_factory, "p2plane")


@pytest.fixture(scope="module")
def ndt_s(tmp_path_factory):
    return device_asm(tmp_path_factory, "ndt")


def test_knn_sweep_and_probe_loads_are_in_flight_together(knn_s):
    # the batched and the single-scan form of the 19-cell stencil: the sweep's four candidate loads in one group, the probe's two home slots in one
    for kernel in ("knn_batch_kernelILi2ELb0E", "knn_kernelILi2ELi0E"):
        for name, (loads, runs) in groups_of(knn_s, kernel).items():
            assert any(r >= 4 for r in runs), (name, runs)            # the sweep step
            assert sum(1 for r in runs if r >= 2) >= 2, (name, runs)  # ... and the home-slot pair (and / or the stencil offsets)


def test_voxelgrid_tile_loads_are_grouped(vg_s):
    (_, (_, runs)), = groups_of(vg_s, "vg_count_heads_kernel").items()
    assert max(runs) >= 16, runs                                      # 8 keys + their 8 predecessors
    for name, (_, runs) in groups_of(vg_s, "vg_heads_").items():
        assert max(runs) >= 12, (name, runs)                          # keys, previous keys, indices of a tile (24 loads: one group, or 15 + 10 in the batched form)
        assert sum(1 for r in runs if r >= 4) >= 2, (name, runs)      # ... and the gathers four at a time


def test_linearize_requests_everything_up_front(p2p_s):
    (_, (_, runs)), = groups_of(p2p_s, "linearize_kernel").items()
    assert max(runs) >= 5, runs                                       # the five neighbours (with the point and its count)


def test_single_alignment_ndt_has_all_seven_records_in_flight(ndt_s):
    for name, (_, runs) in groups_of(ndt_s, "ndt_cost_kernelILb1ELb1ELi7E").items():
        assert max(runs) >= 28, (name, runs)                          # 7 offsets x 4 x 16 bytes
