#!/usr/bin/env python
"""Runs ON THE GPU BOX: records what the reference's own matcher code -- fast_gicp::cuda::NDTCudaCore kernels and the fast_gicp::NDTCuda
registration object, built for gfx950 into oracle/_ref/libref_ndt_cuda.so -- computes for the fixture of tests/test_ndt_gpu.py::_world, so
that the CPU oracle (oracle/ndt_oracle.cpp) can be held against the reference's numbers where there is no GPU:
    gpurun -- 'python tools/make_golden_ndt_gpu.py gpurun_out/ndt_ref_cuda.npz'   then copy the file to tests/golden/."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "lidar-slam-detection_amd", "python"), os.path.join(ROOT, "tests")]
import oracle  # noqa: E402
import ref_ndt_cuda as ref  # noqa: E402
from test_ndt_gpu import _world  # noqa: E402


def main():
    out_path = sys.argv[1]
    mp, raw, T_true, T_guess = _world()
    ds = oracle.voxel_downsample(raw, 0.5)
    out = dict(T_true=T_true, T_guess=T_guess, n_ds=len(ds), n_map=len(mp))
    rng = np.random.default_rng(3)
    guesses = [T_guess]
    from lsd_amd import synth

    for k in range(3):
        G = T_true.copy()
        G[:3, 3] = T_true[:3, 3] + rng.uniform(-0.4, 0.4, 3)
        G[:3, :3] = T_true[:3, :3] @ synth.quat_to_R(synth.quat_from_rotvec(rng.uniform(-0.03, 0.03, 3)))
        guesses.append(G)
    out["guesses"] = np.stack(guesses)
    for m in (1, 7, 27):
        core = ref.NdtCudaCore(1.0, m)
        core.set_target(mp)
        core.set_source(ds)
        co, nn, me, cv = core.voxels()
        out[f"m{m}_num_voxels"] = core.num_voxels
        if m == 7:
            pick = np.random.default_rng(1).choice(len(co), 400, replace=False)
            out["vox_coord"], out["vox_n"], out["vox_mean"], out["vox_cov"] = co[pick], nn[pick], me[pick], cv[pick]
        for name, T in (("guess", T_guess), ("true", T_true)):
            runs = [core.linearize(T) for _ in range(3)]  # its own run-to-run spread is part of the record
            out[f"m{m}_{name}_pairs"] = np.array([r["n_corr"] for r in runs])
            out[f"m{m}_{name}_err"] = np.array([r["err"] for r in runs])
            out[f"m{m}_{name}_H"] = np.stack([r["H"] for r in runs])
            out[f"m{m}_{name}_b"] = np.stack([r["b"] for r in runs])
        core.close()
        reg = ref.NdtCudaRegistration(1.0, m)
        reg.set_target(mp)
        reg.set_source(ds)
        res = [reg.align(G) for G in guesses]
        out[f"m{m}_align_T"] = np.stack([r[0] for r in res])
        out[f"m{m}_align_conv"] = np.array([r[1] for r in res])
        out[f"m{m}_align_iters"] = np.array([r[2] for r in res])
        reg.close()
    np.savez_compressed(out_path, **out, source="fast_gicp::cuda::NDTCudaCore + fast_gicp::NDTCuda (slam/thirdparty/fast_gicp) compiled for gfx950 "
                        "(oracle/ref_ndt_cuda.hip), run on an MI355X; inputs = tests/test_ndt_gpu.py::_world, oracle.voxel_downsample(raw, 0.5)")
    print("wrote", out_path, os.path.getsize(out_path))


if __name__ == "__main__":
    main()
