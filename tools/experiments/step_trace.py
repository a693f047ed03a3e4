"""Phase timing of the device filter pass (step_batch): needs a diagnostic build  make -C lidar-slam-detection_amd/csrc EXTRA=-DLIO_STEP_TRACE
(cycle stamps of workgroup 0 / lane 0 accumulated in a device array).  Prints cycles per call for every stamped phase."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "lidar-slam-detection_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes  # noqa: E402
from lsd_amd import capi, lio, synth  # noqa: E402

NAMES = {1: "copy-in + fold partials", 20: "  wave 0: measure_head", 21: "  wave 0: measure_tail", 2: "wave 0: wait for wave 1 (barrier)", 3: "degeneracy sums / barrier", 4: "measure_tail", 5: "ek_step total (or nothing)", 6: "write back",
         7: "publish", 11: "  wave 1: boxminus", 12: "  wave 1: jacobians + copy P", 13: "  wave 1: P <- J P J^T (6 phases)", 22: "  wave 1: G", 14: "G, M6", 15: "inverse6",
         16: "Pi6, Kx/Kh, dx_out", 17: "boxplus", 18: "log + convergence", 19: "final covariance"}


def main():
    sc = synth.Scene(half=100.0, n_boxes=40, seed=1)
    mp = sc.sample_surface(2_000_000, seed=2, sigma=0.01)
    m = lio.Map(resolution=0.5, stencil=19, max_points=2_000_000, max_voxels=1_000_000)
    m.add(mp)
    jobs_src = [scenes.config_scan(sc, 1000 + k, fov_deg=(-24.8, 2.0), max_range=150.0) for k in range(4)]
    d = [scenes.to_device(j["raw"]) for j in jobs_src]
    P0 = lio.init_cov()
    jobs = [dict(dptr=d[i % 4], n=len(jobs_src[i % 4]["raw"]), t=1.0 + 0.1 * i, state=jobs_src[i % 4]["guess"], cov=P0) for i in range(64)]
    b = lio.Batch(m, n_slots=16, n_groups=1, max_raw=1 << 17, max_ds=100000)
    b.process(jobs)
    out = (C.c_ulonglong * 64)()
    capi.lib().lio_debug_step_trace(out, 1)
    b.process(jobs)
    capi.lib().lio_debug_step_trace(out, 1)
    calls = out[0]
    print("step_batch calls stamped (workgroup 0):", calls)
    for k in sorted(NAMES):
        print(f"  [{k:2d}] {NAMES[k]:34s} {out[k] / max(calls, 1):10.0f} cycles / call")


if __name__ == "__main__":
    main()
