#!/usr/bin/env python
"""tests/golden/voxelgrid_vgc.npz: what the reference tree's own PCL-derived voxel filter (pclomp::VoxelGridCovariance compiled from where it lies,
oracle/_ref/libref_voxelgrid_cov.so, built by `make -C oracle ref` where /root/reference is mounted) makes of tests/voxelgrid_cases.py -- leaf keys,
point counts, f32 centroids of all four fields, the box -- so that the VoxelGrid pin also holds where the reference tree is absent.
    python tools/make_golden_voxelgrid.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "lidar-slam-detection_amd", "python"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import ref_voxelgrid_cov as rv  # noqa: E402
import voxelgrid_cases as vc  # noqa: E402


def main():
    out = {}
    for name, (cloud, leaf, dense) in vc.cases().items():
        r = rv.leaves(cloud, leaf, dense)
        out[name + "/fired"] = np.array(r is None)
        if r is None:
            continue
        keys, cnt, cen, mb, db = r
        out[name + "/keys"], out[name + "/counts"], out[name + "/centroids"], out[name + "/min_b"], out[name + "/div_b"] = keys, cnt, cen, mb, db
        print(name, len(cloud), "->", len(keys), "leaves, largest", int(cnt.max()))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "voxelgrid_vgc.npz"), **out)


if __name__ == "__main__":
    main()
