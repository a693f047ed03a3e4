"""INTEGRATION.md's Option 0 binding -- the replacement of the eight FastLIO entry points a maintainer of the reference would paste into
slam/mapping/fastlio/src/fastlio.cpp -- is extracted from the document and compiled (syntax + types, no link) against include/lio_hip.h and
the reference's type names (PointCloudAttrPtr, ImuType, RTKType as oracle/ref_shims/mapping_types.h declares them from
slam/common/mapping_types.h; the reference's own header needs OpenCV).  Keeps the document honest.  CPU only."""
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EIGEN = "/root/reference/slam/thirdparty/fast_gicp/thirdparty/Eigen"

PREAMBLE = r'''
#include <string>
#include <vector>
#include <Eigen/Geometry>
#include "mapping_types.h"   // oracle/ref_shims: the reference's plain data types
#include "slam_utils.h"      // oracle/ref_shims: getTransformFromRPYT
#include "Logger.h"          // oracle/ref_shims: LOG_ERROR
static Eigen::Matrix3d Lidar_R_wrt_IMU = Eigen::Matrix3d::Identity();   // laserMapping.cpp:139
'''


def test_option0_binding_compiles():
    import pytest

    if not os.path.isdir(EIGEN):
        pytest.skip("the reference's vendored Eigen is not mounted")
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = doc[doc.index("## 2. Option 0"):doc.index("## 2a.")]
    blocks = re.findall(r"```cpp\n(.*?)```", sec, re.S)
    assert len(blocks) == 1 and "lio_fastlio_main" in blocks[0] and "lio_fastlio_pcl_enqueue" in blocks[0]
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "option0.cpp")
        open(src, "w").write(PREAMBLE + blocks[0])
        cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror=return-type", "-I" + os.path.join(ROOT, "include"),
               "-I" + os.path.join(ROOT, "oracle", "ref_shims"), "-I" + EIGEN, src]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
