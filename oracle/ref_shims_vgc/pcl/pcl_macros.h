// stand-in for <pcl/pcl_macros.h> (harness ref_voxelgrid_cov.cpp only)
#pragma once
#include <cmath>
#include <cstdio>
#define pcl_isfinite(x) std::isfinite(x)
#define PCL_WARN(...) ((void)0)
#define PCL_ERROR(...) ((void)0)
#define PCL_EXPORTS
