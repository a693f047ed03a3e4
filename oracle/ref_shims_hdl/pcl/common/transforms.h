// <pcl/common/transforms.h> for slam_base.h: transformPoint (ref_shims) + transformPointCloud as PCL 1.9.1 computes it
#pragma once
#include "../../../ref_shims/pcl/common/transforms.h"
#include <pcl/registration/registration.h>
