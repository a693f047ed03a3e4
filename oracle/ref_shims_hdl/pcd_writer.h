// stand-in for slam/common/pcd_writer.h (PCD export, needs PCL's IO): slam_base.h includes it without HDL_FastLIO using it
#pragma once
