// stand-in for slam/common/slam_base.h (key frames, PCD writer, the SLAM base class): pose_estimator.cpp includes it without using it
#pragma once
#include "mapping_types.h"
