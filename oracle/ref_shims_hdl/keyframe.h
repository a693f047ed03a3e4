// stand-in for slam/common/keyframe.h (g2o vertices, PCD IO): only the name is needed by slam_base.h / backend_api.h signatures
#pragma once
struct KeyFrame;
