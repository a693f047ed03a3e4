// stand-in for <pcl/search/kdtree.h>: included by ndt_cuda.hpp, nothing from it is used
#pragma once
