// stand-in for <pcl/filters/boost.h> (harness ref_voxelgrid_cov.cpp only): the two Boost names voxel_grid_covariance_omp.h uses
#pragma once
#include <iostream>
#include <memory>
#include <random>
namespace boost {
using std::shared_ptr;
using std::make_shared;
// (named by getDisplayCloud, which the harness never calls)
using std::mt19937;
template <typename T = double>
using normal_distribution = std::normal_distribution<T>;
template <typename E, typename D>
struct variate_generator {
    E e; D d;
    variate_generator(E e_, D d_) : e(e_), d(d_) {}
    double operator()() { return d(e); }
};
namespace mpl {
template <typename L>
struct size { static constexpr int value = L::value; };
}  // namespace mpl
}  // namespace boost
