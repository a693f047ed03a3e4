// stand-in for <boost/functional/hash.hpp> (Boost is not installed): boost::hash_combine as fast_vgicp_voxel.hpp's Vector3iHash uses it.
// Only the bucket order of an unordered_map depends on it, never a result.  Test infrastructure.
#pragma once
#include <cstddef>
#include <functional>
namespace boost {
template <typename T>
inline void hash_combine(std::size_t& seed, const T& v) {
    seed ^= std::hash<T>()(v) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
}
}  // namespace boost
