// shim for the one Boost.Math function mtkmath.hpp uses (cos_sinc_sqrt's Taylor bound): machine epsilon
#pragma once
#include <limits>
namespace boost { namespace math { namespace tools {
template <class T> inline T epsilon() { return std::numeric_limits<T>::epsilon(); }
}}}
