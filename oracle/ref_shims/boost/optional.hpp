// stand-in for <boost/optional.hpp> over std::optional (Boost is not installed)
#pragma once
#include <optional>
namespace boost {
template <typename T>
using optional = std::optional<T>;
inline constexpr std::nullopt_t none = std::nullopt;
}  // namespace boost
