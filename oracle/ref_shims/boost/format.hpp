// stand-in for <boost/format.hpp>: lsq_registration_impl.hpp formats its optional LM debug table with it (lm_debug_print_ is off)
#pragma once
#include <ostream>
#include <string>
namespace boost {
class format {
    std::string s_;

   public:
    explicit format(const char* s) : s_(s) {}
    template <typename T>
    format& operator%(const T&) { return *this; }
    friend std::ostream& operator<<(std::ostream& os, const format& f) { return os << f.s_; }
};
template <typename T>
using shared_ptr = std::shared_ptr<T>;
}  // namespace boost
