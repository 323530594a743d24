// TEST INFRASTRUCTURE (boost stand-in, written here): boost::math::log1p for double forwards to the C library's log1p
// wherever the platform has one (BOOST_HAS_LOG1P, all glibc targets); blt_util/math_util.hpp:35-45 is its only user.
#pragma once
#include <cmath>
namespace boost {
namespace math {
template <typename T>
inline T log1p(const T x)
{
  return std::log1p(x);
}
}  // namespace math
}  // namespace boost
