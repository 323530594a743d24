// Test-infrastructure shim (NOT boost): boost::chrono surface used by blt_util/time_util.hpp, mapped onto std::chrono
#pragma once
#include <chrono>
namespace boost { namespace chrono {
using std::chrono::duration_cast;
using std::chrono::microseconds;
using std::chrono::nanoseconds;
}}  // namespace boost::chrono
