// Test-infrastructure shim (NOT boost): only ExceptionData::getContext() wants a timestamp string.
#pragma once
#include <ctime>
#include <string>
namespace boost { namespace posix_time {
struct ptime { std::time_t t; };
struct second_clock { static ptime local_time() { return ptime{std::time(nullptr)}; } };
inline std::string to_simple_string(const ptime& p) {
  char buf[64];
  std::tm tmv;
  localtime_r(&p.t, &tmv);
  std::strftime(buf, sizeof(buf), "%Y-%b-%d %H:%M:%S", &tmv);
  return buf;
}
}}  // namespace boost::posix_time
