// Test-infrastructure shim (NOT boost): stringstream-based lexical_cast, enough for CIGAR printing.
#pragma once
#include <sstream>
#include <stdexcept>
#include <string>
namespace boost {
template <typename Target, typename Source> Target lexical_cast(const Source& s) {
  std::stringstream ss;
  ss << s;
  Target t;
  if (!(ss >> t)) throw std::runtime_error("bad lexical_cast (shim)");
  return t;
}
}  // namespace boost
