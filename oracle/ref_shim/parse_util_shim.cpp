// TEST INFRASTRUCTURE shim: blt_util/parse_util.cpp needs boost::spirit; align_path.cpp only needs this one
// function (cigar string -> path), which the oracle driver never exercises on malformed input.
#include "blt_util/parse_util.hpp"
#include <cstdlib>
namespace illumina { namespace blt_util {
unsigned parse_unsigned(const char*& s) {
  char* end = nullptr;
  const unsigned long v = std::strtoul(s, &end, 10);
  s = end;
  return static_cast<unsigned>(v);
}
}}  // namespace illumina::blt_util
