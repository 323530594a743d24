// TEST INFRASTRUCTURE shim: blt_util/parse_util.cpp needs boost::spirit; align_path.cpp only needs this one
// function (cigar string -> path), which the oracle driver never exercises on malformed input.
#include "blt_util/parse_util.hpp"
#include <cstdlib>
#include <string>
namespace illumina { namespace blt_util {
unsigned parse_unsigned(const char*& s) {
  char* end = nullptr;
  const unsigned long v = std::strtoul(s, &end, 10);
  s = end;
  return static_cast<unsigned>(v);
}
int parse_int_str(const std::string& s) {  // used by htsapi/bam_header_util.cpp on well-formed region strings only
  return static_cast<int>(std::strtol(s.c_str(), nullptr, 10));
}
double parse_double(const char*& s, const char*) {  // blt_util/chrom_depth_map.cpp (the depth file is never given here)
  char* end = nullptr;
  const double v = std::strtod(s, &end);
  s = end;
  return v;
}
}}  // namespace illumina::blt_util
