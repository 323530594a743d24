// Test-only C surface over manta_amd/host/read_pile.hpp: packs BAM-style records and returns the packed arrays, and
// runs packed piles through manta_assemble-free paths of the ABI (the small-SV pipeline) so that tests can compare the
// packed-input path with the 1-byte-per-base path.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "read_pile.hpp"

using namespace manta_amd;
#define MINE_EXPORT extern "C" __attribute__((visibility("default")))

/// n records of one locus; returns the number of reads accepted.  Outputs sized by the caller (len/16+1 dwords per read ...)
MINE_EXPORT int mine_pack_bam_reads(
    unsigned n, const uint8_t* const* seq4, const uint8_t* const* qual, const unsigned* len, const int* isReversed, unsigned minQval,
    uint32_t* codesOut, uint32_t* maskOut, uint64_t* codeOff, uint64_t* maskOff, uint32_t* readLen, int* accepted)
{
  ReadPileBuilder b;
  for (unsigned i = 0; i < n; ++i) accepted[i] = b.addBamRead(seq4[i], qual[i], len[i], uint8_t(minQval), isReversed[i] != 0) ? 1 : 0;
  b.endLocus();
  std::memcpy(codesOut, b.codes.data(), 4 * b.codes.size());
  std::memcpy(maskOut, b.nmask.data(), 4 * b.nmask.size());
  std::memcpy(codeOff, b.codeOff.data(), 8 * b.codeOff.size());
  std::memcpy(maskOff, b.maskOff.data(), 8 * b.maskOff.size());
  std::memcpy(readLen, b.readLen.data(), 4 * b.readLen.size());
  return int(b.nReads());
}

/// text reads -> packed arrays (same output convention)
MINE_EXPORT int mine_pack_text_reads(
    unsigned n, const char* const* reads, uint32_t* codesOut, uint32_t* maskOut, uint64_t* codeOff, uint64_t* maskOff, uint32_t* readLen)
{
  ReadPileBuilder b;
  for (unsigned i = 0; i < n; ++i)
    if (!b.addRead(reads[i])) return -1;
  b.endLocus();
  std::memcpy(codesOut, b.codes.data(), 4 * b.codes.size());
  std::memcpy(maskOut, b.nmask.data(), 4 * b.nmask.size());
  std::memcpy(codeOff, b.codeOff.data(), 8 * b.codeOff.size());
  std::memcpy(maskOff, b.maskOff.data(), 8 * b.maskOff.size());
  std::memcpy(readLen, b.readLen.data(), 4 * b.readLen.size());
  return int(b.nReads());
}
