// Read-pile construction for the device: what SVCandidateAssembler::getBreakendReads keeps of a BAM
// record -- insertAssemblyRead, manta/SVCandidateAssembler.cpp:102-136 -- emitted straight into the packed pile layout of
// include/manta_amd.h (manta_packed_piles_t) instead of one std::string per read:
//     reads.push_back(bamRead.get_bam_read().get_string());      4-bit BAM codes -> text  (htsapi/bam_seq.hpp:41-59,170-177)
//     if (qual[i] < minQval) nread[i] = 'N';                     :130-132
//     if (isReversed) reverseCompStr(reads.back());              :134  (blt_util/seq_util.hpp:150-204)
// Paths relative to /root/reference/src/c++/lib.  The BAM scan itself (region seeks, read classification) stays the
// reference's: this builder is what its loop body calls instead of insertAssemblyRead's string handling.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/manta_amd.h"

namespace manta_amd {

struct ReadPileBuilder {
  std::vector<uint32_t> codes, nmask, readLen, locusBegin{0};
  std::vector<uint64_t> codeOff{0}, maskOff{0};

  void clear() { *this = ReadPileBuilder(); }
  uint32_t nLoci() const { return uint32_t(locusBegin.size() - 1); }
  uint32_t nReads() const { return uint32_t(readLen.size()); }

  /// one BAM record as htslib lays it out: `seq4` = bam_get_seq (two bases per byte, first base in the high nibble),
  /// `qual` = bam_get_qual.  Returns false (and adds nothing) if the record holds the BAM code '=' ("same as reference"):
  /// legal for the reference's byte-generic assembler in forward orientation, a fatal base_error in reverse
  /// (seq_util.hpp:163-165), and not representable in two bits -- the caller reports the locus as unsupported.
  bool addBamRead(const uint8_t* seq4, const uint8_t* qual, const unsigned len, const uint8_t minQval, const bool isReversed)
  {
    // BAM 4-bit code -> 2-bit code (bit 2: 'N', bit 3: '=')   (BAM_BASE: A=1, C=2, G=4, T=8, everything else reads as 'N')
    static const uint8_t lut[16] = {8, 0, 1, 4, 2, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4};
    for (unsigned i = 0; i < len; ++i)
      if (((seq4[i >> 1] >> ((~i & 1) << 2)) & 0xf) == 0) return false;
    const size_t c0 = codes.size(), m0 = nmask.size();
    codes.resize(c0 + (len + 15) / 16, 0u);
    nmask.resize(m0 + (len + 31) / 32, 0u);
    for (unsigned i = 0; i < len; ++i) {
      const unsigned src = isReversed ? (len - 1 - i) : i;  // output position i takes source base `src`
      unsigned       c   = lut[(seq4[src >> 1] >> ((~src & 1) << 2)) & 0xf];
      if (qual[src] < minQval) c = 4;
      if (c & 4) {
        nmask[m0 + (i >> 5)] |= 1u << (i & 31);
      } else {
        if (isReversed) c = 3 - c;  // A<->T, C<->G
        codes[c0 + (i >> 4)] |= c << (30 - 2 * (i & 15));
      }
    }
    finishRead(len);
    return true;
  }

  /// a read that already is text over {A,C,G,T,N} (the mirror refiner's AssemblyReadInput); false on any other byte
  bool addRead(const std::string& read)
  {
    const unsigned len = unsigned(read.size());
    for (const char ch : read)
      if (ch != 'A' && ch != 'C' && ch != 'G' && ch != 'T' && ch != 'N') return false;
    const size_t c0 = codes.size(), m0 = nmask.size();
    codes.resize(c0 + (len + 15) / 16, 0u);
    nmask.resize(m0 + (len + 31) / 32, 0u);
    for (unsigned i = 0; i < len; ++i) {
      const char ch = read[i];
      if (ch == 'N') {
        nmask[m0 + (i >> 5)] |= 1u << (i & 31);
      } else {
        const unsigned c = (ch == 'A') ? 0u : (ch == 'C') ? 1u : (ch == 'G') ? 2u : 3u;
        codes[c0 + (i >> 4)] |= c << (30 - 2 * (i & 15));
      }
    }
    finishRead(len);
    return true;
  }

  /// closes the pile of one candidate locus (one runIterativeAssembler call of the reference, :661-698)
  void endLocus() { locusBegin.push_back(nReads()); }

  manta_packed_piles_t view() const
  {
    manta_packed_piles_t p;
    p.codes            = codes.data();
    p.nmask            = nmask.data();
    p.read_len         = readLen.data();
    p.read_code_off    = codeOff.data();
    p.read_mask_off    = maskOff.data();
    p.locus_read_begin = locusBegin.data();
    return p;
  }

private:
  void finishRead(const unsigned len)
  {
    readLen.push_back(len);
    codeOff.push_back(codes.size());
    maskOff.push_back(nmask.size());
  }
};

}  // namespace manta_amd
