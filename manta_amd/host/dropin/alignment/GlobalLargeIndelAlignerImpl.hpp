// Shadows Manta's alignment/GlobalLargeIndelAlignerImpl.hpp: GlobalLargeIndelAligner<ScoreType>::align on the device.
#pragma once
#include "manta_amd_dropin.hpp"

template <typename ScoreType>
template <typename SymIter>
void GlobalLargeIndelAligner<ScoreType>::align(
    const SymIter queryBegin, const SymIter queryEnd, const SymIter refBegin, const SymIter refEnd, AlignmentResult<ScoreType>& result) const
{
  result.clear();
  std::vector<uint32_t>      cigar;
  const manta_align_result_t r =
      manta_amd_dropin::alignOne(MANTA_ALIGNER_LARGE_INDEL, manta_amd_dropin::toAbi(this->getScores()), int32_t(_largeIndelScore), queryBegin,
                                 queryEnd, refBegin, refEnd, refEnd, refEnd, cigar);
  result.score          = ScoreType(r.score);
  result.isJumped       = r.is_jumped != 0;
  result.align.beginPos = r.begin_pos1;
  manta_amd_dropin::toPath(cigar.data() + r.cigar1_off, r.cigar1_len, result.align.apath);
}
