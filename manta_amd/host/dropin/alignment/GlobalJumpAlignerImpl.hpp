// Shadows Manta's alignment/GlobalJumpAlignerImpl.hpp: GlobalJumpAligner<ScoreType>::align on the device.
#pragma once
#include "manta_amd_dropin.hpp"

template <typename ScoreType>
template <typename SymIter>
void GlobalJumpAligner<ScoreType>::align(
    const SymIter queryBegin, const SymIter queryEnd, const SymIter ref1Begin, const SymIter ref1End, const SymIter ref2Begin,
    const SymIter ref2End, JumpAlignmentResult<ScoreType>& result) const
{
  result.clear();
  std::vector<uint32_t>      cigar;
  const manta_align_result_t r =
      manta_amd_dropin::alignOne(MANTA_ALIGNER_JUMP, manta_amd_dropin::toAbi(this->getScores()), int32_t(this->getJumpScore()), queryBegin, queryEnd,
                                 ref1Begin, ref1End, ref2Begin, ref2End, cigar);
  result.score           = ScoreType(r.score);
  result.jumpInsertSize  = r.jump_insert_size;
  result.jumpRange       = r.jump_range;
  result.align1.beginPos = r.begin_pos1;
  result.align2.beginPos = r.begin_pos2;
  manta_amd_dropin::toPath(cigar.data() + r.cigar1_off, r.cigar1_len, result.align1.apath);
  manta_amd_dropin::toPath(cigar.data() + r.cigar2_off, r.cigar2_len, result.align2.apath);
}
