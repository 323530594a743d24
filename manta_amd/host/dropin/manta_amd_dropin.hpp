// Drop-in glue for a REAL Manta source tree (applied mechanically).
//
// Put this directory in front of Manta's src/c++/lib on the include path and link runIterativeAssembler.cpp instead of
// assembly/IterativeAssembler.cpp: the three alignment/*Impl.hpp files here shadow the reference's implementation headers
// (the class declarations GlobalAligner.hpp / GlobalLargeIndelAligner.hpp / GlobalJumpAligner.hpp stay the reference's own
// and #include "alignment/...Impl.hpp" at their end), so SVCandidateAssemblyRefiner.cpp -- unmodified -- instantiates
// align() bodies that forward to the C ABI of include/manta_amd.h.  Everything here is written against Manta's own types
// (AlignmentResult, JumpAlignmentResult, ALIGNPATH::path_t, AssembledContig, AssemblyReadInfo, IterativeAssemblerOptions,
// illumina::common::GeneralException); nothing of manta_amd/host/'s mirror types is used.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "alignment/Alignment.hpp"
#include "alignment/AlignmentScores.hpp"
#include "blt_util/align_path.hpp"
#include "common/Exceptions.hpp"

#include "manta_amd.h"

namespace manta_amd_dropin {

namespace {
struct ThreadContextHolder {
  manta_ctx_t* ctx = nullptr;
  ~ThreadContextHolder() { manta_ctx_destroy(ctx); }
};
// internal linkage on purpose (a function-local static of an inline function is one object per PROCESS, even across
// differently built copies of this glue loaded side by side)
thread_local ThreadContextHolder g_threadContext;
}  // namespace

/// one ABI context per host thread (GenerateSVCandidates.cpp:232-250: one refiner, hence one aligner set, per worker)
inline manta_ctx_t* threadContext()
{
  ThreadContextHolder& h(g_threadContext);
  if (!h.ctx) {
    if (manta_ctx_create(-1, &h.ctx) != MANTA_OK)
      BOOST_THROW_EXCEPTION(illumina::common::GeneralException(std::string("manta_amd: no usable GPU context: ") + manta_last_error(nullptr)));
  }
  return h.ctx;
}

template <typename ScoreType>
manta_align_scores_t toAbi(const AlignmentScores<ScoreType>& s)
{
  return manta_align_scores_t{int32_t(s.match), int32_t(s.mismatch), int32_t(s.open), int32_t(s.extend), int32_t(s.offEdge),
                              s.isAllowEdgeInsertion ? 1 : 0};
}

/// BAM-packed CIGAR words -> ALIGNPATH::path_t
inline void toPath(const uint32_t* cig, const unsigned n, ALIGNPATH::path_t& path)
{
  static const ALIGNPATH::align_t map[9] = {ALIGNPATH::MATCH,     ALIGNPATH::INSERT,    ALIGNPATH::DELETE,   ALIGNPATH::SKIP,     ALIGNPATH::SOFT_CLIP,
                                            ALIGNPATH::HARD_CLIP, ALIGNPATH::PAD,       ALIGNPATH::SEQ_MATCH, ALIGNPATH::SEQ_MISMATCH};
  path.clear();
  for (unsigned i = 0; i < n; ++i) path.push_back(ALIGNPATH::path_segment(map[cig[i] & 15u], cig[i] >> 4));
}

/// one alignment through manta_align_batch; the reference's own input checks and messages
template <typename SymIter>
manta_align_result_t alignOne(
    const int kind, const manta_align_scores_t& sc, const int32_t extra, SymIter qb, SymIter qe, SymIter r1b, SymIter r1e, SymIter r2b,
    SymIter r2e, std::vector<uint32_t>& cigar)
{
  using illumina::common::GeneralException;
  std::vector<uint8_t> arena(qb, qe);
  manta_align_task_t   t{};
  t.query_len = uint32_t(arena.size());
  t.ref1_off  = arena.size();
  arena.insert(arena.end(), r1b, r1e);
  t.ref1_len = uint32_t(arena.size() - t.ref1_off);
  t.ref2_off = arena.size();
  arena.insert(arena.end(), r2b, r2e);
  t.ref2_len = uint32_t(arena.size() - t.ref2_off);
  if (t.query_len == 0) BOOST_THROW_EXCEPTION(GeneralException("Unexpected empty query sequence"));
  if (t.ref1_len == 0)
    BOOST_THROW_EXCEPTION(GeneralException(kind == MANTA_ALIGNER_JUMP ? "Unexpected empty reference1 sequence" : "Unexpected empty reference sequence"));
  if (kind == MANTA_ALIGNER_JUMP && t.ref2_len == 0) BOOST_THROW_EXCEPTION(GeneralException("Unexpected empty reference2 sequence"));
  arena.push_back(0);
  cigar.assign(2 * size_t(t.query_len) + 16, 0);
  manta_align_result_t res{};
  uint64_t             used = 0;
  manta_ctx_t*         ctx  = threadContext();
  const int rc = manta_align_batch(ctx, kind, &sc, extra, 1, &t, arena.data(), arena.size() - 1, &res, cigar.data(), cigar.size(), &used);
  if (rc != MANTA_OK) BOOST_THROW_EXCEPTION(GeneralException(std::string("manta_amd aligner: ") + manta_last_error(ctx)));
  return res;
}

}  // namespace manta_amd_dropin
