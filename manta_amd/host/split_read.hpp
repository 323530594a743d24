// Host adapter of the device split-read scorer: the reference's own interface
//   void splitReadAligner(flankScoreSize, querySeq, qualConvert, queryQual, targetSeq, targetBpOffsetRange, alignment)
//                                      applications/GenerateSVCandidates/SplitReadAlignment.hpp:57-64
//   struct SRAlignmentInfo             SplitReadAlignment.hpp:33-49
//   struct qscore_snp                  blt_util/qscore_snp.hpp:33-55
// on top of manta_split_read_batch (include/manta_amd.h), plus a batched form for SVScorerSplit's per-read loops
// (SVScorerSplit.cpp scores every read against the alt contig and both reference breakends: three calls per read).
// Paths relative to /root/reference/src/c++/lib.  No CPU path: the scan runs in split_read_kernel; what stays here is
// the scalar tail of the reference function (alignScore, the three float ratio tests of isEvidenceCheck, evidence).
#pragma once
#include <algorithm>
#include <cmath>
#include <sstream>
#include <string>
#include <vector>

#include "manta_amd.hpp"

namespace manta_amd {

/// blt_util/qscore_snp.cpp:26-38 (phred_to_error_prob qscore.hpp:68-71, log1p_switch math_util.hpp:35-45)
struct qscore_snp {
  enum { MAX_QSCORE = 70 };  // blt_util/qscore_cache.hpp:46
  explicit qscore_snp(const double snp_prob)
  {
    const double comp_snp3(1. - (snp_prob / 3.));
    for (int i(0); i <= MAX_QSCORE; ++i) {
      const double qerr(std::pow(10., -static_cast<double>(i) / 10.));
      _q2p[i]         = (qerr * comp_snp3) + ((1 - qerr) * snp_prob);
      const double x  = -_q2p[i];
      _q2lncompe[i]   = (std::abs(x) < 0.01) ? std::log1p(x) : std::log(1 + x);
      _q2lne[i]       = std::log(_q2p[i]);
    }
  }
  double        qphred_to_error_prob(const int q) const { return _q2p[check(q)]; }
  double        qphred_to_ln_comp_error_prob(const int q) const { return _q2lncompe[check(q)]; }
  double        qphred_to_ln_error_prob(const int q) const { return _q2lne[check(q)]; }
  const double* lnCompErrorTable() const { return _q2lncompe; }
  const double* lnErrorTable() const { return _q2lne; }

private:
  static int check(const int q)
  {
    if (q < 0 || q > MAX_QSCORE) throw GeneralException("qscore_snp: basecall quality outside [0,70]");
    return q;
  }
  double _q2p[MAX_QSCORE + 1], _q2lncompe[MAX_QSCORE + 1], _q2lne[MAX_QSCORE + 1];
};

struct SRAlignmentInfo {
  unsigned alignPos = 0, leftSize = 0, homSize = 0, rightSize = 0, leftMismatches = 0, homMismatches = 0, rightMismatches = 0, alignScore = 0;
  float    alignLnLhood    = 0;
  bool     isEvidence      = false;
  bool     isTier2Evidence = false;
  float    evidence        = 0;
};

namespace detail {
inline bool isEvidenceCheck(const SRAlignmentInfo& alignment, const unsigned minFlankSize)  // :123-136
{
  if (alignment.leftSize < minFlankSize) return false;
  if (alignment.rightSize < minFlankSize) return false;
  if ((alignment.leftMismatches / (float)alignment.leftSize) >= 0.25) return false;
  if ((alignment.rightMismatches / (float)alignment.rightSize) >= 0.25) return false;
  const float size(static_cast<float>(alignment.leftSize + alignment.rightSize));
  if ((alignment.alignScore / size) < 0.9) return false;
  return true;
}
inline void setEvidence(SRAlignmentInfo& alignment)  // :138-155
{
  static const unsigned minFlankSize(16);
  static const unsigned minFlankSizeTier2(8);
  alignment.isEvidence      = isEvidenceCheck(alignment, minFlankSize);
  alignment.isTier2Evidence = isEvidenceCheck(alignment, minFlankSizeTier2);
  alignment.evidence        = 0;
  if (!(alignment.isEvidence || alignment.isTier2Evidence)) return;
  const float size(static_cast<float>(alignment.leftSize + alignment.rightSize));
  alignment.evidence = 2 * std::min(alignment.leftSize, alignment.rightSize) / (size);
}
}  // namespace detail

/// one (read, target) pair of a batch; the strings and the quality array must stay alive until the call returns
struct SplitReadTask {
  unsigned           flankScoreSize = 0;
  const std::string* querySeq       = nullptr;
  const uint8_t*     queryQual      = nullptr;
  const std::string* targetSeq      = nullptr;
  known_pos_range2   targetBpOffsetRange;
};

/// splitReadAligner for a list of pairs in one device launch.  `errors` (optional, one slot per task): the exception the
/// reference would have thrown for that pair; without it the first one is thrown after the batch.
inline void splitReadAlignerBatch(
    const qscore_snp& qualConvert, const std::vector<SplitReadTask>& tasks, std::vector<SRAlignmentInfo>& alignments,
    std::vector<std::string>* errors = nullptr)
{
  const size_t n = tasks.size();
  alignments.assign(n, SRAlignmentInfo());
  if (errors) errors->assign(n, std::string());
  if (n == 0) return;
  std::vector<uint8_t>            arena;
  std::vector<manta_split_task_t> abi(n);
  for (size_t i = 0; i < n; ++i) {
    const SplitReadTask& t(tasks[i]);
    manta_split_task_t&  a(abi[i]);
    a.query_off = arena.size();
    arena.insert(arena.end(), t.querySeq->begin(), t.querySeq->end());
    a.qual_off = arena.size();
    arena.insert(arena.end(), t.queryQual, t.queryQual + t.querySeq->size());
    a.target_off = arena.size();
    arena.insert(arena.end(), t.targetSeq->begin(), t.targetSeq->end());
    a.query_len        = uint32_t(t.querySeq->size());
    a.target_len       = uint32_t(t.targetSeq->size());
    a.bp_begin         = t.targetBpOffsetRange.begin_pos();
    a.bp_end           = t.targetBpOffsetRange.end_pos();
    a.flank_score_size = t.flankScoreSize;
    a.reserved         = 0;
  }
  static const float ln_one_third(std::log(1 / 3.f));  // SplitReadAlignment.cpp:50
  static const float lnRandomBase(-std::log(4.f));     // :76
  std::vector<manta_split_result_t> res(n);
  manta_ctx_t*                      ctx = threadContext();
  const int rc = manta_split_read_batch(ctx, qualConvert.lnCompErrorTable(), qualConvert.lnErrorTable(), qscore_snp::MAX_QSCORE + 1, ln_one_third,
                                        lnRandomBase, uint32_t(n), abi.data(), arena.data(), arena.size(), res.data());
  if (rc != MANTA_OK && rc != MANTA_E_SPLIT_QUERY_NOT_SHORTER && rc != MANTA_E_SPLIT_EMPTY_SCAN && rc != MANTA_E_UNSUPPORTED)
    throw GeneralException(std::string("manta_amd split-read scorer: ") + manta_last_error(ctx), rc);
  std::string firstError;
  for (size_t i = 0; i < n; ++i) {
    const manta_split_result_t& r(res[i]);
    const SplitReadTask&        t(tasks[i]);
    std::string                 err;
    const unsigned              querySize = unsigned(t.querySeq->size());
    if (r.status == MANTA_E_SPLIT_QUERY_NOT_SHORTER) {
      std::ostringstream oss;
      oss << "Unexpected split read alignment input. querySize: " << querySize << " targetSize: " << t.targetSeq->size();  // :237-247
      err = oss.str();
    } else if (r.status == MANTA_E_SPLIT_EMPTY_SCAN) {
      err = "Unexpected split read alignment input condition: scanEnd < scanStart.";  // :265-273
    } else if (r.status != MANTA_OK) {
      err = "basecall quality above the supported range";  // qphred_cache::qscore_check (qscore_cache.hpp:49-52)
    } else if (r.left_size > querySize) {
      err = "Unexpected split read alignment outcome.";  // :317-331
    }
    if (!err.empty()) {
      if (errors) (*errors)[i] = err;
      if (firstError.empty()) firstError = err;
      continue;
    }
    SRAlignmentInfo& a(alignments[i]);
    a.alignPos        = r.best_pos;
    a.alignLnLhood    = r.best_ln_lhood;
    a.leftSize        = r.left_size;
    a.homSize         = r.hom_size;
    a.rightSize       = r.right_size;
    a.leftMismatches  = r.left_mismatches;
    a.homMismatches   = r.hom_mismatches;
    a.rightMismatches = r.right_mismatches;
    a.alignScore      = querySize - (a.leftMismatches + a.homMismatches + a.rightMismatches);  // :119-120
    detail::setEvidence(a);                                                                    // :346
  }
  if (!errors && !firstError.empty()) throw GeneralException(firstError);
}

/// the reference's signature: a batch of one
inline void splitReadAligner(
    const unsigned flankScoreSize, const std::string& querySeq, const qscore_snp& qualConvert, const uint8_t* queryQual,
    const std::string& targetSeq, const known_pos_range2& targetBpOffsetRange, SRAlignmentInfo& alignment)
{
  SplitReadTask t;
  t.flankScoreSize      = flankScoreSize;
  t.querySeq            = &querySeq;
  t.queryQual           = queryQual;
  t.targetSeq           = &targetSeq;
  t.targetBpOffsetRange = targetBpOffsetRange;
  std::vector<SRAlignmentInfo> out;
  splitReadAlignerBatch(qualConvert, std::vector<SplitReadTask>(1, t), out);
  alignment = out[0];
}

}  // namespace manta_amd
