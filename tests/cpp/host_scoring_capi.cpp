// Test-only C surface over manta_amd/host/split_read.hpp (and shadow_align.hpp) with the SAME signatures and text format
// as oracle/ref_scoring_driver.cpp / oracle/scoring_oracle.cpp, so tests can compare product and checkers as strings.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "split_read.hpp"

using namespace manta_amd;
#define MINE_EXPORT extern "C" __attribute__((visibility("default")))

namespace {
int emitText(const std::string& s, char* out, int cap)
{
  const int n = static_cast<int>(s.size());
  if (out != nullptr && cap > 0) {
    const int m = (n < cap - 1) ? n : (cap - 1);
    std::memcpy(out, s.data(), m);
    out[m] = '\0';
  }
  return n;
}
std::string infoText(const SRAlignmentInfo& a)
{
  char buf[512];
  std::snprintf(buf, sizeof(buf),
                "alignPos=%u leftSize=%u homSize=%u rightSize=%u leftMismatches=%u homMismatches=%u rightMismatches=%u alignScore=%u "
                "isEvidence=%d isTier2Evidence=%d evidence=%a alignLnLhood=%a\n",
                a.alignPos, a.leftSize, a.homSize, a.rightSize, a.leftMismatches, a.homMismatches, a.rightMismatches, a.alignScore,
                a.isEvidence ? 1 : 0, a.isTier2Evidence ? 1 : 0, double(a.evidence), double(a.alignLnLhood));
  return buf;
}
}  // namespace

MINE_EXPORT int mine_qscore_snp_tables(double snpPrior, double* lnCompError, double* lnError, float* lnOneThird, float* lnRandomBase)
{
  const qscore_snp q(snpPrior);
  for (int i = 0; i <= qscore_snp::MAX_QSCORE; ++i) {
    lnCompError[i] = q.qphred_to_ln_comp_error_prob(i);
    lnError[i]     = q.qphred_to_ln_error_prob(i);
  }
  *lnOneThird   = std::log(1 / 3.f);
  *lnRandomBase = -std::log(4.f);
  return qscore_snp::MAX_QSCORE + 1;
}

MINE_EXPORT int mine_split_read_aligner(
    unsigned flankScoreSize, const char* query, unsigned queryLen, const uint8_t* qual, const char* target, unsigned targetLen, int bpBegin,
    int bpEnd, double snpPrior, char* out, int cap)
{
  try {
    const qscore_snp  q(snpPrior);
    SRAlignmentInfo   a;
    const std::string qs(query, queryLen), ts(target, targetLen);
    splitReadAligner(flankScoreSize, qs, q, qual, ts, known_pos_range2(bpBegin, bpEnd), a);
    return emitText(infoText(a), out, cap);
  } catch (const std::exception&) {
    return emitText("EXCEPTION\n", out, cap);
  }
}

/// n pairs in ONE device launch; texts separated by nothing (each ends in a newline)
MINE_EXPORT int mine_split_read_aligner_batch(
    unsigned n, const unsigned* flankScoreSize, const char* const* query, const unsigned* queryLen, const uint8_t* const* qual,
    const char* const* target, const unsigned* targetLen, const int* bpBegin, const int* bpEnd, double snpPrior, char* out, int cap)
{
  try {
    const qscore_snp           q(snpPrior);
    std::vector<std::string>   qs(n), ts(n);
    std::vector<SplitReadTask> tasks(n);
    for (unsigned i = 0; i < n; ++i) {
      qs[i].assign(query[i], queryLen[i]);
      ts[i].assign(target[i], targetLen[i]);
      tasks[i].flankScoreSize      = flankScoreSize[i];
      tasks[i].querySeq            = &qs[i];
      tasks[i].queryQual           = qual[i];
      tasks[i].targetSeq           = &ts[i];
      tasks[i].targetBpOffsetRange = known_pos_range2(bpBegin[i], bpEnd[i]);
    }
    std::vector<SRAlignmentInfo> res;
    std::vector<std::string>     errors;
    splitReadAlignerBatch(q, tasks, res, &errors);
    std::string text;
    for (unsigned i = 0; i < n; ++i) text += errors[i].empty() ? infoText(res[i]) : std::string("EXCEPTION\n");
    return emitText(text, out, cap);
  } catch (const std::exception& e) {
    return emitText(std::string("FATAL ") + e.what() + "\n", out, cap);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// shadow-read aligner call site (same signature and text as ref_shadow_realign in oracle/ref_scoring_driver.cpp)
// ---------------------------------------------------------------------------------------------------------------
#include "shadow_align.hpp"

MINE_EXPORT int mine_shadow_realign(
    const char* extendedContig, int bp1Begin, int bp1End, int bp2Begin, int bp2End, const char* insertSeq, int isUnknownSizeInsertion,
    const char* unknownLeft, const char* unknownRight, int align1BeginPos, const char* align1Cigar, unsigned nReads, const char* const* reads,
    const int* isLeftOfInsert, const int* anchorPos, char* out, int cap)
{
  std::string text;
  try {
    SVCandidate sv;
    sv.bp1.interval = GenomeInterval(0, bp1Begin, bp1End);
    sv.bp2.interval = GenomeInterval(0, bp2Begin, bp2End);
    sv.bp1.state    = SVBreakendState::RIGHT_OPEN;
    sv.bp2.state    = SVBreakendState::LEFT_OPEN;
    sv.insertSeq    = insertSeq;
    sv.isUnknownSizeInsertion       = isUnknownSizeInsertion != 0;
    sv.unknownSizeInsertionLeftSeq  = unknownLeft;
    sv.unknownSizeInsertionRightSeq = unknownRight;
    sv.setPrecise();
    sv.assemblyAlignIndex = 0;
    SVCandidateAssemblyData data;
    data.isSpanning         = true;
    data.bestAlignmentIndex = 0;
    SVCandidateAssemblyData::JumpAlignmentResultType ja;
    ja.align1.beginPos = align1BeginPos;
    ja.align1.apath    = ALIGNPATH::cigar_to_apath(align1Cigar);
    data.spanningAlignments.push_back(ja);
    data.extendedContigs.push_back(extendedContig);
    const AlignmentScores<int> spanningAlignScores(2, -8, -12, -1, -1);  // SVRefinerOptions.hpp:43
    const ShadowRealigner      sh(spanningAlignScores, 50, data, sv);    // PairOptions::minFragSupport = 50
    std::vector<std::string>   rs(nReads);
    std::vector<ShadowRead>    in(nReads);
    for (unsigned i = 0; i < nReads; ++i) {
      rs[i]                = reads[i];
      in[i].isLeftOfInsert = isLeftOfInsert[i] != 0;
      in[i].floatRead      = &rs[i];
      in[i].anchorPos      = anchorPos[i];
    }
    std::vector<ShadowResult> res;
    sh.realignPairedReads(in, res);
    for (unsigned i = 0; i < nReads; ++i) {
      if (!res[i].error.empty())
        text += "EXCEPTION\n";
      else
        text += "pass=" + std::to_string(res[i].isUsable ? 1 : 0) + " altTemplateSize=" + std::to_string(res[i].isUsable ? res[i].altTemplateSize : 0) + "\n";
    }
  } catch (const std::exception& e) {
    text = std::string("FATAL ") + e.what() + "\n";
  }
  return emitText(text, out, cap);
}
