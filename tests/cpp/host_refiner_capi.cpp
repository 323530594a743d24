// Test-only C surface over manta_amd/host/refiner_util.hpp with the SAME signatures as oracle/ref_refiner_driver.cpp
// (which wraps the unmodified reference statics), so that tests/test_refiner_util.py can fuzz one against the other.
#include <cstdint>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "refiner_util.hpp"

using namespace manta_amd;
#define MINE_EXPORT extern "C" __attribute__((visibility("default")))

namespace {
int emit(const std::string& s, char* out, int cap)
{
  const int n = static_cast<int>(s.size());
  if (out != nullptr && cap > 0) {
    const int m = (n < cap - 1) ? n : (cap - 1);
    std::memcpy(out, s.data(), m);
    out[m] = '\0';
  }
  return n;
}
AlignmentScores<int> mk(const int32_t* s)
{
  return AlignmentScores<int>(s[0], s[1], s[2], s[3], s[4], s[5] != 0);
}
ALIGNPATH::path_t path(const char* cigar)
{
  return ALIGNPATH::cigar_to_apath(cigar);
}
std::string segText(const std::vector<std::pair<unsigned, unsigned>>& segs)
{
  std::ostringstream os;
  for (size_t i = 0; i < segs.size(); ++i) os << (i ? "," : "") << segs[i].first << "-" << segs[i].second;
  return os.str();
}
}  // namespace

/// alignment/AlignmentScoringUtilImpl.hpp:35-155
MINE_EXPORT int mine_path_score(const int32_t* scores, const char* cigar, int isScoreOffEdge)
{
  return getPathScore(mk(scores), path(cigar), isScoreOffEdge != 0);
}
MINE_EXPORT int mine_max_path_score(const int32_t* scores, const char* cigar, int isScoreOffEdge, unsigned* readOff, unsigned* refOff)
{
  return getMaxPathScore(mk(scores), path(cigar), *readOff, *refOff, isScoreOffEdge != 0);
}

/// SVCandidateAssemblyRefiner.cpp:93-163
MINE_EXPORT int mine_is_low_quality_spanning(unsigned maxQCRefSpan, const int32_t* scores, int isLeadingPath, int isRNA, const char* cigar)
{
  return isLowQualitySpanningSVAlignment(maxQCRefSpan, mk(scores), isLeadingPath != 0, isRNA != 0, path(cigar)) ? 1 : 0;
}

/// :173-208 / :210-227 / :230-279
MINE_EXPORT int mine_large_indel_segments(const char* cigar, unsigned minSize, char* out, int cap)
{
  std::vector<std::pair<unsigned, unsigned>> segs;
  const ALIGNPATH::path_t                    p(path(cigar));
  getLargeIndelSegments(p, minSize, segs);
  std::ostringstream os;
  os << segText(segs) << " largest=" << getLargestIndelSize(p, segs);
  std::vector<std::pair<unsigned, unsigned>> ins;
  getLargestInsertSegment(p, minSize, ins);
  os << " largestInsert=" << segText(ins);
  return emit(os.str(), out, cap);
}

/// :318-388 (apath is modified in place: returned as cigar)
MINE_EXPORT int mine_is_low_quality_smallsv(
    unsigned maxQCRefSpan, const int32_t* scores, int isLeadingPath, int isComplex, const char* cigar, char* out, int cap)
{
  ALIGNPATH::path_t p(path(cigar));
  const bool        r = isLowQualitySmallSVAlignment(maxQCRefSpan, mk(scores), isLeadingPath != 0, isComplex != 0, p);
  emit(ALIGNPATH::apath_to_cigar(p), out, cap);
  return r ? 1 : 0;
}

/// :393-418
MINE_EXPORT int mine_query_seq_match_count(const char* target, const char* query, float maxMismatchRate)
{
  return getQuerySeqMatchCount(target, query, maxMismatchRate);
}

/// :430-553
MINE_EXPORT int mine_find_candidate_variants(
    unsigned maxQCRefSpan, const int32_t* scores, int beginPos, const char* cigar, const char* contig, const char* ref,
    unsigned minCandidateIndelSize, char* out, int cap)
{
  Alignment al;
  al.beginPos = beginPos;
  al.apath    = path(cigar);
  std::vector<std::pair<unsigned, unsigned>> segs;
  const bool r = findCandidateVariantsFromComplexSVContigAlignment(maxQCRefSpan, mk(scores), al, contig, ref, minCandidateIndelSize, segs);
  emit(segText(segs), out, cap);
  return r ? 1 : 0;
}

/// :563-665
MINE_EXPORT int mine_is_large_insert_alignment(const int32_t* scores, const char* cigar, int* candidateInsertInfo)
{
  LargeInsertionInfo       info;
  const bool               r = isLargeInsertAlignment(mk(scores), path(cigar), info);
  candidateInsertInfo[0] = info.isLeftCandidate;
  candidateInsertInfo[1] = info.isRightCandidate;
  candidateInsertInfo[2] = int(info.contigOffset);
  candidateInsertInfo[3] = int(info.refOffset);
  candidateInsertInfo[4] = info.score;
  return r ? 1 : 0;
}

/// :1254-1309
MINE_EXPORT int mine_is_low_quality_jump_alignment(
    const int32_t* scores, int begin1, const char* cigar1, int begin2, const char* cigar2, unsigned jumpInsertSize, int isRNA)
{
  JumpAlignmentResult<int> ja;
  ja.align1.beginPos = begin1;
  ja.align1.apath    = path(cigar1);
  ja.align2.beginPos = begin2;
  ja.align2.apath    = path(cigar2);
  ja.jumpInsertSize  = jumpInsertSize;
  return isLowQualityJumpAlignment(ja, mk(scores), isRNA != 0) ? 1 : 0;
}

/// alignment/AlignmentUtil.cpp:98-142
MINE_EXPORT int mine_extended_contig_single(int beginPos, const char* cigar, const char* query, const char* ref, char* out, int cap)
{
  AlignmentResult<int> a;
  a.align.beginPos = beginPos;
  a.align.apath    = path(cigar);
  std::string ext;
  getExtendedContig(a, query, ref, ext);
  return emit(ext, out, cap);
}
MINE_EXPORT int mine_extended_contig_jump(
    int begin1, const char* cigar1, int begin2, const char* cigar2, unsigned jumpInsertSize, const char* query, const char* ref1,
    const char* ref2, int isBp1Reversed, char* out, int cap)
{
  JumpAlignmentResult<int> ja;
  ja.align1.beginPos = begin1;
  ja.align1.apath    = path(cigar1);
  ja.align2.beginPos = begin2;
  ja.align2.apath    = path(cigar2);
  ja.jumpInsertSize  = jumpInsertSize;
  std::string ext, ins;
  getExtendedContig(ja, query, ref1, ref2, ext);
  getFwdStrandInsertSegment(ja, query, isBp1Reversed != 0, ins);
  return emit(ext + " insert=" + ins, out, cap);
}
