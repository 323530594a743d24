// Host-side glue of the hot path: the result-only helper logic of SVCandidateAssemblyRefiner (SURVEY.md section 8a,
// last rows), restated over the adapter types of manta_amd.hpp.  It touches nothing but alignment results and
// strings -- no BAM/FASTA I/O -- and is what a refiner built on the batched ABI runs between the GPU stages and the
// reference's own candidate bookkeeping.  Every function names the reference lines it reproduces (paths relative to
// /root/reference/src/c++/lib); float comparisons use `float` exactly as the reference does.
// Pinned by tests/test_refiner_util.py against the UNMODIFIED reference statics (oracle/ref_refiner_driver.cpp).
#pragma once

#include <algorithm>
#include <string>
#include <utility>
#include <vector>

#include "manta_amd.hpp"

namespace manta_amd {

// ------------------------------------------------------------------------------------------------------
// ALIGNPATH utilities (blt_util/align_path.{hpp,cpp})
// ------------------------------------------------------------------------------------------------------
namespace ALIGNPATH {
inline bool is_segment_type_read_length(const align_t t)
{
  return t == MATCH || t == INSERT || t == SOFT_CLIP || t == SEQ_MATCH || t == SEQ_MISMATCH;
}
inline bool is_segment_type_ref_length(const align_t t)
{
  return t == MATCH || t == DELETE || t == SKIP || t == SEQ_MATCH || t == SEQ_MISMATCH;
}
inline bool is_segment_align_match(const align_t t) { return t == MATCH || t == SEQ_MATCH || t == SEQ_MISMATCH; }

inline unsigned apath_read_length(const path_t& apath)  // align_path.cpp:97-105
{
  unsigned n = 0;
  for (const path_segment& ps : apath)
    if (is_segment_type_read_length(ps.type)) n += ps.length;
  return n;
}
inline unsigned apath_ref_length(const path_t& apath)  // :127-135
{
  unsigned n = 0;
  for (const path_segment& ps : apath)
    if (is_segment_type_ref_length(ps.type)) n += ps.length;
  return n;
}
inline unsigned apath_spliced_length(const path_t& apath)  // :117-124
{
  unsigned n = 0;
  for (const path_segment& ps : apath)
    if (ps.type == SKIP) n += ps.length;
  return n;
}
inline unsigned apath_soft_clip_right_size(const path_t& apath)  // :184-198
{
  unsigned n = 0;
  for (auto it = apath.rbegin(); it != apath.rend(); ++it) {
    if (it->type == HARD_CLIP) continue;
    if (it->type != SOFT_CLIP) break;
    n += it->length;
  }
  return n;
}
/// keep only the prefix of the path that covers the first target_ref_length reference bases (:273-293)
inline void apath_limit_ref_length(const unsigned target_ref_length, path_t& apath)
{
  unsigned covered = 0;
  for (size_t i = 0; i < apath.size(); ++i) {
    if (!is_segment_type_ref_length(apath[i].type)) continue;
    covered += apath[i].length;
    if (covered < target_ref_length) continue;
    apath[i].length -= (covered - target_ref_length);
    apath.resize(i + 1);
    return;
  }
}
/// cigar text -> path, merging equal neighbours and dropping pads / zero lengths like cigar_to_apath (:63-95)
inline path_t cigar_to_apath(const std::string& cigar)
{
  path_t       path;
  path_segment last;
  size_t       i = 0;
  while (i < cigar.size()) {
    unsigned len = 0;
    while (i < cigar.size() && cigar[i] >= '0' && cigar[i] <= '9') len = len * 10 + unsigned(cigar[i++] - '0');
    const char c = cigar[i++];
    align_t    t = NONE;
    switch (c) {
    case 'M': t = MATCH; break;
    case 'I': t = INSERT; break;
    case 'D': t = DELETE; break;
    case 'N': t = SKIP; break;
    case 'S': t = SOFT_CLIP; break;
    case 'H': t = HARD_CLIP; break;
    case 'P': t = PAD; break;
    case '=': t = SEQ_MATCH; break;
    case 'X': t = SEQ_MISMATCH; break;
    default: throw GeneralException(std::string("Can't parse cigar string: ") + cigar);
    }
    if (t == PAD || len == 0) continue;
    if (t == last.type) {
      last.length += len;
    } else {
      if (last.type != NONE) path.push_back(last);
      last = path_segment(t, len);
    }
  }
  if (last.type != NONE) path.push_back(last);
  return path;
}
}  // namespace ALIGNPATH

// ------------------------------------------------------------------------------------------------------
// re-scoring of (sub)paths  (alignment/AlignmentScoringUtilImpl.hpp:35-155)
// The reference knowingly keeps a quirk: its "was the previous segment an indel" flag is re-initialised for every
// segment, so an adjacent insertion+deletion pays the gap-open twice.  Reproduced.
// ------------------------------------------------------------------------------------------------------
struct PathScoreWalk {
  int      val = 0, maxVal = 0;
  unsigned readOffset = 0, refOffset = 0, maxReadOffset = 0, maxRefOffset = 0;
};
inline PathScoreWalk walkPathScore(const AlignmentScores<int>& scores, const ALIGNPATH::path_t& apath, const bool isScoreOffEdge)
{
  using namespace ALIGNPATH;
  PathScoreWalk w;
  for (const path_segment& ps : apath) {
    switch (ps.type) {
    case SEQ_MATCH:
      w.val += scores.match * int(ps.length);
      w.readOffset += ps.length;
      w.refOffset += ps.length;
      break;
    case SEQ_MISMATCH:
      w.val += scores.mismatch * int(ps.length);
      w.readOffset += ps.length;
      w.refOffset += ps.length;
      break;
    case INSERT:
      w.val += scores.open + scores.extend * int(ps.length);
      w.readOffset += ps.length;
      break;
    case DELETE:
      w.val += scores.open + scores.extend * int(ps.length);
      w.refOffset += ps.length;
      break;
    case SOFT_CLIP:
      if (isScoreOffEdge) w.val += scores.offEdge * int(ps.length);
      w.readOffset += ps.length;
      break;
    default:
      break;
    }
    if (w.val > w.maxVal) {
      w.maxVal        = w.val;
      w.maxReadOffset = w.readOffset;
      w.maxRefOffset  = w.refOffset;
    }
  }
  return w;
}
/// NB the defaults differ in the reference (AlignmentScoringUtil.hpp:37,50): off-edge is NOT scored by getPathScore
/// unless asked, but IS scored by getMaxPathScore
inline int getPathScore(const AlignmentScores<int>& scores, const ALIGNPATH::path_t& apath, const bool isScoreOffEdge = false)
{
  return walkPathScore(scores, apath, isScoreOffEdge).val;
}
inline int getMaxPathScore(
    const AlignmentScores<int>& scores, const ALIGNPATH::path_t& apath, unsigned& maxReadOffset, unsigned& maxRefOffset,
    const bool isScoreOffEdge = true)
{
  const PathScoreWalk w = walkPathScore(scores, apath, isScoreOffEdge);
  maxReadOffset         = w.maxReadOffset;
  maxRefOffset          = w.maxRefOffset;
  return w.maxVal;
}

// ------------------------------------------------------------------------------------------------------
// contig / reference stitching  (alignment/AlignmentUtil.cpp:62-142)
// ------------------------------------------------------------------------------------------------------
inline void reverseCompStr(std::string& s)  // blt_util/seq_util.hpp:150-197 (comp_base: ACGT <-> TGCA, N stays, rest -> 'N')
{
  std::reverse(s.begin(), s.end());
  for (char& c : s) {
    switch (c) {
    case 'A': c = 'T'; break;
    case 'C': c = 'G'; break;
    case 'G': c = 'C'; break;
    case 'T': c = 'A'; break;
    case 'N': c = 'N'; break;
    default: c = 'N'; break;
    }
  }
}
inline void getExtendedContig(
    const AlignmentResult<int>& alignment, const std::string& querySeq, const std::string& refSeq, std::string& extendedContig)
{
  const unsigned refEnd = unsigned(alignment.align.beginPos) + ALIGNPATH::apath_ref_length(alignment.align.apath);
  extendedContig        = refSeq.substr(0, alignment.align.beginPos) + querySeq + refSeq.substr(refEnd);
}
inline void getExtendedContig(
    const JumpAlignmentResult<int>& align, const std::string& querySeq, const std::string& ref1Seq, const std::string& ref2Seq,
    std::string& extendedContig)
{
  const unsigned ref2End = unsigned(align.align2.beginPos) + ALIGNPATH::apath_ref_length(align.align2.apath);
  extendedContig         = ref1Seq.substr(0, align.align1.beginPos) + querySeq + ref2Seq.substr(ref2End);
}
inline void getFwdStrandInsertSegment(
    const JumpAlignmentResult<int>& align, const std::string& querySeq, const bool isBp1Reversed, std::string& insertSeq)
{
  insertSeq = querySeq.substr(ALIGNPATH::apath_read_length(align.align1.apath), align.jumpInsertSize);
  if (isBp1Reversed) reverseCompStr(insertSeq);
}

// ------------------------------------------------------------------------------------------------------
// refiner QC / candidate discovery  (applications/GenerateSVCandidates/SVCandidateAssemblyRefiner.cpp)
// ------------------------------------------------------------------------------------------------------
typedef std::pair<unsigned, unsigned> segment_t;  ///< [first,last] indices into a path

/// flank quality shared by the spanning and the small-SV tests: orient the flank away from the breakend, keep its
/// first maxQCRefSpan reference bases, then require a minimum unclipped read length and >= 75 % of the perfect score
/// (:93-163 and :318-388; minRefSpan == 0 disables the reference-span test of the small-SV flavour)
inline bool isLowQualityFlank(
    const unsigned maxQCRefSpan, const AlignmentScores<int>& scores, const bool isLeadingPath, const unsigned minAlignRefSpan,
    const unsigned minAlignReadLength, ALIGNPATH::path_t& apath)
{
  static const float minScoreFrac(0.75);
  if (isLeadingPath) std::reverse(apath.begin(), apath.end());
  ALIGNPATH::apath_limit_ref_length(maxQCRefSpan, apath);
  if (minAlignRefSpan && ALIGNPATH::apath_ref_length(apath) < minAlignRefSpan) return true;
  const unsigned readSize    = ALIGNPATH::apath_read_length(apath);
  const unsigned clippedSize = readSize - ALIGNPATH::apath_soft_clip_right_size(apath);
  if (clippedSize < minAlignReadLength) return true;
  const int   nonClipScore = std::max(0, getPathScore(scores, apath));
  const int   optimalScore = int(clippedSize) * scores.match;
  const float scoreFrac    = static_cast<float>(nonClipScore) / static_cast<float>(optimalScore);
  return scoreFrac < minScoreFrac;
}

/// :93-163
inline bool isLowQualitySpanningSVAlignment(
    const unsigned maxQCRefSpan, const AlignmentScores<int>& scores, const bool isLeadingPath, const bool isRNA,
    const ALIGNPATH::path_t& input_apath)
{
  ALIGNPATH::path_t apath(input_apath);
  return isLowQualityFlank(maxQCRefSpan, scores, isLeadingPath, 0, isRNA ? 20u : 30u, apath);
}

/// :318-388 (the path is modified in place, as in the reference)
inline bool isLowQualitySmallSVAlignment(
    const unsigned maxQCRefSpan, const AlignmentScores<int>& scores, const bool isLeadingPath, const bool isComplex,
    ALIGNPATH::path_t& apath)
{
  const unsigned minSpan = isComplex ? 35u : 30u;
  return isLowQualityFlank(maxQCRefSpan, scores, isLeadingPath, minSpan, minSpan, apath);
}

/// runs of adjacent insert/delete segments that contain at least one indel >= minSize (:173-208)
inline void getLargeIndelSegments(const ALIGNPATH::path_t& apath, const unsigned minSize, std::vector<segment_t>& segments)
{
  segments.clear();
  const unsigned n = unsigned(apath.size());
  unsigned       i = 0;
  while (i < n) {
    if (apath[i].type != ALIGNPATH::INSERT && apath[i].type != ALIGNPATH::DELETE) {
      ++i;
      continue;
    }
    unsigned j   = i;
    bool     big = false;
    while (j < n && (apath[j].type == ALIGNPATH::INSERT || apath[j].type == ALIGNPATH::DELETE)) {
      big = big || (apath[j].length >= minSize);
      ++j;
    }
    if (big) segments.push_back(segment_t(i, j - 1));
    i = j;
  }
}
inline unsigned getLargestIndelSize(const ALIGNPATH::path_t& apath, const std::vector<segment_t>& segments)  // :210-227
{
  unsigned largest = 0;
  for (const segment_t& seg : segments)
    for (unsigned i = seg.first; i <= seg.second; ++i)
      if (apath[i].type == ALIGNPATH::INSERT || apath[i].type == ALIGNPATH::DELETE) largest = std::max(largest, apath[i].length);
  return largest;
}
/// the indel run holding the LAST insertion that is >= every earlier qualifying insertion and >= minSize (:230-279)
inline void getLargestInsertSegment(const ALIGNPATH::path_t& apath, const unsigned minSize, std::vector<segment_t>& segments)
{
  segments.clear();
  const unsigned n = unsigned(apath.size());
  unsigned       maxSize = minSize;
  bool           found   = false;
  segment_t      best;
  unsigned       i = 0;
  while (i < n) {
    if (apath[i].type != ALIGNPATH::INSERT && apath[i].type != ALIGNPATH::DELETE) {
      ++i;
      continue;
    }
    unsigned j    = i;
    bool     cand = false;
    while (j < n && (apath[j].type == ALIGNPATH::INSERT || apath[j].type == ALIGNPATH::DELETE)) {
      if (apath[j].type == ALIGNPATH::INSERT && apath[j].length >= maxSize) {
        maxSize = apath[j].length;
        cand    = true;
        found   = true;
      }
      ++j;
    }
    if (cand) best = segment_t(i, j - 1);
    i = j;
  }
  if (found) segments.push_back(best);
}

/// number of placements of querySeq in targetSeq with mismatch rate <= maxMismatchRate; 'N' in the query always
/// mismatches (:393-418)
inline int getQuerySeqMatchCount(const std::string& targetSeq, const std::string& querySeq, const float maxMismatchRate)
{
  const unsigned querySize = unsigned(querySeq.size()), targetSize = unsigned(targetSeq.size());
  if (querySize > targetSize) return 0;
  // smallest mismatch count that already fails the reference's float test `mismatches/querySize <= maxMismatchRate`
  // (monotone in the count): a placement is abandoned as soon as it gets there -- same verdicts, far fewer compares
  unsigned failCount = 0;
  while (failCount <= querySize && float(failCount) / float(querySize) <= maxMismatchRate) ++failCount;
  const char* q = querySeq.data();
  unsigned    hits = 0;
  for (unsigned i = 0; i + querySize <= targetSize; ++i) {
    const char* t = targetSeq.data() + i;
    unsigned    mismatches = 0;
    for (unsigned j = 0; j < querySize; ++j) {
      if (q[j] != t[j] || q[j] == 'N') {
        if (++mismatches >= failCount) break;
      }
    }
    if (mismatches < failCount) ++hits;
  }
  return int(hits);
}

/// :430-553
inline bool findCandidateVariantsFromComplexSVContigAlignment(
    const unsigned maxQCRefSpan, const AlignmentScores<int>& scores, const Alignment& align, const std::string& contigSeq,
    const std::string& refSeq, const unsigned minCandidateIndelSize, std::vector<segment_t>& candidateSegments)
{
  using namespace ALIGNPATH;
  const path_t& apath(align.apath);
  getLargeIndelSegments(apath, minCandidateIndelSize, candidateSegments);
  if (candidateSegments.empty()) return false;
  const bool isComplex = (candidateSegments.size() > 1) || (candidateSegments[0].first != candidateSegments[0].second);

  // drop candidates from the left until the flank before the first one is clean, then from the right
  while (true) {
    path_t leading(apath.begin(), apath.begin() + candidateSegments.front().first);
    if (!isLowQualitySmallSVAlignment(maxQCRefSpan, scores, true, isComplex, leading)) break;
    if (candidateSegments.size() == 1) return false;
    candidateSegments.erase(candidateSegments.begin());
  }
  while (true) {
    path_t trailing(apath.begin() + candidateSegments.back().second + 1, apath.end());
    if (!isLowQualitySmallSVAlignment(maxQCRefSpan, scores, false, isComplex, trailing)) break;
    if (candidateSegments.size() == 1) return false;
    candidateSegments.pop_back();
  }

  // ambiguity filter: either contig flank placing more than once (<= 5 % mismatches) inside a 500 bp window
  {
    const path_t tillStart(apath.begin(), apath.begin() + candidateSegments.front().first);
    const path_t tillEnd(apath.begin(), apath.begin() + candidateSegments.back().second + 1);
    const int    leftSize  = int(apath_read_length(tillStart));
    const int    endPos    = int(apath_read_length(tillEnd));
    const int    rightSize = int(contigSeq.length()) - endPos;
    const std::string leftContig  = contigSeq.substr(0, leftSize);
    const std::string rightContig = contigSeq.substr(endPos, rightSize);
    const int   searchWindow(500);
    const float mismatchRate(0.05f);
    const int   refAlignStart = align.beginPos;
    const int   refAlignEnd   = align.beginPos + int(apath_ref_length(apath));
    const int   leftSearchStart = std::max(0, refAlignEnd - searchWindow);
    if (getQuerySeqMatchCount(refSeq.substr(leftSearchStart, refAlignEnd - leftSearchStart), leftContig, mismatchRate) > 1) return false;
    const int rightSearchSize = std::min(searchWindow, int(refSeq.length()) - refAlignStart);
    if (getQuerySeqMatchCount(refSeq.substr(refAlignStart, rightSearchSize), rightContig, mismatchRate) > 1) return false;
  }

  // keep only runs that still hold an indel of the minimum size
  std::vector<segment_t> kept;
  for (const segment_t& seg : candidateSegments) {
    for (unsigned i = seg.first; i <= seg.second; ++i) {
      if ((apath[i].type == INSERT || apath[i].type == DELETE) && apath[i].length >= minCandidateIndelSize) {
        kept.push_back(seg);
        break;
      }
    }
  }
  candidateSegments.swap(kept);
  return !candidateSegments.empty();
}

/// the two QC spans of getSmallSVAssembly (:2046-2066): larger segment list wins
inline bool findSmallSVCandidateSegments(
    const AlignmentScores<int>& contigFilterScores, const Alignment& align, const std::string& contigSeq, const std::string& refSeq,
    const unsigned minCandidateVariantSize, std::vector<segment_t>& candidateSegments)
{
  candidateSegments.clear();
  bool isCandidate = false;
  for (const unsigned maxQCRefSpan : {100u, 200u}) {
    std::vector<segment_t> segments;
    if (findCandidateVariantsFromComplexSVContigAlignment(maxQCRefSpan, contigFilterScores, align, contigSeq, refSeq,
                                                          minCandidateVariantSize, segments)) {
      if (segments.size() > candidateSegments.size()) candidateSegments = segments;
      isCandidate = true;
    }
  }
  return isCandidate;
}

/// manta/SVCandidateAssemblyData.hpp:60-78
struct LargeInsertionInfo {
  bool     isLeftCandidate = false, isRightCandidate = false;
  unsigned contigOffset = 0, refOffset = 0;
  int      score = 0;
};

/// :563-608
inline bool isLargeInsertSegment(
    const AlignmentScores<int>& scores, const ALIGNPATH::path_t& apath, unsigned& contigOffset, unsigned& refOffset, int& score,
    const unsigned trimInsertLength = 0)
{
  static const unsigned minAlignReadLength(40), minExtendedReadLength(40), minAlignRefSpan(40);
  static const float    minScoreFrac(0.75);
  const unsigned        pathSize = ALIGNPATH::apath_read_length(apath);
  score = std::max(0, getMaxPathScore(scores, apath, contigOffset, refOffset));
  if (refOffset < minAlignRefSpan) return false;
  if (contigOffset < minAlignReadLength) return false;
  if ((pathSize - contigOffset) < (minExtendedReadLength + trimInsertLength)) return false;
  const int   optimalScore = int(contigOffset) * scores.match;
  const float scoreFrac    = static_cast<float>(score) / static_cast<float>(optimalScore);
  return !(scoreFrac < minScoreFrac);
}
/// :611-639
inline bool isLargeInsertAlignment(const AlignmentScores<int>& scores, const ALIGNPATH::path_t& apath, LargeInsertionInfo& info)
{
  info.isLeftCandidate = isLargeInsertSegment(scores, apath, info.contigOffset, info.refOffset, info.score);
  if (info.isLeftCandidate) return true;
  ALIGNPATH::path_t rev(apath);
  std::reverse(rev.begin(), rev.end());
  info.isRightCandidate = isLargeInsertSegment(scores, rev, info.contigOffset, info.refOffset, info.score);
  if (info.isRightCandidate) {
    info.contigOffset = ALIGNPATH::apath_read_length(apath) - info.contigOffset;
    info.refOffset    = ALIGNPATH::apath_ref_length(apath) - info.refOffset;
    return true;
  }
  return false;
}

/// :1254-1271
inline bool isJumpAlignmentQCFail(const JumpAlignmentResult<int>& ja)
{
  auto bad = [](const Alignment& a) { return !a.isAligned() || ALIGNPATH::apath_ref_length(a.apath) < 20u; };
  return bad(ja.align1) || bad(ja.align2);
}
/// :1287-1309
inline bool isLowQualityJumpAlignment(const JumpAlignmentResult<int>& ja, const AlignmentScores<int>& scores, const bool isRNA)
{
  bool           low1 = true, low2 = true;
  const unsigned dna[] = {75, 100, 200}, rna[] = {36, 75, 100};
  for (int i = 0; i < 3; ++i) {
    const unsigned span = isRNA ? rna[i] : dna[i];
    const unsigned s1   = span + (isRNA ? ALIGNPATH::apath_spliced_length(ja.align1.apath) : 0u);
    const unsigned s2   = span + (isRNA ? ALIGNPATH::apath_spliced_length(ja.align2.apath) : 0u);
    if (!isLowQualitySpanningSVAlignment(s1, scores, true, isRNA, ja.align1.apath)) low1 = false;
    if (!isLowQualitySpanningSVAlignment(s2, scores, false, isRNA, ja.align2.apath)) low2 = false;
  }
  return low1 || low2;
}
/// selectJumpContigDNA (:1364-1398): index of the usable best contig alignment, or -1
inline int selectJumpContigDNA(const std::vector<JumpAlignmentResult<int>>& alignments, const AlignmentScores<int>& scores)
{
  int best = -1;
  for (size_t i = 0; i < alignments.size(); ++i) {
    if (isJumpAlignmentQCFail(alignments[i])) continue;
    if (best == -1 || alignments[i].score > alignments[size_t(best)].score) best = int(i);
  }
  if (best == -1 || isLowQualityJumpAlignment(alignments[size_t(best)], scores, false)) return -1;
  return best;
}

}  // namespace manta_amd
