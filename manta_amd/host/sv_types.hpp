// The slice of Manta's SV object model that SVCandidateAssemblyRefiner::getCandidateAssemblyData reads and writes
// (SURVEY.md section 8b "primary boundary"): same type names, member names and meaning, so that the refiner of
// refiner.hpp reads like the reference's.  Only members the path touches exist; evidence counters, serialization and
// stream operators live on the far side of the boundary.  Paths relative to /root/reference/src/c++/lib.
#pragma once

#include <algorithm>
#include <set>
#include <string>
#include <vector>

#include "manta_amd.hpp"
#include "refiner_util.hpp"

namespace manta_amd {

/// svgraph/GenomeInterval.hpp:30-80
struct GenomeInterval {
  GenomeInterval(const int32_t initTid = 0, const pos_t beginPos = 0, const pos_t endPos = 0) : tid(initTid)
  {
    range.set_begin_pos(beginPos);
    range.set_end_pos(endPos);
  }
  bool isIntersect(const GenomeInterval& gi) const { return tid == gi.tid && range.is_range_intersect(gi.range); }
  int32_t          tid;
  known_pos_range2 range;
};

/// manta/SVBreakend.hpp:146-206
namespace SVBreakendState {
enum index_t { UNKNOWN, RIGHT_OPEN, LEFT_OPEN, COMPLEX };
inline const char* label(const index_t idx)
{
  static const char* names[] = {"UNKNOWN", "RIGHT_OPEN", "LEFT_OPEN", "COMPLEX"};
  return (unsigned(idx) < 4) ? names[idx] : "UNKNOWN";
}
inline bool isSimpleBreakend(const index_t idx) { return idx == RIGHT_OPEN || idx == LEFT_OPEN; }
inline bool isSameOrientation(const index_t a, const index_t b) { return isSimpleBreakend(a) && isSimpleBreakend(b) && a == b; }
inline bool isInnies(const bool isIdx1First, const index_t idx1, const index_t idx2)
{
  return isIdx1First ? (idx1 == RIGHT_OPEN && idx2 == LEFT_OPEN) : (idx2 == RIGHT_OPEN && idx1 == LEFT_OPEN);
}
inline bool isOutties(const bool isIdx1First, const index_t idx1, const index_t idx2) { return isInnies(!isIdx1First, idx1, idx2); }
}  // namespace SVBreakendState

struct SVBreakend {  // manta/SVBreakend.hpp:210-295
  /// low-resolution evidence counts that the candidate VCF reports (lowresEvidence PAIR / LOCAL_PAIR, :248-250); they are
  /// produced upstream of the refiner and only carried through it
  unsigned getPairCount() const { return pairCount; }
  unsigned getLocalPairCount() const { return localPairCount; }
  /// :280-291
  pos_t getLeftSideOfBkptAdjustment() const { return (state == SVBreakendState::LEFT_OPEN) ? -1 : 0; }

  SVBreakendState::index_t state = SVBreakendState::UNKNOWN;
  GenomeInterval           interval;
  unsigned                 pairCount = 0, localPairCount = 0;
};

/// manta/SVCandidate.hpp:33-190
struct SVCandidate {
  bool isImprecise() const { return _isImprecise; }
  void setPrecise() { _isImprecise = false; }
  bool isForward() const { return forwardTranscriptStrandReadCount > reverseTranscriptStrandReadCount; }
  bool isBreakendRangeSameShift() const { return bp1.state != bp2.state; }  // manta/SVCandidate.hpp:124

  SVBreakend        bp1, bp2;
  std::string       insertSeq;
  ALIGNPATH::path_t insertAlignment;
  std::string       contigSeq;
  bool              isUnknownSizeInsertion = false;
  std::string       unknownSizeInsertionLeftSeq, unknownSizeInsertionRightSeq;
  unsigned          candidateIndex = 0, assemblyAlignIndex = 0, assemblySegmentIndex = 0;
  unsigned          forwardTranscriptStrandReadCount = 0, reverseTranscriptStrandReadCount = 0;

private:
  bool _isImprecise = true;
};

/// manta/SVCandidateUtil.{hpp,cpp}
namespace SV_TYPE {
enum index_t { UNKNOWN, INTERTRANSLOC, INVERSION, INDEL, TANDUP };
}
inline SV_TYPE::index_t getSVType(const SVCandidate& sv)  // SVCandidateUtil.cpp:69-94
{
  using namespace SV_TYPE;
  if (sv.bp1.state == SVBreakendState::UNKNOWN || sv.bp2.state == SVBreakendState::UNKNOWN) return UNKNOWN;
  const bool isBp1First = sv.bp1.interval.range.begin_pos() <= sv.bp2.interval.range.begin_pos();
  const bool isBp2First = sv.bp2.interval.range.begin_pos() <= sv.bp1.interval.range.begin_pos();
  if (sv.bp1.interval.tid != sv.bp2.interval.tid) return INTERTRANSLOC;
  if (SVBreakendState::isSameOrientation(sv.bp1.state, sv.bp2.state)) return INVERSION;
  if (isBp1First || isBp2First) {
    if (SVBreakendState::isInnies(isBp1First, sv.bp1.state, sv.bp2.state)) return INDEL;
    if (SVBreakendState::isOutties(isBp1First, sv.bp1.state, sv.bp2.state)) return TANDUP;
  }
  return UNKNOWN;
}
namespace EXTENDED_SV_TYPE {
enum index_t { UNKNOWN, INTERTRANSLOC, INTRATRANSLOC, INVERSION, INSERT, DELETE, TANDUP };
inline bool        isSVTransloc(const index_t idx) { return idx == INTERTRANSLOC || idx == INTRATRANSLOC; }
inline bool        isSVIndel(const index_t idx) { return idx == INSERT || idx == DELETE; }
inline bool        isSVInv(const index_t idx) { return idx == INVERSION; }
inline const char* label(const index_t idx)  // SVCandidateUtil.hpp:109-128
{
  switch (idx) {
  case INTERTRANSLOC:
  case INTRATRANSLOC:
  case INVERSION: return "BND";
  case INSERT: return "INS";
  case DELETE: return "DEL";
  case TANDUP: return "DUP:TANDEM";
  default: return "UNKNOWN";
  }
}
}
inline EXTENDED_SV_TYPE::index_t getExtendedSVType(const SVCandidate& sv, const bool isForceIntraChromBnd = false)  // :96-136
{
  using namespace EXTENDED_SV_TYPE;
  const SV_TYPE::index_t svType(getSVType(sv));
  if (svType == SV_TYPE::INTERTRANSLOC) return INTERTRANSLOC;
  if (isForceIntraChromBnd) return INTRATRANSLOC;
  switch (svType) {
  case SV_TYPE::INVERSION: return INVERSION;
  case SV_TYPE::TANDUP: return TANDUP;
  case SV_TYPE::INDEL: {
    if (sv.isUnknownSizeInsertion) return INSERT;
    const bool        isBp1First = sv.bp1.interval.range.begin_pos() <= sv.bp2.interval.range.begin_pos();
    const SVBreakend& bpA(isBp1First ? sv.bp1 : sv.bp2);
    const SVBreakend& bpB(isBp1First ? sv.bp2 : sv.bp1);
    const unsigned    deleteSize = unsigned(bpB.interval.range.begin_pos() - bpA.interval.range.begin_pos());
    return (deleteSize >= unsigned(sv.insertSeq.size())) ? DELETE : INSERT;
  }
  default: return UNKNOWN;
  }
}
inline bool isSpanningSV(const SVCandidate& sv)  // SVCandidateUtil.hpp:135-139
{
  return SVBreakendState::isSimpleBreakend(sv.bp1.state) && SVBreakendState::isSimpleBreakend(sv.bp2.state);
}
inline bool isComplexSV(const SVCandidate& sv)  // :144-148
{
  return sv.bp1.state == SVBreakendState::COMPLEX && sv.bp2.state == SVBreakendState::UNKNOWN;
}

/// blt_util/reference_contig_segment.hpp:39-80
struct reference_contig_segment {
  std::string&       seq() { return _seq; }
  const std::string& seq() const { return _seq; }
  pos_t              get_offset() const { return _offset; }
  void               set_offset(const pos_t offset) { _offset = offset; }
  pos_t              end() const { return _offset + pos_t(_seq.size()); }
  void               clear()
  {
    _offset = 0;
    _seq.clear();
  }

private:
  pos_t       _offset = 0;
  std::string _seq;
};

/// htsapi/bam_header_info.hpp:60-110
struct bam_header_info {
  struct chrom_info {
    explicit chrom_info(const char* initLabel = nullptr, const unsigned initLength = 0) : label(initLabel ? initLabel : ""), length(initLength) {}
    std::string label;
    unsigned    length;
  };
  std::vector<chrom_info> chrom_data;
};

/// svgraph/GenomeIntervalTracker.hpp:32-63 over blt_util/RegionTracker.{hpp,cpp}: a set of disjoint, merged regions per
/// chromosome ordered by end position
struct GenomeIntervalTracker {
  void clear() { _regions.clear(); }
  void addInterval(const GenomeInterval& gi)
  {
    if (unsigned(gi.tid) >= _regions.size()) _regions.resize(size_t(gi.tid) + 1);
    std::vector<known_pos_range2>& regions(_regions[size_t(gi.tid)]);  // kept sorted; disjoint and non-adjacent
    known_pos_range2               merged(gi.range);
    std::vector<known_pos_range2>  kept;
    for (const known_pos_range2& r : regions) {
      // RegionTracker.cpp:51-69 merges regions that overlap OR touch [begin-1, end]
      if (r.end_pos() >= merged.begin_pos() && r.begin_pos() <= merged.end_pos()) {
        merged.set_begin_pos(std::min(merged.begin_pos(), r.begin_pos()));
        merged.set_end_pos(std::max(merged.end_pos(), r.end_pos()));
      } else {
        kept.push_back(r);
      }
    }
    kept.push_back(merged);
    std::sort(kept.begin(), kept.end(), [](const known_pos_range2& a, const known_pos_range2& b) { return a.end_pos() < b.end_pos(); });
    regions.swap(kept);
  }
  bool isSubsetOfRegion(const GenomeInterval& gi) const  // RegionTracker.cpp:38-49
  {
    if (gi.tid < 0 || unsigned(gi.tid) >= _regions.size()) return false;
    for (const known_pos_range2& r : _regions[size_t(gi.tid)]) {
      if (r.end_pos() > gi.range.begin_pos()) return (r.end_pos() >= gi.range.end_pos()) && (r.begin_pos() <= gi.range.begin_pos());
    }
    return false;
  }

private:
  std::vector<std::vector<known_pos_range2>> _regions;
};

/// options/ReadScannerOptions.hpp:68, options/SVRefinerOptions.hpp:36-93, .../GenerateSVCandidates/GSCOptions.hpp:39-110
struct ReadScannerOptions {
  unsigned minCandidateVariantSize = 10;
};
typedef IterativeAssemblerOptions AssemblerOptions;
struct SVRefinerOptions {
  SVRefinerOptions()
    : largeSVAlignScores(2, -8, -24, -1, -1), largeInsertEdgeAlignScores(2, -8, -18, -1, -1),
      largeInsertCompleteAlignScores(2, -8, -100, 0, -1), spanningAlignScores(2, -8, -12, -1, -1), largeGapOpenScore(-100),
      jumpScore(-100), contigFilterScores(2, -8, -18, 0, -1)
  {
    spanningAssembleOpt.minContigLength = 75;
  }
  AlignmentScores<int> largeSVAlignScores, largeInsertEdgeAlignScores, largeInsertCompleteAlignScores;
  AssemblerOptions     smallSVAssembleOpt;
  AlignmentScores<int> spanningAlignScores;
  const int            largeGapOpenScore, jumpScore;
  AssemblerOptions     spanningAssembleOpt;
  AlignmentScores<int> contigFilterScores;
};
struct GSCOptions {
  ReadScannerOptions scanOpt;
  SVRefinerOptions   refineOpt;
  std::string        referenceFilename;
  bool               enableRemoteReadRetrieval = false;
  bool               isRNA                     = false;  ///< the RNA (intron-aware) spanning path is not on the GPU path: refused
  bool               isOutputContig            = false;
};

/// manta/SVCandidateAssemblyData.hpp:39-182
struct BPOrientation {
  void clear() { *this = BPOrientation(); }
  bool isBp2AlignedFirst = false, isBp1Reversed = false, isBp2Reversed = false, isBp1First = true, isTranscriptStrandKnown = false;
};
struct SVCandidateAssemblyData {
  void clear() { *this = SVCandidateAssemblyData(); }
  typedef AlignmentResult<int>              SmallAlignmentResultType;
  typedef JumpAlignmentResult<int>          JumpAlignmentResultType;
  typedef std::pair<unsigned, unsigned>     CandidateSegmentType;
  typedef std::vector<CandidateSegmentType> CandidateSegmentSetType;

  Assembly                              contigs;
  bool                                  isCandidateSpanning = false, isSpanning = false;
  BPOrientation                         bporient;
  std::vector<std::string>              extendedContigs;
  std::vector<SmallAlignmentResultType> smallSVAlignments;
  std::vector<JumpAlignmentResultType>  spanningAlignments;
  std::vector<CandidateSegmentSetType>  smallSVSegments;
  std::vector<LargeInsertionInfo>       largeInsertInfo;
  unsigned                              bestAlignmentIndex = 0;
  reference_contig_segment              bp1ref, bp2ref;
  std::vector<SVCandidate>              svs;
  bool                                  isOverlapSkip = false;
};

}  // namespace manta_amd
