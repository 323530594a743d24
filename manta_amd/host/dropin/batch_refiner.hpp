// The batched refiner call over Manta's OWN types.
//
// The reference's interface is one candidate at a time (SVCandidateAssemblyRefiner::getCandidateAssemblyData,
// applications/GenerateSVCandidates/SVCandidateAssemblyRefiner.hpp:41-99; called per candidate from SVCandidateProcessor.cpp:304-346 inside
// the edge loop of GenerateSVCandidates.cpp:148-208).  With the shadow headers of this directory that call already runs its arithmetic on
// the device -- one alignment, one assembly per launch.  Throughput needs the whole list of an edge (or of many edges) in ONE device batch:
//
//   manta_amd_dropin::BatchRefiner refiner(opt, header, source);
//   std::vector<SVCandidateAssemblyData> data;                    // Manta's type
//   refiner.getCandidateAssemblyDataBatch(svs, isFindLargeInsertions, data);   // std::vector<SVCandidate>, Manta's type
//
// `source` answers the two questions the reference asks the outside world per candidate: reference bases of a region
// (get_standardized_region_seq, htsapi/samtools_fasta_util.hpp:52-57) and the assembly reads of a breakend
// (SVCandidateAssembler::getBreakendReads, manta/SVCandidateAssembler.cpp:271-659) -- with Manta's SVBreakend /
// reference_contig_segment / AssemblyReadInput.  Inside, the candidates are handed to manta_amd::SVCandidateAssemblyRefiner
// (host/refiner.hpp: plan -> pack -> device batch -> per-candidate glue) and the results are written back into Manta's objects; a
// refined SVCandidate starts as a copy of the caller's candidate, so everything the refiner does not touch (low-resolution evidence,
// filters, ...) is carried through as the reference's `SVCandidate newSV(sv)` does.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "applications/GenerateSVCandidates/GSCOptions.hpp"
#include "assembly/AssemblyReadInfo.hpp"
#include "htsapi/bam_header_info.hpp"
#include "manta/SVCandidate.hpp"
#include "manta/SVCandidateAssemblyData.hpp"

#include "../refiner.hpp"

namespace manta_amd_dropin {

/// the refiner's two input seams, over Manta's types
struct BatchInputSource {
  virtual ~BatchInputSource() {}
  virtual void getReferenceSeq(const std::string& chrom, int beginPos, int endPos, std::string& seq) = 0;
  virtual void getBreakendReads(const ::SVBreakend& bp, bool isReversed, const ::reference_contig_segment& refSeq, ::AssemblyReadInput& reads) = 0;
};

namespace detail {

inline manta_amd::SVBreakendState::index_t toMirror(const ::SVBreakendState::index_t s)
{
  switch (s) {
  case ::SVBreakendState::RIGHT_OPEN: return manta_amd::SVBreakendState::RIGHT_OPEN;
  case ::SVBreakendState::LEFT_OPEN: return manta_amd::SVBreakendState::LEFT_OPEN;
  case ::SVBreakendState::COMPLEX: return manta_amd::SVBreakendState::COMPLEX;
  default: return manta_amd::SVBreakendState::UNKNOWN;
  }
}
inline ::SVBreakendState::index_t toReal(const manta_amd::SVBreakendState::index_t s)
{
  switch (s) {
  case manta_amd::SVBreakendState::RIGHT_OPEN: return ::SVBreakendState::RIGHT_OPEN;
  case manta_amd::SVBreakendState::LEFT_OPEN: return ::SVBreakendState::LEFT_OPEN;
  case manta_amd::SVBreakendState::COMPLEX: return ::SVBreakendState::COMPLEX;
  default: return ::SVBreakendState::UNKNOWN;
  }
}
inline void toMirror(const ::SVBreakend& in, manta_amd::SVBreakend& out)
{
  out.state          = toMirror(in.state);
  out.interval       = manta_amd::GenomeInterval(in.interval.tid, in.interval.range.begin_pos(), in.interval.range.end_pos());
  out.pairCount      = in.getPairCount();
  out.localPairCount = in.getLocalPairCount();
}
inline void toReal(const manta_amd::SVBreakend& in, ::SVBreakend& out)  // (evidence counts stay the caller's)
{
  out.state        = toReal(in.state);
  out.interval.tid = in.interval.tid;
  out.interval.range.set_range(in.interval.range.begin_pos(), in.interval.range.end_pos());
}
inline ::ALIGNPATH::align_t toReal(const manta_amd::ALIGNPATH::align_t t)
{
  switch (t) {
  case manta_amd::ALIGNPATH::MATCH: return ::ALIGNPATH::MATCH;
  case manta_amd::ALIGNPATH::INSERT: return ::ALIGNPATH::INSERT;
  case manta_amd::ALIGNPATH::DELETE: return ::ALIGNPATH::DELETE;
  case manta_amd::ALIGNPATH::SKIP: return ::ALIGNPATH::SKIP;
  case manta_amd::ALIGNPATH::SOFT_CLIP: return ::ALIGNPATH::SOFT_CLIP;
  case manta_amd::ALIGNPATH::HARD_CLIP: return ::ALIGNPATH::HARD_CLIP;
  case manta_amd::ALIGNPATH::PAD: return ::ALIGNPATH::PAD;
  case manta_amd::ALIGNPATH::SEQ_MATCH: return ::ALIGNPATH::SEQ_MATCH;
  case manta_amd::ALIGNPATH::SEQ_MISMATCH: return ::ALIGNPATH::SEQ_MISMATCH;
  default: return ::ALIGNPATH::NONE;
  }
}
inline void toReal(const manta_amd::ALIGNPATH::path_t& in, ::ALIGNPATH::path_t& out)
{
  out.clear();
  for (const manta_amd::ALIGNPATH::path_segment& ps : in) out.push_back(::ALIGNPATH::path_segment(toReal(ps.type), ps.length));
}
inline void toMirror(const ::ALIGNPATH::path_t& in, manta_amd::ALIGNPATH::path_t& out)
{
  out.clear();
  for (const ::ALIGNPATH::path_segment& ps : in) {
    manta_amd::ALIGNPATH::align_t t = manta_amd::ALIGNPATH::NONE;
    switch (ps.type) {
    case ::ALIGNPATH::MATCH: t = manta_amd::ALIGNPATH::MATCH; break;
    case ::ALIGNPATH::INSERT: t = manta_amd::ALIGNPATH::INSERT; break;
    case ::ALIGNPATH::DELETE: t = manta_amd::ALIGNPATH::DELETE; break;
    case ::ALIGNPATH::SKIP: t = manta_amd::ALIGNPATH::SKIP; break;
    case ::ALIGNPATH::SOFT_CLIP: t = manta_amd::ALIGNPATH::SOFT_CLIP; break;
    case ::ALIGNPATH::HARD_CLIP: t = manta_amd::ALIGNPATH::HARD_CLIP; break;
    case ::ALIGNPATH::PAD: t = manta_amd::ALIGNPATH::PAD; break;
    case ::ALIGNPATH::SEQ_MATCH: t = manta_amd::ALIGNPATH::SEQ_MATCH; break;
    case ::ALIGNPATH::SEQ_MISMATCH: t = manta_amd::ALIGNPATH::SEQ_MISMATCH; break;
    default: break;
    }
    out.push_back(manta_amd::ALIGNPATH::path_segment(t, ps.length));
  }
}
inline void toMirror(const ::SVCandidate& in, manta_amd::SVCandidate& out)
{
  out = manta_amd::SVCandidate();
  if (!in.isImprecise()) out.setPrecise();
  toMirror(in.bp1, out.bp1);
  toMirror(in.bp2, out.bp2);
  out.insertSeq = in.insertSeq;
  toMirror(in.insertAlignment, out.insertAlignment);
  out.contigSeq                        = in.contigSeq;
  out.isUnknownSizeInsertion           = in.isUnknownSizeInsertion;
  out.unknownSizeInsertionLeftSeq      = in.unknownSizeInsertionLeftSeq;
  out.unknownSizeInsertionRightSeq     = in.unknownSizeInsertionRightSeq;
  out.candidateIndex                   = in.candidateIndex;
  out.assemblyAlignIndex               = in.assemblyAlignIndex;
  out.assemblySegmentIndex             = in.assemblySegmentIndex;
  out.forwardTranscriptStrandReadCount = in.forwardTranscriptStrandReadCount;
  out.reverseTranscriptStrandReadCount = in.reverseTranscriptStrandReadCount;
}
/// `out` starts as the caller's candidate (the reference's `SVCandidate newSV(sv)`): what the refiner computes is written over it
inline void toReal(const manta_amd::SVCandidate& in, const ::SVCandidate& original, ::SVCandidate& out)
{
  out = original;
  if (!in.isImprecise()) out.setPrecise();
  toReal(in.bp1, out.bp1);
  toReal(in.bp2, out.bp2);
  out.insertSeq = in.insertSeq;
  toReal(in.insertAlignment, out.insertAlignment);
  out.contigSeq                    = in.contigSeq;
  out.isUnknownSizeInsertion       = in.isUnknownSizeInsertion;
  out.unknownSizeInsertionLeftSeq  = in.unknownSizeInsertionLeftSeq;
  out.unknownSizeInsertionRightSeq = in.unknownSizeInsertionRightSeq;
  out.candidateIndex               = in.candidateIndex;
  out.assemblyAlignIndex           = in.assemblyAlignIndex;
  out.assemblySegmentIndex         = in.assemblySegmentIndex;
}
inline void toReal(const manta_amd::Alignment& in, ::Alignment& out)
{
  out.beginPos = in.beginPos;
  toReal(in.apath, out.apath);
}
inline void toReal(const manta_amd::reference_contig_segment& in, ::reference_contig_segment& out)
{
  out.set_offset(in.get_offset());
  out.seq() = in.seq();
}
inline void toReal(const manta_amd::SVCandidateAssemblyData& in, const ::SVCandidate& original, ::SVCandidateAssemblyData& out)
{
  out.clear();
  out.contigs.resize(in.contigs.size());
  for (size_t i = 0; i < in.contigs.size(); ++i) {
    const manta_amd::AssembledContig& c(in.contigs[i]);
    ::AssembledContig&                o(out.contigs[i]);
    o.seq           = c.seq;
    o.seedReadCount = c.seedReadCount;
    o.supportReads  = c.supportReads;
    o.rejectReads   = c.rejectReads;
    o.conservativeRange.set_range(c.conservativeRange.begin_pos(), c.conservativeRange.end_pos());
  }
  out.isCandidateSpanning              = in.isCandidateSpanning;
  out.isSpanning                       = in.isSpanning;
  out.bporient.isBp2AlignedFirst       = in.bporient.isBp2AlignedFirst;
  out.bporient.isBp1Reversed           = in.bporient.isBp1Reversed;
  out.bporient.isBp2Reversed           = in.bporient.isBp2Reversed;
  out.bporient.isBp1First              = in.bporient.isBp1First;
  out.bporient.isTranscriptStrandKnown = in.bporient.isTranscriptStrandKnown;
  out.extendedContigs                  = in.extendedContigs;
  out.smallSVAlignments.resize(in.smallSVAlignments.size());
  for (size_t i = 0; i < in.smallSVAlignments.size(); ++i) {
    out.smallSVAlignments[i].score    = in.smallSVAlignments[i].score;
    out.smallSVAlignments[i].isJumped = in.smallSVAlignments[i].isJumped;
    toReal(in.smallSVAlignments[i].align, out.smallSVAlignments[i].align);
  }
  out.spanningAlignments.resize(in.spanningAlignments.size());
  for (size_t i = 0; i < in.spanningAlignments.size(); ++i) {
    out.spanningAlignments[i].score          = in.spanningAlignments[i].score;
    out.spanningAlignments[i].jumpInsertSize = in.spanningAlignments[i].jumpInsertSize;
    out.spanningAlignments[i].jumpRange      = in.spanningAlignments[i].jumpRange;
    toReal(in.spanningAlignments[i].align1, out.spanningAlignments[i].align1);
    toReal(in.spanningAlignments[i].align2, out.spanningAlignments[i].align2);
  }
  out.smallSVSegments = in.smallSVSegments;  // (vectors of std::pair<unsigned, unsigned> on both sides)
  out.largeInsertInfo.resize(in.largeInsertInfo.size());
  for (size_t i = 0; i < in.largeInsertInfo.size(); ++i) {
    out.largeInsertInfo[i].isLeftCandidate  = in.largeInsertInfo[i].isLeftCandidate;
    out.largeInsertInfo[i].isRightCandidate = in.largeInsertInfo[i].isRightCandidate;
    out.largeInsertInfo[i].contigOffset     = in.largeInsertInfo[i].contigOffset;
    out.largeInsertInfo[i].refOffset        = in.largeInsertInfo[i].refOffset;
    out.largeInsertInfo[i].score            = in.largeInsertInfo[i].score;
  }
  out.bestAlignmentIndex = in.bestAlignmentIndex;
  toReal(in.bp1ref, out.bp1ref);
  toReal(in.bp2ref, out.bp2ref);
  out.svs.resize(in.svs.size());
  for (size_t i = 0; i < in.svs.size(); ++i) toReal(in.svs[i], original, out.svs[i]);
  out.isOverlapSkip = in.isOverlapSkip;
}
template <typename S>
bool sameScores(const ::AlignmentScores<S>& a, const manta_amd::AlignmentScores<S>& b)
{
  return a.match == b.match && a.mismatch == b.mismatch && a.open == b.open && a.extend == b.extend && a.offEdge == b.offEdge &&
         a.isAllowEdgeInsertion == b.isAllowEdgeInsertion;
}
inline void toMirror(const ::IterativeAssemblerOptions& in, manta_amd::IterativeAssemblerOptions& out)
{
  out.alphabet                = in.alphabet;
  out.minWordLength           = in.minWordLength;
  out.maxWordLength           = in.maxWordLength;
  out.wordStepSize            = in.wordStepSize;
  out.minContigLength         = in.minContigLength;
  out.minCoverage             = in.minCoverage;
  out.minConservativeCoverage = in.minConservativeCoverage;
  out.minUnusedReads          = in.minUnusedReads;
  out.minSupportReads         = in.minSupportReads;
  out.maxAssemblyCount        = in.maxAssemblyCount;
}
inline void toMirror(const ::GSCOptions& in, manta_amd::GSCOptions& out)
{
  const ::SVRefinerOptions&   r(in.refineOpt);
  manta_amd::SVRefinerOptions m;  // (the score sets are constants of SVRefinerOptions.hpp:36-60 on both sides)
  if (!sameScores(r.largeSVAlignScores, m.largeSVAlignScores) || !sameScores(r.largeInsertEdgeAlignScores, m.largeInsertEdgeAlignScores) ||
      !sameScores(r.largeInsertCompleteAlignScores, m.largeInsertCompleteAlignScores) || !sameScores(r.spanningAlignScores, m.spanningAlignScores) ||
      !sameScores(r.contigFilterScores, m.contigFilterScores) || r.largeGapOpenScore != m.largeGapOpenScore || r.jumpScore != m.jumpScore)
    BOOST_THROW_EXCEPTION(illumina::common::GeneralException("manta_amd batch refiner: SVRefinerOptions differ from the values of SVRefinerOptions.hpp"));
  toMirror(r.smallSVAssembleOpt, out.refineOpt.smallSVAssembleOpt);
  toMirror(r.spanningAssembleOpt, out.refineOpt.spanningAssembleOpt);
  out.scanOpt.minCandidateVariantSize = in.scanOpt.minCandidateVariantSize;
  out.referenceFilename               = in.referenceFilename;
  out.enableRemoteReadRetrieval       = in.enableRemoteReadRetrieval;
  out.isRNA                           = in.isRNA;
  out.isOutputContig                  = in.isOutputContig;
}
inline void toMirror(const ::bam_header_info& in, manta_amd::bam_header_info& out)
{
  out.chrom_data.clear();
  for (const auto& c : in.chrom_data) out.chrom_data.emplace_back(c.label.c_str(), c.length);
}

/// the caller's source behind the product refiner's source interface
struct SourceAdapter : manta_amd::RefinerInputSource {
  explicit SourceAdapter(BatchInputSource& s) : src(s) {}
  void getReferenceSeq(const std::string& chrom, manta_amd::pos_t beginPos, manta_amd::pos_t endPos, std::string& seq) override
  {
    src.getReferenceSeq(chrom, beginPos, endPos, seq);
  }
  void getBreakendReads(const manta_amd::SVBreakend& bp, bool isReversed, const manta_amd::reference_contig_segment& refSeq,
                        manta_amd::AssemblyReadInput& reads) override
  {
    ::SVBreakend rbp;
    toReal(bp, rbp);
    ::reference_contig_segment rref;
    toReal(refSeq, rref);
    src.getBreakendReads(rbp, isReversed, rref, reads);  // (AssemblyReadInput is std::vector<std::string> on both sides)
  }
  BatchInputSource& src;
};

}  // namespace detail

class BatchRefiner {
public:
  BatchRefiner(const ::GSCOptions& opt, const ::bam_header_info& header, BatchInputSource& source) : _adapter(source)
  {
    manta_amd::GSCOptions      o;
    manta_amd::bam_header_info h;
    detail::toMirror(opt, o);
    detail::toMirror(header, h);
    _refiner.reset(new manta_amd::SVCandidateAssemblyRefiner(o, h, _adapter));
  }
  /// host threads of the per-candidate glue; `planThreads` > 1 only with a source that may be called concurrently
  void setThreads(const unsigned hostThreads, const unsigned planThreads = 1)
  {
    _refiner->setHostThreads(hostThreads);
    _refiner->setPlanThreads(planThreads);
  }
  /// getCandidateAssemblyData for every candidate of `svs`, in list order (the _spanToComplexAssmRegions filter of consecutive calls
  /// included); `errors` (optional, one slot per candidate) as manta_amd::SVCandidateAssemblyRefiner::getCandidateAssemblyDataBatch
  void getCandidateAssemblyDataBatch(
      const std::vector<::SVCandidate>& svs, const bool isFindLargeInsertions, std::vector<::SVCandidateAssemblyData>& out,
      std::vector<std::exception_ptr>* errors = nullptr) const
  {
    std::vector<manta_amd::SVCandidate> msvs(svs.size());
    for (size_t i = 0; i < svs.size(); ++i) detail::toMirror(svs[i], msvs[i]);
    std::vector<manta_amd::SVCandidateAssemblyData> mout;
    _refiner->getCandidateAssemblyDataBatch(msvs, isFindLargeInsertions, mout, errors);
    out.resize(svs.size());
    for (size_t i = 0; i < svs.size(); ++i) detail::toReal(mout[i], svs[i], out[i]);
  }
  void clearEdgeData() { _refiner->clearEdgeData(); }

private:
  detail::SourceAdapter                                  _adapter;
  std::unique_ptr<manta_amd::SVCandidateAssemblyRefiner> _refiner;
};

}  // namespace manta_amd_dropin
