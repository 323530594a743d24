// Host adapter of the shadow-read aligner call site:
//   SVScorePairAltProcessor::realignPairedRead   applications/GenerateSVCandidates/SVScorePairAltProcessor.cpp:147-342
//   ContigParams                                 :53-110
//   testFragOverlap                              :122-131, SVScorePairInitParams SVScorePairProcessor.cpp:26-52
// (paths relative to /root/reference/src/c++/lib).  The reference re-aligns, one read at a time, the unmapped ("shadow")
// or chimeric mate of every read pair anchored next to a large-insertion candidate onto the candidate's extended contig
// with GlobalAligner<int>(spanningAlignScores) and keeps the pair if the alignment passes three gates.  Here all the
// reads of a candidate go through ONE manta_align_batch(MANTA_ALIGNER_GLOBAL) launch (the KIND-0 kernel built for the
// large-insertion alignment); what stays on the host is the scalar glue around the alignment: the pre-checks, the 0.85
// score fraction (a float ratio, formed with the reference's expression), the coordinate translation and the overlap test.
#pragma once
#include <sstream>
#include <string>
#include <vector>

#include "refiner_util.hpp"
#include "sv_types.hpp"

namespace manta_amd {

namespace ALIGNPATH {
inline unsigned apath_soft_clip_left_size(const path_t& apath)  // blt_util/align_path.cpp:167-181
{
  unsigned n = 0;
  for (const path_segment& ps : apath) {
    if (ps.type == HARD_CLIP) continue;
    if (ps.type != SOFT_CLIP) break;
    n += ps.length;
  }
  return n;
}
}  // namespace ALIGNPATH

/// SVScorePairAltProcessor.cpp:53-110
struct ContigParams {
  ContigParams(const SVCandidateAssemblyData& assemblyData, const SVCandidate& sv) : extSeq(assemblyData.extendedContigs[sv.assemblyAlignIndex])
  {
    const bool        isBp1First(sv.bp1.interval.range.begin_pos() <= sv.bp2.interval.range.begin_pos());
    const SVBreakend& bpA(isBp1First ? sv.bp1 : sv.bp2);
    const SVBreakend& bpB(isBp1First ? sv.bp2 : sv.bp1);
    const pos_t       bpAHomLength(static_cast<pos_t>(bpA.interval.range.size()) - 1);
    const pos_t       bpBHomLength(static_cast<pos_t>(bpB.interval.range.size()) - 1);
    segmentSpan.set_range(bpA.interval.range.begin_pos() + 1, bpB.interval.range.begin_pos());  // :74 (off by one on purpose)
    pos_t alignBeginPos(0);
    pos_t readStartPos(0);
    if (assemblyData.isSpanning) {
      const SVCandidateAssemblyData::JumpAlignmentResultType& alignment(assemblyData.spanningAlignments[sv.assemblyAlignIndex]);
      alignBeginPos = alignment.align1.beginPos;
      readStartPos  = pos_t(ALIGNPATH::apath_read_length(alignment.align1.apath));
    } else {
      const AlignmentResult<int>&          alignment(assemblyData.smallSVAlignments[sv.assemblyAlignIndex]);
      const std::pair<unsigned, unsigned>& alignSegment(assemblyData.smallSVSegments[sv.assemblyAlignIndex][sv.assemblySegmentIndex]);
      ALIGNPATH::path_t apathTillSvStart(alignment.align.apath.begin(), alignment.align.apath.begin() + alignSegment.first);
      alignBeginPos = alignment.align.beginPos;
      readStartPos  = pos_t(ALIGNPATH::apath_read_length(apathTillSvStart));
    }
    bpAOffset.set_begin_pos(alignBeginPos + readStartPos - 1);
    bpAOffset.set_end_pos(bpAOffset.begin_pos() + bpAHomLength);
    bpBOffset.set_begin_pos(bpAOffset.begin_pos() + pos_t(sv.insertSeq.size()));
    bpBOffset.set_end_pos(bpBOffset.begin_pos() + bpBHomLength);
  }
  const std::string& extSeq;
  known_pos_range2   segmentSpan, bpAOffset, bpBOffset;
};

struct ShadowRead {
  bool               isLeftOfInsert = false;  ///< the anchor is on the left side of the insertion
  const std::string* floatRead      = nullptr;  ///< already reverse-complemented to the expected orientation (:345-360)
  pos_t              anchorPos      = 0;
};

struct ShadowResult {
  bool        isUsable        = false;  ///< realignPairedRead's return value
  int         altTemplateSize = 0;      ///< valid when isUsable
  std::string error;                    ///< the message of the GeneralException the reference throws for this read, if any
};

struct ShadowRealigner {
  /// `minFragSupport`: PairOptions::minFragSupport (SVScorerPairOptions.hpp:31: 50)
  ShadowRealigner(const AlignmentScores<int>& spanningAlignScores, const pos_t minFragSupport, const SVCandidateAssemblyData& assemblyData,
                  const SVCandidate& sv)
    : _scores(spanningAlignScores), _minFragSupport(minFragSupport), _sv(sv), _contig(assemblyData, sv)
  {
    // SVScorePairInitParams (SVScorePairProcessor.cpp:26-40): breakends approximated by the centre of their ranges
    const pos_t centerPos1 = sv.bp1.interval.range.center_pos(), centerPos2 = sv.bp2.interval.range.center_pos();
    const bool  isBp1Lower(centerPos1 <= centerPos2);
    _centerPosA = (isBp1Lower ? centerPos1 : centerPos2);
    _centerPosB = (isBp1Lower ? centerPos2 : centerPos1);
  }

  const ContigParams& contig() const { return _contig; }

  bool testFragOverlap(const int fragBeginRefPos, const int fragEndRefPos) const  // :122-131
  {
    const pos_t fragOverlap(std::min((1 + _centerPosA - fragBeginRefPos), (fragEndRefPos - _centerPosB)));
    return (fragOverlap >= _minFragSupport);
  }

  /// realignPairedRead for every read of the candidate: one aligner launch
  void realignPairedReads(const std::vector<ShadowRead>& reads, std::vector<ShadowResult>& results) const
  {
    const size_t n = reads.size();
    results.assign(n, ShadowResult());
    std::string::const_iterator contigBegin(_contig.extSeq.begin()), contigEnd(_contig.extSeq.end());
    // ---- pre-checks and alignment targets (:159-209) ----
    std::vector<int>                contigBeginOffset(n, 0);
    std::vector<size_t>             taskOf(n, size_t(-1));
    std::vector<manta_align_task_t> tasks;
    std::vector<uint8_t>            arena;
    // the contig once; every task points into it
    arena.assign(_contig.extSeq.begin(), _contig.extSeq.end());
    for (size_t i = 0; i < n; ++i) {
      const ShadowRead&  r(reads[i]);
      const std::string& floatRead(*r.floatRead);
      if (r.isLeftOfInsert) {
        if (r.anchorPos >= _contig.segmentSpan.begin_pos()) continue;
      } else {
        const pos_t endPos(r.anchorPos + pos_t(floatRead.size()));
        if (endPos <= _contig.segmentSpan.end_pos()) continue;
      }
      if (floatRead.empty()) {
        results[i].error = "Empty read attributed to sequence fragment";  // :168-176
        continue;
      }
      size_t regionBegin = 0, regionEnd = _contig.extSeq.size();
      if (_sv.isUnknownSizeInsertion) {  // :189-198
        if (r.isLeftOfInsert) {
          regionEnd = size_t(_contig.bpAOffset.begin_pos()) + _sv.unknownSizeInsertionLeftSeq.size();
        } else {
          contigBeginOffset[i] = static_cast<int>(_contig.bpBOffset.begin_pos()) - int(_sv.unknownSizeInsertionRightSeq.size());
          regionBegin          = size_t(contigBeginOffset[i]);
        }
      }
      if (regionEnd > _contig.extSeq.size() || regionBegin >= regionEnd) {
        results[i].error = "Unexpected zero-length contig region targeted for alignment.";  // :201-209
        continue;
      }
      manta_align_task_t t{};
      t.query_off = arena.size();
      t.query_len = uint32_t(floatRead.size());
      arena.insert(arena.end(), floatRead.begin(), floatRead.end());
      t.ref1_off = regionBegin;
      t.ref1_len = uint32_t(regionEnd - regionBegin);
      taskOf[i]  = tasks.size();
      tasks.push_back(t);
    }
    (void)contigBegin;
    (void)contigEnd;
    std::vector<manta_align_result_t> res(tasks.size());
    std::vector<uint32_t>             cigar;
    if (!tasks.empty()) {
      uint64_t cigCap = 0;
      for (const manta_align_task_t& t : tasks) cigCap += 2ull * t.query_len + 8;
      cigar.assign(cigCap, 0);
      arena.push_back(0);
      const manta_align_scores_t sc   = detail::toAbi(_scores);
      uint64_t                   used = 0;
      manta_ctx_t*               ctx  = threadContext();
      const int rc = manta_align_batch(ctx, MANTA_ALIGNER_GLOBAL, &sc, 0, uint32_t(tasks.size()), tasks.data(), arena.data(), arena.size() - 1,
                                       res.data(), cigar.data(), cigar.size(), &used);
      if (rc != MANTA_OK && rc != MANTA_E_UNSUPPORTED && rc != MANTA_E_DEVICE_FAULT)
        throw GeneralException(std::string("manta_amd shadow aligner: ") + manta_last_error(ctx), rc);
    }
    // ---- the gates (:211-342) ----
    for (size_t i = 0; i < n; ++i) {
      if (taskOf[i] == size_t(-1)) continue;
      const ShadowRead&           r(reads[i]);
      const manta_align_result_t& a(res[taskOf[i]]);
      if (a.status != MANTA_OK) {
        results[i].error = "manta_amd shadow aligner: read outside the device aligner's envelope";
        continue;
      }
      ALIGNPATH::path_t readPath;
      detail::toPath(cigar.data() + a.cigar1_off, a.cigar1_len, readPath);
      const int      alignBeginPos = a.begin_pos1;
      const unsigned readSize(unsigned(r.floatRead->size()));
      unsigned       clipSize(0);
      if (_sv.isUnknownSizeInsertion) {
        clipSize = r.isLeftOfInsert ? ALIGNPATH::apath_soft_clip_right_size(readPath) : ALIGNPATH::apath_soft_clip_left_size(readPath);
      }
      const unsigned        clippedReadSize(readSize - clipSize);
      static const unsigned minAlignReadLength(40);
      if (clippedReadSize < minAlignReadLength) continue;
      const int          nonClipScore(getPathScore(_scores, readPath));
      static const float minScoreFrac(0.85f);
      const int          optimalScore(clippedReadSize * _scores.match);
      const float        scoreFrac(static_cast<float>(nonClipScore) / static_cast<float>(optimalScore));
      if (scoreFrac < minScoreFrac) continue;

      known_pos_range2 fakeRefSpan;
      if (r.isLeftOfInsert) {
        fakeRefSpan.set_begin_pos(r.anchorPos);
        const unsigned shadowRefSpan(ALIGNPATH::apath_ref_length(readPath));
        const int      readContigEndOffset(contigBeginOffset[i] + alignBeginPos + int(shadowRefSpan));
        if (readContigEndOffset < _contig.bpAOffset.begin_pos()) continue;
        const int readContigEndRefOffset(_contig.segmentSpan.begin_pos() + (readContigEndOffset - _contig.bpAOffset.begin_pos()));
        fakeRefSpan.set_end_pos(readContigEndRefOffset);
      } else {
        fakeRefSpan.set_end_pos(r.anchorPos + pos_t(r.floatRead->size()));
        const int readContigBeginOffset(contigBeginOffset[i] + alignBeginPos);
        if (readContigBeginOffset > _contig.bpBOffset.begin_pos()) continue;
        const int readContigBeginRefOffset(_contig.segmentSpan.end_pos() - (_contig.bpBOffset.begin_pos() - readContigBeginOffset));
        fakeRefSpan.set_begin_pos(readContigBeginRefOffset);
      }
      if (fakeRefSpan.begin_pos() > fakeRefSpan.end_pos()) {
        results[i].error = "Failed to parse fragment range from alignment record.";  // :321-329
        continue;
      }
      const int altTemplateSize = int(fakeRefSpan.size());
      if (!testFragOverlap(fakeRefSpan.begin_pos(), fakeRefSpan.end_pos())) continue;
      results[i].isUsable        = true;
      results[i].altTemplateSize = altTemplateSize;
    }
  }

  /// the reference's single-read form
  bool realignPairedRead(const bool isLeftOfInsert, const std::string& floatRead, const pos_t anchorPos, int& altTemplateSize) const
  {
    ShadowRead r;
    r.isLeftOfInsert = isLeftOfInsert;
    r.floatRead      = &floatRead;
    r.anchorPos      = anchorPos;
    std::vector<ShadowResult> out;
    realignPairedReads(std::vector<ShadowRead>(1, r), out);
    if (!out[0].error.empty()) throw GeneralException(out[0].error);
    if (out[0].isUsable) altTemplateSize = out[0].altTemplateSize;
    return out[0].isUsable;
  }

private:
  const AlignmentScores<int> _scores;
  const pos_t                _minFragSupport;
  const SVCandidate&         _sv;
  const ContigParams         _contig;
  pos_t                      _centerPosA = 0, _centerPosB = 0;
};

}  // namespace manta_amd
