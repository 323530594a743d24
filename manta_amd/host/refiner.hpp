// SVCandidateAssemblyRefiner on the MI355X path: the reference's object interface
//   struct SVCandidateAssemblyRefiner { getCandidateAssemblyData(sv, isFindLargeInsertions, assemblyData); clearEdgeData(); }
//   (/root/reference/src/c++/lib/applications/GenerateSVCandidates/SVCandidateAssemblyRefiner.hpp:41-99)
// plus a batched form of the same call.  All assembly and all DP run in the HIP kernels behind include/manta_amd.h:
//   complex ("small SV") candidates -> manta_smallsv_*  (assemble -> 10-mer trim -> GlobalLargeIndelAligner, fused on device)
//   spanning candidates             -> manta_spanning_* (assemble -> GlobalJumpAligner on the cut references -> re-align rule, fused)
//   large-insertion completion      -> manta_align_batch(MANTA_ALIGNER_GLOBAL)
// The host code here is the refiner's own glue (SVCandidateAssemblyRefiner.cpp:59-86, 677-1007, 1210-1250, 1364-1398,
// 1422-1849, 1860-2303; manta/SVReferenceUtil.cpp:56-205), restated over sv_types.hpp.  The two I/O seams of the
// reference -- faidx reference fetch and the BAM read scan -- are the RefinerInputSource callbacks.
// Pinned against the UNMODIFIED reference refiner run in memory (oracle/ref_refiner_driver.cpp) by
// tests/test_refiner.py.  DNA only: GSCOptions::isRNA is refused (the intron-aware aligner is not on this path).
#pragma once

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <exception>
#include <future>
#include <memory>
#include <mutex>
#include <sstream>
#include <thread>

#include "pile_dump.hpp"
#include "sv_types.hpp"

namespace manta_amd {

/// wall-clock split of the last getCandidateAssemblyDataBatch call, seconds (no reference counterpart; for logs)
struct RefinerTimes {
  double plan = 0, pack = 0, device = 0, post = 0;
};

/// the refiner's two input seams
struct RefinerInputSource {
  virtual ~RefinerInputSource() {}
  /// get_standardized_region_seq (htsapi/samtools_fasta_util.hpp:52-57): [beginPos, endPos] closed, zero-based
  virtual void getReferenceSeq(const std::string& chrom, pos_t beginPos, pos_t endPos, std::string& seq) = 0;
  /// SVCandidateAssembler::getBreakendReads (manta/SVCandidateAssembler.cpp:271-659): append this breakend's assembly
  /// reads (reverse-complemented when isReversed) to `reads`
  virtual void getBreakendReads(
      const SVBreakend& bp, bool isReversed, const reference_contig_segment& refSeq, AssemblyReadInput& reads) = 0;
};

namespace detail {

// ---- manta/SVReferenceUtil.cpp ---------------------------------------------------------------------------------
inline void getBpReferenceInterval(  // :56-98
    const bam_header_info& header, const pos_t extraRefEdgeSize, const GenomeInterval& bpInterval, GenomeInterval& refInterval,
    unsigned& leadingTrim, unsigned& trailingTrim)
{
  const pos_t chromSize = pos_t(header.chrom_data.at(size_t(bpInterval.tid)).length);
  if (bpInterval.range.begin_pos() >= chromSize || bpInterval.range.end_pos() <= 0) {
    std::ostringstream oss;
    oss << "getBpReferenceInterval: requested reference range has no overlap with chromosome\n";
    throw GeneralException(oss.str());
  }
  pos_t beginPos = bpInterval.range.begin_pos() - extraRefEdgeSize;
  leadingTrim    = 0;
  if (beginPos < 0) {
    leadingTrim = unsigned(-beginPos);
    beginPos    = 0;
  }
  pos_t endPos = bpInterval.range.end_pos() + extraRefEdgeSize;
  trailingTrim = 0;
  if (endPos > chromSize) {
    trailingTrim = unsigned(endPos - chromSize);
    endPos       = chromSize;
  }
  refInterval.tid = bpInterval.tid;
  refInterval.range.set_range(beginPos, endPos);
}
inline bool isRefRegionValid(const bam_header_info& header, const GenomeInterval& bpInterval)  // :171-178
{
  const pos_t chromSize = pos_t(header.chrom_data.at(size_t(bpInterval.tid)).length);
  return !(bpInterval.range.begin_pos() >= chromSize || bpInterval.range.end_pos() <= 0);
}
inline bool isRefRegionOverlap(const bam_header_info& header, const pos_t extraRefEdgeSize, const SVCandidate& sv)  // :159-169
{
  if (sv.bp1.interval.tid != sv.bp2.interval.tid) return false;
  GenomeInterval r1, r2;
  unsigned       lt, tt;
  getBpReferenceInterval(header, extraRefEdgeSize, sv.bp1.interval, r1, lt, tt);
  getBpReferenceInterval(header, extraRefEdgeSize, sv.bp2.interval, r2, lt, tt);
  return r1.isIntersect(r2);
}
inline void getIntervalReferenceSegment(  // :113-156
    RefinerInputSource& source, const bam_header_info& header, const pos_t extraRefEdgeSize, const GenomeInterval& bpInterval,
    reference_contig_segment& intervalRefSeq, unsigned& leadingTrim, unsigned& trailingTrim)
{
  GenomeInterval refInterval;
  getBpReferenceInterval(header, extraRefEdgeSize, bpInterval, refInterval, leadingTrim, trailingTrim);
  const known_pos_range2& range(refInterval.range);
  intervalRefSeq.set_offset(range.begin_pos());
  source.getReferenceSeq(header.chrom_data[size_t(refInterval.tid)].label, range.begin_pos(), range.end_pos() - 1, intervalRefSeq.seq());
  if (intervalRefSeq.seq().size() != range.size()) {
    std::ostringstream oss;
    oss << "getIntervalReferenceSegment: Unexpected reference sequence\n\texpected_size: " << range.size()
        << "\treturned_size: " << intervalRefSeq.seq().size() << "\n";
    throw GeneralException(oss.str());
  }
}

// ---- breakend coordinate conversion (SVCandidateAssemblyRefiner.cpp:59-86, 287-311, 677-800, 1210-1250) --------------
inline pos_t alignEnd(const Alignment& align) { return align.beginPos + pos_t(ALIGNPATH::apath_ref_length(align.apath)); }

inline void adjustAssembledBreakend(
    const Alignment& align, const bool isAlign1, const unsigned jumpRange, const reference_contig_segment& ref, const bool isReversed,
    SVBreakend& bp)
{
  const pos_t refSize = pos_t(ref.seq().size());
  // alignment/AlignmentUtil.hpp:38-56
  const pos_t bpBeginOffset = isReversed ? (refSize - alignEnd(align)) : align.beginPos;
  const pos_t bpEndOffset   = isReversed ? (refSize - align.beginPos) : alignEnd(align);
  const bool  isBpAtAlignEnd   = (bp.state == SVBreakendState::RIGHT_OPEN);
  const pos_t bpBreakendOffset = isBpAtAlignEnd ? (bpEndOffset - 1) : bpBeginOffset;
  const pos_t bpBreakendPos    = ref.get_offset() + bpBreakendOffset;
  const bool  isLeftAligned    = (isAlign1 == isBpAtAlignEnd);
  if (isLeftAligned)
    bp.interval.range.set_range(bpBreakendPos, bpBreakendPos + pos_t(jumpRange) + 1);
  else
    bp.interval.range.set_range(bpBreakendPos - pos_t(jumpRange), bpBreakendPos + 1);
}

inline void addCigarToSpanningAlignment(SVCandidate& sv)
{
  if (getSVType(sv) != SV_TYPE::INDEL) return;
  const bool        isBp1First = sv.bp1.interval.range.begin_pos() <= sv.bp2.interval.range.begin_pos();
  const SVBreakend& bpA(isBp1First ? sv.bp1 : sv.bp2);
  const SVBreakend& bpB(isBp1First ? sv.bp2 : sv.bp1);
  const pos_t       deleteSize = (bpB.interval.range.begin_pos() - bpA.interval.range.begin_pos()) - 1;
  const pos_t       insertSize = pos_t(sv.insertSeq.size());
  if (insertSize) sv.insertAlignment.emplace_back(ALIGNPATH::INSERT, unsigned(insertSize));
  if (deleteSize) sv.insertAlignment.emplace_back(ALIGNPATH::DELETE, unsigned(deleteSize));
}

/// how far an indel can slide left/right with unchanged edit distance (:677-717)
inline known_pos_range2 getVariantRange(
    const std::string& ref, const known_pos_range2& refRange, const std::string& read, const known_pos_range2& readRange)
{
  const pos_t maxRightOffset = std::min(pos_t(ref.size()) - refRange.end_pos(), pos_t(read.size()) - readRange.end_pos());
  pos_t       rightOffset    = 0;
  for (; rightOffset < maxRightOffset; ++rightOffset)
    if (ref[size_t(refRange.begin_pos() + rightOffset)] != read[size_t(readRange.begin_pos() + rightOffset)]) break;
  const pos_t minLeftOffset = std::max(-refRange.begin_pos(), -readRange.begin_pos());
  pos_t       leftOffset    = 0;
  for (; leftOffset >= minLeftOffset; --leftOffset)
    if (ref[size_t(refRange.end_pos() + leftOffset - 1)] != read[size_t(readRange.end_pos() + leftOffset - 1)]) break;
  return known_pos_range2(leftOffset, rightOffset);
}

/// :720-800
inline void setSmallCandSV(
    const reference_contig_segment& ref, const std::string& contig, const Alignment& align, const segment_t& segRange, SVCandidate& sv,
    const GSCOptions& opt)
{
  sv.setPrecise();
  known_pos_range2 readRange, refRange;
  pos_t            readPos = 0, refPos = align.beginPos;
  for (unsigned i = 0; i < align.apath.size(); ++i) {
    const ALIGNPATH::path_segment& ps(align.apath[i]);
    if (i == segRange.first) {
      refRange.set_begin_pos(refPos);
      readRange.set_begin_pos(readPos);
    }
    if (ALIGNPATH::is_segment_type_ref_length(ps.type)) refPos += pos_t(ps.length);
    if (ALIGNPATH::is_segment_type_read_length(ps.type)) readPos += pos_t(ps.length);
    if (i == segRange.second) {
      refRange.set_end_pos(refPos);
      readRange.set_end_pos(readPos);
    }
  }
  const known_pos_range2 cipos(getVariantRange(ref.seq(), refRange, contig, readRange));
  if (cipos.begin_pos() != 0) {
    std::ostringstream oss;
    oss << "Attempting to convert alignment to sv candidate. contigSize: " << contig.size() << " segments: [" << segRange.first << ","
        << segRange.second << "]\n";
    throw GeneralException(oss.str());
  }
  sv.bp1.state = SVBreakendState::RIGHT_OPEN;
  const pos_t beginPos = ref.get_offset() + refRange.begin_pos() - 1;
  sv.bp1.interval.range.set_range(beginPos, beginPos + cipos.end_pos() + 1);
  sv.bp2.state = SVBreakendState::LEFT_OPEN;
  const pos_t endPos = ref.get_offset() + refRange.end_pos();
  sv.bp2.interval.range.set_range(endPos, endPos + cipos.end_pos() + 1);
  sv.bp2.interval.tid = sv.bp1.interval.tid;
  sv.insertSeq        = contig.substr(size_t(readRange.begin_pos()), readRange.size());
  sv.insertAlignment  = ALIGNPATH::path_t(align.apath.begin() + segRange.first, align.apath.begin() + segRange.second + 1);
  if (opt.isOutputContig) sv.contigSeq = contig;
}

/// blt_util/align_path.cpp:295-329
inline void apath_limit_read_length(const unsigned target_read_start, const unsigned target_read_end, ALIGNPATH::path_t& apath)
{
  bool           isStartSet  = false;
  unsigned       read_length = 0;
  const unsigned as          = unsigned(apath.size());
  unsigned       startSegment = 0, endSegment = as;
  for (unsigned i = 0; i < as; ++i) {
    ALIGNPATH::path_segment& ps(apath[i]);
    if (!ALIGNPATH::is_segment_type_read_length(ps.type)) continue;
    read_length += ps.length;
    if (!isStartSet && read_length > target_read_start) {
      ps.length -= ps.length - (read_length - target_read_start);
      startSegment = i;
      isStartSet   = true;
    }
    if (read_length >= target_read_end) {
      if (read_length > target_read_end) ps.length -= (read_length - target_read_end);
      endSegment = i + 1;
      break;
    }
  }
  apath = ALIGNPATH::path_t(apath.begin() + startSegment, apath.begin() + endSegment);
}

/// read-coordinate span of an indel run (:802-830)
inline known_pos_range2 getInsertTrim(const ALIGNPATH::path_t& apath, const segment_t& segRange)
{
  known_pos_range2 range;
  pos_t            readPos = 0;
  for (unsigned i = 0; i < apath.size(); ++i) {
    if (i == segRange.first) range.set_begin_pos(readPos);
    if (ALIGNPATH::is_segment_type_read_length(apath[i].type)) readPos += pos_t(apath[i].length);
    if (i == segRange.second) {
      range.set_end_pos(readPos);
      return range;
    }
  }
  return range;
}

/// :642-665
inline bool isFinishedLargeInsertAlignment(
    const AlignmentScores<int>& scores, const ALIGNPATH::path_t& apath, const segment_t& insertSegment, const unsigned middleSize)
{
  const ALIGNPATH::path_t left(apath.begin(), apath.begin() + insertSegment.second + 1);
  LargeInsertionInfo      info;
  info.isLeftCandidate = isLargeInsertSegment(scores, left, info.contigOffset, info.refOffset, info.score, middleSize);
  ALIGNPATH::path_t rev(apath.begin() + insertSegment.first, apath.end());
  std::reverse(rev.begin(), rev.end());
  info.isRightCandidate = isLargeInsertSegment(scores, rev, info.contigOffset, info.refOffset, info.score, middleSize);
  return info.isLeftCandidate && info.isRightCandidate;
}

/// the per-locus host glue after a device batch is independent across loci: spread it over host threads
template <typename F>
void parallelFor(const size_t n, unsigned threads, F&& body)
{
  if (threads > n) threads = unsigned(n);
  if (threads <= 1) {
    for (size_t i = 0; i < n; ++i) body(i);
    return;
  }
  std::atomic<size_t> next(0);
  std::exception_ptr  firstError;
  std::mutex          errorLock;
  auto                worker = [&]() {
    try {
      while (true) {
        const size_t begin = next.fetch_add(16);
        if (begin >= n) return;
        for (size_t i = begin; i < std::min(n, begin + 16); ++i) body(i);
      }
    } catch (...) {
      std::lock_guard<std::mutex> g(errorLock);
      if (!firstError) firstError = std::current_exception();
      next.store(n);
    }
  };
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < threads; ++t) pool.emplace_back(worker);
  worker();
  for (std::thread& t : pool) t.join();
  if (firstError) std::rethrow_exception(firstError);
}

// ---- batched device calls ------------------------------------------------------------------------------------------
struct AlignJob {
  const std::string* query = nullptr;
  const char *       ref1 = nullptr, *ref2 = nullptr;
  size_t             ref1Len = 0, ref2Len = 0;
  manta_align_result_t res{};
  std::vector<uint32_t> cigar;  ///< this job's segments, result offsets rebased to 0
};

inline void alignBatch(const int kind, const AlignmentScores<int>& scores, const int extra, std::vector<AlignJob*>& jobs)
{
  if (jobs.empty()) return;
  std::vector<uint8_t>            arena;
  std::vector<manta_align_task_t> tasks(jobs.size());
  uint64_t                        cigarCap = 64;
  for (size_t i = 0; i < jobs.size(); ++i) {
    AlignJob&           j(*jobs[i]);
    manta_align_task_t& t(tasks[i]);
    if (j.query->empty()) throw GeneralException("Unexpected empty query sequence");
    if (j.ref1Len == 0) throw GeneralException(kind == MANTA_ALIGNER_JUMP ? "Unexpected empty reference1 sequence" : "Unexpected empty reference sequence");
    if (kind == MANTA_ALIGNER_JUMP && j.ref2Len == 0) throw GeneralException("Unexpected empty reference2 sequence");
    t           = manta_align_task_t{};
    t.query_off = arena.size();
    t.query_len = uint32_t(j.query->size());
    arena.insert(arena.end(), j.query->begin(), j.query->end());
    t.ref1_off = arena.size();
    t.ref1_len = uint32_t(j.ref1Len);
    arena.insert(arena.end(), j.ref1, j.ref1 + j.ref1Len);
    t.ref2_off = arena.size();
    t.ref2_len = uint32_t(j.ref2Len);
    if (j.ref2Len) arena.insert(arena.end(), j.ref2, j.ref2 + j.ref2Len);
    cigarCap += 2ull * t.query_len + 16;
  }
  arena.push_back(0);
  std::vector<manta_align_result_t> results(jobs.size());
  std::vector<uint32_t>             cigar(cigarCap);
  uint64_t                          used = 0;
  manta_ctx_t*                      ctx  = threadContext();
  const manta_align_scores_t        sc   = toAbi(scores);
  const int rc = manta_align_batch(ctx, kind, &sc, extra, uint32_t(jobs.size()), tasks.data(), arena.data(), arena.size() - 1,
                                   results.data(), cigar.data(), cigar.size(), &used);
  if (rc != MANTA_OK) throw GeneralException(std::string("manta_amd aligner: ") + manta_last_error(ctx), rc);
  for (size_t i = 0; i < jobs.size(); ++i) {
    AlignJob& j(*jobs[i]);
    j.res = results[i];
    if (j.res.status != MANTA_OK) throw GeneralException("manta_amd aligner: task failed on the device", j.res.status);
    j.cigar.assign(cigar.begin() + j.res.cigar1_off, cigar.begin() + j.res.cigar1_off + j.res.cigar1_len);
    j.cigar.insert(j.cigar.end(), cigar.begin() + j.res.cigar2_off, cigar.begin() + j.res.cigar2_off + j.res.cigar2_len);
    j.res.cigar1_off = 0;
    j.res.cigar2_off = j.res.cigar1_len;
  }
}
inline void toResult(const AlignJob& j, AlignmentResult<int>& result)
{
  result.clear();
  result.score          = j.res.score;
  result.isJumped       = j.res.is_jumped != 0;
  result.align.beginPos = j.res.begin_pos1;
  toPath(j.cigar.data(), j.res.cigar1_len, result.align.apath);
}
inline void toResult(const AlignJob& j, JumpAlignmentResult<int>& result)
{
  result.clear();
  result.score           = j.res.score;
  result.jumpInsertSize  = j.res.jump_insert_size;
  result.jumpRange       = j.res.jump_range;
  result.align1.beginPos = j.res.begin_pos1;
  result.align2.beginPos = j.res.begin_pos2;
  toPath(j.cigar.data(), j.res.cigar1_len, result.align1.apath);
  toPath(j.cigar.data() + j.res.cigar2_off, j.res.cigar2_len, result.align2.apath);
}

/// page-locked staging memory kept by a refiner object across calls (manta_host_alloc: the upload is a plain DMA, and the buffer is
/// not zero-filled and re-faulted on every batch as a std::vector would be); falls back to pageable memory if the driver says no
struct PinnedArena {
  uint8_t* p   = nullptr;
  size_t   cap = 0;
  PinnedArena() = default;
  PinnedArena(const PinnedArena&) = delete;
  PinnedArena& operator=(const PinnedArena&) = delete;
  ~PinnedArena()
  {
    if (p) manta_host_free(p);
  }
  /// at least `n` bytes, or nullptr
  uint8_t* ensure(const size_t n)
  {
    if (n <= cap) return p;
    if (p) manta_host_free(p);
    p   = nullptr;
    cap = 0;
    const size_t want = n + n / 4 + 4096;
    void*        q    = nullptr;
    if (manta_host_alloc(want, &q) != MANTA_OK || !q) return nullptr;
    p   = static_cast<uint8_t*>(q);
    cap = want;
    return p;
  }
};

struct PackedReads {
  std::vector<uint8_t>  bases;
  const uint8_t*        basesPtr = nullptr;  ///< the flattened piles + one pad byte (finish())
  std::vector<uint64_t> readOff{0};
  std::vector<uint32_t> locusBegin{0};
  std::vector<uint64_t> bitsBound;  ///< per locus
  std::vector<const AssemblyReadInput*> piles;
  void addLocus(const AssemblyReadInput& reads, const unsigned maxAssemblyCount)
  {
    uint64_t off = readOff.back();
    for (const std::string& r : reads) {
      off += r.size();
      readOff.push_back(off);
    }
    locusBegin.push_back(uint32_t(readOff.size() - 1));
    piles.push_back(&reads);
    const uint64_t W = (reads.size() + 2ull * maxAssemblyCount + 63) / 64;
    bitsBound.push_back(uint64_t(maxAssemblyCount) * 2 * W + 2ull * maxAssemblyCount + 8);
  }
  /// flatten the piles (the copy is spread over host threads) into `stage` if it can hold them, else into `bases`; the kernels
  /// read input arenas in aligned dwords: one pad byte behind the last base
  void finish(const unsigned threads, PinnedArena* stage = nullptr)
  {
    uint8_t* dst = stage ? stage->ensure(readOff.back() + 16) : nullptr;
    if (!dst) {
      bases.resize(readOff.back() + 1);
      dst = bases.data();
    }
    dst[readOff.back()] = 0;
    basesPtr            = dst;
    parallelFor(piles.size(), threads, [&](const size_t l) {
      size_t r = locusBegin[l];
      for (const std::string& rd : *piles[l]) {
        std::copy(rd.begin(), rd.end(), dst + readOff[r]);
        ++r;
      }
    });
  }
  uint32_t nLoci() const { return uint32_t(locusBegin.size() - 1); }
};

/// reference windows of a batch back to back (+ one pad byte) in page-locked memory: a pageable source is staged by the runtime at a
/// few GB/s, and 10 000 windows of 1 800 bases appended one by one to a std::vector cost more than their DMA
struct PackedRefs {
  const uint8_t*        bytes = nullptr;
  std::vector<uint64_t> off;
  std::vector<uint8_t>  fallback;
  void pack(const std::vector<const std::string*>& refs, const unsigned threads, PinnedArena& stage)
  {
    off.assign(refs.size() + 1, 0);
    for (size_t i = 0; i < refs.size(); ++i) off[i + 1] = off[i] + refs[i]->size();
    uint8_t* dst = stage.ensure(off.back() + 16);
    if (!dst) {
      fallback.resize(off.back() + 16);
      dst = fallback.data();
    }
    parallelFor(refs.size(), threads, [&](const size_t i) { std::copy(refs[i]->begin(), refs[i]->end(), dst + off[i]); });
    dst[off.back()] = 0;
    bytes           = dst;
  }
};

struct AsmOutput {
  std::vector<manta_asm_locus_result_t> loci;
  std::vector<manta_asm_contig_t>       contigs;
  std::vector<uint8_t>                  seq;
  std::vector<uint64_t>                 bits;
  void sizeTo(const uint32_t nLoci, const uint64_t nContigs, const uint64_t seqBytes, const uint64_t bitsWords)
  {
    loci.resize(nLoci);
    contigs.resize(nContigs);
    seq.resize(seqBytes);
    bits.resize(bitsWords);
  }
  void toContigs(const unsigned locus, Assembly& out) const
  {
    const manta_asm_locus_result_t& lr(loci[locus]);
    if (lr.status != MANTA_OK) throw GeneralException("manta_amd assembler: locus failed on the device", lr.status);
    out.clear();
    out.resize(lr.n_contigs);
    for (unsigned c = 0; c < lr.n_contigs; ++c) {
      const manta_asm_contig_t& rec(contigs[lr.first_contig + c]);
      AssembledContig&          ctg(out[c]);
      ctg.seq.assign(reinterpret_cast<const char*>(seq.data() + rec.seq_off), rec.seq_len);
      ctg.seedReadCount = rec.seed_read_count;
      bitsToSet(bits.data() + rec.support_off, lr.n_words, ctg.supportReads);
      bitsToSet(bits.data() + rec.reject_off, lr.n_words, ctg.rejectReads);
      ctg.conservativeRange.set_range(rec.conservative_begin, rec.conservative_end);
    }
  }
};

struct SmallSvOutput : AsmOutput {
  std::vector<manta_smallsv_alignment_t> aligns;
  std::vector<uint32_t>                  cigar;
};

/// the fused device pipeline for a batch of complex loci
inline void smallSvBatch(
    manta_ctx_t* ctx, manta_smallsv_t*& b, const IterativeAssemblerOptions& opt, const AlignmentScores<int>& scores, const int largeIndelScore,
    PackedReads& in, const std::vector<const std::string*>& refs, const std::vector<manta_ref_cuts_t>& cuts, SmallSvOutput& out, const unsigned threads,
    PinnedArena& refStage)
{
  if (in.nLoci() == 0) return;
  const manta_asm_options_t  o   = toAbi(opt);
  const manta_align_scores_t sc  = toAbi(scores);
  PackedRefs                 R;
  R.pack(refs, threads, refStage);
  auto check = [&](const int rc) {
    if (rc == MANTA_OK) return;
    throw GeneralException("manta_amd small-SV pipeline: " + std::string(manta_last_error(ctx)), rc);
  };
  (void)b;  // (the staged pipeline object of earlier rounds: the whole-batch call keeps its worker pipelines in the context)
  // One whole-batch call (manta_smallsv_batch): the read bases stream in behind the running assembler, the results are staged behind the
  // last kernel and compacted straight into these arenas -- the staged calls (upload / run / output_sizes / download) took 2.5x the
  // device time of the same loci (round 5: 26 ms against 10).  The arenas are kept across calls (no re-allocation, no zero-fill) and
  // sized from the previous call's use; a call that reports MANTA_E_CAPACITY runs again with larger ones.
  const uint32_t n      = in.nLoci();
  const uint64_t nReads = in.locusBegin.back();
  const uint64_t nBases = in.readOff.back();
  out.loci.resize(n);
  const uint64_t contigsCap = uint64_t(n) * opt.maxAssemblyCount + 1;
  if (out.contigs.size() < contigsCap) out.contigs.resize(contigsCap);
  if (out.aligns.size() < contigsCap) out.aligns.resize(contigsCap);
  uint64_t seqCap  = std::max<uint64_t>(out.seq.size(), nBases / 4 + 4096ull * n + (1u << 20));
  uint64_t bitsCap = std::max<uint64_t>(out.bits.size(), 40ull * nReads / 64 + 128ull * n + 4096);
  uint64_t cigCap  = std::max<uint64_t>(out.cigar.size(), 512ull * n + 4096);
  for (int attempt = 0;; ++attempt) {
    if (out.seq.size() < seqCap) out.seq.resize(seqCap);
    if (out.bits.size() < bitsCap) out.bits.resize(bitsCap);
    if (out.cigar.size() < cigCap) out.cigar.resize(cigCap);
    uint64_t  su = 0, bu = 0, cu = 0;
    const int rc = manta_smallsv_batch(ctx, &o, &sc, largeIndelScore, n, in.basesPtr, in.readOff.data(), in.locusBegin.data(), R.bytes, R.off.data(),
                                       cuts.data(), nullptr, nullptr, out.loci.data(), out.contigs.data(), out.aligns.data(), out.contigs.size(), out.seq.data(),
                                       out.seq.size(), &su, out.bits.data(), out.bits.size(), &bu, out.cigar.data(), out.cigar.size(), &cu, nullptr, nullptr);
    if (rc == MANTA_E_CAPACITY && attempt < 6) {
      seqCap  = 2 * out.seq.size();
      bitsCap = 2 * out.bits.size();
      cigCap  = 2 * out.cigar.size();
      continue;
    }
    // per-item failures (a contig outside the aligner's envelope, ...) are in the per-locus / per-contig status and become
    // that candidate's exception; only a failure of the call itself is fatal here
    if (rc != MANTA_E_UNSUPPORTED && rc != MANTA_E_DEVICE_FAULT && rc != MANTA_E_EMPTY_SEQ) check(rc);
    break;
  }
}

struct SpanningOutput : AsmOutput {
  std::vector<manta_spanning_alignment_t> aligns;
  std::vector<uint32_t>                   cigar;
};

/// the fused device pipeline for a batch of spanning loci (references already oriented, in alignment order)
inline void spanningBatch(
    manta_ctx_t* ctx, manta_spanning_t*& b, const IterativeAssemblerOptions& opt, const AlignmentScores<int>& scores, const int jumpScore, PackedReads& in,
    const std::vector<const std::string*>& refs1, const std::vector<const std::string*>& refs2, const std::vector<manta_jump_cuts_t>& cuts,
    SpanningOutput& out, const unsigned threads, PinnedArena& ref1Stage, PinnedArena& ref2Stage)
{
  if (in.nLoci() == 0) return;
  const manta_asm_options_t  o   = toAbi(opt);
  const manta_align_scores_t sc  = toAbi(scores);
  PackedRefs                 R1, R2;
  R1.pack(refs1, threads, ref1Stage);
  R2.pack(refs2, threads, ref2Stage);
  auto check = [&](const int rc) {
    if (rc == MANTA_OK) return;
    throw GeneralException("manta_amd spanning pipeline: " + std::string(manta_last_error(ctx)), rc);
  };
  (void)b;
  // one whole-batch call, arenas kept across calls and sized from the previous call's use (see smallSvBatch)
  const uint32_t n      = in.nLoci();
  const uint64_t nReads = in.locusBegin.back();
  const uint64_t nBases = in.readOff.back();
  out.loci.resize(n);
  const uint64_t contigsCap = uint64_t(n) * opt.maxAssemblyCount + 1;
  if (out.contigs.size() < contigsCap) out.contigs.resize(contigsCap);
  if (out.aligns.size() < contigsCap) out.aligns.resize(contigsCap);
  uint64_t seqCap  = std::max<uint64_t>(out.seq.size(), nBases / 4 + 4096ull * n + (1u << 20));
  uint64_t bitsCap = std::max<uint64_t>(out.bits.size(), 40ull * nReads / 64 + 128ull * n + 4096);
  uint64_t cigCap  = std::max<uint64_t>(out.cigar.size(), 1024ull * n + 4096);
  for (int attempt = 0;; ++attempt) {
    if (out.seq.size() < seqCap) out.seq.resize(seqCap);
    if (out.bits.size() < bitsCap) out.bits.resize(bitsCap);
    if (out.cigar.size() < cigCap) out.cigar.resize(cigCap);
    uint64_t  su = 0, bu = 0, cu = 0;
    const int rc = manta_spanning_batch(ctx, &o, &sc, jumpScore, n, in.basesPtr, in.readOff.data(), in.locusBegin.data(), R1.bytes, R1.off.data(),
                                        R2.bytes, R2.off.data(), cuts.data(), nullptr, nullptr, out.loci.data(), out.contigs.data(), out.aligns.data(),
                                        out.contigs.size(), out.seq.data(), out.seq.size(), &su, out.bits.data(), out.bits.size(), &bu, out.cigar.data(),
                                        out.cigar.size(), &cu, nullptr, nullptr);
    if (rc == MANTA_E_CAPACITY && attempt < 6) {
      seqCap  = 2 * out.seq.size();
      bitsCap = 2 * out.bits.size();
      cigCap  = 2 * out.cigar.size();
      continue;
    }
    if (rc != MANTA_E_UNSUPPORTED && rc != MANTA_E_DEVICE_FAULT && rc != MANTA_E_EMPTY_SEQ) check(rc);  // per-item codes: see smallSvBatch
    break;
  }
}

}  // namespace detail

struct SVCandidateAssemblyRefiner {
  SVCandidateAssemblyRefiner(const GSCOptions& opt, const bam_header_info& header, RefinerInputSource& source)
    : _opt(opt), _header(header), _source(source)
  {
    if (opt.isRNA) throw GeneralException("manta_amd::SVCandidateAssemblyRefiner: the RNA (intron-aware) spanning path is not supported");
  }

  ~SVCandidateAssemblyRefiner()
  {
    if (_smallPipe) manta_smallsv_destroy(_smallPipe);
    if (_spanPipe) manta_spanning_destroy(_spanPipe);
    if (_ctx) manta_ctx_destroy(_ctx);
  }
  SVCandidateAssemblyRefiner(const SVCandidateAssemblyRefiner&) = delete;
  SVCandidateAssemblyRefiner& operator=(const SVCandidateAssemblyRefiner&) = delete;

  /// SVCandidateAssemblyRefiner.hpp:56-57 -- a batch of one
  void getCandidateAssemblyData(const SVCandidate& sv, const bool isFindLargeInsertions, SVCandidateAssemblyData& assemblyData) const
  {
    std::vector<SVCandidateAssemblyData> out;
    getCandidateAssemblyDataBatch(std::vector<SVCandidate>(1, sv), isFindLargeInsertions, out);
    assemblyData = std::move(out[0]);
  }

  void clearEdgeData() { _spanToComplexAssmRegions.clear(); }

  /// host threads used for the per-locus glue after each device batch (default: all hardware threads, at most 64)
  void setHostThreads(const unsigned n) { _hostThreads = (n == 0) ? 1 : n; }
  /// host threads that call the input source (reference / read callbacks) of a batch concurrently.  Default 1: the callbacks are
  /// called one candidate after the other, as the reference's worker does.  With more threads the source's two functions must be
  /// safe to call concurrently (a source that keeps one BAM / FASTA handle per thread, or answers from memory); the cross-candidate
  /// state of the call itself -- the geometric _spanToComplexAssmRegions filter -- is still applied in list order.
  void setPlanThreads(const unsigned n) { _planThreads = (n == 0) ? 1 : n; }
  /// every candidate that reaches the assembler + aligner is also written to `w` (pile_dump.hpp: read pile, reference windows, cuts,
  /// options -- the inputs of the whole-batch ABI calls), for tools/replay_piles.py; nullptr switches it off
  void setPileDump(PileDumpWriter* w) { _pileDump = w; }

  /// work counters of this refiner object (no reference counterpart; for logs and tests)
  struct Stats {
    uint64_t smallLoci = 0, spanningLoci = 0, contigAlignments = 0, realignedContigs = 0, largeInsertionAlignments = 0;
  };
  const Stats&        stats() const { return _stats; }
  const RefinerTimes& times() const { return _times; }

  /// The same call for a whole list of candidates (one edge's worth, or many edges' worth: the only cross-candidate
  /// state is the geometric _spanToComplexAssmRegions filter, applied here in list order exactly as consecutive
  /// single calls would).  Every device stage runs once over the whole list.
  /// `errors` (optional): one slot per candidate; a candidate whose processing throws (as the reference's single call
  /// would: off-chromosome regions, empty sequences, an input outside the device path's envelope) gets its exception
  /// there and an empty result, every other candidate is computed.  Without it the first such exception is re-thrown
  /// after the whole batch has been processed.
  void getCandidateAssemblyDataBatch(
      const std::vector<SVCandidate>& svs, const bool isFindLargeInsertions, std::vector<SVCandidateAssemblyData>& out,
      std::vector<std::exception_ptr>* errors = nullptr) const
  {
    const size_t n = svs.size();
    out.resize(n);  // (plan() clears every entry)
    std::vector<Plan> plans(n);
    struct JoinTeardowns {  // (declared behind `plans`: the background frees end before the plans do, whichever way this call is left)
      std::vector<std::unique_ptr<ReadsTeardown>>& v;
      ~JoinTeardowns() { v.clear(); }
    } joinTeardowns{_teardowns};
    _times = RefinerTimes();
    _errors.assign(n, std::exception_ptr());
    const double t0 = now();
    if (_planThreads <= 1) {
      for (size_t i = 0; i < n; ++i) {
        try {
          plan(svs[i], plans[i], out[i], nullptr);
        } catch (...) {
          plans[i].kind = Plan::NONE;
          recordError(i, out);
        }
      }
    } else {
      // the only state that crosses candidates is the interval filter of planSmall: its verdicts in list order first (no callbacks,
      // pure geometry), then every candidate's reference and read callbacks on the host threads
      std::vector<Route> routes(n);
      for (size_t i = 0; i < n; ++i) {
        try {
          route(svs[i], routes[i]);
        } catch (...) {  // (the same exception comes out of plan() below and is recorded there)
        }
      }
      detail::parallelFor(n, _planThreads, [&](const size_t i) {
        try {
          plan(svs[i], plans[i], out[i], &routes[i]);
        } catch (...) {
          plans[i].kind = Plan::NONE;
          recordError(i, out);
        }
      });
    }
    _times.plan = now() - t0;
    runSmall(plans, isFindLargeInsertions, out);
    runSpanning(plans, out);
    // (the plans hold a copy of every read: the candidates that went to the device are being freed in the background since their piles
    // were flattened -- joined here --, what is left -- candidates without a device stage -- is spread over the host threads as well)
    _teardowns.clear();
    detail::parallelFor(n, _hostThreads, [&](const size_t i) { AssemblyReadInput().swap(plans[i].reads); });
    if (errors) {
      *errors = _errors;
      return;
    }
    for (const std::exception_ptr& e : _errors)
      if (e) std::rethrow_exception(e);
  }

private:
  /// the refiner's own ABI context: its pipelines live and report errors through it, whichever host thread calls
  manta_ctx_t* deviceContext() const
  {
    if (!_ctx) {
      const int rc = manta_ctx_create(-1, &_ctx);
      if (rc != MANTA_OK) throw GeneralException(std::string("manta_amd: no usable GPU context: ") + manta_last_error(nullptr), rc);
    }
    return _ctx;
  }
  void recordError(const size_t candidate, std::vector<SVCandidateAssemblyData>& out) const
  {
    std::lock_guard<std::mutex> g(_errorLock);
    if (!_errors[candidate]) _errors[candidate] = std::current_exception();
    out[candidate] = SVCandidateAssemblyData();
  }
  static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

  struct Plan {
    enum Kind { NONE, SMALL, SPANNING } kind = NONE;
    SVCandidate       sv;  ///< the candidate as the chosen sub-path sees it (single-region form after a transfer)
    AssemblyReadInput reads;
    // small
    pos_t leadingCut = 0, trailingCut = 0, maxLeadingCut = 0, maxTrailingCut = 0;
    // spanning (AlignData, SVCandidateAssemblyRefiner.cpp:1400-1411)
    pos_t align1LeadingCut = 0, align1TrailingCut = 0, align2LeadingCut = 0, align2TrailingCut = 0;
  };

  /// everything that precedes the assembler call: getCandidateAssemblyData (:1051-1082), the head of getJumpAssembly
  /// (:1745-1822) with assembleJumpContigs (:1422-1514), and the head of getSmallSVAssembly (:1860-1926)
  /// what the interval filter of planSmall decided for a candidate (threaded plan: decided up front, in list order)
  struct Route {
    bool decided = false, overlapSkip = false;
  };
  /// is a spanning candidate handed to the local assembler in single-region form (:1803-1817) ?
  bool isTransferToSmall(const SVCandidate& sv) const
  {
    const pos_t extraRefSize = 250 + 100;
    if (sv.bp1.interval.tid == sv.bp2.interval.tid && !SVBreakendState::isSameOrientation(sv.bp1.state, sv.bp2.state))
      return (getSVType(sv) == SV_TYPE::INDEL) && detail::isRefRegionOverlap(_header, extraRefSize, sv);
    return false;
  }
  /// the filter's part of plan(), nothing else: spanning candidates that go to the local assembler add their merged interval, complex
  /// candidates inside such an interval are skipped (planSmall's first lines)
  void route(const SVCandidate& sv, Route& r) const
  {
    if (isSpanningSV(sv)) {
      if (isTransferToSmall(sv)) {
        GenomeInterval merged = sv.bp1.interval;
        merged.range.merge_range(sv.bp2.interval.range);
        _spanToComplexAssmRegions.addInterval(merged);
      }
    } else if (isComplexSV(sv)) {
      r.overlapSkip = _spanToComplexAssmRegions.isSubsetOfRegion(sv.bp1.interval);
    }
    r.decided = true;
  }

  void plan(const SVCandidate& sv, Plan& p, SVCandidateAssemblyData& data, const Route* r) const
  {
    data.clear();
    if (isSpanningSV(sv)) {
      data.isCandidateSpanning = true;
      planJump(sv, p, data, r);
    } else if (isComplexSV(sv)) {
      data.isCandidateSpanning = false;
      planSmall(sv, p, data, r);
    } else {
      throw GeneralException("Unknown candidate SV type");
    }
  }

  void planJump(const SVCandidate& sv, Plan& p, SVCandidateAssemblyData& data, const Route* r) const
  {
    const pos_t extraRefEdgeSize = 250, extraRefSplitSize = 100, extraRefSize = extraRefEdgeSize + extraRefSplitSize;
    if (isTransferToSmall(sv)) {
      // breakend regions too close: hand the problem to the local assembler in single-region form (:1803-1817)
      SVCandidate singleSV = sv;
      singleSV.bp1.state   = SVBreakendState::COMPLEX;
      singleSV.bp2.state   = SVBreakendState::UNKNOWN;
      singleSV.bp1.interval.range.merge_range(sv.bp2.interval.range);
      planSmall(singleSV, p, data, r);
      return;
    }
    data.isSpanning = true;
    BPOrientation& bporient(data.bporient);
    bporient.isBp1First              = sv.isForward();
    {  // SVCandidate::isTranscriptStrandKnown (manta/SVCandidate.hpp:112-118)
      const unsigned f = sv.forwardTranscriptStrandReadCount, r = sv.reverseTranscriptStrandReadCount;
      bporient.isTranscriptStrandKnown = ((std::max(f, r) + 1) / (std::min(f, r) + 1) >= 2);
    }
    if (sv.bp1.state != sv.bp2.state) {
      if (sv.bp2.state == SVBreakendState::RIGHT_OPEN) bporient.isBp2AlignedFirst = true;
    } else {
      if (sv.bp1.state == SVBreakendState::RIGHT_OPEN)
        bporient.isBp2Reversed = true;
      else
        bporient.isBp1Reversed = true;
    }
    if (!detail::isRefRegionValid(_header, sv.bp1.interval)) return;
    if (!detail::isRefRegionValid(_header, sv.bp2.interval)) return;
    unsigned bp1LeadingTrim, bp1TrailingTrim, bp2LeadingTrim, bp2TrailingTrim;
    detail::getIntervalReferenceSegment(_source, _header, extraRefSize, sv.bp1.interval, data.bp1ref, bp1LeadingTrim, bp1TrailingTrim);
    detail::getIntervalReferenceSegment(_source, _header, extraRefSize, sv.bp2.interval, data.bp2ref, bp2LeadingTrim, bp2TrailingTrim);
    p.align1LeadingCut  = std::max(0, extraRefSplitSize - pos_t(bp1LeadingTrim));
    p.align1TrailingCut = std::max(0, extraRefSplitSize - pos_t(bp1TrailingTrim));
    p.align2LeadingCut  = std::max(0, extraRefSplitSize - pos_t(bp2LeadingTrim));
    p.align2TrailingCut = std::max(0, extraRefSplitSize - pos_t(bp2TrailingTrim));
    // manta/SVCandidateAssembler.cpp:677-698
    _source.getBreakendReads(sv.bp1, bporient.isBp1Reversed, data.bp1ref, p.reads);
    _source.getBreakendReads(sv.bp2, bporient.isBp2Reversed, data.bp2ref, p.reads);
    p.kind = Plan::SPANNING;
    p.sv   = sv;
  }

  void planSmall(const SVCandidate& sv, Plan& p, SVCandidateAssemblyData& data, const Route* r) const
  {
    data.isSpanning = false;
    if (r && r->decided) {  // (threaded plan: the filter was applied in list order before the callbacks were spread over threads)
      if (!data.isCandidateSpanning && r->overlapSkip) {
        data.isOverlapSkip = true;
        return;
      }
    } else if (data.isCandidateSpanning) {
      _spanToComplexAssmRegions.addInterval(sv.bp1.interval);
    } else if (_spanToComplexAssmRegions.isSubsetOfRegion(sv.bp1.interval)) {
      data.isOverlapSkip = true;
      return;
    }
    const pos_t extraRefEdgeSize = 700, extraRefSplitSize = 100, extraRefSize = extraRefEdgeSize + extraRefSplitSize;
    if (!detail::isRefRegionValid(_header, sv.bp1.interval)) return;
    unsigned leadingTrim, trailingTrim;
    detail::getIntervalReferenceSegment(_source, _header, extraRefSize, sv.bp1.interval, data.bp1ref, leadingTrim, trailingTrim);
    p.maxLeadingCut  = std::max(0, extraRefSize - pos_t(leadingTrim));
    p.maxTrailingCut = std::max(0, extraRefSize - pos_t(trailingTrim));
    p.leadingCut     = std::max(0, p.maxLeadingCut - extraRefEdgeSize);
    p.trailingCut    = std::max(0, p.maxTrailingCut - extraRefEdgeSize);
    // manta/SVCandidateAssembler.cpp:661-675 (remote-read retrieval happens inside the read scan, beyond the boundary)
    _source.getBreakendReads(sv.bp1, false, data.bp1ref, p.reads);
    p.kind = Plan::SMALL;
    p.sv   = sv;
  }

  // ---------------------------------------------------------------------------------------------------------------
  // complex candidates: getSmallSVAssembly (:1921-2303)
  // ---------------------------------------------------------------------------------------------------------------
  struct ContigScoringInfo {  // :1851-1857
    bool     isDefined = false;
    int      score = 0;
    unsigned index = 0, variantSize = 0;
    bool     isJumped = false;
  };
  struct LargeInsertionWork {
    size_t                planIndex = 0;
    unsigned              leftIndex = 0, rightIndex = 0;
    std::set<pos_t>       insPos;
    AssembledContig       fakeContig;
    detail::AlignJob      job;
  };

  /// frees the read copies of the given plans on half the host threads, in the background (the plans are the batch call's own objects; the
  /// device stage and the per-candidate glue that follow only read the flattened piles and the results).  The batch call joins them before
  /// it returns (`_teardowns`, cleared by a guard declared behind the plans); the destructor waits.
  struct ReadsTeardown {
    std::future<void>   done;
    std::vector<size_t> mine;  ///< (the caller's index list is a local of runSmall / runSpanning: this object outlives it)
    ReadsTeardown(const SVCandidateAssemblyRefiner& r, const std::vector<Plan>& plans, const std::vector<size_t>& which) : mine(which)
    {
      const unsigned             threads = std::max(1u, r._hostThreads / 2);
      const std::vector<size_t>* idx     = &mine;
      done = std::async(std::launch::async, [&plans, idx, threads] {
        detail::parallelFor(idx->size(), threads, [&](const size_t w) { AssemblyReadInput().swap(const_cast<Plan&>(plans[(*idx)[w]]).reads); });
      });
    }
    void wait()
    {
      if (done.valid()) done.get();
    }
    ~ReadsTeardown()
    {
      if (done.valid()) done.wait();
    }
  };
  void runSmall(const std::vector<Plan>& plans, const bool isFindLargeInsertions, std::vector<SVCandidateAssemblyData>& out) const
  {
    const double                    tStart = now();
    std::vector<size_t>             which;
    detail::PackedReads             packed;
    std::vector<const std::string*> refs;
    std::vector<manta_ref_cuts_t>   cuts;
    const unsigned                  maxAsm = _opt.refineOpt.smallSVAssembleOpt.maxAssemblyCount;
    for (size_t i = 0; i < plans.size(); ++i) {
      if (plans[i].kind != Plan::SMALL) continue;
      which.push_back(i);
      packed.addLocus(plans[i].reads, maxAsm);
      refs.push_back(&out[i].bp1ref.seq());
      cuts.push_back(manta_ref_cuts_t{plans[i].leadingCut, plans[i].trailingCut, plans[i].maxLeadingCut, plans[i].maxTrailingCut});
    }
    if (which.empty()) return;
    if (_pileDump)
      for (size_t w = 0; w < which.size(); ++w)
        _pileDump->small(_opt.refineOpt.smallSVAssembleOpt, _opt.refineOpt.largeSVAlignScores, _opt.refineOpt.largeGapOpenScore, plans[which[w]].reads, *refs[w],
                         cuts[w].leading_cut, cuts[w].trailing_cut, cuts[w].max_leading_cut, cuts[w].max_trailing_cut);
    packed.finish(_hostThreads, &_stage);
    _stats.smallLoci += which.size();
    // the plans' own copies of the reads (80 strings per candidate) are flattened now: their teardown -- a tenth of a second of `free` on a
    // large batch -- runs on a few host threads WHILE the device works, not behind it
    _teardowns.emplace_back(new ReadsTeardown(*this, plans, which));  // (joined at the end of the batch call)
    const double          tPacked = now();
    _times.pack += tPacked - tStart;
    detail::SmallSvOutput& dev(_smallDev);  // (kept across calls: no re-allocation and zero-fill of tens of MB per batch)
    detail::smallSvBatch(deviceContext(), _smallPipe, _opt.refineOpt.smallSVAssembleOpt, _opt.refineOpt.largeSVAlignScores, _opt.refineOpt.largeGapOpenScore, packed,
                         refs, cuts, dev, _hostThreads, _refStage);
    const double tDevice = now();
    _times.device += tDevice - tPacked;

    std::vector<std::unique_ptr<LargeInsertionWork>> liWorkByLocus(which.size());
    std::atomic<uint64_t>                            nContigs(0);
    auto smallLocus = [&](const size_t w) {
      const Plan&              p(plans[which[w]]);
      SVCandidateAssemblyData& data(out[which[w]]);
      dev.toContigs(unsigned(w), data.contigs);
      nContigs += data.contigs.size();
      const std::string& align1RefStr(data.bp1ref.seq());
      const unsigned     contigCount = unsigned(data.contigs.size());
      data.smallSVAlignments.resize(contigCount);
      data.smallSVSegments.resize(contigCount);
      data.largeInsertInfo.resize(contigCount);
      data.extendedContigs.resize(contigCount);
      ContigScoringInfo     rank1Contig, rank2Contig;
      std::vector<unsigned> largeInsertionCandidateIndex;

      for (unsigned contigIndex = 0; contigIndex < contigCount; ++contigIndex) {
        const AssembledContig&                 contig(data.contigs[contigIndex]);
        AlignmentResult<int>&                  alignment(data.smallSVAlignments[contigIndex]);
        std::vector<segment_t>&                candidateSegments(data.smallSVSegments[contigIndex]);
        const manta_smallsv_alignment_t&       da(dev.aligns[dev.loci[w].first_contig + contigIndex]);
        if (da.align.status != MANTA_OK) throw GeneralException("manta_amd small-SV pipeline: contig alignment failed on the device", da.align.status);
        alignment.clear();
        alignment.score          = da.align.score;
        alignment.isJumped       = da.align.is_jumped != 0;
        alignment.align.beginPos = da.align.begin_pos1;  // adjustedLeadingCut already added on the device (:2039)
        detail::toPath(dev.cigar.data() + da.align.cigar1_off, da.align.cigar1_len, alignment.align.apath);
        getExtendedContig(alignment, contig.seq, align1RefStr, data.extendedContigs[contigIndex]);

        const bool isSmallSVCandidate = findSmallSVCandidateSegments(
            _opt.refineOpt.contigFilterScores, alignment.align, contig.seq, align1RefStr, _opt.scanOpt.minCandidateVariantSize,
            candidateSegments);

        if (isFindLargeInsertions) {  // :2072-2113
          LargeInsertionInfo insertInfo;
          ALIGNPATH::path_t  apath_conservative(alignment.align.apath);
          detail::apath_limit_read_length(unsigned(std::max(contig.conservativeRange.begin_pos(), 0)),
                                          unsigned(std::max(contig.conservativeRange.end_pos(), 0)), apath_conservative);
          bool isCandidate = isLargeInsertAlignment(_opt.refineOpt.largeInsertEdgeAlignScores, apath_conservative, insertInfo);
          if (isCandidate) {
            LargeInsertionInfo insertInfo2;
            isCandidate = isLargeInsertAlignment(_opt.refineOpt.largeInsertEdgeAlignScores, alignment.align.apath, insertInfo2);
            if (insertInfo.isLeftCandidate != insertInfo2.isLeftCandidate || insertInfo.isRightCandidate != insertInfo2.isRightCandidate)
              isCandidate = false;
            insertInfo.contigOffset = insertInfo2.contigOffset;
            insertInfo.refOffset    = insertInfo2.refOffset;
          }
          if (isCandidate) {
            data.largeInsertInfo[contigIndex] = insertInfo;
            largeInsertionCandidateIndex.push_back(contigIndex);
          }
        }

        if (isSmallSVCandidate) {  // :2115-2166
          auto refresh = [&](ContigScoringInfo& info) {
            info.isDefined   = true;
            info.index       = contigIndex;
            info.score       = alignment.score;
            info.variantSize = getLargestIndelSize(alignment.align.apath, candidateSegments);
            info.isJumped    = alignment.isJumped;
          };
          const bool bothJumped    = alignment.isJumped && rank1Contig.isJumped;
          const bool bothNotJumped = (!alignment.isJumped) && (!rank1Contig.isJumped);
          if (!rank1Contig.isDefined || (alignment.isJumped && !rank1Contig.isJumped) ||
              ((bothJumped || bothNotJumped) && alignment.score > rank1Contig.score)) {
            if (rank1Contig.isDefined) rank2Contig = rank1Contig;
            refresh(rank1Contig);
          } else if (!rank2Contig.isDefined || alignment.score > rank2Contig.score) {
            refresh(rank2Contig);
          }
        }
      }

      if (rank2Contig.isDefined) {  // :2169-2219
        const unsigned     n1 = unsigned(data.contigs[rank1Contig.index].supportReads.size());
        const unsigned     n2 = unsigned(data.contigs[rank2Contig.index].supportReads.size());
        static const float minScoreRatio(0.9f), minSupportReadCountRatio(1.2f), minVariantSizeRatio(1.1f);
        const bool         rank1IsSelected = rank1Contig.isJumped && !rank2Contig.isJumped;
        if (!rank1IsSelected) {
          const bool rank2IsBest = (rank2Contig.score > (rank1Contig.score * minScoreRatio)) &&
                                   ((n2 > (n1 * minSupportReadCountRatio)) || (rank2Contig.variantSize > (rank1Contig.variantSize * minVariantSizeRatio)));
          if (rank2IsBest) rank1Contig = rank2Contig;
        }
      }

      std::set<pos_t> insPos;
      if (rank1Contig.isDefined) {  // :2226-2281
        data.bestAlignmentIndex = rank1Contig.index;
        const AssembledContig&        bestContig(data.contigs[data.bestAlignmentIndex]);
        const AlignmentResult<int>&   bestAlign(data.smallSVAlignments[data.bestAlignmentIndex]);
        const std::vector<segment_t>  segs(data.smallSVSegments[data.bestAlignmentIndex]);
        unsigned                      segmentIndex = 0;
        for (const segment_t& segRange : segs) {
          data.svs.push_back(p.sv);
          SVCandidate& newSV(data.svs.back());
          newSV.assemblyAlignIndex   = data.bestAlignmentIndex;
          newSV.assemblySegmentIndex = segmentIndex++;
          detail::setSmallCandSV(data.bp1ref, bestContig.seq, bestAlign.align, segRange, newSV, _opt);
          if (getExtendedSVType(newSV) == EXTENDED_SV_TYPE::INSERT) insPos.insert(newSV.bp1.interval.range.begin_pos());
        }
      }

      if (isFindLargeInsertions) {
        std::unique_ptr<LargeInsertionWork> work(new LargeInsertionWork);
        if (planLargeInsertion(p, data, largeInsertionCandidateIndex, *work)) {
          work->planIndex = which[w];
          work->insPos    = insPos;
          liWorkByLocus[w] = std::move(work);
        }
      }
    };
    // a candidate's failure (the reference would throw from its getCandidateAssemblyData call) stays that candidate's
    detail::parallelFor(which.size(), _hostThreads, [&](const size_t w) {
      try {
        smallLocus(w);
      } catch (...) {
        recordError(which[w], out);
      }
    });
    _stats.contigAlignments += nContigs;
    std::vector<std::unique_ptr<LargeInsertionWork>> liWork;
    for (auto& w : liWorkByLocus)
      if (w) liWork.push_back(std::move(w));

    // large-insertion completion: one GlobalAligner batch over every locus that found a left/right pair
    std::vector<detail::AlignJob*> jobs;
    for (auto& w : liWork) jobs.push_back(&w->job);
    _stats.largeInsertionAlignments += jobs.size();
    detail::alignBatch(MANTA_ALIGNER_GLOBAL, _opt.refineOpt.largeInsertCompleteAlignScores, 0, jobs);
    for (auto& w : liWork) finishLargeInsertion(plans[w->planIndex], *w, out[w->planIndex]);
    _times.post += now() - tDevice;
  }

  /// processLargeInsertion, first half (:833-935): choose the left/right pair and set up the fake-contig alignment
  bool planLargeInsertion(
      const Plan& p, const SVCandidateAssemblyData& data, const std::vector<unsigned>& candIndex, LargeInsertionWork& work) const
  {
    if (candIndex.empty()) return false;
    bool             isPair = false;
    int              bestBreakDist = 0, bestBreakScore = 0;
    static const int maxBreakDist(35);
    const unsigned   candCount = unsigned(candIndex.size());
    for (unsigned c1 = 0; (c1 + 1) < candCount; ++c1) {
      const unsigned            i1 = candIndex[c1];
      const Alignment&          align1(data.smallSVAlignments[i1].align);
      const LargeInsertionInfo& insert1(data.largeInsertInfo[i1]);
      for (unsigned c2 = c1 + 1; c2 < candCount; ++c2) {
        const unsigned            i2 = candIndex[c2];
        const Alignment&          align2(data.smallSVAlignments[i2].align);
        const LargeInsertionInfo& insert2(data.largeInsertInfo[i2]);
        if (!((insert1.isLeftCandidate && insert2.isRightCandidate) || (insert2.isLeftCandidate && insert1.isRightCandidate))) continue;
        const int breakDist = int(std::labs(long(align1.beginPos + pos_t(insert1.refOffset)) - long(align2.beginPos + pos_t(insert2.refOffset))));
        if (breakDist > maxBreakDist) continue;
        const int  breakScore = insert1.score + insert2.score;
        const bool isBetter   = (breakDist < bestBreakDist) || ((breakDist == bestBreakDist) && (breakScore > bestBreakScore));
        if (!isPair || isBetter) {
          isPair          = true;
          work.leftIndex  = i1;
          work.rightIndex = i2;
          if (insert1.isRightCandidate) std::swap(work.leftIndex, work.rightIndex);
          bestBreakDist  = breakDist;
          bestBreakScore = breakScore;
        }
      }
    }
    if (!isPair) return false;
    static const std::string middle(100, 'N');
    work.fakeContig = data.contigs[work.leftIndex];
    work.fakeContig.seq += (middle + data.contigs[work.rightIndex].seq);
    const std::string& ref(data.bp1ref.seq());
    work.job.query   = &work.fakeContig.seq;
    work.job.ref1    = ref.data() + p.leadingCut;
    work.job.ref1Len = ref.size() - size_t(p.leadingCut) - size_t(p.trailingCut);
    return true;
  }

  /// processLargeInsertion, second half (:917-1006)
  void finishLargeInsertion(const Plan& p, LargeInsertionWork& work, SVCandidateAssemblyData& data) const
  {
    const unsigned     middleSize  = 100;
    const unsigned     contigCount = unsigned(data.contigs.size());
    const std::string  leftSeq(data.contigs[work.leftIndex].seq), rightSeq(data.contigs[work.rightIndex].seq);
    data.contigs.resize(contigCount + 1);
    data.smallSVAlignments.resize(contigCount + 1);
    data.smallSVSegments.resize(contigCount + 1);
    data.extendedContigs.resize(contigCount + 1);
    data.contigs[contigCount] = work.fakeContig;
    AlignmentResult<int>&   fakeAlignment(data.smallSVAlignments[contigCount]);
    std::vector<segment_t>& fakeSegments(data.smallSVSegments[contigCount]);
    detail::toResult(work.job, fakeAlignment);
    fakeAlignment.align.beginPos += p.leadingCut;
    fakeSegments.clear();
    getLargestInsertSegment(fakeAlignment.align.apath, middleSize, fakeSegments);
    if (fakeSegments.size() != 1 || fakeSegments[0].second < fakeSegments[0].first) return;
    if (!detail::isFinishedLargeInsertAlignment(_opt.refineOpt.largeInsertCompleteAlignScores, fakeAlignment.align.apath, fakeSegments[0], middleSize))
      return;
    const known_pos_range2 insertTrim(detail::getInsertTrim(fakeAlignment.align.apath, fakeSegments[0]));
    static const int       minFlankSize(40);  // minSemiLargeInsertionLength, SVCandidateAssemblyRefiner.cpp:558
    if ((insertTrim.begin_pos() + minFlankSize) > pos_t(leftSeq.size())) return;
    const pos_t rightOffset = pos_t(leftSeq.size() + middleSize);
    if ((rightOffset + minFlankSize) > insertTrim.end_pos()) return;
    getExtendedContig(fakeAlignment, work.fakeContig.seq, data.bp1ref.seq(), data.extendedContigs[contigCount]);
    SVCandidate newSV(p.sv);
    newSV.assemblyAlignIndex   = contigCount;
    newSV.assemblySegmentIndex = 0;
    detail::setSmallCandSV(data.bp1ref, work.fakeContig.seq, fakeAlignment.align, fakeSegments[0], newSV, _opt);
    if (work.insPos.count(newSV.bp1.interval.range.begin_pos())) return;
    newSV.isUnknownSizeInsertion       = true;
    newSV.unknownSizeInsertionLeftSeq  = leftSeq.substr(size_t(insertTrim.begin_pos()));
    newSV.unknownSizeInsertionRightSeq = rightSeq.substr(0, size_t(insertTrim.end_pos() - rightOffset));
    data.svs.push_back(newSV);
  }

  // ---------------------------------------------------------------------------------------------------------------
  // spanning candidates: alignJumpContigs (:1525-1743), selectJumpContigDNA (:1364-1398),
  // generateRefinedVCFSVCandidateFromJumpAlignment (:1175-1250)
  // ---------------------------------------------------------------------------------------------------------------
  struct SpanningLocus {
    size_t             planIndex = 0;
    std::string        bp1refSeq, bp2refSeq;  ///< orientation applied
    const std::string *align1Ref = nullptr, *align2Ref = nullptr;
    pos_t              a1Lead = 0, a1Trail = 0, a2Lead = 0, a2Trail = 0;
  };

  void runSpanning(const std::vector<Plan>& plans, std::vector<SVCandidateAssemblyData>& out) const
  {
    const double               tStart = now();
    std::vector<SpanningLocus> loci;
    detail::PackedReads        packed;
    const unsigned             maxAsm = _opt.refineOpt.spanningAssembleOpt.maxAssemblyCount;
    for (size_t i = 0; i < plans.size(); ++i) {
      if (plans[i].kind != Plan::SPANNING) continue;
      loci.emplace_back();
      loci.back().planIndex = i;
      packed.addLocus(plans[i].reads, maxAsm);
    }
    if (loci.empty()) return;
    packed.finish(_hostThreads, &_stage);
    _stats.spanningLoci += loci.size();
    std::vector<size_t> spanPlans;
    for (const SpanningLocus& sl : loci) spanPlans.push_back(sl.planIndex);

    // orientation step of alignJumpContigs (:1533-1550)
    std::vector<const std::string*> refs1, refs2;
    std::vector<manta_jump_cuts_t>  cuts;
    for (SpanningLocus& sl : loci) {
      const Plan&                    p(plans[sl.planIndex]);
      const SVCandidateAssemblyData& data(out[sl.planIndex]);
      sl.bp1refSeq = data.bp1ref.seq();
      sl.bp2refSeq = data.bp2ref.seq();
      sl.a1Lead = p.align1LeadingCut, sl.a1Trail = p.align1TrailingCut, sl.a2Lead = p.align2LeadingCut, sl.a2Trail = p.align2TrailingCut;
      if (data.bporient.isBp1Reversed) {
        reverseCompStr(sl.bp1refSeq);
        std::swap(sl.a1Lead, sl.a1Trail);
      }
      if (data.bporient.isBp2Reversed) {
        reverseCompStr(sl.bp2refSeq);
        std::swap(sl.a2Lead, sl.a2Trail);
      }
      sl.align1Ref = &sl.bp1refSeq;
      sl.align2Ref = &sl.bp2refSeq;
      if (data.bporient.isBp2AlignedFirst) {
        std::swap(sl.align1Ref, sl.align2Ref);
        std::swap(sl.a1Lead, sl.a2Lead);
        std::swap(sl.a1Trail, sl.a2Trail);
      }
    }
    for (const SpanningLocus& sl : loci) {  // pointers taken only now: `loci` no longer reallocates
      refs1.push_back(sl.align1Ref);
      refs2.push_back(sl.align2Ref);
      cuts.push_back(manta_jump_cuts_t{sl.a1Lead, sl.a1Trail, sl.a2Lead, sl.a2Trail});
    }

    if (_pileDump)
      for (size_t l = 0; l < loci.size(); ++l)
        _pileDump->spanning(_opt.refineOpt.spanningAssembleOpt, _opt.refineOpt.spanningAlignScores, _opt.refineOpt.jumpScore, plans[loci[l].planIndex].reads, *refs1[l],
                            *refs2[l], cuts[l].align1_leading_cut, cuts[l].align1_trailing_cut, cuts[l].align2_leading_cut, cuts[l].align2_trailing_cut);
    _teardowns.emplace_back(new ReadsTeardown(*this, plans, spanPlans));  // (see runSmall; the pile dump above was the reads' last user)
    // assemble -> jump-align (cut references) -> re-align rule -> jump-align (uncut), all on the device
    const double           tPacked = now();
    _times.pack += tPacked - tStart;
    detail::SpanningOutput& dev(_spanDev);
    detail::spanningBatch(deviceContext(), _spanPipe, _opt.refineOpt.spanningAssembleOpt, _opt.refineOpt.spanningAlignScores, _opt.refineOpt.jumpScore, packed, refs1,
                          refs2, cuts, dev, _hostThreads, _refStage, _ref2Stage);
    const double tDevice = now();
    _times.device += tDevice - tPacked;

    std::atomic<uint64_t> nContigs(0), nRealigned(0);
    auto spanningLocus = [&](const size_t l) {
      const SpanningLocus&     sl(loci[l]);
      const Plan&              p(plans[sl.planIndex]);
      SVCandidateAssemblyData& data(out[sl.planIndex]);
      dev.toContigs(unsigned(l), data.contigs);
      const unsigned contigCount = unsigned(data.contigs.size());
      nContigs += contigCount;
      data.spanningAlignments.resize(contigCount);
      for (unsigned c = 0; c < contigCount; ++c) {
        JumpAlignmentResult<int>&         alignment(data.spanningAlignments[c]);
        const manta_spanning_alignment_t& da(dev.aligns[dev.loci[l].first_contig + c]);
        if (da.align.status == MANTA_E_EMPTY_SEQ)  // the reference throws from GlobalJumpAligner::align (GlobalJumpAlignerImpl.hpp:50-58)
          throw GeneralException("Unexpected empty reference sequence");
        if (da.align.status != MANTA_OK) throw GeneralException("manta_amd spanning pipeline: contig alignment failed on the device", da.align.status);
        nRealigned += da.is_uncut ? 1 : 0;
        alignment.clear();
        alignment.score           = da.align.score;
        alignment.jumpInsertSize  = da.align.jump_insert_size;
        alignment.jumpRange       = da.align.jump_range;
        alignment.align1.beginPos = da.align.begin_pos1;  // leading cuts already added on the device side (:1716-1717)
        alignment.align2.beginPos = da.align.begin_pos2;
        detail::toPath(dev.cigar.data() + da.align.cigar1_off, da.align.cigar1_len, alignment.align1.apath);
        detail::toPath(dev.cigar.data() + da.align.cigar2_off, da.align.cigar2_len, alignment.align2.apath);
        std::string extendedContig;
        getExtendedContig(alignment, data.contigs[c].seq, *sl.align1Ref, *sl.align2Ref, extendedContig);
        data.extendedContigs.push_back(extendedContig);
      }
      const int best = selectJumpContigDNA(data.spanningAlignments, _opt.refineOpt.contigFilterScores);
      if (best < 0) return;  // bestAlignmentIndex stays 0, no refined candidate (:1392-1397, 1829)
      data.bestAlignmentIndex = unsigned(best);
      data.svs.push_back(p.sv);
      SVCandidate&                    sv(data.svs.back());
      const JumpAlignmentResult<int>& align(data.spanningAlignments[data.bestAlignmentIndex]);
      const Alignment*                bp1AlignPtr(&align.align1);
      const Alignment*                bp2AlignPtr(&align.align2);
      if (data.bporient.isBp2AlignedFirst) std::swap(bp1AlignPtr, bp2AlignPtr);
      sv.assemblyAlignIndex   = data.bestAlignmentIndex;
      sv.assemblySegmentIndex = 0;
      sv.setPrecise();
      detail::adjustAssembledBreakend(*bp1AlignPtr, !data.bporient.isBp2AlignedFirst, align.jumpRange, data.bp1ref,
                                      data.bporient.isBp1Reversed, sv.bp1);
      detail::adjustAssembledBreakend(*bp2AlignPtr, data.bporient.isBp2AlignedFirst, align.jumpRange, data.bp2ref,
                                      data.bporient.isBp2Reversed, sv.bp2);
      sv.insertSeq.clear();
      if (align.jumpInsertSize > 0)
        getFwdStrandInsertSegment(align, data.contigs[data.bestAlignmentIndex].seq, data.bporient.isBp1Reversed, sv.insertSeq);
      if (_opt.isOutputContig) sv.contigSeq = data.contigs[data.bestAlignmentIndex].seq;
      detail::addCigarToSpanningAlignment(sv);
    };
    detail::parallelFor(loci.size(), _hostThreads, [&](const size_t l) {
      try {
        spanningLocus(l);
      } catch (...) {
        recordError(loci[l].planIndex, out);
      }
    });
    _stats.contigAlignments += nContigs;
    _stats.realignedContigs += nRealigned;
    _times.post += now() - tDevice;
  }

  const GSCOptions              _opt;
  const bam_header_info         _header;
  RefinerInputSource&           _source;
  mutable GenomeIntervalTracker _spanToComplexAssmRegions;
  mutable Stats                 _stats;
  mutable RefinerTimes          _times;
  mutable std::vector<std::exception_ptr> _errors;  ///< per candidate of the running batch call
  mutable std::mutex                      _errorLock;
  mutable manta_ctx_t*          _ctx       = nullptr;
  mutable manta_smallsv_t*      _smallPipe = nullptr;  ///< device pipelines of this refiner (and of the thread that
  mutable manta_spanning_t*     _spanPipe  = nullptr;  ///< first used it: one ABI context per host thread)
  mutable detail::PinnedArena   _stage;                ///< page-locked staging of a batch's read bases, reused
  mutable detail::PinnedArena   _refStage, _ref2Stage; ///< ... and of its reference windows
  mutable detail::SmallSvOutput  _smallDev;            ///< the device results of the last batch (reused: capacity stays)
  mutable detail::SpanningOutput _spanDev;
  unsigned                      _hostThreads = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
  unsigned                      _planThreads = 1;
  PileDumpWriter*               _pileDump = nullptr;
  mutable std::vector<std::unique_ptr<ReadsTeardown>> _teardowns;  ///< background frees of the current batch call
};

}  // namespace manta_amd
