// C++ host adapter: the reference's own object interface for the hot path, on top of the C ABI
// (include/manta_amd.h).  A Manta source file that today includes
//     assembly/IterativeAssembler.hpp, alignment/Global{,LargeIndel,Jump}Aligner.hpp
// can include this header instead and `using namespace manta_amd;`: the type and function names,
// argument meaning, ownership (callee clears then fills) and error behaviour (exceptions) are the reference's:
//   runIterativeAssembler            /root/reference/src/c++/lib/assembly/IterativeAssembler.hpp:43-47
//   AssembledContig / Assembly       assembly/AssembledContig.hpp:38-54
//   AssemblyReadInfo & typedefs      assembly/AssemblyReadInfo.hpp:31-46
//   IterativeAssemblerOptions        options/IterativeAssemblerOptions.hpp:26-59
//   AlignmentScores<T>               alignment/AlignmentScores.hpp:23-57
//   ALIGNPATH::{align_t,path_segment,path_t,apath_to_cigar}   blt_util/align_path.hpp:35-163
//   Alignment, AlignmentResult<T>, JumpAlignmentResult<T>     alignment/Alignment.hpp:27-45,
//                                    SingleRefAlignerShared.hpp:30-46, JumpAlignerBase.hpp:37-56
//   GlobalAligner<T>, GlobalLargeIndelAligner<T>, GlobalJumpAligner<T>  (align() signatures as the reference's)
// All arithmetic happens in the HIP kernels behind the ABI; this header only converts containers.
// Like the reference's aligner objects (mutable scratch, GlobalJumpAligner.hpp:117-124) the adapter is not
// re-entrant per thread: each host thread owns one lazily created context (GenerateSVCandidates.cpp:232-250).
#pragma once

#include <algorithm>
#include <cstdint>
#include <iterator>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/manta_amd.h"

namespace manta_amd {

typedef int32_t pos_t;

/// illumina::common::GeneralException stand-in (common/Exceptions.hpp:54-85): a std::logic_error
struct GeneralException : public std::logic_error {
  explicit GeneralException(const std::string& msg, int code = 0) : std::logic_error(msg), errorCode(code) {}
  int errorCode;
};

namespace detail {
struct ThreadContextHolder {
  manta_ctx_t* ctx = nullptr;
  ~ThreadContextHolder() { manta_ctx_destroy(ctx); }
};
// internal linkage on purpose: a function-local static of an inline function would be ONE object per process
// (STB_GNU_UNIQUE), shared even between two differently built copies of this library loaded side by side (the test
// suite loads the device build and the wave-emulator build into one process)
namespace {
thread_local ThreadContextHolder g_threadContext;
}
}  // namespace detail

/// one ABI context per host thread (and per translation unit that includes this header)
inline manta_ctx_t* threadContext()
{
  detail::ThreadContextHolder& h(detail::g_threadContext);
  if (!h.ctx) {
    const int rc = manta_ctx_create(-1, &h.ctx);
    if (rc != MANTA_OK) throw GeneralException(std::string("manta_amd: no usable GPU context: ") + manta_last_error(nullptr), rc);
  }
  return h.ctx;
}

// ------------------------------------------------------------------------------------------------------
// assembly
// ------------------------------------------------------------------------------------------------------
struct known_pos_range2 {  // blt_util/known_pos_range2.hpp:33-135 (the members the path uses)
  known_pos_range2() {}
  known_pos_range2(const pos_t b, const pos_t e) : _begin(b), _end(e) {}
  void  set_begin_pos(pos_t p) { _begin = p; }
  void  set_end_pos(pos_t p) { _end = p; }
  void  set_range(pos_t b, pos_t e)
  {
    _begin = b;
    _end   = e;
  }
  pos_t    begin_pos() const { return _begin; }
  pos_t    end_pos() const { return _end; }
  unsigned size() const { return unsigned(_end > _begin ? _end - _begin : 0); }
  pos_t    center_pos() const { return _begin + pos_t((std::max(size(), 1u) - 1) / 2); }
  bool     is_range_intersect(const known_pos_range2& pr) const { return (pr._end > _begin) && (pr._begin < _end); }
  void     merge_range(const known_pos_range2& kpr)
  {
    if (kpr._begin < _begin) _begin = kpr._begin;
    if (kpr._end > _end) _end = kpr._end;
  }
  pos_t _begin = 0, _end = 0;
};

struct AssembledContig {
  std::string        seq;
  unsigned           seedReadCount = 0;
  std::set<unsigned> supportReads;
  std::set<unsigned> rejectReads;
  known_pos_range2   conservativeRange;
};
typedef std::vector<AssembledContig> Assembly;

struct AssemblyReadInfo {
  bool                  isUsed     = false;
  bool                  isFiltered = false;
  bool                  isPseudo   = false;
  std::vector<unsigned> contigIds;
};
typedef std::vector<std::string>      AssemblyReadInput;
typedef std::vector<bool>             AssemblyReadReversal;
typedef std::vector<AssemblyReadInfo> AssemblyReadOutput;

struct IterativeAssemblerOptions {
  std::string alphabet                = "ACGT";
  int         minQval                 = 5;
  unsigned    minWordLength           = 41;
  unsigned    maxWordLength           = 76;
  unsigned    wordStepSize            = 5;
  unsigned    minContigLength         = 15;
  unsigned    minCoverage             = 1;
  unsigned    minConservativeCoverage = 2;
  double      maxError                = 0.35;
  unsigned    minUnusedReads          = 3;
  unsigned    minSupportReads         = 2;
  unsigned    maxAssemblyCount        = 10;
};

inline manta_asm_options_t toAbi(const IterativeAssemblerOptions& o)
{
  if (o.alphabet != "ACGT") throw GeneralException("manta_amd: only the default assembly alphabet \"ACGT\" is supported");
  return manta_asm_options_t{o.minWordLength, o.maxWordLength, o.wordStepSize, o.minContigLength, o.minCoverage,
                             o.minConservativeCoverage, o.minUnusedReads, o.minSupportReads, o.maxAssemblyCount};
}

namespace detail {
inline void bitsToSet(const uint64_t* w, unsigned nWords, std::set<unsigned>& out)
{
  out.clear();
  for (unsigned i = 0; i < nWords; ++i) {
    uint64_t x = w[i];
    while (x) {
      const unsigned b = unsigned(__builtin_ctzll(x));
      out.insert(out.end(), i * 64 + b);
      x &= x - 1;
    }
  }
}
}  // namespace detail

/// assembly/IterativeAssembler.hpp:43-47 -- same contract: `reads` is in-out (pseudo reads may remain appended,
/// IterativeAssembler.cpp:902), `assembledReadInfo` and `contigs` are cleared then filled.
inline void runIterativeAssembler(
    const IterativeAssemblerOptions& opt, AssemblyReadInput& reads, AssemblyReadOutput& assembledReadInfo, Assembly& contigs)
{
  manta_ctx_t*              ctx = threadContext();
  const manta_asm_options_t o   = toAbi(opt);
  const unsigned            nReads = unsigned(reads.size());
  std::vector<uint8_t>      bases;
  std::vector<uint64_t>     readOff(nReads + 1, 0);
  for (unsigned r = 0; r < nReads; ++r) {
    bases.insert(bases.end(), reads[r].begin(), reads[r].end());
    readOff[r + 1] = bases.size();
  }
  bases.push_back(0);
  const uint32_t                  locusBegin[2] = {0, nReads};
  manta_asm_locus_result_t        locus;
  std::vector<manta_asm_contig_t> recs(o.max_assembly_count + 1);
  std::vector<uint8_t>            seqArena(bases.size() * (2 * size_t(o.max_assembly_count) + 2) + 65536);
  std::vector<uint64_t>           bitsArena(size_t(o.max_assembly_count) * 2 * 16 + 2 * o.max_assembly_count + 64);
  uint64_t                        seqUsed = 0, bitsUsed = 0;
  const int rc = manta_assemble_batch(
      ctx, &o, 1, bases.data(), readOff.data(), locusBegin, &locus, recs.data(), recs.size(), seqArena.data(), seqArena.size(),
      &seqUsed, bitsArena.data(), bitsArena.size(), &bitsUsed);
  if (rc != MANTA_OK) throw GeneralException(std::string("manta_amd::runIterativeAssembler: ") + manta_last_error(ctx), rc);

  contigs.clear();
  contigs.resize(locus.n_contigs);
  for (unsigned c = 0; c < locus.n_contigs; ++c) {
    const manta_asm_contig_t& rc2(recs[locus.first_contig + c]);
    AssembledContig&          ctg(contigs[c]);
    ctg.seq.assign(reinterpret_cast<const char*>(seqArena.data() + rc2.seq_off), rc2.seq_len);
    ctg.seedReadCount = rc2.seed_read_count;
    detail::bitsToSet(bitsArena.data() + rc2.support_off, locus.n_words, ctg.supportReads);
    detail::bitsToSet(bitsArena.data() + rc2.reject_off, locus.n_words, ctg.rejectReads);
    ctg.conservativeRange.set_begin_pos(rc2.conservative_begin);
    ctg.conservativeRange.set_end_pos(rc2.conservative_end);
  }
  // pseudo reads the reference leaves behind in `reads`
  uint64_t off = locus.pseudo_seq_off;
  for (unsigned p = 0; p < locus.n_pseudo; ++p) {
    const uint64_t len = bitsArena[locus.pseudo_len_off + p];
    reads.emplace_back(reinterpret_cast<const char*>(seqArena.data() + off), len);
    off += len;
  }
  // readInfo (IterativeAssembler.cpp:826-834)
  assembledReadInfo.clear();
  assembledReadInfo.resize(reads.size());
  for (unsigned r = nReads; r < reads.size(); ++r) assembledReadInfo[r].isPseudo = true;
  for (unsigned c = 0; c < contigs.size(); ++c) {
    for (const unsigned rd : contigs[c].supportReads) {
      if (rd >= assembledReadInfo.size()) continue;  // stale pseudo-read index
      assembledReadInfo[rd].isUsed = true;
      assembledReadInfo[rd].contigIds.push_back(c);
    }
  }
}

/// options/SmallAssemblerOptions.hpp:24-56
struct SmallAssemblerOptions {
  std::string alphabet                = "ACGT";
  uint8_t     minQval                 = 5;
  unsigned    minWordLength           = 41;
  unsigned    maxWordLength           = 76;
  unsigned    wordStepSize            = 5;
  unsigned    minContigLength         = 15;
  unsigned    minCoverage             = 1;
  unsigned    minConservativeCoverage = 2;
  double      maxError                = 0.35;
  unsigned    minSeedReads            = 3;
  unsigned    maxAssemblyIterations   = 10;
};

/// assembly/SmallAssembler.hpp:43-47 -- same contract: `assembledReadInfo` and `contigs` are cleared then filled.
inline void runSmallAssembler(
    const SmallAssemblerOptions& opt, const AssemblyReadInput& reads, AssemblyReadOutput& assembledReadInfo, Assembly& contigs)
{
  if (opt.alphabet != "ACGT") throw GeneralException("manta_amd: only the default assembly alphabet \"ACGT\" is supported");
  manta_ctx_t*                    ctx = threadContext();
  const manta_small_asm_options_t o{opt.minWordLength, opt.maxWordLength, opt.wordStepSize, opt.minContigLength, opt.minCoverage,
                                    opt.minConservativeCoverage, opt.minSeedReads, opt.maxAssemblyIterations};
  const unsigned        nReads = unsigned(reads.size());
  std::vector<uint8_t>  bases;
  std::vector<uint64_t> readOff(nReads + 1, 0);
  for (unsigned r = 0; r < nReads; ++r) {
    bases.insert(bases.end(), reads[r].begin(), reads[r].end());
    readOff[r + 1] = bases.size();
  }
  bases.push_back(0);
  const uint32_t                  locusBegin[2] = {0, nReads};
  const unsigned                  nSlots = o.max_assembly_iterations + 1;  // one per iteration's contig + the isFiltered record
  manta_asm_locus_result_t        locus;
  std::vector<manta_asm_contig_t> recs(nSlots + 1);
  std::vector<uint8_t>            seqArena(bases.size() * (size_t(nSlots) + 1) + 65536);
  std::vector<uint64_t>           bitsArena(size_t(nSlots) * 2 * 16 + 64);
  uint64_t                        seqUsed = 0, bitsUsed = 0;
  const int rc = manta_small_assemble_batch(
      ctx, &o, 1, bases.data(), readOff.data(), locusBegin, &locus, recs.data(), recs.size(), seqArena.data(), seqArena.size(),
      &seqUsed, bitsArena.data(), bitsArena.size(), &bitsUsed);
  if (rc != MANTA_OK) throw GeneralException(std::string("manta_amd::runSmallAssembler: ") + manta_last_error(ctx), rc);

  contigs.clear();
  assembledReadInfo.clear();
  assembledReadInfo.resize(nReads);
  for (unsigned c = 0; c < locus.n_contigs; ++c) {
    const manta_asm_contig_t& rec(recs[locus.first_contig + c]);
    std::set<unsigned>        support;
    detail::bitsToSet(bitsArena.data() + rec.support_off, locus.n_words, support);
    if (rec.seed_read_count == 0xffffffffu) {  // reads dropped for holding a word twice (SmallAssembler.cpp:496-503)
      for (const unsigned rd : support) {
        assembledReadInfo[rd].isUsed     = true;
        assembledReadInfo[rd].isFiltered = true;
      }
      continue;
    }
    contigs.emplace_back();
    AssembledContig& ctg(contigs.back());
    ctg.seq.assign(reinterpret_cast<const char*>(seqArena.data() + rec.seq_off), rec.seq_len);
    ctg.seedReadCount = rec.seed_read_count;
    ctg.supportReads  = support;
    detail::bitsToSet(bitsArena.data() + rec.reject_off, locus.n_words, ctg.rejectReads);
    ctg.conservativeRange.set_begin_pos(rec.conservative_begin);
    ctg.conservativeRange.set_end_pos(rec.conservative_end);
    for (const unsigned rd : support) {  // :594-606 (a contig's support holds reads that were unused until then)
      assembledReadInfo[rd].isUsed = true;
      assembledReadInfo[rd].contigIds.push_back(unsigned(contigs.size() - 1));
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// alignment
// ------------------------------------------------------------------------------------------------------
template <typename ScoreType>
struct AlignmentScores {
  AlignmentScores(
      ScoreType initMatch, ScoreType initMismatch, ScoreType initOpen, ScoreType initExtend, ScoreType initOffEdge,
      bool initIsAllowEdgeInsertion = false)
    : match(initMatch), mismatch(initMismatch), open(initOpen), extend(initExtend), offEdge(initOffEdge),
      isAllowEdgeInsertion(initIsAllowEdgeInsertion)
  {
  }
  const ScoreType match, mismatch, open, extend, offEdge;
  const bool      isAllowEdgeInsertion;
};

namespace ALIGNPATH {
enum align_t { NONE, MATCH, INSERT, DELETE, SKIP, SOFT_CLIP, HARD_CLIP, PAD, SEQ_MATCH, SEQ_MISMATCH };
struct path_segment {
  path_segment(const align_t t = NONE, const unsigned l = 0) : type(t), length(l) {}
  bool operator==(const path_segment& rhs) const { return (type == rhs.type) && (length == rhs.length); }
  align_t  type;
  unsigned length;
};
typedef std::vector<path_segment> path_t;

inline char segment_type_to_cigar_code(const align_t id)
{
  static const char code[] = {'X', 'M', 'I', 'D', 'N', 'S', 'H', 'P', '=', 'X'};
  return code[id];
}
inline std::string apath_to_cigar(const path_t& apath)
{
  std::string s;
  for (const path_segment& ps : apath) {
    s += std::to_string(ps.length);
    s.push_back(segment_type_to_cigar_code(ps.type));
  }
  return s;
}
/// BAM op numbering of the ABI -> align_t
inline align_t fromBamOp(unsigned op)
{
  static const align_t map[] = {MATCH, INSERT, DELETE, SKIP, SOFT_CLIP, HARD_CLIP, PAD, SEQ_MATCH, SEQ_MISMATCH};
  return (op < 9) ? map[op] : NONE;
}
}  // namespace ALIGNPATH

struct Alignment {
  void clear()
  {
    beginPos = 0;
    apath.clear();
  }
  bool              isAligned() const { return !apath.empty(); }
  pos_t             beginPos = 0;
  ALIGNPATH::path_t apath;
};

template <typename ScoreType>
struct AlignmentResult {
  AlignmentResult() { clear(); }
  void clear()
  {
    score    = 0;
    isJumped = false;
    align.clear();
  }
  ScoreType score;
  bool      isJumped;
  Alignment align;
};

template <typename ScoreType>
struct JumpAlignmentResult {
  JumpAlignmentResult() { clear(); }
  void clear()
  {
    score          = 0;
    jumpInsertSize = 0;
    jumpRange      = 0;
    align1.clear();
    align2.clear();
  }
  ScoreType score;
  unsigned  jumpInsertSize;
  unsigned  jumpRange;
  Alignment align1;
  Alignment align2;
};

namespace detail {
template <typename ScoreType>
manta_align_scores_t toAbi(const AlignmentScores<ScoreType>& s)
{
  return manta_align_scores_t{int32_t(s.match), int32_t(s.mismatch), int32_t(s.open), int32_t(s.extend), int32_t(s.offEdge),
                              s.isAllowEdgeInsertion ? 1 : 0};
}
inline void toPath(const uint32_t* cig, unsigned n, ALIGNPATH::path_t& path)
{
  path.clear();
  for (unsigned i = 0; i < n; ++i) path.emplace_back(ALIGNPATH::fromBamOp(cig[i] & 15u), cig[i] >> 4);
}

/// one alignment through manta_align_batch
template <typename SymIter>
manta_align_result_t alignOne(
    int kind, const manta_align_scores_t& sc, int32_t extra, SymIter qb, SymIter qe, SymIter r1b, SymIter r1e, SymIter r2b,
    SymIter r2e, std::vector<uint32_t>& cigar)
{
  std::vector<uint8_t> arena(qb, qe);
  manta_align_task_t   t{};
  t.query_off = 0;
  t.query_len = uint32_t(arena.size());
  t.ref1_off  = arena.size();
  arena.insert(arena.end(), r1b, r1e);
  t.ref1_len = uint32_t(arena.size() - t.ref1_off);
  t.ref2_off = arena.size();
  arena.insert(arena.end(), r2b, r2e);
  t.ref2_len = uint32_t(arena.size() - t.ref2_off);
  // the reference's own checks, same messages (GlobalJumpAlignerImpl.hpp:50-58, GlobalAlignerImpl.hpp:44-49)
  if (t.query_len == 0) throw GeneralException("Unexpected empty query sequence");
  if (t.ref1_len == 0) throw GeneralException(kind == MANTA_ALIGNER_JUMP ? "Unexpected empty reference1 sequence" : "Unexpected empty reference sequence");
  if (kind == MANTA_ALIGNER_JUMP && t.ref2_len == 0) throw GeneralException("Unexpected empty reference2 sequence");
  arena.push_back(0);
  cigar.assign(2 * size_t(t.query_len) + 16, 0);
  manta_align_result_t res{};
  uint64_t             used = 0;
  manta_ctx_t*         ctx  = threadContext();
  const int rc = manta_align_batch(ctx, kind, &sc, extra, 1, &t, arena.data(), arena.size() - 1, &res, cigar.data(), cigar.size(), &used);
  if (rc != MANTA_OK) throw GeneralException(std::string("manta_amd aligner: ") + manta_last_error(ctx), rc);
  return res;
}
}  // namespace detail

template <typename ScoreType>
struct AlignerBase {
  explicit AlignerBase(const AlignmentScores<ScoreType>& scores) : _scores(scores) {}
  const AlignmentScores<ScoreType>& getScores() const { return _scores; }

protected:
  const AlignmentScores<ScoreType> _scores;
};

/// alignment/GlobalAligner.hpp:36-46
template <typename ScoreType>
struct GlobalAligner : public AlignerBase<ScoreType> {
  explicit GlobalAligner(const AlignmentScores<ScoreType>& scores) : AlignerBase<ScoreType>(scores) {}
  template <typename SymIter>
  void align(const SymIter queryBegin, const SymIter queryEnd, const SymIter refBegin, const SymIter refEnd,
             AlignmentResult<ScoreType>& result) const
  {
    result.clear();
    std::vector<uint32_t>      cig;
    const manta_align_result_t r = detail::alignOne(MANTA_ALIGNER_GLOBAL, detail::toAbi(this->_scores), 0, queryBegin, queryEnd,
                                                    refBegin, refEnd, refEnd, refEnd, cig);
    result.score          = ScoreType(r.score);
    result.isJumped       = r.is_jumped != 0;
    result.align.beginPos = r.begin_pos1;
    detail::toPath(cig.data() + r.cigar1_off, r.cigar1_len, result.align.apath);
  }
};

/// alignment/GlobalLargeIndelAligner.hpp:39-54
template <typename ScoreType>
struct GlobalLargeIndelAligner : public AlignerBase<ScoreType> {
  GlobalLargeIndelAligner(const AlignmentScores<ScoreType>& scores, const ScoreType largeIndelScore)
    : AlignerBase<ScoreType>(scores), _largeIndelScore(largeIndelScore)
  {
  }
  template <typename SymIter>
  void align(const SymIter queryBegin, const SymIter queryEnd, const SymIter refBegin, const SymIter refEnd,
             AlignmentResult<ScoreType>& result) const
  {
    result.clear();
    std::vector<uint32_t>      cig;
    const manta_align_result_t r = detail::alignOne(MANTA_ALIGNER_LARGE_INDEL, detail::toAbi(this->_scores), int32_t(_largeIndelScore),
                                                    queryBegin, queryEnd, refBegin, refEnd, refEnd, refEnd, cig);
    result.score          = ScoreType(r.score);
    result.isJumped       = r.is_jumped != 0;
    result.align.beginPos = r.begin_pos1;
    detail::toPath(cig.data() + r.cigar1_off, r.cigar1_len, result.align.apath);
  }

private:
  const ScoreType _largeIndelScore;
};

/// alignment/GlobalJumpAligner.hpp:36-53
template <typename ScoreType>
struct GlobalJumpAligner : public AlignerBase<ScoreType> {
  GlobalJumpAligner(const AlignmentScores<ScoreType>& scores, const ScoreType jumpScore)
    : AlignerBase<ScoreType>(scores), _jumpScore(jumpScore)
  {
    if (scores.isAllowEdgeInsertion) throw GeneralException("GlobalJumpAligner does not support isAllowEdgeInsertion");
  }
  const ScoreType& getJumpScore() const { return _jumpScore; }
  template <typename SymIter>
  void align(const SymIter queryBegin, const SymIter queryEnd, const SymIter ref1Begin, const SymIter ref1End,
             const SymIter ref2Begin, const SymIter ref2End, JumpAlignmentResult<ScoreType>& result) const
  {
    result.clear();
    std::vector<uint32_t>      cig;
    const manta_align_result_t r = detail::alignOne(MANTA_ALIGNER_JUMP, detail::toAbi(this->_scores), int32_t(_jumpScore), queryBegin,
                                                    queryEnd, ref1Begin, ref1End, ref2Begin, ref2End, cig);
    result.score           = ScoreType(r.score);
    result.jumpInsertSize  = r.jump_insert_size;
    result.jumpRange       = r.jump_range;
    result.align1.beginPos = r.begin_pos1;
    result.align2.beginPos = r.begin_pos2;
    detail::toPath(cig.data() + r.cigar1_off, r.cigar1_len, result.align1.apath);
    detail::toPath(cig.data() + r.cigar2_off, r.cigar2_len, result.align2.apath);
  }

private:
  const ScoreType _jumpScore;
};

}  // namespace manta_amd
