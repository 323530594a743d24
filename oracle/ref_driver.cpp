// TEST INFRASTRUCTURE ONLY -- never linked into or called from the product path.
//
// Thin extern-"C" driver around the UNMODIFIED reference translation units, compiled where they lie under
// /root/reference (see oracle/Makefile).  It exposes exactly the three seams of the hot path
// (SURVEY.md section 8b):
//   * runIterativeAssembler            (assembly/IterativeAssembler.hpp:43-47)
//   * GlobalAligner / GlobalLargeIndelAligner / GlobalJumpAligner ::align
//                                      (alignment/GlobalAligner.hpp:36-46, GlobalLargeIndelAligner.hpp:39-54,
//                                       GlobalJumpAligner.hpp:36-53)
// and renders the reference's result objects as canonical text (oracle/FORMAT.md) so that the reference,
// the CPU restatement (oracle/manta_oracle.cpp) and the HIP path can be compared with string equality.
//
// No reference source is copied here; the reference headers are #included from /root/reference at build
// time and the resulting shared object lives in oracle/_ref/ (git-ignored).

#include "bench_harness.hpp"
#include <memory>

#include "alignment/GlobalAligner.hpp"
#include "alignment/GlobalJumpAligner.hpp"
#include "alignment/GlobalLargeIndelAligner.hpp"
#include "assembly/IterativeAssembler.hpp"
#include "assembly/SmallAssembler.hpp"
#include "blt_util/align_path.hpp"

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

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

IterativeAssemblerOptions makeAsmOpt(const uint32_t* o)
{
  IterativeAssemblerOptions opt;
  opt.minWordLength           = o[0];
  opt.maxWordLength           = o[1];
  opt.wordStepSize            = o[2];
  opt.minContigLength         = o[3];
  opt.minCoverage             = o[4];
  opt.minConservativeCoverage = o[5];
  opt.minUnusedReads          = o[6];
  opt.minSupportReads         = o[7];
  opt.maxAssemblyCount        = o[8];
  return opt;
}

void joinSet(std::ostream& os, const std::set<unsigned>& s)
{
  bool first = true;
  for (const unsigned v : s) {
    if (!first) os << ',';
    os << v;
    first = false;
  }
}

std::string assemblyText(
    const unsigned normalReadCount, const AssemblyReadInput& reads, const AssemblyReadOutput& readInfo,
    const Assembly& contigs)
{
  std::ostringstream os;
  os << "contigs " << contigs.size() << '\n';
  for (unsigned i = 0; i < contigs.size(); ++i) {
    const AssembledContig& c(contigs[i]);
    os << "contig " << i << " seq=" << c.seq << " seed=" << c.seedReadCount << " cons=" << c.conservativeRange.begin_pos()
       << ',' << c.conservativeRange.end_pos() << " support=";
    joinSet(os, c.supportReads);
    os << " reject=";
    joinSet(os, c.rejectReads);
    os << '\n';
  }
  os << "reads " << readInfo.size() << " normal " << normalReadCount << '\n';
  for (unsigned i = 0; i < readInfo.size(); ++i) {
    const AssemblyReadInfo& r(readInfo[i]);
    os << "read " << i << " used=" << r.isUsed << " filtered=" << r.isFiltered << " pseudo=" << r.isPseudo << " ids=";
    for (unsigned j = 0; j < r.contigIds.size(); ++j) {
      if (j) os << ',';
      os << r.contigIds[j];
    }
    os << '\n';
  }
  for (unsigned i = normalReadCount; i < reads.size(); ++i) {
    os << "pseudo " << i << " seq=" << reads[i] << '\n';
  }
  return os.str();
}

std::string alignText(const AlignmentResult<int>& r)
{
  std::ostringstream os;
  os << "score=" << r.score << " jumped=" << r.isJumped << " begin=" << r.align.beginPos
     << " cigar=" << ALIGNPATH::apath_to_cigar(r.align.apath) << '\n';
  return os.str();
}

std::string jumpText(const JumpAlignmentResult<int>& r)
{
  std::ostringstream os;
  os << "score=" << r.score << " jumpInsertSize=" << r.jumpInsertSize << " jumpRange=" << r.jumpRange
     << " begin1=" << r.align1.beginPos << " cigar1=" << ALIGNPATH::apath_to_cigar(r.align1.apath)
     << " begin2=" << r.align2.beginPos << " cigar2=" << ALIGNPATH::apath_to_cigar(r.align2.apath) << '\n';
  return os.str();
}

AlignmentScores<int> makeScores(const int32_t* s)
{
  return AlignmentScores<int>(s[0], s[1], s[2], s[3], s[4], s[5] != 0);
}

}  // namespace

extern "C" {

/// opts = {minWordLength,maxWordLength,wordStepSize,minContigLength,minCoverage,minConservativeCoverage,
///         minUnusedReads,minSupportReads,maxAssemblyCount}
/// returns length of the canonical text (may exceed cap), or -1 on exception (message in out)
int ref_assemble(
    const uint32_t* opts, int n_reads, const char* const* reads, const uint32_t* read_lens, char* out, int cap)
{
  try {
    const IterativeAssemblerOptions opt(makeAsmOpt(opts));
    AssemblyReadInput               in;
    in.reserve(n_reads);
    for (int i = 0; i < n_reads; ++i) in.emplace_back(reads[i], read_lens[i]);
    AssemblyReadOutput info;
    Assembly           contigs;
    runIterativeAssembler(opt, in, info, contigs);
    return emit(assemblyText(static_cast<unsigned>(n_reads), in, info, contigs), out, cap);
  } catch (const std::exception& e) {
    emit(std::string("EXCEPTION ") + e.what(), out, cap);
    return -1;
  }
}

/// runSmallAssembler (assembly/SmallAssembler.cpp:622-685), same canonical text.
/// opts = {minWordLength,maxWordLength,wordStepSize,minCoverage,minConservativeCoverage,minSeedReads,maxAssemblyIterations}
int ref_small_assemble(
    const uint32_t* opts, int n_reads, const char* const* reads, const uint32_t* read_lens, char* out, int cap)
{
  try {
    SmallAssemblerOptions opt;
    opt.minWordLength           = opts[0];
    opt.maxWordLength           = opts[1];
    opt.wordStepSize            = opts[2];
    opt.minCoverage             = opts[3];
    opt.minConservativeCoverage = opts[4];
    opt.minSeedReads            = opts[5];
    opt.maxAssemblyIterations   = opts[6];
    AssemblyReadInput in;
    in.reserve(n_reads);
    for (int i = 0; i < n_reads; ++i) in.emplace_back(reads[i], read_lens[i]);
    AssemblyReadOutput info;
    Assembly           contigs;
    runSmallAssembler(opt, in, info, contigs);
    return emit(assemblyText(static_cast<unsigned>(n_reads), in, info, contigs), out, cap);
  } catch (const std::exception& e) {
    emit(std::string("EXCEPTION ") + e.what(), out, cap);
    return -1;
  }
}

/// kind 0 = GlobalAligner, 1 = GlobalLargeIndelAligner, 2 = GlobalJumpAligner
/// scores = {match,mismatch,open,extend,offEdge,isAllowEdgeInsertion}; extra = largeIndelScore / jumpScore
int ref_align(
    int kind, const int32_t* scores, int32_t extra, const char* q, int qlen, const char* r1, int r1len, const char* r2,
    int r2len, char* out, int cap)
{
  try {
    const AlignmentScores<int> sc(makeScores(scores));
    const std::string          query(q, qlen), ref1(r1, r1len), ref2(r2 ? r2 : "", r2 ? r2len : 0);
    if (kind == 0) {
      GlobalAligner<int>   aln(sc);
      AlignmentResult<int> res;
      aln.align(query.begin(), query.end(), ref1.begin(), ref1.end(), res);
      return emit(alignText(res), out, cap);
    } else if (kind == 1) {
      GlobalLargeIndelAligner<int> aln(sc, extra);
      AlignmentResult<int>         res;
      aln.align(query.begin(), query.end(), ref1.begin(), ref1.end(), res);
      return emit(alignText(res), out, cap);
    } else if (kind == 2) {
      GlobalJumpAligner<int>   aln(sc, extra);
      JumpAlignmentResult<int> res;
      aln.align(query.begin(), query.end(), ref1.begin(), ref1.end(), ref2.begin(), ref2.end(), res);
      return emit(jumpText(res), out, cap);
    }
    emit("EXCEPTION unknown aligner kind", out, cap);
    return -1;
  } catch (const std::exception& e) {
    emit(std::string("EXCEPTION ") + e.what(), out, cap);
    return -1;
  }
}

/// The "small SV" locus pipeline of SVCandidateAssemblyRefiner::getSmallSVAssembly
/// (SVCandidateAssemblyRefiner.cpp:1860-2038) reduced to its arithmetic core:
/// runIterativeAssembler -> per contig: 10-mer reference trim (:1984-2011, restated here because that TU needs
/// htslib + full boost and cannot be built) -> GlobalLargeIndelAligner::align (:2032-2038) -> beginPos += cut.
///
/// ref is the full fetched window; leadingCut/trailingCut/maxLeadingCut/maxTrailingCut as computed at :1912-1915.
static std::string smallSvLocus(
    const IterativeAssemblerOptions& opt, const GlobalLargeIndelAligner<int>& aligner, AssemblyReadInput& in,
    const std::string& ref, const int leadingCut, const int trailingCut, const int maxLeadingCut,
    const int maxTrailingCut, const bool wantText)
{
  const unsigned     normalReadCount(in.size());
  AssemblyReadOutput info;
  Assembly           contigs;
  runIterativeAssembler(opt, in, info, contigs);

  std::ostringstream os;
  if (wantText) os << assemblyText(normalReadCount, in, info, contigs);

  for (unsigned ci = 0; ci < contigs.size(); ++ci) {
    const std::string& seq(contigs[ci].seq);
    int                adjLead(leadingCut), adjTrail(trailingCut);
    {
      static const int                merSize(10);
      std::unordered_set<std::string> contigHash;
      const unsigned                  contigSize(seq.size());
      for (unsigned i = 0; i < (contigSize - (merSize - 1)); ++i) contigHash.insert(seq.substr(i, merSize));
      const int refSize(ref.size());
      const int minRefIndex(leadingCut);
      const int maxRefIndex(refSize - (trailingCut + merSize));
      const int maxFwdRefIndex(std::min(maxLeadingCut, maxRefIndex));
      int       refIndex = minRefIndex;
      for (refIndex = minRefIndex; refIndex <= maxFwdRefIndex; refIndex++) {
        if (contigHash.count(ref.substr(refIndex, merSize)) != 0) break;
      }
      adjLead = refIndex;
      const int minRevRefIndex(std::max(minRefIndex, refSize - maxTrailingCut));
      for (refIndex = maxRefIndex; refIndex >= minRevRefIndex; refIndex--) {
        if (contigHash.count(ref.substr(refIndex, merSize)) != 0) break;
      }
      adjTrail = (refSize - (refIndex + merSize));
    }
    AlignmentResult<int> res;
    aligner.align(seq.begin(), seq.end(), ref.begin() + adjLead, ref.end() - adjTrail, res);
    res.align.beginPos += adjLead;
    if (wantText) os << "align " << ci << " lead=" << adjLead << " trail=" << adjTrail << ' ' << alignText(res);
  }
  return os.str();
}

int ref_small_sv_locus(
    const uint32_t* opts, const int32_t* scores, int32_t largeIndelScore, int n_reads, const char* const* reads,
    const uint32_t* read_lens, const char* ref, int ref_len, int leadingCut, int trailingCut, int maxLeadingCut,
    int maxTrailingCut, char* out, int cap)
{
  try {
    const IterativeAssemblerOptions    opt(makeAsmOpt(opts));
    const GlobalLargeIndelAligner<int> aligner(makeScores(scores), largeIndelScore);
    AssemblyReadInput                  in;
    for (int i = 0; i < n_reads; ++i) in.emplace_back(reads[i], read_lens[i]);
    return emit(
        smallSvLocus(opt, aligner, in, std::string(ref, ref_len), leadingCut, trailingCut, maxLeadingCut, maxTrailingCut, true),
        out, cap);
  } catch (const std::exception& e) {
    emit(std::string("EXCEPTION ") + e.what(), out, cap);
    return -1;
  }
}

/// CPU-baseline timer: the same pipeline over a packed batch on n_threads host threads
/// (one aligner per thread, mirroring GenerateSVCandidates.cpp:232-266).  Packed batch layout:
///   bases[]            all reads of all loci back to back, then nothing else
///   read_off[n_reads_total+1]   offsets into bases
///   locus_read_begin[n_loci+1]  read-index ranges
///   refs[] / ref_off[n_loci+1]  reference windows
/// Returns elapsed seconds (wall) for the whole batch; *n_contigs_out = total contigs aligned.
double ref_bench_small_sv(
    const uint32_t* opts, const int32_t* scores, int32_t largeIndelScore, int n_loci, const char* bases,
    const uint64_t* read_off, const uint32_t* locus_read_begin, const char* refs, const uint64_t* ref_off,
    int leadingCut, int trailingCut, int maxLeadingCut, int maxTrailingCut, int n_threads, uint64_t* n_contigs_out)
{
  const IterativeAssemblerOptions opt(makeAsmOpt(opts));
  std::atomic<int>                next(0);
  std::atomic<uint64_t>           ncontigs(0);
  const AlignmentScores<int>      sc(makeScores(scores));
  auto                            worker = [&]() {
    const GlobalLargeIndelAligner<int> aligner(sc, largeIndelScore);
    while (true) {
      const int li = next.fetch_add(1);
      if (li >= n_loci) break;
      AssemblyReadInput in;
      for (uint32_t r = locus_read_begin[li]; r < locus_read_begin[li + 1]; ++r) {
        in.emplace_back(bases + read_off[r], read_off[r + 1] - read_off[r]);
      }
      const std::string ref(refs + ref_off[li], ref_off[li + 1] - ref_off[li]);
      const std::string txt(
          smallSvLocus(opt, aligner, in, ref, leadingCut, trailingCut, maxLeadingCut, maxTrailingCut, false));
      (void)txt;
      ncontigs += 1;
    }
  };
  const auto               t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> pool;
  for (int t = 1; t < n_threads; ++t) pool.emplace_back(worker);
  worker();
  for (auto& th : pool) th.join();
  const auto t1 = std::chrono::steady_clock::now();
  if (n_contigs_out) *n_contigs_out = ncontigs.load();
  return std::chrono::duration<double>(t1 - t0).count();
}

/// the same with the thread harness of bench_harness.hpp: threads started (and pinned if `pin`) before the clock, `loci_total` loci taken
/// round robin from the batch, at most `max_seconds`.  Returns wall seconds; *n_done = loci processed.
double ref_bench_small_sv_timed(
    const uint32_t* opts, const int32_t* scores, int32_t largeIndelScore, int n_loci, const char* bases,
    const uint64_t* read_off, const uint32_t* locus_read_begin, const char* refs, const uint64_t* ref_off,
    int leadingCut, int trailingCut, int maxLeadingCut, int maxTrailingCut, int n_threads, uint64_t loci_total, double max_seconds,
    int pin, uint64_t* n_done)
{
  const IterativeAssemblerOptions opt(makeAsmOpt(opts));
  const AlignmentScores<int>      sc(makeScores(scores));
  auto makeWorker = [&]() {
    auto aligner = std::make_shared<GlobalLargeIndelAligner<int>>(sc, largeIndelScore);
    return [&, aligner](const int li) {
      AssemblyReadInput in;
      for (uint32_t r = locus_read_begin[li]; r < locus_read_begin[li + 1]; ++r) in.emplace_back(bases + read_off[r], read_off[r + 1] - read_off[r]);
      const std::string ref(refs + ref_off[li], ref_off[li + 1] - ref_off[li]);
      const std::string txt(smallSvLocus(opt, *aligner, in, ref, leadingCut, trailingCut, maxLeadingCut, maxTrailingCut, false));
      (void)txt;
    };
  };
  return bench_harness::run(n_threads, loci_total, max_seconds, pin != 0, n_loci, makeWorker, n_done);
}

}  // extern "C"
