// TEST INFRASTRUCTURE ONLY -- fourth reference driver: the reference's REAL read-pile construction on real BAM files.
//   manta/SVCandidateAssembler.cpp (UNMODIFIED: getBreakendReads :271-659, insertAssemblyRead :102-136), the htsapi BAM layer
//   and htslib 1.9 from the reference's own redist tarball (built into oracle/_ref/htslib by the Makefile).
// Used only by tests/golden/make_demo_golden.py in the authoring container to dump the assembly read piles of the bundled demo
// (src/demo/data, BASELINE config 1) exactly as the reference gathers them.  Test doubles defined here:
//   * runIterativeAssembler -> records the pile it is handed (the pile is what we are after; the real assembler lives in the
//     other _ref libraries),
//   * SVLocusScanner's constructor (alignment statistics file, boost serialization) -> empty object; getBreakendReads never
//     consults it (grep _readScanner manta/SVCandidateAssembler.cpp).
#include <algorithm>
#include <map>
#include <set>

#include "manta/SVCandidateAssembler.hpp"

#include "assembly/IterativeAssembler.hpp"
#include "htsapi/bam_header_info.hpp"
#include "htsapi/bam_streamer.hpp"
#include "htsapi/samtools_fasta_util.hpp"
#include "manta/SVReferenceUtil.hpp"

#include <cstring>
#include <sstream>
#include <string>

#define REF_EXPORT extern "C" __attribute__((visibility("default")))

namespace {
thread_local AssemblyReadInput* g_captured = nullptr;
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
}  // namespace

void runIterativeAssembler(const IterativeAssemblerOptions&, AssemblyReadInput& reads, AssemblyReadOutput& readInfo, Assembly& contigs)
{
  if (g_captured) *g_captured = reads;
  readInfo.clear();
  contigs.clear();
}

SVLocusScanner::SVLocusScanner(const ReadScannerOptions& opt, const std::string&, const std::vector<std::string>&, const bool)
  : _opt(opt), _dopt(opt, false)
{
}

/// One candidate of the demo: the reference's own read gathering.
///   spanning (state2 != UNKNOWN): assembleSpanningSVCandidate with the orientation rule of getJumpAssembly
///       (SVCandidateAssemblyRefiner.cpp:1795-1808: isBp1Reversed / isBp2Reversed from the breakend states)
///   complex: assembleComplexSVCandidate
/// Text: "reads <n>\n" then one read per line; then "ref1 <beginPos> <seq>" / "ref2 ..." = the reference windows the refiner
/// fetches (getSVReferenceSegments, extraRefEdgeSize 250 / getIntervalReferenceSegment 700).
REF_EXPORT int ref_demo_pile(
    int nBams, const char* const* bamPaths, const int* isTumor, const char* fastaPath, int tid1, int begin1, int end1, int state1, int tid2,
    int begin2, int end2, int state2, char* out, int cap)
{
  std::ostringstream os;
  try {
    AlignmentFileOptions alignOpt;
    for (int i = 0; i < nBams; ++i) {
      alignOpt.alignmentFilenames.push_back(bamPaths[i]);
      alignOpt.isAlignmentTumor.push_back(isTumor[i] != 0);
    }
    bam_streamer          first(bamPaths[0], fastaPath);
    const bam_header_info header(first.get_header());
    ReadScannerOptions        scanOpt;
    IterativeAssemblerOptions asmOpt;
    AllSampleReadCounts       counts;  // only feeds the remote-read recovery rate (not used: isSearchRemoteInsertionReads = false)
    counts.setSampleCount(nBams);
    TimeTracker               tt;
    SVCandidateAssembler      assembler(scanOpt, asmOpt, alignOpt, fastaPath, "", "", header, counts, false, tt);
    SVBreakend bp1, bp2;
    bp1.interval = GenomeInterval(tid1, begin1, end1);
    bp1.state    = static_cast<SVBreakendState::index_t>(state1);
    bp2.interval = GenomeInterval(tid2, begin2, end2);
    bp2.state    = static_cast<SVBreakendState::index_t>(state2);
    AssemblyReadInput pile;
    g_captured = &pile;
    Assembly as;
    reference_contig_segment r1, r2;
    if (state2 != SVBreakendState::UNKNOWN) {
      SVCandidate sv;
      sv.bp1 = bp1;
      sv.bp2 = bp2;
      // SVCandidateAssemblyRefiner.cpp:1760-1790: reference windows of a spanning candidate
      static const pos_t extraRefEdgeSize(250);
      unsigned t1, t2, t3, t4;
      getSVReferenceSegments(fastaPath, header, extraRefEdgeSize, sv, r1, r2, t1, t2, t3, t4);
      // :1795-1808
      bool isBp1Reversed(false), isBp2Reversed(false);
      if (bp1.state == bp2.state) {
        if (bp1.state == SVBreakendState::RIGHT_OPEN)
          isBp2Reversed = true;
        else
          isBp1Reversed = true;
      }
      assembler.assembleSpanningSVCandidate(bp1, bp2, isBp1Reversed, isBp2Reversed, r1, r2, as);
    } else {
      static const pos_t extraRefEdgeSize(700);  // :1884
      getIntervalReferenceSegment(fastaPath, header, extraRefEdgeSize, bp1.interval, r1);
      RemoteReadCache remote;
      assembler.assembleComplexSVCandidate(bp1, r1, false, remote, as);
    }
    g_captured = nullptr;
    os << "reads " << pile.size() << "\n";
    for (const std::string& r : pile) os << r << "\n";
    os << "ref1 " << r1.get_offset() << " " << r1.seq() << "\n";
    os << "ref2 " << r2.get_offset() << " " << r2.seq() << "\n";
  } catch (const std::exception& e) {
    os.str("");
    os << "EXCEPTION " << e.what() << "\n";
  }
  return emitText(os.str(), out, cap);
}

/// every record of a BAM region through the reference's own string handling (bam_seq.hpp get_string, the Q mask and
/// reverseCompStr of insertAssemblyRead) in both orientations, next to the raw 4-bit sequence and qualities -- the real-data
/// vectors for manta_amd/host/read_pile.hpp.  Lines: "<seq4 hex> <qual hex> <forward text> <reverse text or '-'>"
REF_EXPORT int ref_bam_records(const char* bamPath, const char* fastaPath, int tid, int begin, int end, int minQval, int maxRecords, char* out, int cap)
{
  std::ostringstream os;
  try {
    bam_streamer bs(bamPath, fastaPath);
    bs.resetRegion(tid, begin, end);
    int n = 0;
    static const char hex[] = "0123456789abcdef";
    while (bs.next() && n < maxRecords) {
      const bam_record& rec(*bs.get_record_ptr());
      const unsigned    len = rec.read_size();
      if (len == 0) continue;
      const uint8_t* seq  = bam_get_seq(rec.get_data());
      const uint8_t* qual = rec.qual();
      std::string    s4, sq;
      for (unsigned i = 0; i < (len + 1) / 2; ++i) {
        s4 += hex[seq[i] >> 4];
        s4 += hex[seq[i] & 15];
      }
      for (unsigned i = 0; i < len; ++i) {
        sq += hex[qual[i] >> 4];
        sq += hex[qual[i] & 15];
      }
      std::string fwd(rec.get_bam_read().get_string());
      for (unsigned i = 0; i < len; ++i)
        if (qual[i] < minQval) fwd[i] = 'N';
      std::string rev("-");
      if (fwd.find('=') == std::string::npos) {
        rev = fwd;
        reverseCompStr(rev);
      }
      os << s4 << " " << sq << " " << fwd << " " << rev << "\n";
      ++n;
    }
  } catch (const std::exception& e) {
    os.str("");
    os << "EXCEPTION " << e.what() << "\n";
  }
  return emitText(os.str(), out, cap);
}

/// Every record of one region query, field by field, for the read-gathering tests (tests/test_read_class.py):
///   "<qname> <flag> <tid> <pos0> <mapq> <mtid> <mpos0> <cigar|*> <seq4 hex> <qual hex> <hasSA> <MC|*>"
/// exactly what bamStream.resetRegion(tid, begin, end) + next() hands getBreakendReads (:381-395), in file order.
REF_EXPORT int ref_region_records(const char* bamPath, const char* fastaPath, int tid, int begin, int end, char* out, int cap)
{
  std::ostringstream os;
  try {
    bam_streamer bs(bamPath, fastaPath);
    bs.resetRegion(tid, begin, end);
    static const char hex[] = "0123456789abcdef";
    while (bs.next()) {
      const bam_record& rec(*bs.get_record_ptr());
      const bam1_t*     b   = rec.get_data();
      const unsigned    len = rec.read_size();
      os << rec.qname() << " " << b->core.flag << " " << b->core.tid << " " << b->core.pos << " " << int(b->core.qual) << " " << b->core.mtid
         << " " << b->core.mpos << " ";
      if (b->core.n_cigar == 0) os << "*";
      const uint32_t* cig = bam_get_cigar(b);
      for (unsigned i = 0; i < b->core.n_cigar; ++i) os << (cig[i] >> 4) << "MIDNSHP=XB"[cig[i] & 15];
      os << " ";
      const uint8_t* seq  = bam_get_seq(b);
      const uint8_t* qual = rec.qual();
      if (len == 0) os << "*";
      for (unsigned i = 0; i < (len + 1) / 2; ++i) os << hex[seq[i] >> 4] << hex[seq[i] & 15];
      os << " ";
      if (len == 0) os << "*";
      for (unsigned i = 0; i < len; ++i) os << hex[qual[i] >> 4] << hex[qual[i] & 15];
      static const char mc[] = {'M', 'C'};
      const char*       mcs  = rec.get_string_tag(mc);
      os << " " << (rec.isSASplit() ? 1 : 0) << " " << (mcs ? mcs : "*") << "\n";
    }
  } catch (const std::exception& e) {
    os.str("");
    os << "EXCEPTION " << e.what() << "\n";
  }
  return emitText(os.str(), out, cap);
}

/// SAM text -> BAM + index with the htslib of the reference's redist (so that synthetic alignments can go through the reference's
/// own BAM layer)
REF_EXPORT int ref_sam_to_bam(const char* samPath, const char* bamPath)
{
  samFile* in = sam_open(samPath, "r");
  if (!in) return -1;
  bam_hdr_t* hdr = sam_hdr_read(in);
  if (!hdr) return -2;
  samFile* outf = sam_open(bamPath, "wb");
  if (!outf) return -3;
  if (sam_hdr_write(outf, hdr) != 0) return -4;
  bam1_t* b = bam_init1();
  int     rc;
  while ((rc = sam_read1(in, hdr, b)) >= 0)
    if (sam_write1(outf, hdr, b) < 0) return -5;
  bam_destroy1(b);
  bam_hdr_destroy(hdr);
  sam_close(in);
  sam_close(outf);
  if (rc < -1) return -6;
  return sam_index_build(bamPath, 0) == 0 ? 0 : -7;
}

/// The reference's read gathering for one candidate with everything that shapes it under the caller's control:
/// chromDepthPath ("" = no depth filter), isSearchRemote (complex candidates), minCandidateVariantSize.  state2 < 0: complex
/// candidate on breakend 1 (assembleComplexSVCandidate); else spanning (assembleSpanningSVCandidate, orientation rule of
/// getJumpAssembly as in ref_demo_pile).  Text: "reads <n>" + the pile + "ref1 <offset> <seq>" / "ref2 ...".
REF_EXPORT int ref_breakend_pile(
    int nBams, const char* const* bamPaths, const int* isTumor, const char* fastaPath, const char* chromDepthPath, int minCandidateVariantSize,
    int isSearchRemote, int tid1, int begin1, int end1, int state1, int tid2, int begin2, int end2, int state2, char* out, int cap)
{
  std::ostringstream os;
  try {
    AlignmentFileOptions alignOpt;
    for (int i = 0; i < nBams; ++i) {
      alignOpt.alignmentFilenames.push_back(bamPaths[i]);
      alignOpt.isAlignmentTumor.push_back(isTumor[i] != 0);
    }
    bam_streamer          first(bamPaths[0], fastaPath);
    const bam_header_info header(first.get_header());
    ReadScannerOptions    scanOpt;
    scanOpt.minCandidateVariantSize = unsigned(minCandidateVariantSize);
    IterativeAssemblerOptions asmOpt;
    AllSampleReadCounts       counts;
    counts.setSampleCount(nBams);
    TimeTracker          tt;
    SVCandidateAssembler assembler(scanOpt, asmOpt, alignOpt, fastaPath, "", chromDepthPath, header, counts, false, tt);
    SVBreakend           bp1, bp2;
    bp1.interval = GenomeInterval(tid1, begin1, end1);
    bp1.state    = static_cast<SVBreakendState::index_t>(state1);
    AssemblyReadInput pile;
    g_captured = &pile;
    Assembly                 as;
    reference_contig_segment r1, r2;
    RemoteReadCache          remote;
    if (state2 >= 0) {
      bp2.interval = GenomeInterval(tid2, begin2, end2);
      bp2.state    = static_cast<SVBreakendState::index_t>(state2);
      SVCandidate sv;
      sv.bp1 = bp1;
      sv.bp2 = bp2;
      static const pos_t extraRefEdgeSize(250);
      unsigned           t1, t2, t3, t4;
      getSVReferenceSegments(fastaPath, header, extraRefEdgeSize, sv, r1, r2, t1, t2, t3, t4);
      bool isBp1Reversed(false), isBp2Reversed(false);
      if (bp1.state == bp2.state) {
        if (bp1.state == SVBreakendState::RIGHT_OPEN)
          isBp2Reversed = true;
        else
          isBp1Reversed = true;
      }
      assembler.assembleSpanningSVCandidate(bp1, bp2, isBp1Reversed, isBp2Reversed, r1, r2, as);
    } else {
      static const pos_t extraRefEdgeSize(700);
      getIntervalReferenceSegment(fastaPath, header, extraRefEdgeSize, bp1.interval, r1);
      assembler.assembleComplexSVCandidate(bp1, r1, isSearchRemote != 0, remote, as);
    }
    g_captured = nullptr;
    os << "reads " << pile.size() << "\n";
    for (const std::string& r : pile) os << r << "\n";
    {  // the RemoteReadCache retrieveRemoteReads filled (SVCandidateAssembler.cpp:241), by name
      std::vector<std::string> keys;
      for (const auto& kv : remote) keys.push_back(kv.first);
      std::sort(keys.begin(), keys.end());
      for (const std::string& k : keys) os << "remote " << k << " " << int(remote[k].readNo) << " " << remote[k].readSeq << "\n";
    }
    os << "ref1 " << r1.get_offset() << " " << r1.seq() << "\n";
    os << "ref2 " << r2.get_offset() << " " << r2.seq() << "\n";
    os << "maxdepth " << (scanOpt.maxDepthFactor) << " " << scanOpt.maxLocalDepthFactorForRemoteReadRetrieval << "\n";
  } catch (const std::exception& e) {
    os.str("");
    os << "EXCEPTION " << e.what() << "\n";
  }
  return emitText(os.str(), out, cap);
}
