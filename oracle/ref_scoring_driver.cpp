// TEST INFRASTRUCTURE ONLY -- third reference driver: the scoring-side functions of SURVEY.md 8(f) #2 and #3, compiled
// from the UNMODIFIED reference sources where they lie under /root/reference:
//   applications/GenerateSVCandidates/SplitReadAlignment.cpp   (splitReadAligner and its statics; #included the way the
//       reference's own unit test does it, .../test/SplitReadAlignmentTest.cpp:30)
//   blt_util/qscore_snp.cpp, blt_util/qscore_cache.cpp          (the quality -> log-probability tables)
// Linked into oracle/_ref/libmanta_ref_refiner.so (hidden visibility + --gc-sections: the BAM-bound getRefAlignment is
// never referenced and is dropped).
#include <map>
#include <set>  // SVEvidenceWriter.hpp:38 uses std::set without including <set> (it arrives through real boost headers)

#include "applications/GenerateSVCandidates/SplitReadAlignment.cpp"

#include <cstdio>
#include <cstring>
#include <string>

#define REF_EXPORT extern "C" __attribute__((visibility("default")))

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
}  // namespace

/// blt_util/qscore_snp.cpp:26-38 + the two float constants of SplitReadAlignment.cpp:50,76
REF_EXPORT int ref_qscore_snp_tables(double snpPrior, double* lnCompError, double* lnError, float* lnOneThird, float* lnRandomBase)
{
  const qscore_snp q(snpPrior);
  for (int i = 0; i <= qphred_cache::MAX_QSCORE; ++i) {
    lnCompError[i] = q.qphred_to_ln_comp_error_prob(i);
    lnError[i]     = q.qphred_to_ln_error_prob(i);
  }
  *lnOneThird   = std::log(1 / 3.f);
  *lnRandomBase = -std::log(4.f);
  return qphred_cache::MAX_QSCORE + 1;
}

/// splitReadAligner (SplitReadAlignment.cpp:223-350); canonical text, floats as C99 hexfloat (bit-exact)
REF_EXPORT int ref_split_read_aligner(
    unsigned flankScoreSize, const char* query, unsigned queryLen, const uint8_t* qual, const char* target, unsigned targetLen, int bpBegin,
    int bpEnd, double snpPrior, char* out, int cap)
{
  try {
    const qscore_snp  q(snpPrior);
    SRAlignmentInfo   a;
    const std::string qs(query, queryLen), ts(target, targetLen);
    splitReadAligner(flankScoreSize, qs, q, qual, ts, known_pos_range2(bpBegin, bpEnd), a);
    char buf[512];
    std::snprintf(buf, sizeof(buf),
                  "alignPos=%u leftSize=%u homSize=%u rightSize=%u leftMismatches=%u homMismatches=%u rightMismatches=%u alignScore=%u "
                  "isEvidence=%d isTier2Evidence=%d evidence=%a alignLnLhood=%a\n",
                  a.alignPos, a.leftSize, a.homSize, a.rightSize, a.leftMismatches, a.homMismatches, a.rightMismatches, a.alignScore,
                  a.isEvidence ? 1 : 0, a.isTier2Evidence ? 1 : 0, double(a.evidence), double(a.alignLnLhood));
    return emitText(buf, out, cap);
  } catch (const std::exception&) {
    return emitText("EXCEPTION\n", out, cap);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// the shadow-read aligner call site, SVScorePairAltProcessor::realignPairedRead
// (applications/GenerateSVCandidates/SVScorePairAltProcessor.cpp:147-342), reached the way the reference's own unit test
// reaches the private members: through the friend struct the class declares (SVScorePairAltProcessor.hpp:92).
// ---------------------------------------------------------------------------------------------------------------
#include "applications/GenerateSVCandidates/SVScorePairAltProcessor.cpp"
#include "applications/GenerateSVCandidates/SVScorePairProcessor.cpp"

// Test double: the only BAM-record accessor the linked (never executed) record-processing members reach that lives in
// htsapi/bam_record.cpp, i.e. behind htslib.  realignPairedRead works on strings and never touches a bam_record.
const char* bam_record::get_string_tag(const char*) const
{
  return nullptr;
}

struct TestSVScorerAltProcessor {
  static bool realign(
      SVScorePairAltProcessor& p, const bam_header_info& header, const bool isLeftOfInsert, const std::string& floatRead, const pos_t anchorPos,
      int& altTemplateSize)
  {
    return p.realignPairedRead(header, "frag", isLeftOfInsert, floatRead, 0, anchorPos, altTemplateSize);
  }
};

/// One precise same-chromosome indel-type candidate with its extended contig, then n shadow reads through the unmodified
/// realignPairedRead.  Candidate description (all the function reads): the two breakend intervals, insertSeq, the
/// unknown-size-insertion sequences, and the contig alignment (spanning form: align1 begin + cigar).
/// Text: one line per read "pass=<0|1> altTemplateSize=<n>" or "EXCEPTION".
REF_EXPORT int ref_shadow_realign(
    const char* extendedContig, int bp1Begin, int bp1End, int bp2Begin, int bp2End, const char* insertSeq, int isUnknownSizeInsertion,
    const char* unknownLeft, const char* unknownRight, int align1BeginPos, const char* align1Cigar, unsigned nReads, const char* const* reads,
    const int* isLeftOfInsert, const int* anchorPos, char* out, int cap)
{
  std::string text;
  try {
    bam_header_info header;
    header.chrom_data.emplace_back("chr1", 1000000);
    const std::vector<bool> isTumor = {false};
    ReadScannerOptions      scanOpt;
    SVRefinerOptions        refineOpt;
    const std::vector<std::string> noFiles;
    SVLocusScanner          scanner(scanOpt, std::string(), noFiles, false);
    const PairOptions       pairOpt(false);
    SVCandidate             sv;
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
    ALIGNPATH::cigar_to_apath(align1Cigar, ja.align1.apath);
    data.spanningAlignments.push_back(ja);
    data.extendedContigs.push_back(extendedContig);
    SVEvidence evidence;
    evidence.samples.resize(1);
    SVScorePairAltProcessor proc(header, scanOpt, refineOpt, isTumor, scanner, pairOpt, data, sv, true, evidence);
    for (unsigned i = 0; i < nReads; ++i) {
      try {
        int        alt = 0;
        const bool ok  = TestSVScorerAltProcessor::realign(proc, header, isLeftOfInsert[i] != 0, reads[i], anchorPos[i], alt);
        text += "pass=" + std::to_string(ok ? 1 : 0) + " altTemplateSize=" + std::to_string(ok ? alt : 0) + "\n";
      } catch (const std::exception&) {
        text += "EXCEPTION\n";
      }
    }
  } catch (const std::exception& e) {
    text = std::string("FATAL ") + e.what() + "\n";
  }
  return emitText(text, out, cap);
}
