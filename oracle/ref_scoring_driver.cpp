// TEST INFRASTRUCTURE ONLY -- third reference driver: the scoring-side functions of SURVEY.md 8(f) #2 and #3, compiled
// from the UNMODIFIED reference sources where they lie under /root/reference:
//   applications/GenerateSVCandidates/SplitReadAlignment.cpp   (splitReadAligner and its statics; #included the way the
//       reference's own unit test does it, .../test/SplitReadAlignmentTest.cpp:30)
//   blt_util/qscore_snp.cpp, blt_util/qscore_cache.cpp          (the quality -> log-probability tables)
// Linked into oracle/_ref/libmanta_ref_refiner.so (hidden visibility + --gc-sections: the BAM-bound getRefAlignment is
// never referenced and is dropped).
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
