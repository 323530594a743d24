// TEST INFRASTRUCTURE -- CPU restatement of the scoring-side functions of SURVEY.md 8(f) #2 and #3 (linked into
// oracle/libmanta_oracle.so next to manta_oracle.cpp).  Paths relative to /root/reference/src/c++/lib.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it; the product never does.
//
//   orc_qscore_snp_tables   blt_util/qscore_snp.cpp:26-38, blt_util/math_util.hpp:35-45, blt_util/qscore.hpp:68-71
//   orc_split_read_aligner  applications/GenerateSVCandidates/SplitReadAlignment.cpp:223-350
//                           (getLnLhood :52-93, calculateAlignScore :95-121, isEvidenceCheck :123-136, setEvidence :138-155)
//
// Parity status: pinned -- tests/test_split_read.py compares this file with the unmodified reference sources
// (oracle/_ref, ref_scoring_driver.cpp) on the reference's own unit-test vectors and on random cases, floats as bit patterns.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

namespace {

const int kMaxQscore = 70;  // blt_util/qscore_cache.hpp:46

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

struct Tables {
  double lnComp[kMaxQscore + 1], lnErr[kMaxQscore + 1];
  float  lnOneThird, lnRandomBase;
  explicit Tables(const double snpProb)
  {
    const double compSnp3 = 1. - (snpProb / 3.);
    for (int i = 0; i <= kMaxQscore; ++i) {
      const double qerr = std::pow(10., -static_cast<double>(i) / 10.);  // phred_to_error_prob
      const double p    = (qerr * compSnp3) + ((1 - qerr) * snpProb);
      const double x    = -p;
      lnComp[i]         = (std::abs(x) < 0.01) ? std::log1p(x) : std::log(1 + x);  // log1p_switch
      lnErr[i]          = std::log(p);
    }
    lnOneThird   = std::log(1 / 3.f);
    lnRandomBase = -std::log(4.f);
  }
};

struct Info {
  unsigned alignPos = 0, leftSize = 0, homSize = 0, rightSize = 0, leftMismatches = 0, homMismatches = 0, rightMismatches = 0, alignScore = 0;
  float    alignLnLhood = 0;
  bool     isEvidence = false, isTier2Evidence = false;
  float    evidence = 0;
};

bool isEvidenceCheck(const Info& a, const unsigned minFlankSize)
{
  if (a.leftSize < minFlankSize) return false;
  if (a.rightSize < minFlankSize) return false;
  if ((a.leftMismatches / (float)a.leftSize) >= 0.25) return false;
  if ((a.rightMismatches / (float)a.rightSize) >= 0.25) return false;
  const float size(static_cast<float>(a.leftSize + a.rightSize));
  if ((a.alignScore / size) < 0.9) return false;
  return true;
}

/// one placement's log-likelihood, without the early break (see manta_amd/csrc/split_kernels.hpp for why it is immaterial)
float placementLnLhood(
    const Tables& t, const uint8_t* query, const uint8_t* qual, const int querySize, const uint8_t* target, const int pos, const int scoreBegin,
    const int scoreEnd, bool& badQ)
{
  float lnLhood = 0;
  for (int i = 0; i < querySize; ++i) {
    if (pos + i > scoreEnd) break;
    if (pos + i <= scoreBegin) continue;
    int           bq   = std::max(2, static_cast<int>(qual[i]));
    const bool    over = bq > kMaxQscore;  // the reference's table lookup throws (qscore_cache.hpp:49-52) -- where it looks up
    if (over) bq = kMaxQscore;
    const uint8_t qb = query[i], tb = target[pos + i];
    if (qb != tb || qb == 'N') {
      if (qb == 'N' || tb == 'N') {
        lnLhood += t.lnRandomBase;  // (SplitReadAlignment.cpp:75-78: no table lookup at an 'N')
      } else {
        badQ = badQ || over;
        lnLhood += t.lnErr[bq] + t.lnOneThird;
      }
    } else {
      badQ = badQ || over;
      lnLhood += t.lnComp[bq];
    }
  }
  return lnLhood;
}

}  // namespace

extern "C" {

int orc_qscore_snp_tables(double snpPrior, double* lnCompError, double* lnError, float* lnOneThird, float* lnRandomBase)
{
  const Tables t(snpPrior);
  for (int i = 0; i <= kMaxQscore; ++i) {
    lnCompError[i] = t.lnComp[i];
    lnError[i]     = t.lnErr[i];
  }
  *lnOneThird   = t.lnOneThird;
  *lnRandomBase = t.lnRandomBase;
  return kMaxQscore + 1;
}

int orc_split_read_aligner(
    unsigned flankScoreSize, const char* query, unsigned queryLen, const uint8_t* qual, const char* target, unsigned targetLen, int bpBegin,
    int bpEnd, double snpPrior, char* out, int cap)
{
  const Tables t(snpPrior);
  const int    querySize = int(queryLen), targetSize = int(targetLen);
  if (querySize >= targetSize) return emitText("EXCEPTION\n", out, cap);
  const unsigned scanStart = unsigned(std::max(0, bpBegin - querySize + 2));
  const unsigned scanEnd   = unsigned(std::max(0, std::min(bpEnd, targetSize - querySize)));
  if (scanEnd < scanStart) return emitText("EXCEPTION\n", out, cap);
  const int scoreBegin = bpBegin - int(flankScoreSize), scoreEnd = bpEnd + int(flankScoreSize);
  const uint8_t* q  = reinterpret_cast<const uint8_t*>(query);
  const uint8_t* tg = reinterpret_cast<const uint8_t*>(target);
  bool     isBest = false, badQ = false;
  float    best = 0;
  unsigned bestPos = 0;
  for (unsigned i = scanStart; i <= scanEnd; ++i) {
    const float v = placementLnLhood(t, q, qual, querySize, tg, int(i), scoreBegin, scoreEnd, badQ);
    if (!isBest || v > best) {
      best    = v;
      bestPos = i;
      isBest  = true;
    }
  }
  if (badQ) return emitText("EXCEPTION\n", out, cap);  // qphred_cache::qscore_check throws (qscore_cache.hpp:49-52)
  Info a;
  if (int(bestPos) <= bpBegin + 1) a.leftSize = unsigned(bpBegin + 1 - int(bestPos));
  if (a.leftSize > unsigned(querySize)) return emitText("EXCEPTION\n", out, cap);
  a.homSize = std::min(unsigned(querySize) - a.leftSize, (unsigned(bpEnd + 1) - bestPos) - a.leftSize);  // unsigned operands as in the reference (:320-322)
  if (a.leftSize + a.homSize < unsigned(querySize)) a.rightSize = unsigned(querySize) - (a.leftSize + a.homSize);
  a.alignLnLhood = best;
  a.alignPos     = bestPos;
  for (int i = 0; i < querySize; ++i) {
    if (q[i] != tg[bestPos + unsigned(i)] || q[i] == 'N') {
      if (unsigned(i) <= a.leftSize)
        a.leftMismatches++;
      else if (unsigned(i) <= a.leftSize + a.homSize)
        a.homMismatches++;
      else
        a.rightMismatches++;
    }
  }
  a.alignScore      = unsigned(querySize) - (a.leftMismatches + a.homMismatches + a.rightMismatches);
  a.isEvidence      = isEvidenceCheck(a, 16);
  a.isTier2Evidence = isEvidenceCheck(a, 8);
  a.evidence        = 0;
  if (a.isEvidence || a.isTier2Evidence) {
    const float size(static_cast<float>(a.leftSize + a.rightSize));
    a.evidence = 2 * std::min(a.leftSize, a.rightSize) / (size);
  }
  char buf[512];
  std::snprintf(buf, sizeof(buf),
                "alignPos=%u leftSize=%u homSize=%u rightSize=%u leftMismatches=%u homMismatches=%u rightMismatches=%u alignScore=%u "
                "isEvidence=%d isTier2Evidence=%d evidence=%a alignLnLhood=%a\n",
                a.alignPos, a.leftSize, a.homSize, a.rightSize, a.leftMismatches, a.homMismatches, a.rightMismatches, a.alignScore,
                a.isEvidence ? 1 : 0, a.isTier2Evidence ? 1 : 0, double(a.evidence), double(a.alignLnLhood));
  return emitText(buf, out, cap);
}

}  // extern "C"
