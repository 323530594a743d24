// HIP kernel (gfx950) for Manta's split-read scoring
//   splitReadAligner   applications/GenerateSVCandidates/SplitReadAlignment.cpp:223-350
//     getLnLhood       :52-93      (the O(window x read) scan)
//     calculateAlignScore :95-121  (mismatch counts of the winning placement)
// (paths relative to /root/reference/src/c++/lib).  Consumer: SVScorerSplit.cpp -- every read near a breakend is slid,
// ungapped, along the extended contig and along the reference; the placement with the highest log-likelihood wins.
//
// Mapping: ONE 64-lane wavefront per (read, target) task; lane l owns the candidate placements scanStart + l,
// scanStart + l + 64, ...  A lane runs the reference's inner loop for its placement unchanged: query positions in
// increasing order, the float accumulator updated exactly as the C++ expression does it --
//     match      lnLhood = float(double(lnLhood) + lnCompError[q])          (qscore_snp returns double, :82)
//     mismatch   lnLhood = float(double(lnLhood) + (lnError[q] + double(ln_one_third)))   (:79)
//     N          lnLhood = lnLhood + lnRandomBase                          (float + float, :77)
// so every partial sum is bit-identical to the reference's (IEEE add, round-to-nearest conversion; no reassociation,
// nothing fused).  The reference's early break (:89) only abandons placements whose sum is already below the best:
// terms are <= 0, such a placement can never win, and the winner's value is always a complete sum -- skipping the
// break changes neither bestPos nor bestLnLhood.  "First best wins" (strict '>', :305) = highest value, lowest
// placement on ties.
#pragma once
#include "wave.hpp"

namespace manta_dev {

struct SplitTaskDev {
  const uint8_t* query;   ///< read bases, 1 byte each
  const uint8_t* qual;    ///< basecall qualities
  const uint8_t* target;  ///< contig / reference window
  uint32_t       query_len, target_len;
  int32_t        bp_begin, bp_end;  ///< targetBpOffsetRange
  uint32_t       flank_score_size;
  uint32_t       reserved;
};

struct SplitResultDev {
  int32_t  status;  ///< 0 ok, 1 querySize >= targetSize (:237), 2 scanEnd < scanStart (:265), 3 quality above the table
  uint32_t best_pos;
  float    best_ln_lhood;
  uint32_t left_mismatches, hom_mismatches, right_mismatches;  ///< calculateAlignScore for the sizes below
  uint32_t left_size, hom_size, right_size;                    ///< :310-337 (left_size > query_len: the caller throws, :317)
  uint32_t reserved;
};

struct SplitParams {
  const SplitTaskDev* tasks;
  SplitResultDev*     results;
  uint32_t            n_tasks;
  uint32_t            n_q;            ///< entries in the two tables (MAX_QSCORE + 1)
  const double*       ln_comp_error;  ///< qscore_snp::qphred_to_ln_comp_error_prob
  const double*       ln_error;       ///< qscore_snp::qphred_to_ln_error_prob
  float               ln_one_third;   ///< std::log(1 / 3.f)
  float               ln_random_base; ///< -std::log(4.f)
  uint32_t*           counter;
};

WV_DEV void splitReadTask(const SplitParams& P, const unsigned t)
{
  const unsigned     lane = unsigned(wv::lane());
  const SplitTaskDev T    = P.tasks[t];
  SplitResultDev     R;
  R.status = 0;
  R.best_pos = 0;
  R.best_ln_lhood = 0.f;
  R.left_mismatches = R.hom_mismatches = R.right_mismatches = 0;
  R.left_size = R.hom_size = R.right_size = 0;
  R.reserved = 0;
  const int querySize = int(T.query_len), targetSize = int(T.target_len);
  if (querySize >= targetSize) {
    R.status = 1;
    if (lane == 0) P.results[t] = R;
    return;
  }
  // :250-258
  const int scanStartI = T.bp_begin - querySize + 2;
  const unsigned scanStart = unsigned(scanStartI > 0 ? scanStartI : 0);
  const int seI = (T.bp_end < (targetSize - querySize)) ? T.bp_end : (targetSize - querySize);
  const unsigned scanEnd = unsigned(seI > 0 ? seI : 0);
  if (scanEnd < scanStart) {
    R.status = 2;
    if (lane == 0) P.results[t] = R;
    return;
  }
  const int scoreBegin = T.bp_begin - int(T.flank_score_size), scoreEnd = T.bp_end + int(T.flank_score_size);  // :260-262
  const double thirdD  = double(P.ln_one_third);

  bool     have = false, badQ = false;
  float    best = 0.f;
  unsigned bestPos = 0;
  for (unsigned base = scanStart; base <= scanEnd; base += 64) {
    const unsigned pos = base + lane;
    if (pos <= scanEnd) {
      float lnLhood = 0.f;
      // only query positions with scoreBegin < pos + i <= scoreEnd contribute (:68-69)
      int iLo = scoreBegin + 1 - int(pos);
      if (iLo < 0) iLo = 0;
      int iHi = scoreEnd - int(pos);
      if (iHi > querySize - 1) iHi = querySize - 1;
      for (int i = iLo; i <= iHi; ++i) {
        const uint8_t qb = T.query[i], tb = T.target[pos + unsigned(i)];
        int           bq = int(T.qual[i]);
        if (bq < 2) bq = 2;  // :65
        // (the quality tables are consulted only where the reference consults them: not at 'N' positions, :75-78)
        const bool over = unsigned(bq) >= P.n_q;
        if (over) bq = int(P.n_q) - 1;
        if (qb != tb || qb == 'N') {
          if (qb == 'N' || tb == 'N') {
            lnLhood += P.ln_random_base;
          } else {
            badQ    = badQ || over;
            lnLhood = float(double(lnLhood) + (P.ln_error[bq] + thirdD));
          }
        } else {
          badQ    = badQ || over;
          lnLhood = float(double(lnLhood) + P.ln_comp_error[bq]);
        }
      }
      // lanes see their placements in increasing order: strict '>' keeps the earliest of equal values
      if (!have || lnLhood > best) {
        have    = true;
        best    = lnLhood;
        bestPos = pos;
      }
    }
  }
  // wave arg-max: highest value, lowest placement on ties
  for (int off = 1; off < 64; off <<= 1) {
    const int      src   = wv::lane() ^ off;
    const float    oval  = __builtin_bit_cast(float, wv::shfl(__builtin_bit_cast(unsigned, best), src));
    const unsigned opos  = wv::shfl(bestPos, src);
    const bool     ohave = wv::shfl(int(have), src) != 0;
    if (ohave && (!have || oval > best || (oval == best && opos < bestPos))) {
      have    = true;
      best    = oval;
      bestPos = opos;
    }
  }
  if (wv::any(badQ)) {
    R.status = 3;
    if (lane == 0) P.results[t] = R;
    return;
  }
  // :309-337
  unsigned leftSize = 0;
  if (int(bestPos) <= T.bp_begin + 1) leftSize = unsigned(T.bp_begin + 1 - int(bestPos));
  unsigned homSize = 0, rightSize = 0;
  if (leftSize <= unsigned(querySize)) {
    // std::min over UNSIGNED operands in the reference (:320-322): a range that ends before the placement wraps around and
    // the first operand wins
    const unsigned a = unsigned(querySize) - leftSize, b = (unsigned(T.bp_end + 1) - bestPos) - leftSize;
    homSize          = (a < b) ? a : b;
    if (leftSize + homSize < unsigned(querySize)) rightSize = unsigned(querySize) - (leftSize + homSize);
  }
  // calculateAlignScore (:95-121): i <= leftSize counts as left, i <= leftSize + homSize as hom
  unsigned lm = 0, hm = 0, rm = 0;
  for (int i0 = 0; i0 < querySize; i0 += 64) {
    const int  i   = i0 + int(lane);
    bool       mis = false;
    if (i < querySize) {
      const uint8_t qb = T.query[i];
      mis              = (qb != T.target[bestPos + unsigned(i)]) || (qb == 'N');
    }
    lm += unsigned(wv::popc(wv::ballot(mis && unsigned(i) <= leftSize)));
    hm += unsigned(wv::popc(wv::ballot(mis && unsigned(i) > leftSize && unsigned(i) <= leftSize + homSize)));
    rm += unsigned(wv::popc(wv::ballot(mis && unsigned(i) > leftSize + homSize)));
  }
  R.best_pos         = bestPos;
  R.best_ln_lhood    = best;
  R.left_size        = leftSize;
  R.hom_size         = homSize;
  R.right_size       = rightSize;
  R.left_mismatches  = lm;
  R.hom_mismatches   = hm;
  R.right_mismatches = rm;
  if (lane == 0) P.results[t] = R;
}

#if !MANTA_TU_DEFINES(MANTA_TU_GLUE)
WV_KERNEL void split_read_kernel(const SplitParams P);
#else
WV_KERNEL void split_read_kernel(const SplitParams P)
{
  while (true) {
    unsigned t = 0;
    if (wv::lane() == 0) t = wv::atomic_add(P.counter, 1u);
    t = wv::first(t);
    if (t >= P.n_tasks) break;
    splitReadTask(P, t);
    wv::sync();
  }
}
#endif

}  // namespace manta_dev
