// GlobalLargeIndelAligner (alignment/GlobalLargeIndelAlignerImpl.hpp:35-225), TWO alignments per wavefront in packed 16-bit
// arithmetic.
//
// align_kernel<1,E> spends ~50 VALU instructions per DP cell: five states, each an arg-max over up to five candidates, every
// candidate packed as value*8 + (7 - state) so that one integer max carries the reference's strict-'>' tie order
// (GlobalLargeIndelAligner.hpp:124-151).  The contigs of small-SV loci are short (a few hundred bases), so those values are
// small: a cell of such an alignment fits 13 bits + the 3 index bits = one int16.  This kernel runs the same recurrence on
// v_pk_add_i16 / v_pk_max_i16 with the low half of every register belonging to task A and the high half to task B: the same
// instruction stream, half the instructions per cell.
//
// Exactness.  States live in the x8 domain (value * 8, index bits clear).  The reference's finite sentinel badVal = -10000
// becomes INT16_MIN = -4096 * 8, and every addition SATURATES, so sentinel-derived cells stay at or just above it.  A cell
// whose value derives from the boundary row / column without the sentinel ("real") is bounded below by
//     Q * min(mismatch, offEdge, 0) + open + extend + largeIndel   (the all-mismatch diagonal; a gap state always has the
//                                                                   fresh-open candidate of its neighbour)
// and sentinel-derived cells are bounded above by  -4096 + Q * max(match, 0).  pairEligible() admits an E bucket only when the
// first bound clears the second with a margin for every query the bucket can hold: then every arg-max among candidates that
// include a real one picks the same winner as the reference's int arithmetic, and the traceback (which starts at a real cell
// and follows winners) only ever reads such cells.  Pointers of sentinel-only cells may differ from the reference's; nothing
// reads them.  The '='/'X' expansion and the traceback are align_kernel's own (Aligner<1, E, true>: the two tasks' cells
// interleave in one slab, a task reads every second 16-bit word).
//
// Rows.  Both tasks sweep max(G_A, G_B) reference rows; the shorter one computes rows past its reference against base 0 (never a
// match), its start candidates stop at its own last row.  Start candidates are tracked as in align_kernel (AlignerUtil.hpp:53-67).
#pragma once
#include "align_kernels.hpp"

namespace manta_dev {

static const int PAIR_BAD8 = -32768;  // the sentinel in the x8 domain

/// may a bucket of E columns per lane (queries up to 64 E bases) run on the packed kernel with these scores ?
WV_HD bool pairEligible(const int E, const int match, const int mismatch, const int open, const int extend, const int offEdge,
                        const int largeIndel, const int allowEdgeIns)
{
  if (E > 6 || allowEdgeIns) return false;
  const long q      = 64L * E;
  const long perCol = -long((mismatch < offEdge) ? ((mismatch < 0) ? mismatch : 0) : ((offEdge < 0) ? offEdge : 0));
  const long gaps   = -long((open < 0) ? open : 0) - long((extend < 0) ? extend : 0) - long((largeIndel < 0) ? largeIndel : 0);
  const long up     = q * long((match > 0) ? match : 0);
  // (also: open / largeIndel must not be positive beyond the margin, and every per-step addend must fit the packed constants)
  if (open > 0 || extend > 0 || largeIndel > 0 || match < 0 || match > 64 || mismatch < -512 || offEdge < -512 || open < -2048 || largeIndel < -2048 || extend < -512)
    return false;
  // (a sentinel-derived insert cell may rise by largeIndel - open through the jump-deletion candidate `ins + L - open`: part of the margin)
  const long liftJD = (largeIndel - open > 0) ? long(largeIndel - open) : 0L;
  return q * perCol + gaps + up + liftJD + 64 < 4096;
}

template <int E>
struct PairAligner {
  static const int   NS = 5;
  const AlignParams& P;

  WV_DEV PairAligner(const AlignParams& p) : P(p) {}

  WV_DEV static uint32_t pk(const int lo, const int hi) { return (uint32_t(lo) & 0xffffu) | (uint32_t(hi) << 16); }
  WV_DEV static uint32_t pk2(const int v) { return pk(v, v); }
  WV_DEV static int      half(const uint32_t v, const int h) { return h ? (int(v) >> 16) : int(int16_t(v & 0xffffu)); }

  /// both tasks' sweeps; returns their traceback starts.  ptr32: cell pairs, layout ptr32[(t * E + e) * 64 + lane]
  WV_DEV void sweep(const AlignTaskDev& TA, const AlignTaskDev& TB, uint32_t* ptr32, StartCand& outA, StartCand& outB)
  {
    const int      lane = wv::lane();
    const unsigned QA = TA.query_len, QB = TB.query_len, GA = TA.ref1_len, GB = TB.ref1_len;
    const unsigned G  = (GA > GB) ? GA : GB;
    const int      open = P.open, extend = P.extend, L = P.extra, offEdge = P.off_edge;
    // packed constants (x8 domain; the index bits of candidate i are 7 - i)
    const uint32_t cM0 = pk2(7), cD1 = pk2(6), cI2 = pk2(5), cJ3 = pk2(4), cJI4 = pk2(3);
    const uint32_t cOpen0   = pk2(open * 8 + 7);        // match + open, candidate 0
    const uint32_t cL0      = pk2(L * 8 + 7);           // match + L, candidate 0
    const uint32_t cLmOpen2 = pk2((L - open) * 8 + 5);  // insert + L - open, candidate 2
    const uint32_t cL4      = pk2(L * 8 + 3);           // jumpIns + L, candidate 4
    const uint32_t cBad1 = pk2(PAIR_BAD8 + 6), cBad3 = pk2(PAIR_BAD8 + 4);
    const uint32_t cExt     = pk2(extend * 8);
    const uint32_t cMatch   = pk2(P.match * 8);
    const uint32_t cMisDiff = pk2((P.mismatch - P.match) * 8);
    const uint32_t clrIdx = 0xfff8fff8u, idxBits = 0x00070007u, bad = pk2(PAIR_BAD8);

    // traceback start candidates per task.  Rows at q == Q: one 32-bit key per task, (value x 8) << 16 | (0xffff - row), kept by
    // a plain integer max on the lane that owns column Q: the higher value wins, among equal values the EARLIER row (the
    // reference's strict '>' scan, AlignerUtil.hpp:53-67).  Off-edge candidates: the match states of a task's LAST row are
    // captured into lastRow[] when a lane passes that row and evaluated once after the sweep.
    const unsigned Qs[2] = {QA, QB}, Gs[2] = {GA, GB};
    unsigned       lQ[2], eQ[2];
    for (int h = 0; h < 2; ++h) {
      lQ[h] = (Qs[h] - 1) / E;
      eQ[h] = (Qs[h] - 1) % E;
    }
    int      rowKey[2] = {int(0x80000000u), int(0x80000000u)};
    uint32_t lastRow[E];
    for (int e = 0; e < E; ++e) lastRow[e] = bad;
    const uint32_t fcMask = (lane == 0) ? 0xffffffffu : 0u;  // column 1 of the query: gap states are reset (:130, :143, ...)

    uint32_t st[NS][E], lcur[NS], lprev[NS], qc[E];
    for (int e = 0; e < E; ++e) {
      const unsigned q0 = unsigned(lane) * E + e;
      const unsigned a = (q0 < QA) ? TA.query[q0] : 0u, b = (q0 < QB) ? TB.query[q0] : 0u;
      qc[e]            = a | (b << 16);
      const int row0   = int((q0 + 1) * unsigned(offEdge)) * 8;  // row 0 (GlobalAlignerImpl.hpp:66-80); beyond a task's query: unused columns
      for (int s = 0; s < NS; ++s) st[s][e] = bad;
      st[ST_MATCH][e] = pk2(row0 < PAIR_BAD8 ? PAIR_BAD8 : row0);
    }
    for (int s = 0; s < NS; ++s) lcur[s] = lprev[s] = bad;
    lcur[ST_MATCH] = lprev[ST_MATCH] = 0;  // column 0, row 0

    unsigned curA = 0, curB = 0, nextA = 0, nextB = 0;
    {
      const unsigned i0 = unsigned(lane);
      nextA             = (i0 < GA) ? TA.ref1[i0] : 0u;
      nextB             = (i0 < GB) ? TB.ref1[i0] : 0u;
    }
    uint32_t rc = 0;  // this lane's reference symbols for its current row (A | B << 16)

    const unsigned nSteps = G + 63;
    for (unsigned t = 1; t <= nSteps; ++t) {
      if (((t - 1) & 63) == 0) {
        curA              = nextA;
        curB              = nextB;
        const unsigned i0 = t - 1 + 64 + unsigned(lane);
        nextA             = (i0 < GA) ? TA.ref1[i0] : 0u;
        nextB             = (i0 < GB) ? TB.ref1[i0] : 0u;
      }
      const unsigned c0 = wv::readlane(curA, int((t - 1) & 63)) | (wv::readlane(curB, int((t - 1) & 63)) << 16);
      rc                = wv::shr1(rc, c0);
      uint32_t incoming[NS];
      // lane 0 receives column 0 from the shift's fill value: rows >= 1 are (0, bad, bad, ..) (GlobalAlignerImpl.hpp:98-107); row 0 is
      // the initial lprev, and from row 2 on lprev is the previous step's lcur, i.e. the same fill
      for (int s = 0; s < NS; ++s) incoming[s] = wv::shr1(st[s][E - 1], (s == ST_MATCH) ? 0u : bad);
      const int g = int(t) - lane;  // this lane's row
      for (int s = 0; s < NS; ++s) {
        lprev[s] = lcur[s];
        lcur[s]  = incoming[s];
      }
      const bool active = (g >= 1) && (unsigned(g) <= G);
      if (!active) continue;

      uint32_t diag[NS], left[NS];
      for (int s = 0; s < NS; ++s) {
        diag[s] = lprev[s];
        left[s] = lcur[s];
      }
      uint32_t cells[E];
      for (int e = 0; e < E; ++e) {
        uint32_t up[NS];
        for (int s = 0; s < NS; ++s) up[s] = st[s][e];
        const uint32_t fc = (e == 0) ? fcMask : 0u;  // (compile-time zero for e > 0)
        // substitution score per half: match where the symbols agree
        const uint32_t differ = wv::pk_min_u16(qc[e] ^ rc, 0x00010001u);
        const uint32_t sub8   = wv::pk_mad_u16(differ, cMisDiff, cMatch);
        uint32_t       nv[NS], code;
        {  // match: max5 over the diagonal cell's states
          uint32_t m = wv::pk_max_i16(wv::pk_max_i16(wv::pk_add_sat_i16(diag[ST_MATCH], cM0), wv::pk_add_sat_i16(diag[ST_DELETE], cD1)), wv::pk_add_sat_i16(diag[ST_INSERT], cI2));
          m          = wv::pk_max_i16(wv::pk_max_i16(m, wv::pk_add_sat_i16(diag[ST_JUMP], cJ3)), wv::pk_add_sat_i16(diag[ST_JUMPINS], cJI4));
          nv[ST_MATCH] = wv::pk_add_sat_i16(m & clrIdx, sub8);
          code         = m & idxBits;
        }
        {  // delete (:121-135)
          uint32_t m = wv::pk_max_i16(wv::pk_max_i16(wv::pk_add_sat_i16(up[ST_MATCH], cOpen0), wv::pk_add_sat_i16(up[ST_DELETE], cD1)), wv::pk_add_sat_i16(up[ST_INSERT], cI2));
          m          = wv::pk_max_i16(wv::pk_max_i16(m, cBad3), wv::pk_add_sat_i16(up[ST_JUMPINS], cJI4));
          uint32_t b = wv::pk_add_sat_i16(m & clrIdx, cExt);
          b = (b & ~fc) | (bad & fc);
          nv[ST_DELETE] = b;
          code |= (m & idxBits) << 3;
        }
        {  // insert (:137-146)
          uint32_t m = wv::pk_max_i16(wv::pk_max_i16(wv::pk_add_sat_i16(left[ST_MATCH], cOpen0), cBad1), wv::pk_add_sat_i16(left[ST_INSERT], cI2));
          uint32_t b = wv::pk_add_sat_i16(m & clrIdx, cExt);
          b = (b & ~fc) | (bad & fc);
          nv[ST_INSERT] = b;
          code |= (m & idxBits) << 6;
        }
        {  // jumpDel (:148-166)
          uint32_t m = wv::pk_max_i16(wv::pk_max_i16(wv::pk_add_sat_i16(up[ST_MATCH], cL0), cBad1), wv::pk_add_sat_i16(up[ST_INSERT], cLmOpen2));
          m          = wv::pk_max_i16(wv::pk_max_i16(m, wv::pk_add_sat_i16(up[ST_JUMP], cJ3)), wv::pk_add_sat_i16(up[ST_JUMPINS], cL4));
          uint32_t b = m & clrIdx;
          b = (b & ~fc) | (bad & fc);
          nv[ST_JUMP] = b;
          code |= (m & idxBits) << 9;
        }
        {  // jumpIns (:169-176)
          uint32_t m = wv::pk_max_i16(wv::pk_max_i16(wv::pk_add_sat_i16(left[ST_MATCH], cL0), cBad1), wv::pk_add_sat_i16(left[ST_JUMPINS], cJI4));
          uint32_t b = m & clrIdx;
          b = (b & ~fc) | (bad & fc);
          nv[ST_JUMPINS] = b;
          code |= (m & idxBits) << 12;
        }
        for (int s = 0; s < NS; ++s) {
          diag[s]  = up[s];
          left[s]  = nv[s];
          st[s][e] = nv[s];
        }
        cells[e] = code;
      }
#ifndef MANTA_PAIR_EXPERIMENT_NO_STORE  // (developer experiment: what the back-pointer stream costs; results are garbage without it)
      for (int e = 0; e < E; ++e) ptr32[(uint64_t(t) * E + e) * 64 + unsigned(lane)] = cells[e];
#else
      if (t == 0xffffffffu) for (int e = 0; e < E; ++e) ptr32[e] = cells[e];
#endif

      // traceback start candidates (see above)
      {
        const uint32_t cap = ((unsigned(g) == GA) ? 0x0000ffffu : 0u) | ((unsigned(g) == GB) ? 0xffff0000u : 0u);
        for (int e = 0; e < E; ++e) lastRow[e] = (lastRow[e] & ~cap) | (st[ST_MATCH][e] & cap);
        for (int h = 0; h < 2; ++h) {
          uint32_t vM = st[ST_MATCH][0];
          for (int e = 1; e < E; ++e) vM = (unsigned(e) == eQ[h]) ? st[ST_MATCH][e] : vM;
          const int  key = int((uint32_t(half(vM, h)) << 16) | (0xffffu - unsigned(g)));
          const bool on  = unsigned(lane) == lQ[h] && unsigned(g) <= Gs[h];
          rowKey[h]      = imax(rowKey[h], on ? key : int(0x80000000u));
        }
      }
    }
    for (int h = 0; h < 2; ++h) {
      // off-edge candidates of the task's last row (q < Q; the reference's extra q == Q term can never win the strict '>');
      // a lane's columns in ascending q, then the lanes: first best wins
      bool     haveOff = false;
      int      offVal  = 0;
      unsigned offQ    = 0;
      for (int e = 0; e < E; ++e) {
        const unsigned q = unsigned(lane) * E + e + 1;
        if (q < Qs[h]) {
          const int v = (half(lastRow[e], h) >> 3) + int((Qs[h] - q) * unsigned(offEdge));
          if (!haveOff || v > offVal) {
            haveOff = true;
            offVal  = v;
            offQ    = q;
          }
        }
      }
      // q == 0 off-edge candidate (column 0 holds match == 0 on every row >= 1); it precedes every other q in scan order
      if (lane == 0) {
        const int v0 = int(Qs[h] * unsigned(offEdge));
        if (!haveOff || v0 >= offVal) {
          haveOff = true;
          offVal  = v0;
          offQ    = 0;
        }
      }
      waveArgmaxFirst(haveOff, offVal, offQ);
      const int key = wv::readlane(rowKey[h], int(lQ[h]));
      StartCand best;
      best.val   = (key >> 16) >> 3;
      best.ref   = 0xffffu - (unsigned(key) & 0xffffu);
      best.query = Qs[h];
      best.state = ST_MATCH;
      best.init  = true;
      candUpdate(best, offVal, Gs[h], offQ, ST_MATCH);
      if (h == 0)
        outA = best;
      else
        outB = best;
    }
  }

  WV_DEV void run(const AlignTaskDev& TA, const AlignTaskDev& TB, AlignResultDev& resA, AlignResultDev& resB, const bool haveB, uint8_t* slab)
  {
    StartCand sa, sb;
    sweep(TA, TB, reinterpret_cast<uint32_t*>(slab), sa, sb);
    wv::sync();  // back-pointers written by all lanes are read by all lanes below
    for (int h = 0; h < (haveB ? 2 : 1); ++h) {
      const AlignTaskDev& T = h ? TB : TA;
      Aligner<1, E, true> al(P);
      al.query      = T.query;
      al.ref1       = T.ref1;
      al.ref2       = nullptr;
      al.Q          = T.query_len;
      al.R1         = T.ref1_len;
      al.R2         = 0;
      al.G          = T.ref1_len;
      al.ptr        = reinterpret_cast<uint16_t*>(slab) + h;
      al.nStrips    = 1;
      al.stripCells = 0;
      AlignResultDev r;
      al.tracebackSingle(h ? sb : sa, T, r);
      if (wv::lane() == 0) (h ? resB : resA) = r;
      wv::sync();
    }
  }
};

/// the work unit is a PAIR of tasks of one E bucket (task_ids[2 i], task_ids[2 i + 1]; an odd last task runs against itself).
/// A wave's slab holds cell pairs: P.ptr_ws_stride is twice the single-alignment stride (api.cpp: alignUsesPairs).
template <int E>
WV_KERNEL void align_pair_kernel(const AlignParams P)
{
  uint8_t*       slab   = P.ptr_ws + uint64_t(wv::block()) * P.ptr_ws_stride;
  const unsigned nTasks = P.n_tasks_dev ? *P.n_tasks_dev : P.n_tasks;
  const unsigned nPairs = (nTasks + 1) / 2;
  while (true) {
    unsigned slot = 0;
    if (wv::lane() == 0) slot = wv::atomic_add(P.counter, 1u);
    slot = wv::first(slot);
    if (slot >= nPairs) break;
    const unsigned ia = 2 * slot, ib = (2 * slot + 1 < nTasks) ? 2 * slot + 1 : 2 * slot;
    const unsigned ta = P.task_ids ? P.task_ids[ia] : ia, tb = P.task_ids ? P.task_ids[ib] : ib;
    PairAligner<E> al(P);
    al.run(P.tasks[ta], P.tasks[tb], P.results[ta], P.results[tb], ib != ia, slab);
    wv::sync();
  }
}

/// Every packed bucket of a block in ONE persistent launch.  Per-bucket launches on side streams do not share the device well:
/// the widest bucket's grid fills the wave slots and the narrow buckets -- few pairs, long dependent chains (short contigs against
/// whole reference windows) -- either wait behind it or crawl beside it (measured: 1.9 ms at 79 % of the VALU issue rate, then
/// 1.1 ms at 28 %).  Here one queue covers the buckets in the order the host gives (fewest tasks first: the long chains start at
/// once, the bulk of the widest bucket fills in behind), and a wave runs whichever width its pair needs.
struct PairMultiParams {
  AlignParams     A;           ///< tasks, results, cigar, scores, counter; ptr_ws / ptr_ws_stride sized for the widest bucket
  const uint32_t* bucket_ids;  ///< [bucket][n_slots]: the buckets' task ids, reference length descending (bucket_sort_kernel)
  const uint32_t* counts;      ///< device: tasks per bucket
  uint32_t        n_slots;
  uint32_t        n_order;     ///< <= 6
  uint8_t         order[8];    ///< bucket indices in queue order
  uint8_t         e_of[8];     ///< their E
  uint32_t        prio_work[3];  ///< a pair of at least this much work (E x (G + 64)) issues at priority 1 / 2 / 3 (s_setprio); 0 = never
};

template <int E>
WV_DEV void runPairWidth(const AlignParams& A, const unsigned ta, const unsigned tb, const bool haveB, uint8_t* slab)
{
  PairAligner<E>(A).run(A.tasks[ta], A.tasks[tb], A.results[ta], A.results[tb], haveB, slab);
}

#if !MANTA_TU_DEFINES(MANTA_TU_ALIGN_PAIR)
WV_KERNEL_OCC(4) void align_pair_multi_kernel(const PairMultiParams M);
#else
WV_KERNEL_OCC(4) void align_pair_multi_kernel(const PairMultiParams M)
{
  uint8_t* slab = M.A.ptr_ws + uint64_t(wv::block()) * M.A.ptr_ws_stride;
  unsigned n0 = 0, n1 = 0, n2 = 0, n3 = 0, n4 = 0, n5 = 0;  // tasks per queue position
  if (M.n_order > 0) n0 = M.counts[M.order[0]];
  if (M.n_order > 1) n1 = M.counts[M.order[1]];
  if (M.n_order > 2) n2 = M.counts[M.order[2]];
  if (M.n_order > 3) n3 = M.counts[M.order[3]];
  if (M.n_order > 4) n4 = M.counts[M.order[4]];
  if (M.n_order > 5) n5 = M.counts[M.order[5]];
  const unsigned f1 = (n0 + 1) / 2, f2 = f1 + (n1 + 1) / 2, f3 = f2 + (n2 + 1) / 2, f4 = f3 + (n3 + 1) / 2, f5 = f4 + (n4 + 1) / 2,
                 f6 = f5 + (n5 + 1) / 2;
  while (true) {
    unsigned slot = 0;
    if (wv::lane() == 0) slot = wv::atomic_add(M.A.counter, 1u);
    slot = wv::first(slot);
    if (slot >= f6) break;
    unsigned j = 0, base = 0, n = n0;
    if (slot >= f1) { j = 1; base = f1; n = n1; }
    if (slot >= f2) { j = 2; base = f2; n = n2; }
    if (slot >= f3) { j = 3; base = f3; n = n3; }
    if (slot >= f4) { j = 4; base = f4; n = n4; }
    if (slot >= f5) { j = 5; base = f5; n = n5; }
    const unsigned  p   = slot - base;
    const uint32_t* ids = M.bucket_ids + uint64_t(M.order[j]) * M.n_slots;
    const unsigned  ia = 2 * p, ib = (2 * p + 1 < n) ? 2 * p + 1 : 2 * p;
    const unsigned  ta = ids[ia], tb = ids[ib];
    {
      // The queue hands out the long sweeps first, but a short contig against a whole reference window is one dependent chain of
      // E x (G + 63) steps' worth of work -- longer than the AVERAGE load of a wave in a block of ~2 pairs per wave -- and under
      // an even share of a saturated SIMD it would finish last with the device idling around it.  Long pairs therefore issue
      // ahead of the short ones of their SIMD; the short ones fill the gaps.
      const unsigned ga = M.A.tasks[ta].ref1_len, gb = M.A.tasks[tb].ref1_len;
      const unsigned work = unsigned(M.e_of[j]) * (((ga > gb) ? ga : gb) + 64u);
      wv::setprio((M.prio_work[2] && work >= M.prio_work[2]) ? 3u : (M.prio_work[1] && work >= M.prio_work[1]) ? 2u : (M.prio_work[0] && work >= M.prio_work[0]) ? 1u : 0u);
    }
    switch (M.e_of[j]) {
    case 1: runPairWidth<1>(M.A, ta, tb, ib != ia, slab); break;
    case 2: runPairWidth<2>(M.A, ta, tb, ib != ia, slab); break;
    case 3: runPairWidth<3>(M.A, ta, tb, ib != ia, slab); break;
    case 4: runPairWidth<4>(M.A, ta, tb, ib != ia, slab); break;
    case 5: runPairWidth<5>(M.A, ta, tb, ib != ia, slab); break;
    default: runPairWidth<6>(M.A, ta, tb, ib != ia, slab); break;
    }
    wv::sync();
  }
}
#endif

#if MANTA_TU != MANTA_TU_ALL
#if MANTA_TU == MANTA_TU_ALIGN_PAIR
#define MANTA_X
#else
#define MANTA_X extern
#endif
MANTA_X template __global__ void align_pair_kernel<1>(const AlignParams);
MANTA_X template __global__ void align_pair_kernel<2>(const AlignParams);
MANTA_X template __global__ void align_pair_kernel<3>(const AlignParams);
MANTA_X template __global__ void align_pair_kernel<4>(const AlignParams);
MANTA_X template __global__ void align_pair_kernel<5>(const AlignParams);
MANTA_X template __global__ void align_pair_kernel<6>(const AlignParams);
#undef MANTA_X
#endif

}  // namespace manta_dev
