// GlobalJumpAligner (alignment/GlobalJumpAlignerImpl.hpp:33-333), TWO alignments per wavefront in the two int16 halves of every
// register -- the packed form align_pair.hpp gives GlobalLargeIndelAligner, for the aligner of the spanning path.
//
// Why: the sweep's instructions are the half-rate ones of this part (v_max, DPP moves, pointer packing: profiles/r05_valu_ceiling.txt);
// a packed instruction does two alignments' cells in one such slot, and the back-pointer stream takes half the stores.
//
// Domain.  Four states (match, delete, insert, jump) = four arg-max candidates at most = TWO index bits: a cell holds value * 4 with the
// candidate tag (3 - index) below it, so one v_pk_max_i16 is the reference's strict-'>' scan in state order (JumpAlignerBase.hpp:93-111) for
// both alignments.  The sentinel is INT16_MIN (a multiple of 4); saturating adds keep it there.  jumpPairEligible() admits a bucket of E
// columns per lane only if no cell the traceback can visit leaves the range (the bound of align_pair.hpp:16-19 with 8 192 for 4 096).
//
// Two references.  The combined rows of a task are ref1's, then ref2's (G = R1 + R2); the two tasks of a pair have their seams on
// different rows, so everything the unpacked kernel does "in ref2" or "at the seam" is a per-half MASK here: the jump candidate of match and
// insert exists behind the seam only, the first-column reset before it only, the jump state itself is computed before and carried behind
// it, and the seam's re-seed (:181-204: match = q * offEdge, delete = insert = bad, jump PRESERVED) blends into the half whose row it is.
// The shorter task of a pair computes rows past its end against base 0; its start candidates stop at its own last row.
//
// Back-pointers: 4 states x 2 bits = one byte per cell and alignment; a pair's two bytes interleave in one 16-bit stream
// (ptr16[(t * E + e) * 64 + lane]), and the traceback is align_kernel's own (Aligner<2, E, true>::tracebackJump reads every second byte).
#pragma once
#include "align_pair.hpp"

namespace manta_dev {

static const int JP_BAD4 = -32768;  // the sentinel in the x4 domain

/// may a bucket of E columns per lane run on the packed jump kernel with these scores ?
WV_HD bool jumpPairEligible(const int E, const int match, const int mismatch, const int open, const int extend, const int offEdge, const int jump)
{
  if (E > 8) return false;
  if (open > 0 || extend > 0 || jump > 0 || match < 0 || match > 64 || mismatch < -1024 || offEdge < -1024 || open < -4096 || jump < -4096 || extend < -1024)
    return false;
  const long q      = 64L * E;
  const long perCol = -long((mismatch < offEdge) ? ((mismatch < 0) ? mismatch : 0) : ((offEdge < 0) ? offEdge : 0));
  const long gaps   = -long(open) - long(extend) - long(jump);
  const long up     = q * long(match);
  return q * perCol + gaps + up + 64 < 8192;
}

template <int E>
struct JumpPairAligner {
  static const int   NS = 4;
  const AlignParams& P;

  WV_DEV JumpPairAligner(const AlignParams& p) : P(p) {}

  WV_DEV static uint32_t pk(const int lo, const int hi) { return (uint32_t(lo) & 0xffffu) | (uint32_t(hi) << 16); }
  WV_DEV static uint32_t pk2(const int v) { return pk(v, v); }
  WV_DEV static int      half(const uint32_t v, const int h) { return h ? (int(v) >> 16) : int(int16_t(v & 0xffffu)); }
  /// (a & m) | (b & ~m)
  WV_DEV static uint32_t blend(const uint32_t m, const uint32_t a, const uint32_t b) { return (a & m) | (b & ~m); }

  /// both tasks' sweeps; returns their traceback starts.  ptr16: cell pairs, layout ptr16[(t * E + e) * 64 + lane]
  WV_DEV void sweep(const AlignTaskDev& TA, const AlignTaskDev& TB, uint16_t* ptr16, StartCand& outA, StartCand& outB)
  {
    const int      lane = wv::lane();
    const unsigned Qs[2] = {TA.query_len, TB.query_len}, R1s[2] = {TA.ref1_len, TB.ref1_len};
    const unsigned Gs[2] = {TA.ref1_len + TA.ref2_len, TB.ref1_len + TB.ref2_len};
    const unsigned G     = (Gs[0] > Gs[1]) ? Gs[0] : Gs[1];
    const int      open = P.open, extend = P.extend, J = P.extra, offEdge = P.off_edge;
    // packed constants (x4 domain; the tag of candidate i is 3 - i)
    const uint32_t cM0 = pk2(3), cD1 = pk2(2), cI2 = pk2(1), cJ3 = pk2(0);
    const uint32_t cOpen0 = pk2(open * 4 + 3);  // match + open, candidate 0
    const uint32_t cJ0    = pk2(J * 4 + 3);     // match + jump score, candidate 0
    const uint32_t cJ2    = pk2(J * 4 + 1);     // insert + jump score, candidate 2
    const uint32_t cBad1  = pk2(JP_BAD4 + 2);
    const uint32_t cExt = pk2(extend * 4), cMatch = pk2(P.match * 4), cMisDiff = pk2((P.mismatch - P.match) * 4);
    const uint32_t clrIdx = 0xfffcfffcu, idxBits = 0x00030003u, bad = pk2(JP_BAD4);

    unsigned lQ[2], eQ[2];
    for (int h = 0; h < 2; ++h) {
      lQ[h] = (Qs[h] - 1) / E;
      eQ[h] = (Qs[h] - 1) % E;
    }
    // traceback start candidates (AlignerUtil.hpp:53-67): rows at q == Q as one 32-bit key per task and reference, (value x 4) << 16 |
    // (0xffff - row), kept by a plain integer max on the lane that owns column Q (higher value, among equal values the EARLIER row);
    // the off-edge candidates of a task's last ref1 row (taken at the seam, :181-196) and of its last row are captured and evaluated
    // after the sweep
    int      rowKey1[2] = {int(0x80000000u), int(0x80000000u)}, rowKey2[2] = {int(0x80000000u), int(0x80000000u)};
    uint32_t seamRow[E], lastRow[E], seedRow[E];
    const uint32_t fcMask = (lane == 0) ? 0xffffffffu : 0u;  // column 1 of the query: gap states are reset in ref1 (:130, :143)

    uint32_t st[NS][E], lcur[NS], lprev[NS], qc[E];
    for (int e = 0; e < E; ++e) {
      const unsigned q0 = unsigned(lane) * E + e;
      const unsigned a = (q0 < Qs[0]) ? TA.query[q0] : 0u, b = (q0 < Qs[1]) ? TB.query[q0] : 0u;
      qc[e]            = a | (b << 16);
      const int row0   = int((q0 + 1) * unsigned(offEdge)) * 4;  // row 0 and the seam's re-seed: q * offEdge (:77-94, :197-204)
      seedRow[e]       = pk2(row0 < JP_BAD4 ? JP_BAD4 : row0);
      for (int s = 0; s < NS; ++s) st[s][e] = bad;
      st[ST_MATCH][e] = seedRow[e];
      seamRow[e] = lastRow[e] = bad;
    }
    // the seam's value of the column to the left of this lane's first one (q = lane * E; column 0: 0)
    const uint32_t seedLeft = pk2((int(unsigned(lane) * E * unsigned(offEdge)) * 4 < JP_BAD4) ? JP_BAD4 : int(unsigned(lane) * E * unsigned(offEdge)) * 4);
    for (int s = 0; s < NS; ++s) lcur[s] = lprev[s] = bad;
    lcur[ST_MATCH] = lprev[ST_MATCH] = 0;  // column 0, row 0

    auto refCharA = [&](const unsigned i) -> unsigned { return (i < R1s[0]) ? TA.ref1[i] : TA.ref2[i - R1s[0]]; };
    auto refCharB = [&](const unsigned i) -> unsigned { return (i < R1s[1]) ? TB.ref1[i] : TB.ref2[i - R1s[1]]; };
    unsigned curA = 0, curB = 0, nextA = 0, nextB = 0;
    {
      const unsigned i0 = unsigned(lane);
      nextA             = (i0 < Gs[0]) ? refCharA(i0) : 0u;
      nextB             = (i0 < Gs[1]) ? refCharB(i0) : 0u;
    }
    uint32_t rc = 0;  // this lane's reference symbols for its current row (A | B << 16)

    const unsigned nSteps = G + 63;
    for (unsigned t = 1; t <= nSteps; ++t) {
      if (((t - 1) & 63) == 0) {
        curA              = nextA;
        curB              = nextB;
        const unsigned i0 = t - 1 + 64 + unsigned(lane);
        nextA             = (i0 < Gs[0]) ? refCharA(i0) : 0u;
        nextB             = (i0 < Gs[1]) ? refCharB(i0) : 0u;
      }
      const unsigned c0 = wv::readlane(curA, int((t - 1) & 63)) | (wv::readlane(curB, int((t - 1) & 63)) << 16);
      rc                = wv::shr1(rc, c0);
      uint32_t incoming[NS];
      // lane 0 receives column 0 from the shift's fill value: rows >= 1 are (0, bad, bad, bad) (:108-118)
      for (int s = 0; s < NS; ++s) incoming[s] = wv::shr1(st[s][E - 1], (s == ST_MATCH) ? 0u : bad);
      const int g = int(t) - lane;  // this lane's row
      for (int s = 0; s < NS; ++s) {
        lprev[s] = lcur[s];
        lcur[s]  = incoming[s];
      }
      const bool active = (g >= 1) && (unsigned(g) <= G);
      if (!active) continue;

      // per-half masks of this row: behind the seam / the seam's own row (the first row of ref2)
      const uint32_t in2  = ((unsigned(g) > R1s[0]) ? 0x0000ffffu : 0u) | ((unsigned(g) > R1s[1]) ? 0xffff0000u : 0u);
      const uint32_t seam = ((unsigned(g) == R1s[0] + 1) ? 0x0000ffffu : 0u) | ((unsigned(g) == R1s[1] + 1) ? 0xffff0000u : 0u);
      if (seam) {
        // (:181-204) the last ref1 row's match states are the off-edge candidates of ref1; then match / delete / insert of the live row are
        // re-seeded while jump stays.  The diagonal neighbour (row R1 of the column to the left) is re-seeded the same way.
        for (int e = 0; e < E; ++e) {
          seamRow[e]       = blend(seam, st[ST_MATCH][e], seamRow[e]);
          st[ST_MATCH][e]  = blend(seam, seedRow[e], st[ST_MATCH][e]);
          st[ST_DELETE][e] = blend(seam, bad, st[ST_DELETE][e]);
          st[ST_INSERT][e] = blend(seam, bad, st[ST_INSERT][e]);
        }
        lprev[ST_MATCH]  = blend(seam, seedLeft, lprev[ST_MATCH]);
        lprev[ST_DELETE] = blend(seam, bad, lprev[ST_DELETE]);
        lprev[ST_INSERT] = blend(seam, bad, lprev[ST_INSERT]);
      }
      const uint32_t fcRef1 = fcMask & ~in2;  // the first-column reset applies in ref1 only (:240-246)

      uint32_t diag[NS], left[NS];
      for (int s = 0; s < NS; ++s) {
        diag[s] = lprev[s];
        left[s] = lcur[s];
      }
      uint32_t cells[E];
      for (int e = 0; e < E; ++e) {
        uint32_t up[NS];
        for (int s = 0; s < NS; ++s) up[s] = st[s][e];
        const uint32_t fc = (e == 0) ? fcRef1 : 0u;  // (compile-time zero for e > 0)
        // substitution score per half: match where the symbols agree
        const uint32_t differ = wv::pk_min_u16(qc[e] ^ rc, 0x00010001u);
        const uint32_t sub4   = wv::pk_mad_u16(differ, cMisDiff, cMatch);
        uint32_t       nv[NS], code;
        {  // match: max over the diagonal cell's match / delete / insert, behind the seam also its jump (:122-128, :228-238)
          uint32_t m = wv::pk_max_i16(wv::pk_max_i16(wv::pk_add_sat_i16(diag[ST_MATCH], cM0), wv::pk_add_sat_i16(diag[ST_DELETE], cD1)), wv::pk_add_sat_i16(diag[ST_INSERT], cI2));
          m          = wv::pk_max_i16(m, blend(in2, wv::pk_add_sat_i16(diag[ST_JUMP], cJ3), bad));
          nv[ST_MATCH] = wv::pk_add_sat_i16(m & clrIdx, sub4);
          code         = m & idxBits;
        }
        {  // delete (:130-140, :240-246)
          const uint32_t m = wv::pk_max_i16(wv::pk_max_i16(wv::pk_add_sat_i16(up[ST_MATCH], cOpen0), wv::pk_add_sat_i16(up[ST_DELETE], cD1)), wv::pk_add_sat_i16(up[ST_INSERT], cI2));
          uint32_t       b = wv::pk_add_sat_i16(m & clrIdx, cExt);
          b                = blend(fc, bad, b);
          nv[ST_DELETE]    = b;
          code |= (m & idxBits) << 2;
        }
        {  // insert (:142-151); behind the seam a jump turns into an insertion without an open (:248-259)
          uint32_t m = wv::pk_max_i16(wv::pk_max_i16(wv::pk_add_sat_i16(left[ST_MATCH], cOpen0), cBad1), wv::pk_add_sat_i16(left[ST_INSERT], cI2));
          m          = wv::pk_max_i16(m, blend(in2, wv::pk_add_sat_i16(left[ST_JUMP], cJ3), bad));
          uint32_t b = wv::pk_add_sat_i16(m & clrIdx, cExt);
          b          = blend(fc, bad, b);
          nv[ST_INSERT] = b;
          code |= (m & idxBits) << 4;
        }
        {  // jump: before the seam from THIS cell's final match / insert or carried down the column (:153-161); behind it carried, pointer JUMP (:262-267)
          uint32_t m = wv::pk_max_i16(wv::pk_max_i16(wv::pk_add_sat_i16(nv[ST_MATCH], cJ0), cBad1), wv::pk_add_sat_i16(nv[ST_INSERT], cJ2));
          m          = wv::pk_max_i16(m, wv::pk_add_sat_i16(up[ST_JUMP], cJ3));
          nv[ST_JUMP] = blend(in2, up[ST_JUMP], m & clrIdx);
          code |= ((m & idxBits) & ~in2) << 6;  // (behind the seam the stored tag is 3 - ST_JUMP = 0)
        }
        for (int s = 0; s < NS; ++s) {
          diag[s]  = up[s];
          left[s]  = nv[s];
          st[s][e] = nv[s];
        }
        cells[e] = (code & 0xffu) | ((code >> 8) & 0xff00u);  // A's byte | B's byte
      }
      for (int e = 0; e < E; ++e) ptr16[(uint64_t(t) * E + e) * 64 + unsigned(lane)] = uint16_t(cells[e]);

      // traceback start candidates (see above)
      {
        const uint32_t cap = ((unsigned(g) == Gs[0]) ? 0x0000ffffu : 0u) | ((unsigned(g) == Gs[1]) ? 0xffff0000u : 0u);
        if (cap)
          for (int e = 0; e < E; ++e) lastRow[e] = blend(cap, st[ST_MATCH][e], lastRow[e]);
        for (int h = 0; h < 2; ++h) {
          uint32_t vM = st[ST_MATCH][0];
          for (int e = 1; e < E; ++e) vM = (unsigned(e) == eQ[h]) ? st[ST_MATCH][e] : vM;
          const int  key = int((uint32_t(half(vM, h)) << 16) | (0xffffu - unsigned(g)));
          const bool on  = unsigned(lane) == lQ[h] && unsigned(g) <= Gs[h];
          if (unsigned(g) <= R1s[h])
            rowKey1[h] = imax(rowKey1[h], on ? key : int(0x80000000u));
          else
            rowKey2[h] = imax(rowKey2[h], on ? key : int(0x80000000u));
        }
      }
    }
    for (int h = 0; h < 2; ++h) {
      // off-edge candidates (q < Q) of the last ref1 row and of the last row; a lane's columns in ascending q, then the lanes:
      // first best wins; q == 0 (column 0 holds match == 0 on every row >= 1) precedes every other q in scan order
      auto offEdgeBest = [&](const uint32_t (&row)[E], int& val, unsigned& qBest) {
        bool have = false;
        val       = 0;
        qBest     = 0;
        for (int e = 0; e < E; ++e) {
          const unsigned q = unsigned(lane) * E + e + 1;
          if (q < Qs[h]) {
            const int v = (half(row[e], h) >> 2) + int((Qs[h] - q) * unsigned(offEdge));
            if (!have || v > val) {
              have  = true;
              val   = v;
              qBest = q;
            }
          }
        }
        if (lane == 0) {
          const int v0 = int(Qs[h] * unsigned(offEdge));
          if (!have || v0 >= val) {
            have  = true;
            val   = v0;
            qBest = 0;
          }
        }
        waveArgmaxFirst(have, val, qBest);
      };
      int      off1Val, off2Val;
      unsigned off1Q, off2Q;
      offEdgeBest(seamRow, off1Val, off1Q);
      offEdgeBest(lastRow, off2Val, off2Q);
      // combine in the reference's evaluation order, first best wins: rows of ref1, ref1's off-edge, rows of ref2, the last row's off-edge
      const int key1 = wv::readlane(rowKey1[h], int(lQ[h])), key2 = wv::readlane(rowKey2[h], int(lQ[h]));
      StartCand best;
      best.val   = (key1 >> 16) >> 2;
      best.ref   = 0xffffu - (unsigned(key1) & 0xffffu);
      best.query = Qs[h];
      best.state = ST_MATCH;
      best.init  = true;
      candUpdate(best, off1Val, R1s[h], off1Q, ST_MATCH);
      candUpdate(best, (key2 >> 16) >> 2, 0xffffu - (unsigned(key2) & 0xffffu), Qs[h], ST_MATCH);
      candUpdate(best, off2Val, Gs[h], off2Q, ST_MATCH);
      if (h == 0)
        outA = best;
      else
        outB = best;
    }
  }

  WV_DEV void run(const AlignTaskDev& TA, const AlignTaskDev& TB, AlignResultDev& resA, AlignResultDev& resB, const bool haveB, uint8_t* slab)
  {
    StartCand sa, sb;
    sweep(TA, TB, reinterpret_cast<uint16_t*>(slab), sa, sb);
    wv::sync();  // back-pointers written by all lanes are read by all lanes below
    for (int h = 0; h < (haveB ? 2 : 1); ++h) {
      const AlignTaskDev& T = h ? TB : TA;
      Aligner<2, E, true> al(P);
      al.query      = T.query;
      al.ref1       = T.ref1;
      al.ref2       = T.ref2;
      al.Q          = T.query_len;
      al.R1         = T.ref1_len;
      al.R2         = T.ref2_len;
      al.G          = T.ref1_len + T.ref2_len;
      al.ptr        = slab + h;
      al.nStrips    = 1;
      al.stripCells = 0;
      AlignResultDev r;
      al.tracebackJump(h ? sb : sa, T, r);
      if (wv::lane() == 0) (h ? resB : resA) = r;
      wv::sync();
    }
  }
};

/// the work unit is a PAIR of tasks of one E bucket (task_ids[2 i], task_ids[2 i + 1]; an odd last task runs against itself).
/// A wave's slab holds cell pairs: P.ptr_ws_stride is twice the single-alignment stride (api.cpp: alignUsesPairs).
template <int E>
WV_KERNEL_OCC(E >= 6 ? 4 : 1) void align_jump_pair_kernel(const AlignParams P)
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
    JumpPairAligner<E> al(P);
    al.run(P.tasks[ta], P.tasks[tb], P.results[ta], P.results[tb], ib != ia, slab);
    wv::sync();
  }
}

#if MANTA_TU != MANTA_TU_ALL
#if MANTA_TU == MANTA_TU_JUMP_PAIR
#define MANTA_X
#else
#define MANTA_X extern
#endif
MANTA_X template __global__ void align_jump_pair_kernel<1>(const AlignParams);
MANTA_X template __global__ void align_jump_pair_kernel<2>(const AlignParams);
MANTA_X template __global__ void align_jump_pair_kernel<3>(const AlignParams);
MANTA_X template __global__ void align_jump_pair_kernel<4>(const AlignParams);
MANTA_X template __global__ void align_jump_pair_kernel<5>(const AlignParams);
MANTA_X template __global__ void align_jump_pair_kernel<6>(const AlignParams);
MANTA_X template __global__ void align_jump_pair_kernel<8>(const AlignParams);
#undef MANTA_X
#endif

}  // namespace manta_dev
