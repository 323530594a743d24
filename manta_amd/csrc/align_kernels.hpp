// HIP kernels (gfx950) for Manta's contig-to-reference aligners:
//   KIND 0  GlobalAligner            (alignment/GlobalAlignerImpl.hpp:29-181)            3 states
//   KIND 1  GlobalLargeIndelAligner  (alignment/GlobalLargeIndelAlignerImpl.hpp:35-225)  5 states
//   KIND 2  GlobalJumpAligner        (alignment/GlobalJumpAlignerImpl.hpp:33-333)        4 states, two references
// plus their tracebacks (SingleRefAlignerSharedImpl.hpp:75-168, JumpAlignerBaseImpl.hpp:86-242) and the '='/'X'
// expansion (blt_util/align_path_impl.hpp:33-72).  Paths are relative to /root/reference/src/c++/lib.
//
// Mapping to the hardware: ONE 64-lane wavefront per alignment.  Lane l owns E consecutive query columns
// (q = l*E+1 .. l*E+E) in registers and sweeps the reference rows as a skewed anti-diagonal wavefront: at step
// t lane l computes row g = t - l.  All cross-lane traffic is one `v_mov_b32_dpp wave_shr:1` per DP state per
// step (the right-most column of lane l-1) -- no LDS, no scans.  The reference base is fed the same way
// (a 64-byte coalesced block load every 64 steps + v_readlane into lane 0, then shifted along the lanes).
// The reference's back-pointer matrix (its dominant memory traffic) is streamed to HBM in a
// private step-major layout ptr[(t*E + e)*64 + lane], i.e. every store instruction writes 64 consecutive
// cells; the traceback then walks it with 64-lane speculative look-ahead (one HBM round trip per alignment
// state change instead of one per cell).
//
// Queries longer than 64 lanes x 32 columns (2048 bases; the reference has no limit) run in STRIPS of 2048 columns on the
// E = 32 kernel: strip s sweeps all reference rows like a whole alignment, but its left boundary is not the constant
// column 0 -- it is the right-most column of strip s-1, which that strip left in a small boundary array (one row of DP
// states per reference row, ping-pong between strips); every strip has its own back-pointer slab, the traceback finds a
// cell's slab from its query column.
//
// Scores are int32 exactly as the reference instantiates them (AlignmentScores<int>); badVal = -10000 is a
// finite sentinel that takes part in arithmetic (GlobalJumpAlignerImpl.hpp:68).  Arg-max ties are resolved by
// the reference's strict '>' scan in state order (AlignerBase.hpp:46-59, JumpAlignerBase.hpp:93-111,
// GlobalLargeIndelAligner.hpp:124-151).
#pragma once
#include "wave.hpp"

namespace manta_dev {

enum { ST_MATCH = 0, ST_DELETE = 1, ST_INSERT = 2, ST_JUMP = 3, ST_JUMPINS = 4 };  // alignment/Alignment.hpp:47-56
// CIGAR ops, BAM numbering (type in low 4 bits, length << 4)
enum { OP_M = 0, OP_I = 1, OP_D = 2, OP_N = 3, OP_S = 4, OP_H = 5, OP_P = 6, OP_EQ = 7, OP_X = 8, OP_NONE = 15 };

static const int ALIGN_BAD = -10000;

struct AlignTaskDev {
  const uint8_t* query;  ///< device pointers (1 byte per base, exactly as the reference's std::string)
  const uint8_t* ref1;
  const uint8_t* ref2;
  uint32_t       query_len, ref1_len, ref2_len;
  uint32_t       cigar_off;  ///< first u32 of this task's cigar region; region size = 4*query_len+16 u32
};

struct AlignResultDev {
  int32_t  status;
  int32_t  score;
  int32_t  is_jumped;
  int32_t  begin1, begin2;
  uint32_t jump_insert_size, jump_range;
  uint32_t cigar1_len, cigar2_len;  ///< cigar1 at cigar_off, cigar2 directly after it
};

struct AlignParams {
  const AlignTaskDev* tasks;
  AlignResultDev*     results;
  uint32_t*           cigar;
  const uint32_t*     task_ids;  ///< tasks of this launch (one E bucket); nullptr = identity
  uint32_t            n_tasks;
  const uint32_t*     n_tasks_dev;  ///< if non-null, the task count is read from here (filled by an earlier kernel)
  uint32_t*           counter;  ///< work-queue head (zeroed before launch)
  uint8_t*            ptr_ws;   ///< back-pointer slabs, one per workgroup
  uint64_t            ptr_ws_stride;
  int32_t             match, mismatch, open, extend, off_edge, allow_edge_ins, extra;
};

template <int KIND>
struct KindTraits;
template <>
struct KindTraits<0> {
  static const int NS = 3, BITS = 2;
  typedef uint8_t cell_t;
};
template <>
struct KindTraits<1> {
  static const int NS = 5, BITS = 3;
  typedef uint16_t cell_t;
};
template <>
struct KindTraits<2> {
  static const int NS = 4, BITS = 2;
  typedef uint8_t cell_t;
};

WV_DEV int imax(const int a, const int b) { return (a > b) ? a : b; }

/// strict-'>' running arg-max step
WV_DEV void amax(int& best, int& ptr, const int v, const int idx)
{
  if (v > best) {
    best = v;
    ptr  = idx;
  }
}

/// "first best wins" candidate for the traceback start (alignment/AlignerUtil.hpp:53-67)
struct StartCand {
  int      val;
  unsigned ref, query;
  int      state;
  bool     init;
};
WV_DEV void candUpdate(StartCand& c, const int v, const unsigned ref, const unsigned query, const int state)
{
  if (!c.init || v > c.val) {
    c.val   = v;
    c.ref   = ref;
    c.query = query;
    c.state = state;
    c.init  = true;
  }
}

/// wave-wide (max value, lowest index on ties) over lanes that have `have`; result uniform on all lanes
WV_DEV void waveArgmaxFirst(bool& have, int& val, unsigned& idx)
{
  for (int off = 1; off < 64; off <<= 1) {
    const int      src   = wv::lane() ^ off;
    const int      oval  = wv::shfl(val, src);
    const unsigned oidx  = wv::shfl(idx, src);
    const bool     ohave = wv::shfl(int(have), src) != 0;
    if (ohave && (!have || oval > val || (oval == val && oidx < idx))) {
      have = true;
      val  = oval;
      idx  = oidx;
    }
  }
}

/// PAIR: the back-pointer cells of two alignments interleave in one slab (align_pair.hpp); this alignment reads every second word
template <int KIND, int E, bool PAIR = false>
struct Aligner {
  typedef KindTraits<KIND>          KT;
  typedef typename KT::cell_t       cell_t;
  static const int                  NS = KT::NS;
  static const int                  BITS = KT::BITS;
  static const int                  FMASK = (1 << KT::BITS) - 1;  // a stored field is (7 - state) & FMASK

  const AlignParams& P;
  const uint8_t*     query;
  const uint8_t*     ref1;
  const uint8_t*     ref2;
  unsigned           Q, R1, R2, G;
  cell_t*            ptr;
  static const bool     MULTI = (E == 32);   // the widest kernel also takes longer queries, in strips
  static const unsigned STRIPW = 64u * E;    // query columns per strip
  unsigned           nStrips;
  uint64_t           stripCells;             // back-pointer cells of one strip's slab
  int*               bnd[2];                 // boundary columns between strips: (G + 2) rows x NS states each

  WV_DEV Aligner(const AlignParams& p) : P(p) {}

  WV_DEV uint8_t refChar(const unsigned g0) const  // g0 = 0-based combined row
  {
    return (g0 < R1) ? ref1[g0] : ref2[g0 - R1];
  }

  /// back-pointer for `state` stored at cell (q,g); boundary rows/columns are constants
  /// (GlobalAlignerImpl.hpp:66-80,98-107; GlobalJumpAlignerImpl.hpp:77-94,108-118)
  WV_DEV int ptrField(const unsigned q, const unsigned g, const int state) const
  {
    if (q == 0 || g == 0) {
      if (KIND != 2 && g == 0 && state == ST_INSERT && P.allow_edge_ins) return ST_INSERT;
      return ST_MATCH;
    }
    const unsigned strip = MULTI ? (q - 1) / STRIPW : 0u, qs = MULTI ? (q - 1) % STRIPW : (q - 1);
    const unsigned l = qs / E, e = qs % E;
    const uint64_t idx = uint64_t(strip) * stripCells + (uint64_t(g + l) * E + e) * 64 + l;
    return (7 - ((int(ptr[PAIR ? 2 * idx : idx]) >> (state * BITS)) & FMASK)) & FMASK;
  }

  // ------------------------------------------------------------------------------------------------
  // DP sweep.  Returns the traceback start.
  // ------------------------------------------------------------------------------------------------
  WV_DEV StartCand sweep()
  {
    const int      lane = wv::lane();
    const int      open = P.open, extend = P.extend, L = P.extra, offEdge = P.off_edge;

    StartCand candRows1 = {0, 0, 0, ST_MATCH, false};  // rows of ref1 (or the single reference) at q=Q
    StartCand candRows2 = {0, 0, 0, ST_MATCH, false};  // rows of ref2 at q=Q
    bool      haveOff1 = false, haveOff2 = false;      // off-edge candidates of this lane's columns
    int       off1Val = 0, off2Val = 0;
    unsigned  off1Q = 0, off2Q = 0;
    int       lastRowIns = ALIGN_BAD;
    unsigned  lQ = 0;

    for (unsigned strip = 0; strip < nStrips; ++strip) {
    const unsigned qBase     = strip * STRIPW;  // query columns of this strip: qBase + 1 .. qBase + 64 E
    const bool     lastStrip = (strip + 1 == nStrips);
    lQ                       = lastStrip ? (Q - 1 - qBase) / E : 64u;  // lane that owns column Q (last strip only)
    const unsigned eQ        = (Q - 1 - (lastStrip ? qBase : 0u)) % E;
    cell_t*        ptrS      = ptr + uint64_t(strip) * stripCells;
    const int*     bndIn     = bnd[(strip & 1) ^ 1];  // written by the previous strip
    int*           bndOut    = bnd[strip & 1];

    int     st[NS][E];      // own columns, row (g-1) before / row g after the step
    int     lcur[NS];       // lane l-1's right-most column, row g   (as of its previous step)
    int     lprev[NS];      // same column, row g-1
    uint8_t qc[E];
    for (int e = 0; e < E; ++e) {
      const unsigned q0 = qBase + unsigned(lane) * E + e;  // 0-based query index
      qc[e]             = (q0 < Q) ? query[q0] : uint8_t(0);
      // row 0 (GlobalAlignerImpl.hpp:66-80)
      const unsigned q = q0 + 1;
      for (int s = 0; s < NS; ++s) st[s][e] = ALIGN_BAD;
      st[ST_MATCH][e] = int(q * unsigned(offEdge));
      if (KIND != 2 && P.allow_edge_ins) st[ST_INSERT][e] = open + int(q * unsigned(extend));
    }
    // the strip's left boundary column, row 0, for lane 0 (column 0 in the first strip); other lanes get theirs through
    // the first shifts
    for (int s = 0; s < NS; ++s) lcur[s] = lprev[s] = ALIGN_BAD;
    lcur[ST_MATCH] = lprev[ST_MATCH] = int(qBase * unsigned(offEdge));
    if (KIND != 2 && P.allow_edge_ins) lcur[ST_INSERT] = lprev[ST_INSERT] = open + int(qBase * unsigned(extend));

    unsigned curBlk = 0, nextBlk = 0;
    int      curB[NS], nextB[NS];  // boundary rows of the next 64 steps, one row per lane (strips > 0)
    for (int s = 0; s < NS; ++s) curB[s] = nextB[s] = ALIGN_BAD;
    {
      const unsigned i0 = unsigned(lane);
      nextBlk           = (i0 < G) ? refChar(i0) : 0u;
      if (MULTI && strip > 0 && i0 < G)
        for (int s = 0; s < NS; ++s) nextB[s] = bndIn[size_t(i0 + 1) * NS + s];
    }
    unsigned rc = 0;  // this lane's reference symbol for its current row

    const unsigned nSteps = G + 63;
    for (unsigned t = 1; t <= nSteps; ++t) {
      if (((t - 1) & 63) == 0) {
        curBlk            = nextBlk;
        const unsigned i0 = t - 1 + 64 + unsigned(lane);
        nextBlk           = (i0 < G) ? refChar(i0) : 0u;
        if (MULTI && strip > 0) {
          for (int s = 0; s < NS; ++s) {
            curB[s]  = nextB[s];
            nextB[s] = (i0 < G) ? bndIn[size_t(i0 + 1) * NS + s] : ALIGN_BAD;
          }
        }
      }
      const unsigned c0 = wv::readlane(curBlk, int((t - 1) & 63));
      rc                = wv::shr1(rc, c0);

      // right-most column of the left neighbour, as of the end of the previous step; lane 0 of a later strip receives
      // the previous strip's last column of row g = t instead
      int incoming[NS];
      for (int s = 0; s < NS; ++s) {
        const int fill = (MULTI && strip > 0) ? wv::readlane(curB[s], int((t - 1) & 63)) : ALIGN_BAD;
        incoming[s]    = wv::shr1(st[s][E - 1], fill);
      }
      const int g = int(t) - lane;  // this lane's row
      for (int s = 0; s < NS; ++s) {
        lprev[s] = lcur[s];
        lcur[s]  = incoming[s];
      }
      if (lane == 0 && strip == 0) {
        // column 0: rows >= 1 are (0,bad,bad,..); row 0 handled by the initial lprev
        // (GlobalAlignerImpl.hpp:98-107)
        if (g >= 2) {
          for (int s = 0; s < NS; ++s) lprev[s] = ALIGN_BAD;
          lprev[ST_MATCH] = 0;
        }
        for (int s = 0; s < NS; ++s) lcur[s] = ALIGN_BAD;
        lcur[ST_MATCH] = 0;
      }
      const bool active = (g >= 1) && (unsigned(g) <= G);
      if (!active) continue;

      const bool inRef2 = (KIND == 2) && (unsigned(g) > R1);
      if (KIND == 2 && unsigned(g) == R1 + 1) {
        // seam (GlobalJumpAlignerImpl.hpp:181-204): off-edge candidates of the last ref1 row, then re-seed
        // match/del/ins of the live row while PRESERVING jump.
        for (int e = 0; e < E; ++e) {
          const unsigned q = qBase + unsigned(lane) * E + e + 1;
          if (q < Q) {
            const int v = st[ST_MATCH][e] + int((Q - q) * unsigned(offEdge));
            if (!haveOff1 || v > off1Val) {
              haveOff1 = true;
              off1Val  = v;
              off1Q    = q;
            }
          }
          st[ST_MATCH][e]  = int(q * unsigned(offEdge));
          st[ST_DELETE][e] = ALIGN_BAD;
          st[ST_INSERT][e] = ALIGN_BAD;
        }
        // the diagonal neighbour (row R1 of the column to the left) is re-seeded the same way; its jump value stays
        // (column 0 has none)
        lprev[ST_MATCH]  = int((qBase + unsigned(lane) * E) * unsigned(offEdge));
        lprev[ST_DELETE] = ALIGN_BAD;
        lprev[ST_INSERT] = ALIGN_BAD;
        if (lane == 0 && strip == 0) lprev[ST_JUMP] = ALIGN_BAD;
      }

      int diag[NS], left[NS];
      for (int s = 0; s < NS; ++s) {
        diag[s] = lprev[s];
        left[s] = lcur[s];
      }
      cell_t cells[E];
      for (int e = 0; e < E; ++e) {
        int up[NS];
        for (int s = 0; s < NS; ++s) up[s] = st[s][e];
        const bool firstCol = (e == 0) && (lane == 0) && (strip == 0);
        const int  sub      = (unsigned(qc[e]) == rc) ? P.match : P.mismatch;
        int        nv[NS];
        unsigned   code = 0;
        // Arg-max with "lowest state index wins ties" (the reference's strict-'>' scan) in ONE integer max per
        // candidate: candidate i is scored as value*8 + (7-i), so equal values order by state index; v_max3_i32
        // folds three at a time; value = max >> 3, and the low bits (= 7 - winning state) go straight into the
        // stored back-pointer field (ptrField undoes the 7-x).  |scores| < 2^27, so the shift cannot overflow.
#define MANTA_PK(v, idx) (((v) * 8) + (7 - (idx)))  // (a multiply: left-shifting a negative int is undefined; same instruction)
#define MANTA_PKA(v, add, idx) (((v) * 8) + (((add) * 8) + (7 - (idx))))
        // match
        {
          int m = imax(imax(MANTA_PK(diag[ST_MATCH], 0), MANTA_PK(diag[ST_DELETE], 1)), MANTA_PK(diag[ST_INSERT], 2));
          if (KIND == 1) m = imax(imax(m, MANTA_PK(diag[ST_JUMP], 3)), MANTA_PK(diag[ST_JUMPINS], 4));
          if (KIND == 2 && inRef2) m = imax(m, MANTA_PK(diag[ST_JUMP], 3));
          nv[ST_MATCH] = (m >> 3) + sub;
          code |= unsigned(m & FMASK) << (ST_MATCH * BITS);
        }
        // delete
        {
          int m = imax(imax(MANTA_PKA(up[ST_MATCH], open, 0), MANTA_PK(up[ST_DELETE], 1)), MANTA_PK(up[ST_INSERT], 2));
          if (KIND == 1) m = imax(imax(m, MANTA_PK(ALIGN_BAD, 3)), MANTA_PK(up[ST_JUMPINS], 4));
          int b = (m >> 3) + extend;
          if (firstCol && !inRef2) b = ALIGN_BAD;  // no reset in ref2 (GlobalJumpAlignerImpl.hpp:240-246)
          nv[ST_DELETE] = b;
          code |= unsigned(m & FMASK) << (ST_DELETE * BITS);
        }
        // insert
        {
          int m = imax(imax(MANTA_PKA(left[ST_MATCH], open, 0), MANTA_PK(ALIGN_BAD, 1)), MANTA_PK(left[ST_INSERT], 2));
          if (KIND == 2 && inRef2) m = imax(m, MANTA_PK(left[ST_JUMP], 3));  // jump->ins pays no open (:251-256)
          int b = (m >> 3) + extend;
          if (firstCol && !inRef2) b = ALIGN_BAD;
          nv[ST_INSERT] = b;
          code |= unsigned(m & FMASK) << (ST_INSERT * BITS);
        }
        if (KIND == 1) {
          {  // jumpDel (GlobalLargeIndelAlignerImpl.hpp:148-166)
            int m = imax(imax(MANTA_PKA(up[ST_MATCH], L, 0), MANTA_PK(ALIGN_BAD, 1)), MANTA_PKA(up[ST_INSERT], L - open, 2));
            m     = imax(imax(m, MANTA_PK(up[ST_JUMP], 3)), MANTA_PKA(up[ST_JUMPINS], L, 4));
            int b = m >> 3;
            if (firstCol) b = ALIGN_BAD;
            nv[ST_JUMP] = b;
            code |= unsigned(m & FMASK) << (ST_JUMP * BITS);
          }
          {  // jumpIns (:169-176)
            int m = imax(imax(MANTA_PKA(left[ST_MATCH], L, 0), MANTA_PK(ALIGN_BAD, 1)), MANTA_PK(left[ST_JUMPINS], 4));
            int b = m >> 3;
            if (firstCol) b = ALIGN_BAD;
            nv[ST_JUMPINS] = b;
            code |= unsigned(m & FMASK) << (ST_JUMPINS * BITS);
          }
        }
        if (KIND == 2) {
          if (!inRef2) {  // uses THIS cell's final match / ins (GlobalJumpAlignerImpl.hpp:153-161)
            int m = imax(imax(MANTA_PKA(nv[ST_MATCH], L, 0), MANTA_PK(ALIGN_BAD, 1)), MANTA_PKA(nv[ST_INSERT], L, 2));
            m     = imax(m, MANTA_PK(up[ST_JUMP], 3));
            nv[ST_JUMP] = m >> 3;
            code |= unsigned(m & FMASK) << (ST_JUMP * BITS);
          } else {  // :262-267: pointer = JUMP
            nv[ST_JUMP] = up[ST_JUMP];
            code |= unsigned((7 - ST_JUMP) & FMASK) << (ST_JUMP * BITS);
          }
        }
#undef MANTA_PK
#undef MANTA_PKA
        for (int s = 0; s < NS; ++s) {
          diag[s]  = up[s];
          left[s]  = nv[s];
          st[s][e] = nv[s];
        }
        cells[e] = cell_t(code);
      }
      for (int e = 0; e < E; ++e) ptrS[(uint64_t(t) * E + e) * 64 + unsigned(lane)] = cells[e];
      if (MULTI && !lastStrip && lane == 63)  // the next strip's left boundary: this strip's last column, row g
        for (int s = 0; s < NS; ++s) bndOut[size_t(g) * NS + s] = st[s][E - 1];

      // traceback start candidates at q == Q for this row
      if (lastStrip && unsigned(lane) == lQ) {
        int vM = 0, vI = 0;
        for (int e = 0; e < E; ++e) {
          if (unsigned(e) == eQ) {
            vM = st[ST_MATCH][e];
            vI = st[ST_INSERT][e];
          }
        }
        if (!inRef2)
          candUpdate(candRows1, vM, unsigned(g), Q, ST_MATCH);
        else
          candUpdate(candRows2, vM, unsigned(g), Q, ST_MATCH);
        if (unsigned(g) == G) lastRowIns = vI;
      }
      if (unsigned(g) == G) {
        // off-edge candidates of the last row (q < Q; the reference's extra q==Q term for the large-indel
        // aligner, GlobalLargeIndelAlignerImpl.hpp:211, can never win the strict '>' and is omitted)
        for (int e = 0; e < E; ++e) {
          const unsigned q = qBase + unsigned(lane) * E + e + 1;
          if (q < Q) {
            const int v = st[ST_MATCH][e] + int((Q - q) * unsigned(offEdge));
            if (!haveOff2 || v > off2Val) {
              haveOff2 = true;
              off2Val  = v;
              off2Q    = q;
            }
          }
        }
      }
    }
    if (MULTI && !lastStrip) wv::sync();  // the boundary column is read by other lanes in the next strip
    }  // strips

    // q == 0 off-edge candidates (column 0 holds match == 0 on every row >= 1)
    if (lane == 0) {
      const int v0 = int(Q * unsigned(offEdge));
      if (KIND == 2) {
        if (!haveOff1 || v0 >= off1Val) {  // q=0 precedes every other q in scan order
          haveOff1 = true;
          off1Val  = v0;
          off1Q    = 0;
        }
      }
      if (!haveOff2 || v0 >= off2Val) {
        haveOff2 = true;
        off2Val  = v0;
        off2Q    = 0;
      }
    }
    waveArgmaxFirst(haveOff2, off2Val, off2Q);
    if (KIND == 2) waveArgmaxFirst(haveOff1, off1Val, off1Q);

    // combine in the reference's evaluation order, first best wins
    StartCand best;
    best.val   = wv::readlane(candRows1.val, int(lQ));
    best.ref   = wv::readlane(candRows1.ref, int(lQ));
    best.query = Q;
    best.state = ST_MATCH;
    best.init  = true;
    if (KIND == 2) {
      candUpdate(best, off1Val, R1, off1Q, ST_MATCH);
      const int      v2 = wv::readlane(candRows2.val, int(lQ));
      const unsigned r2 = wv::readlane(candRows2.ref, int(lQ));
      candUpdate(best, v2, r2, Q, ST_MATCH);
    } else if (P.allow_edge_ins) {
      candUpdate(best, wv::readlane(lastRowIns, int(lQ)), G, Q, ST_INSERT);
    }
    candUpdate(best, off2Val, G, off2Q, ST_MATCH);
    return best;
  }

  // ------------------------------------------------------------------------------------------------
  // traceback helpers: raw segments are pushed (in traceback order) into the upper half of the task's
  // cigar region; `emit*` then writes the final '='/'X' cigars forward.
  // ------------------------------------------------------------------------------------------------
  struct RawStack {
    uint32_t* base;
    unsigned  n;
  };
  WV_DEV static void rawPush(RawStack& rs, const int type, const unsigned len, const int pathId)
  {
    if (wv::lane() == 0) rs.base[rs.n] = (len << 5) | (unsigned(pathId) << 4) | unsigned(type);
    rs.n++;
  }
  /// AlignerUtil::updatePath (alignment/AlignerUtil.hpp:31-38)
  WV_DEV static void updatePath(RawStack& rs, const int pathId, int& psType, unsigned& psLen, const int type)
  {
    if (psType == type) return;
    if (psType != OP_NONE) rawPush(rs, psType, psLen, pathId);
    psType = type;
    psLen  = 0;
  }

  struct CigarOut {
    uint32_t* out;
    unsigned  n;
    int       lastType;
    unsigned  lastLen;
  };
  WV_DEV static void cigarFlush(CigarOut& c)
  {
    if (c.lastType != OP_NONE) {
      if (wv::lane() == 0) c.out[c.n] = (c.lastLen << 4) | unsigned(c.lastType);
      c.n++;
    }
    c.lastType = OP_NONE;
    c.lastLen  = 0;
  }
  WV_DEV static void cigarAppendMerged(CigarOut& c, const int type, const unsigned len)
  {
    if (c.lastType == type) {
      c.lastLen += len;
    } else {
      cigarFlush(c);
      c.lastType = type;
      c.lastLen  = len;
    }
  }

  /// Forward pass over the raw segments of one path (they were pushed in reverse): expands alignment matches
  /// into '=' / 'X' runs (any 'N' is a mismatch, align_path_impl.hpp:58-59).  Returns the read length consumed.
  WV_DEV unsigned emitPath(
      const RawStack& rs, const int pathId, const uint8_t* qSeq, const unsigned qLen, const uint8_t* rSeq,
      const unsigned rLen, CigarOut& c) const
  {
    unsigned qpos = 0, rpos = 0;
    for (int i = int(rs.n) - 1; i >= 0; --i) {
      const uint32_t raw = rs.base[i];
      if (int((raw >> 4) & 1) != pathId) continue;
      const int type = int(raw & 15);
      unsigned  len  = raw >> 5;
      if (type == OP_M) {
        while (len > 0) {
          const unsigned chunk = (len < 64) ? len : 64;
          const unsigned j     = unsigned(wv::lane());
          bool           same  = false;
          if (j < chunk && (qpos + j) < qLen && (rpos + j) < rLen) {
            const uint8_t a = qSeq[qpos + j], b = rSeq[rpos + j];
            same            = (a == b) && (a != 'N') && (b != 'N');
          }
          const uint64_t mask = wv::ballot(same);
          unsigned       pos  = 0;
          while (pos < chunk) {
            const uint64_t m      = mask >> pos;
            const bool     isEq   = (m & 1) != 0;
            const uint64_t inv    = isEq ? ~m : m;
            unsigned       run    = (inv == 0) ? 64u : unsigned(wv::ctz(inv));
            if (run > chunk - pos) run = chunk - pos;
            // a non-match segment always separates two match segments, so merging never crosses segments
            cigarAppendMerged(c, isEq ? OP_EQ : OP_X, run);
            pos += run;
          }
          qpos += chunk;
          rpos += chunk;
          len -= chunk;
        }
      } else {
        cigarFlush(c);
        c.lastType = type;
        c.lastLen  = len;
        cigarFlush(c);
        if (type == OP_I || type == OP_S) qpos += len;
        if (type == OP_D || type == OP_N) rpos += len;
      }
    }
    cigarFlush(c);
    return qpos;
  }

  // ------------------------------------------------------------------------------------------------
  // single-reference traceback (SingleRefAlignerSharedImpl.hpp:75-168), 64 cells of look-ahead per round trip
  // ------------------------------------------------------------------------------------------------
  WV_DEV void tracebackSingle(const StartCand& start, const AlignTaskDev& T, AlignResultDev& res)
  {
    unsigned q = start.query, g = start.ref;
    int      state = start.state;
    RawStack rs    = {P.cigar + T.cigar_off + 2 * Q + 8, 0};
    int      psType = OP_NONE;
    unsigned psLen  = 0;
    bool     isJumped = false;
    if (q < Q) {
      psType = OP_S;
      psLen  = Q - q;
    }
    while (true) {
      const bool     isM = (state == ST_MATCH);
      const bool     isD = (state == ST_DELETE) || (state == ST_JUMP);
      const unsigned dq  = isD ? 0u : 1u;
      const unsigned dg  = (isM || isD) ? 1u : 0u;
      unsigned       limit = isM ? ((q < g) ? q : g) : (isD ? g : q);
      if (limit == 0) break;
      const unsigned window = (limit < 64) ? limit : 64;
      const unsigned j      = unsigned(wv::lane());
      int            field  = state;
      if (j < window) field = ptrField(q - j * dq, g - j * dg, state);
      const uint64_t mask = wv::ballot(j < window && field != state);
      unsigned       moves;
      int            next = state;
      if (mask != 0) {
        const unsigned run = unsigned(wv::ctz(mask));
        moves              = run + 1;
        next               = wv::readlane(field, int(run));
      } else {
        moves = window;
      }
      updatePath(rs, 0, psType, psLen, isM ? OP_M : (isD ? OP_D : OP_I));
      psLen += moves;
      if (state == ST_JUMP || state == ST_JUMPINS) isJumped = true;
      q -= moves * dq;
      g -= moves * dg;
      state = next;
    }
    if (psType != OP_NONE) rawPush(rs, psType, psLen, 0);
    if (q != 0) rawPush(rs, OP_S, q, 0);
    wv::sync();

    CigarOut c = {P.cigar + T.cigar_off, 0, OP_NONE, 0};
    emitPath(rs, 0, query, Q, ref1 + g, R1 - g, c);
    res.status           = 0;
    res.score            = start.val;
    res.is_jumped        = isJumped ? 1 : 0;
    res.begin1           = int(g);
    res.begin2           = 0;
    res.jump_insert_size = 0;
    res.jump_range       = 0;
    res.cigar1_len       = c.n;
    res.cigar2_len       = 0;
  }

  // ------------------------------------------------------------------------------------------------
  // jump traceback (JumpAlignerBaseImpl.hpp:86-242)
  // ------------------------------------------------------------------------------------------------
  WV_DEV void tracebackJump(const StartCand& start, const AlignTaskDev& T, AlignResultDev& res)
  {
    unsigned q = start.query, r = start.ref;
    int      state = start.state;
    RawStack rs    = {P.cigar + T.cigar_off + 2 * Q + 8, 0};
    int      psType = OP_NONE;
    unsigned psLen  = 0;
    if (q < Q) {
      psType = OP_S;
      psLen  = Q - q;
    }
    bool     isRef2End = false;
    int      begin2    = 0;
    unsigned jumpInsertSize = 0;
    const unsigned j = unsigned(wv::lane());

    while (q > 0 && r > 0 && !isRef2End) {
      const bool isRef1 = (r <= R1);
      const int  pathId = isRef1 ? 0 : 1;
      if (state != ST_JUMP) {
        const bool     isM = (state == ST_MATCH);
        const bool     isD = (state == ST_DELETE);
        const unsigned dq  = isD ? 0u : 1u;
        const unsigned dg  = (isM || isD) ? 1u : 0u;
        unsigned       limit = isM ? ((q < r) ? q : r) : (isD ? r : q);
        if (!isRef1 && dg && (r - R1) < limit) limit = r - R1;  // handle ref2's first row explicitly
        const unsigned window = (limit < 64) ? limit : 64;
        int            field  = state;
        if (j < window) field = ptrField(q - j * dq, r - j * dg, state);
        const uint64_t mask = wv::ballot(j < window && field != state);
        unsigned       moves;
        int            next = state;
        if (mask != 0) {
          const unsigned run = unsigned(wv::ctz(mask));
          moves              = run + 1;
          next               = wv::readlane(field, int(run));
        } else {
          moves = window;
          // stepped diagonally out of ref2's first row with MATCH->MATCH (:128)
          if (!isRef1 && isM && moves == (r - R1)) isRef2End = true;
        }
        updatePath(rs, pathId, psType, psLen, isM ? OP_M : (isD ? OP_D : OP_I));
        psLen += moves;
        q -= moves * dq;
        r -= moves * dg;
        state = next;
      } else {
        if (psType != OP_NONE) {  // first entry into the jump state (:144-153)
          begin2 = int(r - R1);
          if (psType == OP_I) {
            jumpInsertSize += psLen;
            psType = OP_NONE;
            psLen  = 0;
          } else {
            updatePath(rs, 1, psType, psLen, OP_NONE);
          }
          state = ptrField(q, r, ST_JUMP);
        } else {  // ride the jump state up the reference rows (:154-156)
          const unsigned window = (r < 64) ? r : 64;
          int            field  = ST_JUMP;
          if (j < window) field = ptrField(q, r - j, ST_JUMP);
          const uint64_t mask = wv::ballot(j < window && field != ST_JUMP);
          if (mask != 0) {
            const unsigned run = unsigned(wv::ctz(mask));
            r -= run;
            state = wv::readlane(field, int(run));
          } else {
            r -= window;
          }
        }
      }
    }
    const bool isRef1 = (r < R1);
    const int  pathId = isRef1 ? 0 : 1;
    if (psType != OP_NONE) rawPush(rs, psType, psLen, pathId);
    if (q != 0) rawPush(rs, OP_S, q, pathId);
    int begin1 = 0;
    if (isRef1)
      begin1 = int(r);
    else
      begin2 = int(r - R1);
    wv::sync();

    // read / ref lengths of path 1 (needed for jumpRange and the query offset of path 2)
    unsigned p1Read = 0, p1Ref = 0;
    bool     have1 = false, have2 = false;
    for (unsigned i = 0; i < rs.n; ++i) {
      const uint32_t raw = rs.base[i];
      const int      tp  = int(raw & 15);
      const unsigned len = raw >> 5;
      if (((raw >> 4) & 1) == 0) {
        have1 = true;
        if (tp == OP_M || tp == OP_I || tp == OP_S) p1Read += len;
        if (tp == OP_M || tp == OP_D) p1Ref += len;
      } else {
        have2 = true;
      }
    }
    unsigned jumpRange = 0;
    if (have1 && have2) {  // :204-230
      unsigned i1 = unsigned(begin1) + p1Ref, i2 = unsigned(begin2), iq = p1Read, insCount = jumpInsertSize;
      while (true) {
        // (>=: with offEdge 0 path 1 can run past the end of ref1 through deletions that cross the seam; the reference then
        // compares bytes beyond its string -- undefined, in practice a mismatch: jumpRange 0)
        if (i1 >= R1) break;
        if (insCount > 0) {
          if (iq == Q) break;
          if (ref1[i1] != query[iq]) break;
        } else {
          if (i2 == R2) break;
          if (ref1[i1] != ref2[i2]) break;
        }
        jumpRange++;
        i1++;
        if (insCount > 0) {
          insCount--;
          iq++;
        } else {
          i2++;
        }
      }
    }
    CigarOut c1 = {P.cigar + T.cigar_off, 0, OP_NONE, 0};
    emitPath(rs, 0, query, Q, ref1 + begin1, R1 - unsigned(begin1), c1);
    const unsigned qoff = p1Read + jumpInsertSize;
    CigarOut       c2   = {P.cigar + T.cigar_off + c1.n, 0, OP_NONE, 0};
    emitPath(rs, 1, query + qoff, (qoff < Q) ? (Q - qoff) : 0, ref2 + begin2, R2 - unsigned(begin2), c2);

    res.status           = 0;
    res.score            = start.val;
    res.is_jumped        = 0;
    res.begin1           = begin1;
    res.begin2           = begin2;
    res.jump_insert_size = jumpInsertSize;
    res.jump_range       = jumpRange;
    res.cigar1_len       = c1.n;
    res.cigar2_len       = c2.n;
  }

  WV_DEV void run(const AlignTaskDev& T, AlignResultDev& res, uint8_t* ptrSlab)
  {
    query = T.query;
    ref1  = T.ref1;
    ref2  = T.ref2;
    Q     = T.query_len;
    R1    = T.ref1_len;
    R2    = (KIND == 2) ? T.ref2_len : 0;
    G     = R1 + R2;
    ptr   = reinterpret_cast<cell_t*>(ptrSlab);
    nStrips    = MULTI ? (Q + STRIPW - 1) / STRIPW : 1u;
    stripCells = uint64_t(G + 64 + 1) * E * 64;
    {
      const uint64_t cellsBytes = (uint64_t(nStrips) * stripCells * sizeof(cell_t) + 15) & ~uint64_t(15);
      bnd[0] = reinterpret_cast<int*>(ptrSlab + cellsBytes);
      bnd[1] = bnd[0] + size_t(G + 2) * NS;
    }
    const StartCand start = sweep();
    wv::sync();  // back-pointers written by all lanes are read by all lanes below
    AlignResultDev r;
    if (KIND == 2)
      tracebackJump(start, T, r);
    else
      tracebackSingle(start, T, r);
    if (wv::lane() == 0) res = r;
  }
};

/// bytes of back-pointer slab one alignment needs (host + device agree on this)
WV_HD uint64_t alignPtrSlabBytes(const int kind, const int E, const uint64_t totalRefLen)
{
  const uint64_t cellBytes = (kind == 1) ? 2 : 1;
  return (totalRefLen + 64 + 1) * uint64_t(E) * 64 * cellBytes;
}

/// The reference length to size a slab for: the real one for a query that fits one strip; for a query of several strips
/// (E = 32 only) a larger stand-in, such that alignPtrSlabBytes() covers every strip's slab plus the two boundary columns.
WV_HD uint64_t alignSlabRefLen(const int kind, const int E, const uint64_t queryLen, const uint64_t totalRefLen)
{
  const uint64_t stripW = 64ull * uint64_t(E);
  if (E != 32 || queryLen <= stripW) return totalRefLen;
  const uint64_t nStrips = (queryLen + stripW - 1) / stripW;
  const uint64_t unit    = alignPtrSlabBytes(kind, E, 0) / 65;  // bytes per reference row
  const uint64_t ns      = (kind == 1) ? 5 : ((kind == 2) ? 4 : 3);
  const uint64_t need    = nStrips * alignPtrSlabBytes(kind, E, totalRefLen) + 2 * (totalRefLen + 2) * ns * 4 + 64;
  return (need + unit - 1) / unit;  // (generous by the 65 rows alignPtrSlabBytes adds)
}

// E = 8 is held to 128 VGPRs (it needs 137 unconstrained): 4 waves per SIMD, and it still fits the slot a pipelined
// batch call's assembler leaves free (api.cpp: StageGates)
template <int KIND, int E>
WV_KERNEL_OCC(E == 8 ? 4 : 1) void align_kernel(const AlignParams P)
{
  uint8_t*       slab   = P.ptr_ws + uint64_t(wv::block()) * P.ptr_ws_stride;
  const unsigned nTasks = P.n_tasks_dev ? *P.n_tasks_dev : P.n_tasks;
  while (true) {
    unsigned slot = 0;
    if (wv::lane() == 0) slot = wv::atomic_add(P.counter, 1u);
    slot = wv::first(slot);
    if (slot >= nTasks) break;
    const unsigned      tid = P.task_ids ? P.task_ids[slot] : slot;
    Aligner<KIND, E>    al(P);
    al.run(P.tasks[tid], P.results[tid], slab);
    wv::sync();
  }
}

#if MANTA_TU != MANTA_TU_ALL
// one translation unit per aligner kind (wave.hpp: MANTA_TU); E = the bucket widths of api.cpp's kESet
#define MANTA_ALIGN_INST(X, KIND) \
  X template __global__ void align_kernel<KIND, 1>(const AlignParams);  X template __global__ void align_kernel<KIND, 2>(const AlignParams);  \
  X template __global__ void align_kernel<KIND, 3>(const AlignParams);  X template __global__ void align_kernel<KIND, 4>(const AlignParams);  \
  X template __global__ void align_kernel<KIND, 5>(const AlignParams);  X template __global__ void align_kernel<KIND, 6>(const AlignParams);  \
  X template __global__ void align_kernel<KIND, 8>(const AlignParams);  X template __global__ void align_kernel<KIND, 10>(const AlignParams); \
  X template __global__ void align_kernel<KIND, 12>(const AlignParams); X template __global__ void align_kernel<KIND, 16>(const AlignParams); \
  X template __global__ void align_kernel<KIND, 24>(const AlignParams); X template __global__ void align_kernel<KIND, 32>(const AlignParams);
#if MANTA_TU == MANTA_TU_ALIGN0
MANTA_ALIGN_INST(, 0)
#else
MANTA_ALIGN_INST(extern, 0)
#endif
#if MANTA_TU == MANTA_TU_ALIGN1
MANTA_ALIGN_INST(, 1)
#else
MANTA_ALIGN_INST(extern, 1)
#endif
#if MANTA_TU == MANTA_TU_ALIGN2
MANTA_ALIGN_INST(, 2)
#else
MANTA_ALIGN_INST(extern, 2)
#endif
#undef MANTA_ALIGN_INST
#endif

}  // namespace manta_dev
