// contig_kernel: the serial half of the LDS assembler pipeline (see asm_lds.hpp).  One single-wave workgroup per locus; LDS
// holds the compact graph graph_kernel left in the locus' slab -- 16-byte node records in SEED ORDER, the bitsets of the words
// with more than one read -- and a few hundred bytes of state.  No pile, no table, no keys.
//
//   cycle test     two-sided Kahn peel over the records (a cyclic graph needs the reference's order-dependent repeat search:
//                  general path)
//   contig loop    buildContigs :685-713 on SPECULATION: next to the first seed's walk (one lane) up to 63 lanes walk the words
//                  most likely to be the next seeds (graph_kernel's list).  A walk (:149-501) reads only immutable graph data,
//                  so its result is valid whenever its seed turns out to be the reference's next seed; results are cached by
//                  word and the reference's seed sequence is replayed over them.  With ids in seed order "the next unused seeds"
//                  are the lowest set bits of a bitmap.
//   selectContigs  :722-842, lane = candidate
//
// In an acyclic graph a walk cannot meet one of its own words again except through a self loop, so the per-walk visited set is
// "chosen word == current word".  Every walk LOGS its words (16-bit ids, four per 8-byte store into the workgroup's workspace);
// accepting a walk clears its words from the unused bitmap, so "was this seed consumed by an accepted contig" is one bit test.
// (Round 4's first cut kept a 64-bit lane mask per word, updated by one L2 atomic per lane and step: 230 M atomics per 10 000
// loci -- the L2's atomic rate, not the waves in flight, set the kernel's speed.)
#pragma once
#ifdef MANTA_WAVE_EMU
#include <map>
#include <set>
#endif
#include "asm_lds.hpp"

namespace manta_dev {

#ifndef MANTA_CK_FAST_CAP
#define MANTA_CK_FAST_CAP 3
#endif
static const unsigned CK_FAST_CAP = MANTA_CK_FAST_CAP;  // fast steps a lane may run ahead between two general steps of its wave
static const unsigned CK_MAX_EXT = 1020;  // extension steps of one walk (both directions); longer contigs (never seen on piles this small): general path

struct CkWsLayout {
  uint64_t lane_seq, lane_bits, lane_meta, lane_log, total;
};
static const unsigned CK_SEQ_WORDS = CK_MAX_EXT / 16 + 2;   // per direction: 2-bit codes, 16 per dword
static const unsigned CK_LOG_QW    = (CK_MAX_EXT + 4) / 4;  // per walk: 16-bit word ids, 4 per qword (entry 0 = the seed)
/// `setw`: qwords of a read set (2: the small class, 4: the big one)
WV_HD CkWsLayout ckWorkspaceLayout(const unsigned setw = 2)
{
  CkWsLayout L;
  uint64_t   o = 0;
  L.lane_seq  = asmPut(o, 64ull * 2 * 4 * CK_SEQ_WORDS);
  L.lane_bits = asmPut(o, 64ull * 2 * setw * 8);
  L.lane_meta = asmPut(o, 64ull * 8 * 4);
  L.lane_log  = asmPut(o, 64ull * 8 * CK_LOG_QW);
  L.total     = (o + 255) & ~uint64_t(255);
  return L;
}

enum { CK_DONE = 0, CK_PUNT = 1, CK_NEXT = 2 };  // CK_NEXT (big class): the locus goes on to the next word length, nothing emitted yet

template <class C>
struct LdsContig {
  typedef LgRec<C>  R;
  typedef FSetT<C>  Set;
  typedef CkMap<C>  M;
  static const unsigned SW  = C::SETW;
  static const unsigned IDB = C::ID_BITS;
  static const unsigned IDM = (1u << C::ID_BITS) - 1u;
  static const unsigned UPL = C::UNUSED_DW / 64;  ///< dwords of the unused-words bitmap a lane owns (consecutive ones)
  const AsmParams& P;
  const LgParams&  G;
  char*            lds;
  const uint8_t*   slab;
  FRec8*           nodes;
  Set*             pool;
  uint8_t*         rd1;  ///< big class: the only read of a single-read word, by id
  uint32_t*        unused_bits;
  uint16_t *       tent, *slotNode, *sib, *sovf, *povf;
  LgSlab           SL;
  uint8_t*         tbl;
  uint32_t*        lane_seq;
  uint64_t*        lane_bits;
  int32_t*         lane_meta;
  uint64_t*        lane_log;
  unsigned         lane, nNormal, W, k, nNodes, nFat, nEligible, nSpec, nSib, nSovf, nPovf, codeWords, nCand, maxLen, seqWords;
  bool             acyclic;
  // big class, word-length rounds: a CYCLIC graph (repeat_big_kernel left the core and repeat-word bitmaps) and pseudo reads
  bool             cyclic, anyRep;
  unsigned         nPseudo, nCore, visDw;
  uint32_t *       coreBits, *repBits, *vis;  ///< vis: one bitmap over the core per lane, dword w of lane l at [w * 64 + l]
  uint16_t*        corePrefix;
  unsigned         candSlotV;  // lane c: cache slot that holds candidate c's walk
  uint64_t         tMark;

  WV_DEV LdsContig(const AsmParams& p, const LgParams& g, char* base, uint8_t* ws) : P(p), G(g), lds(base)
  {
    lane        = unsigned(wv::lane());
    unused_bits = reinterpret_cast<uint32_t*>(lds + M::UNUSED);
    tent        = reinterpret_cast<uint16_t*>(lds + M::TENT);
    slotNode    = reinterpret_cast<uint16_t*>(lds + M::SLOTND);
    tbl         = reinterpret_cast<uint8_t*>(lds + M::TBL);
    sib         = reinterpret_cast<uint16_t*>(lds + M::SIB);
    sovf        = reinterpret_cast<uint16_t*>(lds + M::SOVF);
    povf        = reinterpret_cast<uint16_t*>(lds + M::POVF);
    nodes       = reinterpret_cast<FRec8*>(lds + M::RECS);
    pool        = nullptr;
    rd1         = nullptr;
    const CkWsLayout L = ckWorkspaceLayout(SW);
    lane_seq  = reinterpret_cast<uint32_t*>(ws + L.lane_seq);
    lane_bits = reinterpret_cast<uint64_t*>(ws + L.lane_bits);
    lane_meta = reinterpret_cast<int32_t*>(ws + L.lane_meta);
    lane_log  = reinterpret_cast<uint64_t*>(ws + L.lane_log);
    maxLen    = p.max_contig_len;
    seqWords  = CK_SEQ_WORDS;
    candSlotV = 0;
    nCand     = 0;
    cyclic = anyRep = false;
    nPseudo = nCore = visDw = 0;
    coreBits = repBits = vis = nullptr;
    corePrefix = nullptr;
  }

  WV_DEV void tick(const int phase)
  {
#if defined(MANTA_ASM_PROFILE) && !defined(MANTA_LG_PROFILE_GRAPH)
    const uint64_t now = wv::clock();
    if (P.phase_cycles && lane == 0) wv::atomic_add(&P.phase_cycles[phase], (unsigned long long)(now - tMark));
    tMark = now;
#else
    (void)phase;
#endif
  }

  WV_DEV bool isUnused(const unsigned nd) const { return (unused_bits[nd >> 5] >> (nd & 31)) & 1u; }
  WV_DEV char*    scratch() const { return lds + M::RECS + ((8 * nNodes + 15) & ~15u) + (C::BIG ? ((nNodes + 15) & ~15u) : 0u); }

  /// the only read of single-read word nd (record w)
  WV_DEV unsigned onlyRead(const unsigned nd, const FRec8 w) const { return C::BIG ? unsigned(rd1[nd]) : lg8Read(w); }
  /// read support of word `nd` (record w)
  WV_DEV Set supOf(const unsigned nd, const FRec8 w) const
  {
    Set s;
    if (nd < nFat) {
      s = pool[nd];
    } else {
      const unsigned ref = onlyRead(nd, w);
      for (unsigned q = 0; q < SW; ++q) s.w[q] = ((ref >> 6) == q) ? (uint64_t(1) << (ref & 63)) : 0;
    }
    return s;
  }
  /// successors / predecessors of word nd as 4 x ID_BITS (id + 1)
  WV_DEV uint64_t succOf(const unsigned nd, const FRec8 w) const { return R::links(w, nd, true, sovf, nSovf); }
  WV_DEV uint64_t predOf(const unsigned nd, const FRec8 w) const { return R::links(w, nd, false, povf, nPovf); }
  // set algebra
  WV_DEV static unsigned popSet(const Set& a)
  {
    unsigned n = 0;
    for (unsigned q = 0; q < SW; ++q) n += unsigned(wv::popc(a.w[q]));
    return n;
  }
  WV_DEV static Set setAnd(const Set& a, const Set& b)
  {
    Set r;
    for (unsigned q = 0; q < SW; ++q) r.w[q] = a.w[q] & b.w[q];
    return r;
  }
  WV_DEV static Set setAndNot(const Set& a, const Set& b)
  {
    Set r;
    for (unsigned q = 0; q < SW; ++q) r.w[q] = a.w[q] & ~b.w[q];
    return r;
  }
  WV_DEV static void setOrIn(Set& a, const Set& b)
  {
    for (unsigned q = 0; q < SW; ++q) a.w[q] |= b.w[q];
  }
  WV_DEV static Set setZero()
  {
    Set r;
    for (unsigned q = 0; q < SW; ++q) r.w[q] = 0;
    return r;
  }
  WV_DEV static Set setSel(const bool c, const Set& a, const Set& b)
  {
    Set r;
    for (unsigned q = 0; q < SW; ++q) r.w[q] = c ? a.w[q] : b.w[q];
    return r;
  }

  // ------------------------------------------------------------------------------------------------
  // the slab -> LDS
  // ------------------------------------------------------------------------------------------------
  WV_DEV bool load(const unsigned locus)
  {
    slab            = G.arena + G.slab_off[locus];
    const LgHdr* gh = reinterpret_cast<const LgHdr*>(slab);
    nNodes          = wv::first(gh->nNodes);
    nFat            = wv::first(gh->nFat);
    k               = wv::first(gh->k);
    nNormal         = wv::first(gh->nNormal);
    nEligible       = wv::first(gh->nEligible);
    nSpec           = wv::first(gh->nSpec);
    nSib            = wv::first(gh->nSib);
    nSovf           = wv::first(gh->nSovf);
    nPovf           = wv::first(gh->nPovf);
    codeWords       = wv::first(gh->codeWords);
    W               = wv::first(gh->W);
    acyclic         = wv::first(gh->acyclic) != 0;
    cyclic          = C::BIG && wv::first(gh->cyclic) != 0;
    nPseudo         = C::BIG ? wv::first(gh->nPseudo) : 0u;
    nCore           = cyclic ? wv::first(gh->nCore) : 0u;
    if (wv::first(gh->need) > P.lds_bytes || nNodes > C::MAX_NODES) return false;
    SL               = lgSlabOf<C>(nNodes, nFat, codeWords);
    const FRec8* gRec = reinterpret_cast<const FRec8*>(slab + SL.recs);
    for (unsigned i = lane; i < nNodes; i += 64) nodes[i] = gRec[i];
    if (C::BIG) {
      rd1                 = reinterpret_cast<uint8_t*>(lds + M::RECS + ((8 * nNodes + 15) & ~15u));
      const uint32_t* g4  = reinterpret_cast<const uint32_t*>(slab + SL.rd1);
      uint32_t*       l4  = reinterpret_cast<uint32_t*>(rd1);
      for (unsigned i = lane; i < (nNodes + 3) / 4; i += 64) l4[i] = g4[i];
    }
    // the three side tables lie back to back, in the slab and here
    const uint16_t* gt = reinterpret_cast<const uint16_t*>(slab + SL.sib);
    for (unsigned i = lane; i < 4 * (LG_SIB_CAP + 2 * C::OVF_CAP); i += 64) sib[i] = gt[i];
    // seed eligibility (:679-682): ids in count order, so the seeds are the ids below nEligible
    for (unsigned u = 0; u < UPL; ++u) {
      const unsigned d = UPL * lane + u, lo = 32 * d;
      unused_bits[d]   = (nEligible >= lo + 32) ? 0xffffffffu : ((nEligible > lo) ? ((1u << (nEligible - lo)) - 1u) : 0u);
    }
    if (cyclic) {
      // behind the bitset pool: core bitmap, its prefix counts (a core word's index = words of the core below it), repeat words, and
      // the walks' visited bitmaps.  A core word carries bit 63 of its record: every step sees it without a further read.
      char* x    = scratch() + 8 * SW * (nFat ? nFat : 1u);
      coreBits   = reinterpret_cast<uint32_t*>(x);
      corePrefix = reinterpret_cast<uint16_t*>(x + 4 * C::UNUSED_DW);
      repBits    = reinterpret_cast<uint32_t*>(x + 6 * C::UNUSED_DW);
      vis        = reinterpret_cast<uint32_t*>(x + 10 * C::UNUSED_DW);
      visDw      = (nCore + 31) / 32;
      const uint32_t* gf = reinterpret_cast<const uint32_t*>(slab + SL.flags);
      unsigned        c  = 0;
      uint32_t        cb[UPL];
      for (unsigned u = 0; u < UPL; ++u) {
        const unsigned d = UPL * lane + u;
        cb[u]            = gf[d];
        coreBits[d]      = cb[u];
        repBits[d]       = gf[C::UNUSED_DW + d];
        c += unsigned(wv::popc(cb[u]));
      }
      unsigned inc = c;
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = wv::shfl(inc, int(lane) - off);
        if (int(lane) >= off) inc += o;
      }
      unsigned at = inc - c;
      for (unsigned u = 0; u < UPL; ++u) {
        corePrefix[UPL * lane + u] = uint16_t(at);
        at += unsigned(wv::popc(cb[u]));
      }
      wv::sync();  // (the records are in)
      for (unsigned u = 0; u < UPL; ++u) {
        uint32_t b32 = cb[u];
        while (b32) {
          const unsigned b = unsigned(wv::ctz(uint64_t(b32)));
          b32 &= b32 - 1;
          nodes[32 * (UPL * lane + u) + b] |= FRec8(1) << 63;
        }
      }
    }
    wv::sync();
    return true;
  }
  /// a word the two-sided peel left (cyclic graphs of the big class only: bit 63 of the LDS copy of its record)
  WV_DEV static bool coreFlag(const FRec8 w) { return C::BIG && (w >> 63) != 0; }
  WV_DEV unsigned    coreIndex(const unsigned nd) const
  {
    return unsigned(corePrefix[nd >> 5]) + unsigned(wv::popc(coreBits[nd >> 5] & ((1u << (nd & 31)) - 1u)));
  }
  WV_DEV const Set*      gPool() const { return reinterpret_cast<const Set*>(slab + SL.pool); }
  WV_DEV const uint16_t* gSpecList() const { return reinterpret_cast<const uint16_t*>(slab + SL.spec); }
  WV_DEV const uint16_t* gPb() const { return reinterpret_cast<const uint16_t*>(slab + SL.pb); }
  WV_DEV const uint32_t* gCodes() const { return reinterpret_cast<const uint32_t*>(slab + SL.codes); }

  WV_DEV void loadPool()
  {
    pool              = reinterpret_cast<Set*>(scratch());
    const Set* gp     = gPool();
    for (unsigned i = lane; i < nFat; i += 64) pool[i] = gp[i];
    if (nFat == 0 && lane == 0) pool[0] = setZero();  // (a word without a bitset reads entry 0 and masks it out)
    wv::sync();
  }

  // ------------------------------------------------------------------------------------------------
  // cycle test: two-sided Kahn peel, per-node state byte {in:3, out:3, peeled}, one append-only queue.  True: cyclic.
  // ------------------------------------------------------------------------------------------------
  WV_DEV bool graphHasCycle()
  {
    const unsigned stDw = (nNodes + 3) / 4;
    uint32_t*      st    = reinterpret_cast<uint32_t*>(scratch());
    uint32_t*      qTail = st + stDw;
    uint16_t*      queue = reinterpret_cast<uint16_t*>(qTail + 4);
    if (lane == 0) *qTail = 0;
    for (unsigned w = lane; w < stDw; w += 64) st[w] = 0;
    wv::sync();
    for (unsigned nb = 0; nb < nNodes; nb += 64) {
      const unsigned nd = nb + lane;
      if (nd >= nNodes) continue;
      const FRec8    w  = nodes[nd];
      const uint64_t sl = succOf(nd, w), pl = predOf(nd, w);
      unsigned       id = 0, od = 0;
      for (unsigned c = 0; c < 4; ++c) {
        const unsigned s = R::linkId(sl, c), p = R::linkId(pl, c);
        if (s != ASM_NONE && s != nd) od++;
        if (p != ASM_NONE && p != nd) id++;
      }
      const bool     src = (id == 0 || od == 0);
      const unsigned v   = id | (od << 3) | (src ? 0x40u : 0u);
      wv::atomic_or(&st[nd >> 2], v << (8 * (nd & 3)));
      if (src) queue[wv::atomic_add(qTail, 1u)] = uint16_t(nd);
    }
    wv::sync();
    unsigned head = 0, removed = 0;
    while (true) {
      const unsigned tail = wv::first(wv::atomic_load(qTail));
      if (tail == head) break;
      removed += tail - head;
      for (unsigned i = head + lane; i < tail; i += 64) {
        const unsigned nd = queue[i];
        const FRec8    w  = nodes[nd];
        const uint64_t sl = succOf(nd, w), pl = predOf(nd, w);
        for (unsigned c = 0; c < 4; ++c) {
          const unsigned s = R::linkId(sl, c);
          if (s != ASM_NONE && s != nd) {
            const unsigned sh  = 8 * (s & 3);
            const unsigned old = wv::atomic_sub(&st[s >> 2], 1u << sh) >> sh;
            if ((old & 0x7u) == 1u && !(wv::atomic_or(&st[s >> 2], 0x40u << sh) & (0x40u << sh))) queue[wv::atomic_add(qTail, 1u)] = uint16_t(s);
          }
          const unsigned p = R::linkId(pl, c);
          if (p != ASM_NONE && p != nd) {
            const unsigned sh  = 8 * (p & 3);
            const unsigned old = wv::atomic_sub(&st[p >> 2], 8u << sh) >> sh;
            if ((old & 0x38u) == 8u && !(wv::atomic_or(&st[p >> 2], 0x40u << sh) & (0x40u << sh))) queue[wv::atomic_add(qTail, 1u)] = uint16_t(p);
          }
        }
      }
      wv::sync();
      head = tail;
    }
    return removed != nNodes;
  }

  // ------------------------------------------------------------------------------------------------
  // the next <= T unused words in seed order (:686-696) into tent[]: the lowest set bits of the bitmap
  // ------------------------------------------------------------------------------------------------
  WV_DEV unsigned firstUnused(const unsigned T)
  {
    uint32_t bits[UPL];
    unsigned c = 0;
    for (unsigned u = 0; u < UPL; ++u) {
      bits[u] = unused_bits[UPL * lane + u];
      c += unsigned(wv::popc(bits[u]));
    }
    unsigned inc = c;
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned o = wv::shfl(inc, int(lane) - off);
      if (int(lane) >= off) inc += o;
    }
    const unsigned total = wv::readlane(inc, 63);
    unsigned       at    = inc - c;
    for (unsigned u = 0; u < UPL; ++u) {
      uint32_t b32 = bits[u];
      while (b32 && at < T) {
        const unsigned b = unsigned(wv::ctz(uint64_t(b32)));
        tent[at++]       = uint16_t(32 * (UPL * lane + u) + b);
        b32 &= b32 - 1;
      }
    }
    wv::sync();
    return (total < T) ? total : T;
  }

  /// the lowest unused word (ASM_NONE: none)
  WV_DEV unsigned lowestUnused() const
  {
    unsigned mine = ASM_NONE;
    for (unsigned u = UPL; u-- > 0;) {
      const uint32_t b32 = unused_bits[UPL * lane + u];
      if (b32) mine = 32 * (UPL * lane + u) + unsigned(wv::ctz(uint64_t(b32)));
    }
    const uint64_t m = wv::ballot(mine != ASM_NONE);
    return m ? wv::readlane(mine, wv::ctz(m)) : ASM_NONE;
  }

  /// Big class, every walk round after the first: the next unused words in seed order, ONE PER UNBRANCHED STRETCH -- the words of a
  /// stretch come out of the same walk, so only the lowest of them can be the next seed; with every word listed (the small class'
  /// list) a round's 64 walks cover a dozen stretches and a config-5 pile needs a third round for its last candidates.  256 unused
  /// words are looked at, each is followed to the end of its stretch (the label), later words under a label are dropped, the first 64
  /// kept ones are the list.  The replay stays exact because it only ever accepts THE lowest unused word (contigRounds): a dropped
  /// word that its stretch's walk did not consume simply ends the replay and heads the next list.
#ifdef MANTA_WAVE_EMU
  struct Dbg { std::set<unsigned> looked, listed, dropped, walked, evicted; };
  static Dbg& dbg() { static Dbg d; return d; }
#endif
  // Round 6 ("the walk rounds of a tandem pile's passes"; counted with MANTA_EMU_WHY on the emulator):
  //   PLAIN  the first PLAIN looked-at words are all kept, duplicates of a stretch or not.  Used when MANTA_STRETCH_LATE or fewer candidates
  //          are still wanted: the last seeds are the next few unused words, and a word dropped as the duplicate of a stretch whose first
  //          word went another way cost a walk round of its own (0.7 rounds per tandem pass).
  //   MANTA_STRETCH_EVICT  contigRoundsStretch: eviction by seed order when a round is short of slots (graphs without a proof only).
#ifndef MANTA_STRETCH_PLAIN
#define MANTA_STRETCH_PLAIN 24
#endif
#ifndef MANTA_STRETCH_LATE
#define MANTA_STRETCH_LATE 7
#endif
#ifndef MANTA_STRETCH_EVICT
#define MANTA_STRETCH_EVICT 1
#endif
  WV_DEV unsigned stretchSeedList(const unsigned PLAIN = 0)
  {
    static const unsigned WIN = LGL_STRETCH_WIN;  // entries per lane: 64 x WIN unused words are looked at
    const unsigned nW = firstUnused(64 * WIN);
    if (nW <= 1) return nW;
    unsigned ent[WIN], lab[WIN];
    for (unsigned h = 0; h < WIN; ++h) {
      const unsigned i = lane + 64 * h;
      ent[h]           = (i < nW) ? unsigned(tent[i]) : ASM_NONE;
      unsigned cur     = (i < nW) ? ent[h] : 0u;
      for (unsigned steps = 0; steps < 256 && i < nW; ++steps) {
        const FRec8    w  = nodes[cur];
        const unsigned f0 = R::succ(w, 0);
        if (f0 == 0 || R::succ(w, 1) != 0 || R::sOvf(w) || f0 - 1 == cur) break;  // no successor, a branch, a self loop
        const FRec8 wn = nodes[f0 - 1];
        if (R::pred(wn, 1) != 0 || R::pOvf(wn)) break;  // the successor has another way in
        cur = f0 - 1;
      }
      lab[h] = (i < nW) ? cur : (ASM_NONE - 1u - i);  // (entries past the end: labels of their own)
    }
    bool dup[WIN];
    for (unsigned h = 0; h < WIN; ++h) dup[h] = false;
    for (unsigned g = 0; g < WIN; ++g) {
      const unsigned gEnd = (nW > 64 * g) ? ((nW - 64 * g < 64) ? (nW - 64 * g) : 64u) : 0u;
      for (unsigned j = 0; j < gEnd; ++j) {
        const unsigned v = wv::readlane(lab[g], int(j));
        for (unsigned h = g; h < WIN; ++h)
          if (64 * g + j < lane + 64 * h && v == lab[h] && lane + 64 * h >= PLAIN) dup[h] = true;
      }
    }
    unsigned base = 0;
    uint64_t keepM[WIN];
    for (unsigned h = 0; h < WIN; ++h) keepM[h] = wv::ballot(ent[h] != ASM_NONE && !dup[h]);
#ifdef MANTA_WAVE_EMU
    if (std::getenv("MANTA_EMU_WHY")) {
      for (unsigned h = 0; h < WIN; ++h)
        if (ent[h] != ASM_NONE) {
          dbg().looked.insert(ent[h]);
          if (dup[h]) dbg().dropped.insert(ent[h]);
        }
    }
#endif
    const uint64_t below = (uint64_t(1) << lane) - 1;
    wv::sync();
    for (unsigned h = 0; h < WIN; ++h) {
      if ((keepM[h] >> lane) & 1u) tent[base + unsigned(wv::popc(keepM[h] & below))] = uint16_t(ent[h]);
      base += unsigned(wv::popc(keepM[h]));
    }
    wv::sync();
    return (base < 64) ? base : 64u;
  }

  // ------------------------------------------------------------------------------------------------
  // walks (:149-501), one lane per cache slot
  // ------------------------------------------------------------------------------------------------
  struct Cand {
    FRec8 w;
    Set   s;
  };

  /// Fetching the word behind a link field `f` (id + 1; 0 = no word) takes two LDS reads: its record and -- if the word has
  /// one (id < nFat) -- its bitset, both addressed by the id, so they go out together.  The walk issues the reads of everything a
  /// step needs first, then combines with mask arithmetic (no selects on loaded values: the compiler would turn those into
  /// branches around the loads and serialise the round trips).  A word without a bitset reads pool entry 0 and masks it out.
  /// (Big class: the only read of a single-read word is a third read by the id, from rd1.)
  WV_DEV FRec8    candRec(const unsigned f) const { return nodes[f ? f - 1 : 0]; }
  WV_DEV Set      candPool(const unsigned f) const { return pool[(f != 0 && f - 1 < nFat) ? f - 1 : 0u]; }
  WV_DEV unsigned candRd1(const unsigned f) const { return C::BIG ? unsigned(rd1[f ? f - 1 : 0]) : 0u; }
  WV_DEV Set      candSup(const unsigned f, const FRec8 w, const Set& p, const unsigned r1) const
  {
    const unsigned ref  = C::BIG ? r1 : lg8Read(w);
    const bool     fat  = f != 0 && f - 1 < nFat;
    const uint64_t useM = fat ? ~uint64_t(0) : 0;
    const uint64_t bit  = uint64_t((f != 0 && !fat) ? 1u : 0u) << (ref & 63);
    Set            s;
    if (SW == 2) {
      const uint64_t hiM = (ref & 64u) ? ~uint64_t(0) : 0;
      s.w[0]             = (p.w[0] & useM) | (bit & ~hiM);
      s.w[1]             = (p.w[1] & useM) | (bit & hiM);
    } else {
      for (unsigned q = 0; q < SW; ++q) s.w[q] = (p.w[q] & useM) | (((ref >> 6) == q) ? bit : 0);
    }
    return s;
  }
  WV_DEV Cand loadCand(const unsigned f) const
  {
    Cand           c;
    const Set      p  = candPool(f);
    const unsigned r1 = candRd1(f);
    c.w               = candRec(f);
    c.s               = candSup(f, c.w, p, r1);
    return c;
  }

  /// the lanes of walkMask walk slotNode[lane]; results go to the slot's records in the workspace (lane_bits / lane_meta /
  /// lane_seq) and the walk's words to its log (lane_log: entry 0 = the seed, then the chosen words in order).
  ///
  /// One lane executes the instruction stream of all 64, so the step is written for the union: the first two candidates
  /// of a step (the record's own two links) are always fetched and compared branch-free, a third or fourth one
  /// (three-way branches are rare: overflow table) sits behind a wave vote; likewise the backward check (:377-427) fetches one
  /// "other" neighbour of the chosen word branch-free and further ones behind a vote.  Appended bases come from the chosen
  /// word's record (first / last base), not from the link position.
  WV_DEV void walkSlots(const uint64_t walkMask)
  {
    const bool has = (walkMask >> lane) & 1u;
    LG_STAT(1, 1);
    LG_STAT(2, unsigned(wv::popc(walkMask)));
    const unsigned seed = has ? unsigned(slotNode[lane]) : 0u;
    // The walk buffers are LANE-INTERLEAVED (entry i of lane l at [i * 64 + l]): the lanes of a round advance in step, so the stores of
    // a step -- one qword of the log every fourth extension, one dword of bases every sixteenth -- fall into one run of consecutive
    // addresses instead of 64 sectors a buffer's length apart.
    uint64_t* const logBuf   = lane_log + lane;              // [q * 64]
    uint64_t        logAcc   = seed;
    uint32_t* const rightBuf = lane_seq + lane;              // [i * 128]
    uint32_t* const leftBuf  = lane_seq + 64 + lane;         // [i * 128]
    uint32_t       accR = 0, accL = 0;
    Set            S = setZero(), RJ = setZero();  // contig support (:213) / reject (:214) reads
    bool           active = has, rep = false, tooLong = false;
    unsigned       mode = 0, cur = seed, consOffset = 0, nLeft = 0, nRight = 0;
    int            consEnd = 0, consBegin = 0;
    FRec8          seedRec = 0;
    if (cyclic && has)
      for (unsigned w = 0; w < visDw; ++w) vis[w * 64 + lane] = 0;  // wordsInContig (:182): only words of the core can come back
    if (has) {
      seedRec = nodes[seed];
      S       = supOf(seed, seedRec);
      const bool repeatSeed = R::selfLoop(seedRec) || (cyclic && ((repBits[seed >> 5] >> (seed & 31)) & 1u));
      if (repeatSeed) {  // :172-179 (repeatWords of an acyclic graph = the self loops; of a cyclic one: repeat_big_kernel's bitmap)
        rep    = true;
        active = false;
      } else {
        if (coreFlag(seedRec)) {
          const unsigned ci = coreIndex(seed);
          vis[(ci >> 5) * 64 + lane] |= 1u << (ci & 31);
        }
        // unselected siblings of the seed reject the contig (:185-210).  The words that differ from the seed in the last base
        // only are the other successors of any predecessor of the seed; a seed without a predecessor has them in the side table.
        const unsigned pf = R::pred(seedRec, 0);
        if (pf) {
          const uint64_t zs = succOf(pf - 1, nodes[pf - 1]);
          for (unsigned c = 0; c < 4; ++c) {
            const unsigned f = R::linkField(zs, c);
            if (f && f - 1 != seed) setOrIn(RJ, supOf(f - 1, nodes[f - 1]));
          }
        } else {
          for (unsigned e = 0; e < nSib; ++e) {
            if (unsigned(sib[4 * e]) != seed) continue;
            for (unsigned q = 1; q < 4; ++q) {
              const unsigned n = sib[4 * e + q];
              if (n == LG_NO_SLOT) continue;
              setOrIn(RJ, supOf(n, nodes[n]));
            }
          }
        }
      }
    }
    FRec8 curRec = seedRec;  // record of `cur`

    while (wv::any(active)) {
      // ---- fast steps.  Exactly one word ahead of the current one, and that word has exactly one word behind it (the current
      // one): nothing to choose (:241-336 see one candidate), nobody to reject (:377-427 find no other neighbour).  If the word
      // shares a read with the contig the general step below would do exactly this: take it, add its reads that do not reject
      // the contig.  Four of five steps of a walk are of this kind, and they cost a tenth of the general step -- so every lane
      // runs ahead through its unbranched stretch (up to CK_FAST_CAP words) before the wave takes one general step together.
      for (unsigned it = 0; it < CK_FAST_CAP; ++it) {
        const bool     fwd = (mode == 0);
        const unsigned f   = fwd ? R::succ(curRec, 0) : R::pred(curRec, 0);
        const bool     one = active && f != 0 && (fwd ? R::succ(curRec, 1) : R::pred(curRec, 1)) == 0 && !(fwd ? R::sOvf(curRec) : R::pOvf(curRec));
        const unsigned ff = one ? f : 0u;
        const FRec8    w  = candRec(ff);
        const Set      p  = candPool(ff);
        const unsigned r1 = candRd1(ff);
        const Set      a  = candSup(ff, w, p, r1);
        const bool     backOne = (fwd ? R::pred(w, 1) : R::succ(w, 1)) == 0 && !(fwd ? R::pOvf(w) : R::sOvf(w));
        const unsigned shared  = popSet(setAnd(S, a));
        const unsigned wc      = R::cnt(w);
        const bool     go = one && backOne && shared != 0 && wc >= P.opt.minCoverage && f - 1 != cur && !coreFlag(w) &&
                        (k + nRight + nLeft + 1 < maxLen) && (nRight + nLeft < CK_MAX_EXT);
        if (!wv::any(go)) break;
        if (go) {
          const unsigned pz = 1 + nRight + nLeft;
          logAcc |= uint64_t(f - 1) << (16 * (pz & 3));
          if ((pz & 3) == 3) {
            logBuf[size_t(pz >> 2) * 64] = logAcc;
            logAcc          = 0;
          }
          const unsigned sym = fwd ? R::lastBase(w) : R::firstBase(w);
          if (fwd) {
            accR |= sym << (2 * (nRight & 15));
            if ((nRight & 15) == 15) {
              rightBuf[size_t(nRight >> 4) * 128] = accR;
              accR                  = 0;
            }
            nRight++;
          } else {
            accL |= sym << (2 * (nLeft & 15));
            if ((nLeft & 15) == 15) {
              leftBuf[size_t(nLeft >> 4) * 128] = accL;
              accL                = 0;
            }
            nLeft++;
          }
          if ((consOffset != 0) || (wc < P.opt.minConservativeCoverage)) consOffset += 1;
          setOrIn(S, setAndNot(a, RJ));
          cur    = f - 1;
          curRec = w;
        }
      }
      // ---- the general step ----
      const bool     isEnd = (mode == 0);
      const uint64_t link  = active ? (isEnd ? succOf(cur, curRec) : predOf(cur, curRec)) : 0;  // candidates of the current word in walking direction
      const Cand     ca = loadCand(R::linkField(link, 0)), cb = loadCand(R::linkField(link, 1));
      // ---- choose the extension (:241-336): candidates a, b in alphabet order, strict '>' on the shared-read count ----
      const Set      A = setAnd(S, ca.s), B = setAnd(S, cb.s);
      const unsigned cntA = popSet(A), cntB = popSet(B);
      const bool     bWins = cntB > cntA;
      const Set      SH = setAnd(A, cb.s);
      // the loser's shared reads leave the contig, its reads reject it (an ignored candidate -- count 0 -- loses nothing)
      const bool     loserOn = bWins ? (cntA != 0) : (cntB != 0);
      Set            rm  = setAndNot(setSel(bWins, A, B), SH);
      Set            add = loserOn ? setAndNot(setSel(bWins, ca.s, cb.s), SH) : setZero();
      Set            maxWR = setSel(bWins, cb.s, ca.s);
      Set            maxCW = setSel(bWins, B, A);  // (empty when neither candidate shares a read)
      FRec8          maxW   = bWins ? cb.w : ca.w;
      unsigned       maxCnt = bWins ? cntB : cntA;
      unsigned       maxF   = bWins ? R::linkField(link, 1) : R::linkField(link, 0);  // id + 1 of the chosen word
      if (maxCnt == 0) maxF = 0;
      if (wv::any(active && (link >> (2 * IDB)) != 0)) {  // a third / fourth candidate somewhere in the wave (rare)
        if (maxCnt == 0) maxWR = setZero();  // (nothing chosen so far: nothing to lose to a later candidate)
        for (unsigned i = 2; i < 4; ++i) {
          const unsigned f = active ? R::linkField(link, i) : 0u;
          if (!wv::any(f != 0)) continue;
          const Cand     c   = loadCand(f);
          const Set      Cs  = setAnd(S, c.s);
          const unsigned cnt = popSet(Cs);
          if (cnt == 0) continue;  // :280
          const Set T = setAnd(maxCW, c.s);
          if (cnt > maxCnt) {  // :283-316
            setOrIn(rm, setAndNot(maxCW, T));
            setOrIn(add, setAndNot(maxWR, T));
            maxWR  = c.s;
            maxCW  = Cs;
            maxCnt = cnt;
            maxF   = f;
            maxW   = c.w;
          } else {  // :317-335
            setOrIn(rm, setAndNot(Cs, T));
            setOrIn(add, setAndNot(c.s, T));
          }
        }
      }
      const unsigned maxNode      = maxF - 1;  // (ASM_NONE when nothing was chosen)
      const unsigned maxBaseCount = maxF ? R::cnt(maxW) : 0u;
      // :352: the chosen word is one of the contig's own.  Only a word of the core can be (see repeat_big_kernel).
      const bool     onCore = active && maxF != 0 && coreFlag(maxW);
      unsigned       ci = 0;
      bool           seen = false;
      if (wv::any(onCore)) {
        ci   = onCore ? coreIndex(maxNode) : 0u;
        seen = onCore && ((vis[(ci >> 5) * 64 + lane] >> (ci & 31)) & 1u);
      }
      bool           stop = false, extend = false;
      if (active) {
        if (maxBaseCount < P.opt.minCoverage) {  // :343 (also "no candidate")
          stop = true;
        } else if (maxNode == cur || seen) {  // :352-358: in an acyclic graph a walk meets its own words again only through a self loop
          rep  = true;
          stop = true;
        } else if (k + nRight + nLeft + 1 >= maxLen || nRight + nLeft >= CK_MAX_EXT) {
          tooLong = true;
          active  = false;
        } else {
          extend = true;
        }
      }
      // ---- requests: the neighbours of the chosen word against the walking direction (:377-427) ... ----
      const uint64_t back = extend ? (isEnd ? predOf(maxNode, maxW) : succOf(maxNode, maxW)) : 0;
      unsigned       o0 = 0, nOther = 0;
      for (unsigned c = 0; c < 4; ++c) {
        const unsigned f  = R::linkField(back, c);
        const bool     ok = f != 0 && f != cur + 1 && f != maxF;  // :381, :389
        if (ok && nOther == 0) o0 = f;
        nOther += ok ? 1u : 0u;
      }
      const FRec8    ow  = candRec(o0);
      const Set      po  = candPool(o0);
      const unsigned or1 = candRd1(o0);
      Set            bs  = setAndNot(candSup(o0, ow, po, or1), maxCW);  // :400-414
      if (wv::any(nOther > 1)) {  // more than one other neighbour (rare)
        unsigned seen = 0;
        for (unsigned c = 0; c < 4; ++c) {
          const unsigned f  = R::linkField(back, c);
          const bool     ok = f != 0 && f != cur + 1 && f != maxF;
          const bool     want = ok && seen >= 1;
          seen += ok ? 1u : 0u;
          if (!wv::any(want)) continue;
          const Cand x = loadCand(want ? f : 0u);
          setOrIn(bs, setAndNot(x.s, maxCW));
        }
      }
      // ---- finish this step ----
      if (extend) {
        if (onCore) vis[(ci >> 5) * 64 + lane] |= 1u << (ci & 31);  // :485
        {  // :482-484: the word leaves unusedWords when this walk is accepted
          const unsigned p = 1 + nRight + nLeft;
          logAcc |= uint64_t(maxNode) << (16 * (p & 3));
          if ((p & 3) == 3) {
            logBuf[size_t(p >> 2) * 64] = logAcc;
            logAcc         = 0;
          }
        }
        const unsigned sym = isEnd ? R::lastBase(maxW) : R::firstBase(maxW);
        if (isEnd) {  // :363
          accR |= sym << (2 * (nRight & 15));
          if ((nRight & 15) == 15) {
            rightBuf[size_t(nRight >> 4) * 128] = accR;
            accR                  = 0;
          }
          nRight++;
        } else {
          accL |= sym << (2 * (nLeft & 15));
          if ((nLeft & 15) == 15) {
            leftBuf[size_t(nLeft >> 4) * 128] = accL;
            accL                = 0;
          }
          nLeft++;
        }
        if ((consOffset != 0) || (maxBaseCount < P.opt.minConservativeCoverage)) consOffset += 1;  // :368-369
        setOrIn(add, bs);
        setOrIn(rm, bs);
        setOrIn(RJ, add);                   // :440-442
        setOrIn(S, setAndNot(maxWR, RJ));   // :458-464
        S      = setAndNot(S, rm);          // :471-473
        cur    = maxNode;
        curRec = maxW;
      }
      if (stop) {
        if (mode == 0) {  // :488-491: on to the left, from the seed
          consEnd    = int(consOffset);
          mode       = 1;
          cur        = seed;
          curRec     = seedRec;
          consOffset = 0;
        } else {
          consBegin = int(consOffset);
          active    = false;
        }
      }
    }

    if (has) {
      uint64_t* lb = lane_bits + size_t(lane) * 2 * SW;
      for (unsigned q = 0; q < SW; ++q) {
        lb[q]      = S.w[q];
        lb[SW + q] = RJ.w[q];
      }
      if (nRight & 15) rightBuf[size_t(nRight >> 4) * 128] = accR;
      if (nLeft & 15) leftBuf[size_t(nLeft >> 4) * 128] = accL;
      logBuf[size_t((nRight + nLeft + 1) >> 2) * 64] = logAcc;  // (the last, partly filled qword; a full one was stored and this one is empty)
      int32_t* m = lane_meta + lane * 8;
      m[0]       = int(nLeft);
      m[1]       = int(nRight);
      m[2]       = consBegin;
      m[3]       = consEnd;
      m[4]       = (rep ? 1 : 0) | (tooLong ? 2 : 0);
    }
    wv::sync();
  }

  /// buildContigs' contig loop (:685-713): the reference's seed sequence replayed over cached speculative walks.
  /// Returns 0 = all contigs built (candidate c's walk sits in cache slot candSlotV of lane c; anyRep: one of them hit a repeat,
  /// the reference goes on to the next word length), 1 = not for this path (contig too long).
  WV_DEV int contigRounds()
  {
    const unsigned capCand = 2 * P.opt.maxAssemblyCount;
    nCand                  = 0;
    if (nNodes == 0 || nEligible == 0) return 0;  // no word at this length (:522) / no seed: no contig, no repeat
    slotNode[lane] = uint16_t(LG_NO_SLOT);
    uint64_t cached = 0;  // cache slots in use
    uint64_t accAll = 0;  // slots that hold accepted candidates (never evicted)
    // ---- round 0: the first seed (id 0) and beside it graph_kernel's speculation list ----
    {
      const uint16_t* spec = gSpecList();
      const unsigned  n0   = (nSpec < 1) ? 1u : ((nSpec > 64) ? 64u : nSpec);
      if (lane < n0) slotNode[lane] = (lane == 0) ? uint16_t(0) : spec[lane];
      cached = (n0 >= 64) ? ~uint64_t(0) : ((uint64_t(1) << n0) - 1);
      wv::sync();
      tick(5);
      walkSlots(cached);
      tick(6);
    }
    bool first = true;
    while (nCand < capCand) {
      unsigned nL = 1;
      if (first) {
        if (lane == 0) tent[0] = uint16_t(0);
        wv::sync();
      } else {
        nL = firstUnused(64);
        if (nL == 0) break;
      }
      first = false;
      // cache slot of every list entry (lane i: entry i)
      const unsigned node = (lane < nL) ? unsigned(tent[lane]) : ASM_NONE;
      unsigned       slot = LG_NO_SLOT;
      auto findSlots = [&]() {
        const unsigned mineNode = unsigned(slotNode[lane]);  // lane s: the word of slot s (LG_NO_SLOT: none)
        slot                    = LG_NO_SLOT;
        for (unsigned s = 0; s < 64; ++s) {
          const unsigned v = wv::readlane(mineNode, int(s));
          if (v == node) slot = s;
        }
      };
      findSlots();
      uint64_t miss = wv::ballot(lane < nL && slot == LG_NO_SLOT);
#ifdef MANTA_WAVE_EMU
      if (std::getenv("MANTA_EMU_ROUND_TRACE") && lane == 0) {
        std::fprintf(stderr, "  replay list (%u cands so far, nEligible %u, nFat %u): %u entries, missing mask %016llx; entries (id:count):", nCand, nEligible, nFat, nL, (unsigned long long)miss);
        for (unsigned i = 0; i < nL && i < 24; ++i) std::fprintf(stderr, " %u:%u%s", unsigned(tent[i]), R::cnt(nodes[tent[i]]), ((miss >> i) & 1u) ? "*" : "");
        std::fprintf(stderr, "\n");
      }
#endif
      tick(5);
      // Walk only when the very next seed has no walk yet.  Otherwise replay first: most rounds end there (enough candidates,
      // or the unwalked seeds turn out consumed -- the other words of a branch whose first word was walked), and a walk
      // round is the expensive thing.
      if (!(miss & 1u)) miss = 0;
      if (miss) {
        if (unsigned(wv::popc(~cached)) < unsigned(wv::popc(miss))) {
          // short of slots: take back those whose seed has been consumed since (such a walk can never be accepted)
          const unsigned sn   = unsigned(slotNode[lane]);
          const uint64_t dead = wv::ballot(((cached & ~accAll) >> lane) & 1u && sn != LG_NO_SLOT && !isUnused(sn));
          if (dead) {
            if ((dead >> lane) & 1u) slotNode[lane] = uint16_t(LG_NO_SLOT);
            cached &= ~dead;
            LG_STAT(5, unsigned(wv::popc(dead)));
            wv::sync();
          }
        }
        if (cached == ~uint64_t(0) && (miss & 1u)) {
          // still full and the very next seed has no walk: drop every cached walk that is not an accepted candidate
          if (!((accAll >> lane) & 1u)) slotNode[lane] = uint16_t(LG_NO_SLOT);
          cached = accAll;
          LG_STAT(4, 1);
          wv::sync();
          findSlots();
          miss = wv::ballot(lane < nL && slot == LG_NO_SLOT);
        }
        // the r-th missing entry takes the r-th free slot; entries past the free slots (and everything behind the first
        // of them) wait for the next round
        const uint64_t freeMask = ~cached;
        const unsigned nFree    = unsigned(wv::popc(freeMask));
        if ((freeMask >> lane) & 1u) tbl[wv::popc(freeMask & ((uint64_t(1) << lane) - 1))] = uint8_t(lane);
        wv::sync();
        const bool     isMiss = (miss >> lane) & 1u;
        const unsigned rnk    = unsigned(wv::popc(miss & ((uint64_t(1) << lane) - 1)));
        const uint64_t late   = wv::ballot(isMiss && rnk >= nFree);
        if (late) nL = unsigned(wv::ctz(late));
        uint64_t walkMask = 0;
        if (isMiss && rnk < nFree && lane < nL) {
          slot           = tbl[rnk];
          slotNode[slot] = uint16_t(node);
        }
        {
          // (bits of the slots just taken, gathered from the lanes that took them)
          unsigned mineSlot = (isMiss && rnk < nFree && lane < nL) ? slot : 64u;
          for (int off = 0; off < 64; ++off) {
            const unsigned v = wv::readlane(mineSlot, off);
            if (v < 64) walkMask |= uint64_t(1) << v;
          }
        }
        cached |= walkMask;
        wv::sync();
        if (nL == 0) return 1;  // (cannot happen: a missing first entry finds a free slot after the eviction)
        if (walkMask) {
          walkSlots(walkMask);
          tick(6);
        }
      }
      // replay: entry i is the reference's next seed iff it is still unused when its turn comes (every entry was unused when the
      // list was made; accepting a walk takes the walk's words out of unusedWords, :170,482)
      unsigned flI = 0, lenI = 0;
      if (lane < nL && slot != LG_NO_SLOT) {
        const int32_t* m = lane_meta + slot * 8;
        flI              = unsigned(m[4]);
        lenI             = 1u + unsigned(m[0]) + unsigned(m[1]);
      }
      uint64_t acc = 0;
      bool     bad = false;
      for (unsigned i = 0; i < nL && nCand < capCand; ++i) {
        const unsigned nd = wv::readlane(node, int(i));
        if (!isUnused(nd)) continue;  // consumed by an accepted walk: not a seed for the reference either
        const unsigned sl = wv::readlane(slot, int(i));
        if (sl == LG_NO_SLOT) break;  // the next seed has not been walked: next round
        const unsigned fl = wv::readlane(flI, int(i));
        if (fl & 2u) {  // contig too long for this path
          bad = true;
          break;
        }
        if (fl & 1u) anyRep = true;  // a repeat hit (:699): the contig stays a candidate, the reference moves on to the next word length afterwards
        acc |= uint64_t(1) << sl;
        if (lane == nCand) candSlotV = sl;
        nCand++;
        // unusedWords.erase for every word of the accepted walk
        const unsigned  n   = wv::readlane(lenI, int(i));
        const uint64_t* log = lane_log + sl;  // (interleaved: qword q of slot sl at [q * 64 + sl])
        for (unsigned j = lane; j < n; j += 64) {
          const unsigned w = unsigned(log[size_t(j >> 2) * 64] >> (16 * (j & 3))) & 0xffffu;
          wv::atomic_and(&unused_bits[w >> 5], ~(1u << (w & 31)));
        }
        wv::sync();
      }
      if (bad) return 1;
      accAll |= acc;
      tick(7);
    }
    return 0;
  }

  /// The big class' form of the loop above.  The replay is driven by the bitmap, not by a list: the reference's next seed is THE lowest
  /// unused word (:686-696 with ids in seed order); if a cache slot holds its walk the walk is accepted, otherwise a walk round fills the
  /// free slots from stretchSeedList -- whose first entry is that word.  Exact for any choice of what else gets walked.
  WV_DEV int contigRoundsStretch()
  {
    const unsigned capCand = 2 * P.opt.maxAssemblyCount;
    nCand                  = 0;
    if (nNodes == 0 || nEligible == 0) return 0;
    slotNode[lane] = uint16_t(LG_NO_SLOT);
    uint64_t cached = 0, accAll = 0;
#ifdef MANTA_WAVE_EMU
    if (lane == 0) dbg() = Dbg();
#endif
    {  // round 0: the first seed (id 0) and beside it graph_big_kernel's speculation list
      const uint16_t* spec = gSpecList();
      const unsigned  n0   = (nSpec < 1) ? 1u : ((nSpec > 64) ? 64u : nSpec);
      if (lane < n0) slotNode[lane] = (lane == 0) ? uint16_t(0) : spec[lane];
      cached = (n0 >= 64) ? ~uint64_t(0) : ((uint64_t(1) << n0) - 1);
      wv::sync();
      tick(5);
      walkSlots(cached);
      tick(6);
    }
    while (nCand < capCand) {
      const unsigned nd = lowestUnused();
      if (nd == ASM_NONE) break;
      const uint64_t hit = wv::ballot(((cached >> lane) & 1u) && unsigned(slotNode[lane]) == nd);
      if (hit) {
        const unsigned sl = unsigned(wv::ctz(hit));
        const int32_t* m  = lane_meta + sl * 8;
        const unsigned fl = unsigned(m[4]), n = 1u + unsigned(m[0]) + unsigned(m[1]);
        if (fl & 2u) return 1;       // contig too long for this path
        if (fl & 1u) anyRep = true;  // a repeat hit (:699): the contig stays a candidate, the reference moves on to the next word length afterwards
        accAll |= uint64_t(1) << sl;
        if (lane == nCand) candSlotV = sl;
        nCand++;
        const uint64_t* log = lane_log + sl;  // unusedWords.erase for every word of the accepted walk
        for (unsigned j = lane; j < n; j += 64) {
          const unsigned w = unsigned(log[size_t(j >> 2) * 64] >> (16 * (j & 3))) & 0xffffu;
          wv::atomic_and(&unused_bits[w >> 5], ~(1u << (w & 31)));
        }
        wv::sync();
        continue;
      }
      tick(7);
#ifdef MANTA_WAVE_EMU
      if (std::getenv("MANTA_EMU_WHY") && lane == 0) {
        const char* why = dbg().evicted.count(nd) ? "walked-then-evicted" : dbg().walked.count(nd) ? "walked(?)" : dbg().listed.count(nd) ? "listed-not-walked" : dbg().dropped.count(nd) ? "dropped-as-stretch-duplicate" : dbg().looked.count(nd) ? "looked" : "outside-every-window";
        std::fprintf(stderr, "WHY round at %u cands: seed %u count %u: %s\n", nCand, nd, R::cnt(nodes[nd]), why);
      }
#endif
      // ---- a walk round: the next seed and, beside it, the words most likely to follow it ----
      unsigned       nL   = stretchSeedList((capCand - nCand <= MANTA_STRETCH_LATE) ? unsigned(MANTA_STRETCH_PLAIN) : 0u);
#ifdef MANTA_WAVE_EMU
      if (std::getenv("MANTA_EMU_WHY") && lane < nL) dbg().listed.insert(unsigned(tent[lane]));
#endif
      const unsigned node = (lane < nL) ? unsigned(tent[lane]) : ASM_NONE;
      unsigned       slot = LG_NO_SLOT;
      auto findSlots = [&]() {
        const unsigned mineNode = ((cached >> lane) & 1u) ? unsigned(slotNode[lane]) : unsigned(LG_NO_SLOT);
        slot                    = LG_NO_SLOT;
        for (unsigned s = 0; s < 64; ++s) {
          const unsigned v = wv::readlane(mineNode, int(s));
          if (v == node && v != LG_NO_SLOT) slot = s;
        }
      };
      findSlots();
      uint64_t miss = wv::ballot(lane < nL && slot == LG_NO_SLOT);
#ifdef MANTA_WAVE_EMU
      if (std::getenv("MANTA_EMU_ROUND_TRACE") && lane == 0) {
        std::fprintf(stderr, "  walk round (%u cands so far): list of %u, missing %016llx; entries (id:count):", nCand, nL, (unsigned long long)miss);
        for (unsigned i = 0; i < nL && i < 24; ++i) std::fprintf(stderr, " %u:%u%s", unsigned(tent[i]), R::cnt(nodes[tent[i]]), ((miss >> i) & 1u) ? "*" : "");
        std::fprintf(stderr, "\n");
      }
#endif
      if (unsigned(wv::popc(~cached)) < unsigned(wv::popc(miss))) {
        // short of slots: take back those whose seed has been consumed since (such a walk can never be accepted)
        const unsigned sn   = unsigned(slotNode[lane]);
        const uint64_t dead = wv::ballot(((cached & ~accAll) >> lane) & 1u && sn != LG_NO_SLOT && !isUnused(sn));
        if (dead) {
          if ((dead >> lane) & 1u) slotNode[lane] = uint16_t(LG_NO_SLOT);
          cached &= ~dead;
          LG_STAT(5, unsigned(wv::popc(dead)));
          wv::sync();
        }
      }
#if MANTA_STRETCH_EVICT
      if (!acyclic && unsigned(wv::popc(~cached)) < unsigned(wv::popc(miss))) {
        // still short, on a graph without a proof of acyclicity (a pile with a tandem repeat: its first contigs end at the repeat and the next
        // seeds are well-covered words far up the order): the cached walks nobody asks for now, latest in seed order first, make room for
        // the list's entries -- an entry that waits for a slot costs a walk round of its own as soon as its turn comes.  (On the piles
        // without a repeat the round-0 speculation is what the late seeds need: evicting it cost them 7 % more rounds.)
        const unsigned sn     = unsigned(slotNode[lane]);
        bool           listed = false;
        for (unsigned j = 0; j < nL; ++j) {
          const unsigned v = wv::readlane(node, int(j));
          listed           = listed || (v == sn);
        }
        bool     evictable = (((cached & ~accAll) >> lane) & 1u) && sn != LG_NO_SLOT && !listed;
        unsigned need      = unsigned(wv::popc(miss)) - unsigned(wv::popc(~cached));
        while (need) {
          const unsigned key = evictable ? sn + 1u : 0u;
          unsigned       mx  = key;
          for (int off = 1; off < 64; off <<= 1) {
            const unsigned o = wv::shfl(mx, int(lane) ^ off);
            mx               = (o > mx) ? o : mx;
          }
          if (mx == 0) break;
          const uint64_t out = wv::ballot(evictable && key == mx);
          if ((out >> lane) & 1u) {
            slotNode[lane] = uint16_t(LG_NO_SLOT);
            evictable      = false;
          }
          cached &= ~out;
          need -= (unsigned(wv::popc(out)) < need) ? unsigned(wv::popc(out)) : need;
        }
        wv::sync();
      }
#endif
      if (cached == ~uint64_t(0)) {
        // still full (and the next seed has no walk): drop every cached walk that is not an accepted candidate
        // (making room for the first few missing entries only, at the cost of the cached walks latest in seed order, was tried: fewer
        // walks, the same number of rounds -- the single-read words' walks it gives up are the ones needed last)
#ifdef MANTA_WAVE_EMU
        if (std::getenv("MANTA_EMU_WHY") && !((accAll >> lane) & 1u) && slotNode[lane] != LG_NO_SLOT) dbg().evicted.insert(unsigned(slotNode[lane]));
#endif
        if (!((accAll >> lane) & 1u)) slotNode[lane] = uint16_t(LG_NO_SLOT);
        cached = accAll;
        LG_STAT(4, 1);
        wv::sync();
        findSlots();
        miss = wv::ballot(lane < nL && slot == LG_NO_SLOT);
      }
      // the r-th missing entry takes the r-th free slot; entries past the free slots wait for a later round
      const uint64_t freeMask = ~cached;
      const unsigned nFree    = unsigned(wv::popc(freeMask));
      if ((freeMask >> lane) & 1u) tbl[wv::popc(freeMask & ((uint64_t(1) << lane) - 1))] = uint8_t(lane);
      wv::sync();
      const bool     isMiss = (miss >> lane) & 1u;
      const unsigned rnk    = unsigned(wv::popc(miss & ((uint64_t(1) << lane) - 1)));
      const bool     take   = isMiss && rnk < nFree;
      if (take) {
        slot           = tbl[rnk];
        slotNode[slot] = uint16_t(node);
#ifdef MANTA_WAVE_EMU
        if (std::getenv("MANTA_EMU_WHY")) dbg().walked.insert(node);
#endif
      }
      uint64_t walkMask = 0;
      {
        const unsigned mineSlot = take ? slot : 64u;
        for (int off = 0; off < 64; ++off) {
          const unsigned v = wv::readlane(mineSlot, off);
          if (v < 64) walkMask |= uint64_t(1) << v;
        }
      }
      cached |= walkMask;
      wv::sync();
      if (!(miss & 1u) || nFree == 0 || walkMask == 0) return 1;  // (cannot happen: the list's first entry is the word without a walk, and the eviction frees a slot)
      tick(5);
      walkSlots(walkMask);
      tick(6);
    }
    return 0;
  }

  // ------------------------------------------------------------------------------------------------
  // selectContigs (:722-842) + output, lane c = candidate c (see Assembler::selectAndEmit for the general form)
  // ------------------------------------------------------------------------------------------------
  /// base i of candidate contig c's text (slot sl, nL left extensions, seed at packed base seedPb): reverse(left) + seed + right
  WV_DEV unsigned contigCode(const unsigned sl, const unsigned nL, const unsigned seedPb, const unsigned i) const
  {
    const uint32_t* gc       = gCodes();
    const uint32_t* rightBuf = lane_seq + sl;  // (interleaved: dword i of slot sl at [i * 128 + sl], the left half 64 further)
    const uint32_t* leftBuf  = lane_seq + 64 + sl;
    if (i < nL) {
      const unsigned j = nL - 1 - i;
      return (leftBuf[size_t(j >> 4) * 128] >> (2 * (j & 15))) & 3;
    }
    if (i < nL + k) {
      const unsigned pb = seedPb + (i - nL);
      return (gc[pb >> 4] >> (30 - 2 * (pb & 15))) & 3u;
    }
    const unsigned j = i - nL - k;
    return (rightBuf[size_t(j >> 4) * 128] >> (2 * (j & 15))) & 3;
  }

  /// :882-910: this word length's contigs longer than k + wordStepSize become the next length's pseudo reads.  They go to the pseudo
  /// arena as 2-bit codes in the pile's layout and are described in G.iter[locus] (nPseudo, len, off, codeWords).  False: arena full.
  WV_DEV bool writePseudo(const unsigned locus)
  {
    unsigned nL = 0, nR = 0, len = 0;
    if (lane < nCand) {
      const int32_t* m = lane_meta + candSlotV * 8;
      nL               = unsigned(m[0]);
      nR               = unsigned(m[1]);
      len              = nL + k + nR;
    }
    const bool     isP  = lane < nCand && len > k + P.opt.wordStepSize;  // :898
    const uint64_t mP   = wv::ballot(isP);
    const unsigned nP   = unsigned(wv::popc(mP));
    const unsigned myC  = isP ? (len + 15) / 16 + 1 : 0u;
    unsigned       inc  = myC;
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned o = wv::shfl(inc, int(lane) - off);
      if (int(lane) >= off) inc += o;
    }
    const unsigned total = wv::readlane(inc, 63);
    const unsigned myOff = inc - myC;
    if (nP > LGL_MAX_PSEUDO) return false;
    unsigned long long base = 0;
    if (lane == 0) base = wv::atomic_add(G.parena_used, (unsigned long long)total);
    base = wv::readlane(uint64_t(base), 0);
    if (base + total > G.parena_cap) return false;
    LgIter* st = &G.iter[locus];
    if (isP) st->len[wv::popc(mP & ((uint64_t(1) << lane) - 1))] = uint16_t(len);
    if (lane == 0) {
      st->nPseudo   = nP;
      st->off       = base;
      st->codeWords = total;
    }
    const uint16_t* gp = gPb();
    for (uint64_t m = mP; m; m &= m - 1) {
      const int      c      = wv::ctz(m);
      const unsigned sl     = wv::readlane(candSlotV, c);
      const unsigned cL     = wv::readlane(nL, c), cLen = wv::readlane(len, c), cOff = wv::readlane(myOff, c);
      const unsigned seedPb = gp[slotNode[sl]];
      const unsigned nCw    = (cLen + 15) / 16 + 1;
      for (unsigned wi = lane; wi < nCw; wi += 64) {
        uint32_t code = 0;
        for (unsigned b = 0; b < 16; ++b) {
          const unsigned i = wi * 16 + b;
          if (i < cLen) code |= contigCode(sl, cL, seedPb, i) << (30 - 2 * b);
        }
        G.parena[base + cOff + wi] = code;
      }
    }
    wv::sync();
    return true;
  }

  /// `st`: the pseudo reads the result carries (:898-905; nullptr: none), nIter / cycIters: word lengths tried / of them with a cyclic graph
  WV_DEV void selectAndEmit(const unsigned locus, const LgIter* st, const unsigned nIter, const unsigned cycIters)
  {
    Set      sup = setZero(), rej = setZero();
    unsigned nLeft = 0, nRight = 0, myLen = 0;
    int      consB = 0, consE = 0;
    if (lane < nCand) {
      const uint64_t* lb = lane_bits + size_t(candSlotV) * 2 * SW;
      for (unsigned q = 0; q < SW; ++q) {
        sup.w[q] = lb[q];
        rej.w[q] = lb[SW + q];
      }
      const int32_t* m   = lane_meta + candSlotV * 8;
      nLeft              = unsigned(m[0]);
      nRight             = unsigned(m[1]);
      consB              = m[2];
      consE              = m[3];
      myLen              = nLeft + k + nRight;
    }
    Set      used = setZero();  // wave-uniform
    bool     aliveL     = lane < nCand;
    unsigned finalCount = 0;
    uint64_t chosen     = 0;  // chosen candidates in order, 6 bits each (maxAssemblyCount <= 10 fits a qword; more: second word)
    uint64_t chosenHi   = 0;
    // index >= nNormal <=> pseudo read (oracle/manta_oracle.cpp selectContigs on stale indices)
    Set normalS;
    for (unsigned q = 0; q < SW; ++q) {
      const unsigned lo = 64 * q;
      normalS.w[q]      = (nNormal >= lo + 64) ? ~uint64_t(0) : ((nNormal > lo) ? ((uint64_t(1) << (nNormal - lo)) - 1) : uint64_t(0));
    }
    while (finalCount < P.opt.maxAssemblyCount) {
      if (!wv::any(aliveL)) break;
      const unsigned usedNormal = popSet(setAnd(used, normalS));
      if (nNormal - usedNormal < P.opt.minUnusedReads) break;  // :750
      const Set      fresh  = setAndNot(sup, used);
      const unsigned nFresh = popSet(fresh);
      if (aliveL && popSet(setAnd(fresh, normalS)) < P.opt.minSupportReads) aliveL = false;  // :779-788
      uint64_t key = aliveL ? ((uint64_t(nFresh) << 40) | (uint64_t(myLen) << 8) | uint64_t(63u - lane)) : 0;
      for (int off = 1; off < 64; off <<= 1) {
        const uint64_t o = wv::shfl(key, wv::lane() ^ off);
        key              = (o > key) ? o : key;
      }
      if ((key >> 40) == 0) break;  // :807
      const int selected = wv::first(int(63u - unsigned(key & 63u)));
      if (finalCount < 10)
        chosen |= uint64_t(selected) << (6 * finalCount);
      else
        chosenHi |= uint64_t(selected) << (6 * (finalCount - 10));
      if (int(lane) == selected) aliveL = false;
      for (unsigned q = 0; q < SW; ++q) used.w[q] |= wv::readlane(sup.w[q], selected);
      finalCount++;
    }
    auto chosenAt = [&](const unsigned f) { return unsigned(((f < 10) ? (chosen >> (6 * f)) : (chosenHi >> (6 * (f - 10)))) & 63u); };

    AsmLocusOut out;
    out.status            = ASM_OK;
    out.n_contigs         = finalCount;
    out.n_words           = W;
    const unsigned nP     = st ? wv::first(st->nPseudo) : 0u;
    out.n_pseudo          = nP;
    out.final_word_length = k;
    out.n_iterations      = nIter;
    out.cyclic_iterations = cycIters;
    out.reserved          = 0;
    uint64_t seqBytes = 0;
    for (unsigned f = 0; f < finalCount; ++f) seqBytes += wv::readlane(myLen, int(chosenAt(f)));
    const unsigned pLen = (lane < nP) ? unsigned(st->len[lane]) : 0u;  // (nP <= 40: lane p holds pseudo read p's length)
    {
      unsigned t = pLen;
      for (int off = 1; off < 64; off <<= 1) t += wv::shfl(t, wv::lane() ^ off);
      seqBytes += t;
    }
    const uint64_t     bitsWords = uint64_t(finalCount) * 2 * W + nP;
    unsigned long long seqBase = 0, bitsBase = 0;
    if (lane == 0) {
      seqBase  = wv::atomic_add(P.seq_used, (unsigned long long)seqBytes);
      bitsBase = wv::atomic_add(P.bits_used, (unsigned long long)bitsWords);
    }
    seqBase  = wv::readlane(uint64_t(seqBase), 0);
    bitsBase = wv::readlane(uint64_t(bitsBase), 0);
    if (seqBase + seqBytes > P.seq_cap || bitsBase + bitsWords > P.bits_cap) {
      out.status         = ASM_E_OUT_CAPACITY;
      out.n_contigs      = 0;
      out.pseudo_off     = 0;
      out.pseudo_len_off = 0;
      if (lane == 0) P.loci[locus] = out;
      return;
    }
    uint64_t        so = seqBase, bo = bitsBase;
    const uint16_t* gp = gPb();
    for (unsigned f = 0; f < finalCount; ++f) {
      const int      c   = int(chosenAt(f));
      const unsigned sl  = wv::readlane(candSlotV, c);
      const unsigned nL  = wv::readlane(nLeft, c), nR = wv::readlane(nRight, c), len = nL + k + nR;
      const unsigned seedPb = gp[slotNode[sl]];
      for (unsigned i = lane; i < len; i += 64) P.seq_arena[so + i] = uint8_t("ACGT"[contigCode(sl, nL, seedPb, i)]);
      {
        // lane h * W + w holds word w of the support (h = 0) / reject (h = 1) set
        const unsigned half = lane / W, w = lane % W;
        uint64_t       mine = 0;
        for (unsigned q = 0; q < SW; ++q) {
          const uint64_t sq = wv::readlane(sup.w[q], c), rq = wv::readlane(rej.w[q], c);
          if (w == q) mine = half ? rq : sq;
        }
        if (lane < 2 * W) P.bits_arena[bo + lane] = mine;
      }
      const int cb = wv::readlane(consB, c), ce = wv::readlane(consE, c);
      if (lane == 0) {
        AsmContigOut o;
        o.seq_off    = so;
        o.bits_off   = bo;
        o.seq_len    = len;
        o.cons_begin = cb;
        o.cons_end   = int(len) - ce;  // :498
        o.reserved   = 0;
        P.contigs[size_t(locus) * P.opt.maxAssemblyCount + f] = o;
      }
      so += len;
      bo += 2 * W;
    }
    out.pseudo_off     = so;
    out.pseudo_len_off = bo;
    if (nP) {  // the pseudo reads' text behind the contigs', their lengths behind the contigs' sets (as assemble_kernel's selectAndEmit)
      const uint32_t* pc = G.parena + st->off;
      unsigned        cw = 0;
      for (unsigned p = 0; p < nP; ++p) {
        const unsigned len = wv::readlane(pLen, int(p));
        for (unsigned i = lane; i < len; i += 64) P.seq_arena[so + i] = uint8_t("ACGT"[(pc[cw + (i >> 4)] >> (30 - 2 * (i & 15))) & 3u]);
        if (lane == 0) P.bits_arena[bo + p] = len;
        so += len;
        cw += (len + 15) / 16 + 1;
      }
    }
    if (lane == 0) P.loci[locus] = out;
  }

  /// CK_DONE: results emitted.  CK_PUNT: nothing emitted, the general path takes the locus.
  WV_DEV int run(const unsigned locus)
  {
    tMark = wv::clock();
#ifdef MANTA_WAVE_EMU
#define CK_TRACE(why) do { if (std::getenv("MANTA_EMU_PUNT_TRACE") && lane == 0) std::fprintf(stderr, "  contig_kernel punts locus %u: %s (%u words, %u with a set, k %u)\n", locus, why, nNodes, nFat, k); } while (0)
#else
#define CK_TRACE(why) do { } while (0)
#endif
    if (!load(locus)) {
      CK_TRACE("does not fit this launch's LDS");
      if (C::BIG && lane == 0 && G.stats) wv::atomic_add(&G.stats[7], 1u);
      return CK_PUNT;
    }
    tick(4);
    if (!acyclic && !cyclic && graphHasCycle()) {  // the exact repeat search is repeat_big_kernel's (big class, rounds on) or the general path's
      CK_TRACE("cyclic graph");
      return CK_PUNT;
    }
    tick(3);
    loadPool();
    tick(4);
    anyRep = false;
    if ((C::BIG ? contigRoundsStretch() : contigRounds()) != 0) {
      CK_TRACE("contig too long");
      if (C::BIG && lane == 0 && G.stats) wv::atomic_add(&G.stats[7], 1u);
      return CK_PUNT;
    }
    const bool rounds = C::BIG && G.iter != nullptr;
    if (anyRep && !rounds) {
      CK_TRACE("repeat hit");
      return CK_PUNT;
    }
    // runIterativeAssembler :856-910
    const unsigned round    = rounds ? G.round : 0u;
    const unsigned nIter    = round + 1;
    const unsigned cycIters = ((round > 0) ? wv::first(G.iter[locus].cyclicIters) : 0u) + (cyclic ? 1u : 0u);
    const LgIter*  st       = (round > 0) ? &G.iter[locus] : nullptr;  // the pseudo reads this graph was built with
    if (anyRep) {
      if (!writePseudo(locus)) {
        CK_TRACE("pseudo arena full");
        if (lane == 0 && G.stats) wv::atomic_add(&G.stats[8], 1u);
        return CK_PUNT;
      }
      st                   = &G.iter[locus];
      const unsigned maxWL = P.locus_max_wl ? P.locus_max_wl[locus] : P.opt.maxWordLength;
      if (k + P.opt.wordStepSize <= maxWL) {
        if (round >= G.last_round) {
          CK_TRACE("more word lengths than rounds");
          if (lane == 0 && G.stats) wv::atomic_add(&G.stats[9], 1u);
          return CK_PUNT;
        }
        if (lane == 0) {
          LgIter* w      = &G.iter[locus];
          w->k           = k + P.opt.wordStepSize;
          w->nIter       = nIter;
          w->cyclicIters = cycIters;
          G.next_ids[wv::atomic_add(G.next_count, 1u)] = locus;
        }
        LG_STAT(7, 1);
        return CK_NEXT;
      }
    }
    selectAndEmit(locus, st, nIter, cycIters);
    tick(7);
    LG_STAT(0, 1);
    LG_STAT(3, nCand);
    LG_STAT(6, acyclic ? 1u : 0u);
    return CK_DONE;
  }
};

/// persistent single-wave workgroups with P.lds_bytes of dynamic LDS; works through the loci of size class G.cls
/// (G.class_ids / G.class_count, filled by graph_kernel); P.counter is this launch's own work counter.
template <class C>
WV_DEV void contigKernelBody(const LgArgs& A)
{
  const AsmParams& P = A.P;
  const LgParams&  G = A.G;
  uint8_t*        ws    = G.cws + uint64_t(wv::block_single()) * G.cws_stride;
  char*           lds   = wv::lds_single();
  const unsigned  nLoci = wv::first(wv::atomic_load(&G.class_count[G.cls]));
  const uint32_t* ids   = G.class_ids + size_t(G.cls) * G.class_stride;
  while (true) {
    unsigned slot = 0;
    if (wv::lane() == 0) slot = wv::atomic_add(P.counter, 1u);
    slot = wv::first(slot);
    if (slot >= nLoci) break;
    const unsigned locus = ids[slot];
    LdsContig<C>   c(P, G, lds, ws);
    const int      rc = c.run(locus);
    wv::sync();
    if (rc == CK_PUNT && wv::lane() == 0) P.punt_ids[wv::atomic_add(P.punt_count, 1u)] = locus;
    wv::sync();
  }
}
#if !MANTA_TU_DEFINES(MANTA_TU_CONTIG)
WV_KERNEL_SINGLE WV_WAVES_PER_SIMD(2) void contig_kernel(const LgArgs A);
#else
WV_KERNEL_SINGLE WV_WAVES_PER_SIMD(2) void contig_kernel(const LgArgs A)
{
  contigKernelBody<LgS>(A);
}
#endif
/// the big class (asm_lds_big.hpp): read sets of four qwords, 13-bit ids; one or two loci per CU
#if !MANTA_TU_DEFINES(MANTA_TU_CONTIG)
WV_KERNEL_SINGLE WV_WAVES_PER_SIMD(1) void contig_big_kernel(const LgArgs A);
#else
WV_KERNEL_SINGLE WV_WAVES_PER_SIMD(1) void contig_big_kernel(const LgArgs A)
{
  contigKernelBody<LgL>(A);
}
#endif

}  // namespace manta_dev

#include "asm_lds_big.hpp"
#include "asm_repeat_big.hpp"
