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
WV_HD CkWsLayout ckWorkspaceLayout()
{
  CkWsLayout L;
  uint64_t   o = 0;
  L.lane_seq  = asmPut(o, 64ull * 2 * 4 * CK_SEQ_WORDS);
  L.lane_bits = asmPut(o, 64ull * 4 * 8);
  L.lane_meta = asmPut(o, 64ull * 8 * 4);
  L.lane_log  = asmPut(o, 64ull * 8 * CK_LOG_QW);
  L.total     = (o + 255) & ~uint64_t(255);
  return L;
}

enum { CK_DONE = 0, CK_PUNT = 1 };

struct LdsContig {
  const AsmParams& P;
  const LgParams&  G;
  char*            lds;
  const uint8_t*   slab;
  FRec8*           nodes;
  FSet*            pool;
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
  unsigned         candSlotV;  // lane c: cache slot that holds candidate c's walk
  uint64_t         tMark;

  WV_DEV LdsContig(const AsmParams& p, const LgParams& g, char* base, uint8_t* ws) : P(p), G(g), lds(base)
  {
    lane        = unsigned(wv::lane());
    unused_bits = reinterpret_cast<uint32_t*>(lds + CK_OFF_UNUSED);
    tent        = reinterpret_cast<uint16_t*>(lds + CK_OFF_TENT);
    slotNode    = reinterpret_cast<uint16_t*>(lds + CK_OFF_SLOTND);
    tbl         = reinterpret_cast<uint8_t*>(lds + CK_OFF_TBL);
    sib         = reinterpret_cast<uint16_t*>(lds + CK_OFF_SIB);
    sovf        = reinterpret_cast<uint16_t*>(lds + CK_OFF_SOVF);
    povf        = reinterpret_cast<uint16_t*>(lds + CK_OFF_POVF);
    nodes       = reinterpret_cast<FRec8*>(lds + CK_OFF_RECS);
    pool        = nullptr;
    const CkWsLayout L = ckWorkspaceLayout();
    lane_seq  = reinterpret_cast<uint32_t*>(ws + L.lane_seq);
    lane_bits = reinterpret_cast<uint64_t*>(ws + L.lane_bits);
    lane_meta = reinterpret_cast<int32_t*>(ws + L.lane_meta);
    lane_log  = reinterpret_cast<uint64_t*>(ws + L.lane_log);
    maxLen    = p.max_contig_len;
    seqWords  = CK_SEQ_WORDS;
    candSlotV = 0;
    nCand     = 0;
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
  WV_DEV char*    scratch() const { return lds + CK_OFF_RECS + ((8 * nNodes + 15) & ~15u); }

  /// read support of word `nd` (record w) as two set words
  WV_DEV void supOf(const unsigned nd, const FRec8 w, uint64_t& s0, uint64_t& s1) const
  {
    if (nd < nFat) {
      const FSet v = pool[nd];
      s0           = v.w[0];
      s1           = v.w[1];
    } else {
      const unsigned ref = lg8Read(w);
      s0 = (ref < 64) ? (uint64_t(1) << ref) : 0;
      s1 = (ref >= 64) ? (uint64_t(1) << (ref - 64)) : 0;
    }
  }
  /// successors / predecessors of word nd as 4 x 11 bits (id + 1)
  WV_DEV uint64_t succOf(const unsigned nd, const FRec8 w) const { return lg8Links(w, nd, true, sovf, nSovf); }
  WV_DEV uint64_t predOf(const unsigned nd, const FRec8 w) const { return lg8Links(w, nd, false, povf, nPovf); }

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
    if (wv::first(gh->need) > P.lds_bytes || nNodes > LG_MAX_NODES) return false;
    SL               = lgSlab(nNodes, nFat, codeWords);
    const FRec8* gRec = reinterpret_cast<const FRec8*>(slab + SL.recs);
    for (unsigned i = lane; i < nNodes; i += 64) nodes[i] = gRec[i];
    // the three side tables lie back to back, in the slab and here
    const uint16_t* gt = reinterpret_cast<const uint16_t*>(slab + SL.sib);
    for (unsigned i = lane; i < 4 * (LG_SIB_CAP + 2 * LG_OVF_CAP); i += 64) sib[i] = gt[i];
    // seed eligibility (:679-682): ids in count order, so the seeds are the ids below nEligible
    {
      const unsigned lo = 32 * lane;
      unused_bits[lane] = (nEligible >= lo + 32) ? 0xffffffffu : ((nEligible > lo) ? ((1u << (nEligible - lo)) - 1u) : 0u);
    }
    wv::sync();
    return true;
  }
  WV_DEV const FSet*     gPool() const { return reinterpret_cast<const FSet*>(slab + SL.pool); }
  WV_DEV const uint16_t* gSpecList() const { return reinterpret_cast<const uint16_t*>(slab + SL.spec); }
  WV_DEV const uint16_t* gPb() const { return reinterpret_cast<const uint16_t*>(slab + SL.pb); }
  WV_DEV const uint32_t* gCodes() const { return reinterpret_cast<const uint32_t*>(slab + SL.codes); }

  WV_DEV void loadPool()
  {
    pool              = reinterpret_cast<FSet*>(scratch());
    const FSet* gp    = gPool();
    for (unsigned i = lane; i < nFat; i += 64) pool[i] = gp[i];
    if (nFat == 0 && lane == 0) {  // (a word without a bitset reads entry 0 and masks it out)
      pool[0].w[0] = 0;
      pool[0].w[1] = 0;
    }
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
        const unsigned s = lgLinkId(sl, c), p = lgLinkId(pl, c);
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
          const unsigned s = lgLinkId(sl, c);
          if (s != ASM_NONE && s != nd) {
            const unsigned sh  = 8 * (s & 3);
            const unsigned old = wv::atomic_sub(&st[s >> 2], 1u << sh) >> sh;
            if ((old & 0x7u) == 1u && !(wv::atomic_or(&st[s >> 2], 0x40u << sh) & (0x40u << sh))) queue[wv::atomic_add(qTail, 1u)] = uint16_t(s);
          }
          const unsigned p = lgLinkId(pl, c);
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
    uint32_t       bits = unused_bits[lane];
    const unsigned c    = unsigned(wv::popc(bits));
    unsigned       inc  = c;
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned o = wv::shfl(inc, int(lane) - off);
      if (int(lane) >= off) inc += o;
    }
    const unsigned total = wv::readlane(inc, 63);
    unsigned       at    = inc - c;
    while (bits && at < T) {
      const unsigned b = unsigned(wv::ctz(uint64_t(bits)));
      tent[at++]       = uint16_t(32 * lane + b);
      bits &= bits - 1;
    }
    wv::sync();
    return (total < T) ? total : T;
  }

  // ------------------------------------------------------------------------------------------------
  // walks (:149-501), one lane per cache slot
  // ------------------------------------------------------------------------------------------------
  struct Cand {
    FRec8    w;
    uint64_t s0, s1;
  };

  /// Fetching the word behind a link field `f` (id + 1; 0 = no word) takes two LDS reads: its record and -- if the word has
  /// one (id < nFat) -- its bitset, both addressed by the id, so they go out together.  The walk issues the reads of everything a
  /// step needs first, then combines with mask arithmetic (no selects on loaded values: the compiler would turn those into
  /// branches around the loads and serialise the round trips).  A word without a bitset reads pool entry 0 and masks it out.
  WV_DEV FRec8 candRec(const unsigned f) const { return nodes[f ? f - 1 : 0]; }
  WV_DEV FSet  candPool(const unsigned f) const { return pool[(f != 0 && f - 1 < nFat) ? f - 1 : 0u]; }
  WV_DEV void  candSup(const unsigned f, const FRec8 w, const FSet& p, uint64_t& s0, uint64_t& s1) const
  {
    const unsigned ref  = lg8Read(w);
    const bool     fat  = f != 0 && f - 1 < nFat;
    const uint64_t useM = fat ? ~uint64_t(0) : 0;
    const uint64_t bit  = uint64_t((f != 0 && !fat) ? 1u : 0u) << (ref & 63);
    const uint64_t hiM  = (ref & 64u) ? ~uint64_t(0) : 0;
    s0                  = (p.w[0] & useM) | (bit & ~hiM);
    s1                  = (p.w[1] & useM) | (bit & hiM);
  }
  WV_DEV Cand loadCand(const unsigned f) const
  {
    Cand       c;
    const FSet p = candPool(f);
    c.w          = candRec(f);
    candSup(f, c.w, p, c.s0, c.s1);
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
    uint64_t       S0 = 0, S1 = 0, R0 = 0, R1 = 0;
    bool           active = has, rep = false, tooLong = false;
    unsigned       mode = 0, cur = seed, consOffset = 0, nLeft = 0, nRight = 0;
    int            consEnd = 0, consBegin = 0;
    FRec8          seedRec = 0;
    if (has) {
      seedRec = nodes[seed];
      supOf(seed, seedRec, S0, S1);
      if (lg8SelfLoop(seedRec)) {  // :172-179 (repeatWords of an acyclic graph = the self loops)
        rep    = true;
        active = false;
      } else {
        // unselected siblings of the seed reject the contig (:185-210).  The words that differ from the seed in the last base
        // only are the other successors of any predecessor of the seed; a seed without a predecessor has them in the side table.
        const unsigned pf = lg8Pred(seedRec, 0);
        if (pf) {
          const uint64_t zs = succOf(pf - 1, nodes[pf - 1]);
          for (unsigned c = 0; c < 4; ++c) {
            const unsigned f = unsigned(zs >> (11 * c)) & 0x7ffu;
            if (f && f - 1 != seed) {
              uint64_t a, b;
              supOf(f - 1, nodes[f - 1], a, b);
              R0 |= a;
              R1 |= b;
            }
          }
        } else {
          for (unsigned e = 0; e < nSib; ++e) {
            if (unsigned(sib[4 * e]) != seed) continue;
            for (unsigned q = 1; q < 4; ++q) {
              const unsigned n = sib[4 * e + q];
              if (n == LG_NO_SLOT) continue;
              uint64_t a, b;
              supOf(n, nodes[n], a, b);
              R0 |= a;
              R1 |= b;
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
        const unsigned f   = unsigned(fwd ? curRec : (curRec >> 22)) & 0x7ffu;
        const bool     one = active && f != 0 && (unsigned(fwd ? (curRec >> 11) : (curRec >> 33)) & 0x7ffu) == 0 &&
                         !(fwd ? lg8SOvf(curRec) : lg8POvf(curRec));
        const unsigned ff = one ? f : 0u;
        const FRec8    w  = candRec(ff);
        const FSet     p  = candPool(ff);
        uint64_t       a0, a1;
        candSup(ff, w, p, a0, a1);
        const bool     backOne = (unsigned(fwd ? (w >> 33) : (w >> 11)) & 0x7ffu) == 0 && !(fwd ? lg8POvf(w) : lg8SOvf(w));
        const unsigned shared  = unsigned(wv::popc(S0 & a0)) + unsigned(wv::popc(S1 & a1));
        const unsigned wc      = lg8Cnt(w);
        const bool     go = one && backOne && shared != 0 && wc >= P.opt.minCoverage && f - 1 != cur &&
                        (k + nRight + nLeft + 1 < maxLen) && (nRight + nLeft < CK_MAX_EXT);
        if (!wv::any(go)) break;
        if (go) {
          const unsigned pz = 1 + nRight + nLeft;
          logAcc |= uint64_t(f - 1) << (16 * (pz & 3));
          if ((pz & 3) == 3) {
            logBuf[size_t(pz >> 2) * 64] = logAcc;
            logAcc          = 0;
          }
          const unsigned sym = fwd ? lg8LastBase(w) : lg8FirstBase(w);
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
          S0 |= a0 & ~R0;
          S1 |= a1 & ~R1;
          cur    = f - 1;
          curRec = w;
        }
      }
      // ---- the general step ----
      const bool     isEnd = (mode == 0);
      const uint64_t link  = active ? (isEnd ? succOf(cur, curRec) : predOf(cur, curRec)) : 0;  // candidates of the current word in walking direction
      const Cand     ca = loadCand(unsigned(link) & 0x7ffu), cb = loadCand(unsigned(link >> 11) & 0x7ffu);
      // ---- choose the extension (:241-336): candidates a, b in alphabet order, strict '>' on the shared-read count ----
      const uint64_t A0 = S0 & ca.s0, A1 = S1 & ca.s1, B0 = S0 & cb.s0, B1 = S1 & cb.s1;
      const unsigned cntA = unsigned(wv::popc(A0)) + unsigned(wv::popc(A1)), cntB = unsigned(wv::popc(B0)) + unsigned(wv::popc(B1));
      const bool     bWins = cntB > cntA;
      const uint64_t SH0 = A0 & cb.s0, SH1 = A1 & cb.s1;
      // the loser's shared reads leave the contig, its reads reject it (an ignored candidate -- count 0 -- loses nothing)
      const bool     loserOn = bWins ? (cntA != 0) : (cntB != 0);
      uint64_t       rm0  = (bWins ? A0 : B0) & ~SH0, rm1 = (bWins ? A1 : B1) & ~SH1;
      uint64_t       add0 = loserOn ? ((bWins ? ca.s0 : cb.s0) & ~SH0) : 0, add1 = loserOn ? ((bWins ? ca.s1 : cb.s1) & ~SH1) : 0;
      uint64_t       maxWR0 = bWins ? cb.s0 : ca.s0, maxWR1 = bWins ? cb.s1 : ca.s1;
      uint64_t       maxCW0 = bWins ? B0 : A0, maxCW1 = bWins ? B1 : A1;  // (empty when neither candidate shares a read)
      FRec8          maxW   = bWins ? cb.w : ca.w;
      unsigned       maxCnt = bWins ? cntB : cntA;
      unsigned       maxF   = bWins ? (unsigned(link >> 11) & 0x7ffu) : (unsigned(link) & 0x7ffu);  // id + 1 of the chosen word
      if (maxCnt == 0) maxF = 0;
      if (wv::any(active && (link >> 22) != 0)) {  // a third / fourth candidate somewhere in the wave (rare)
        if (maxCnt == 0) maxWR0 = maxWR1 = 0;  // (nothing chosen so far: nothing to lose to a later candidate)
        for (unsigned i = 2; i < 4; ++i) {
          const unsigned f = active ? (unsigned(link >> (11 * i)) & 0x7ffu) : 0u;
          if (!wv::any(f != 0)) continue;
          const Cand c = loadCand(f);
          const uint64_t C0 = S0 & c.s0, C1 = S1 & c.s1;
          const unsigned cnt = unsigned(wv::popc(C0)) + unsigned(wv::popc(C1));
          if (cnt == 0) continue;  // :280
          const uint64_t T0 = maxCW0 & c.s0, T1 = maxCW1 & c.s1;
          if (cnt > maxCnt) {  // :283-316
            rm0 |= maxCW0 & ~T0;
            rm1 |= maxCW1 & ~T1;
            add0 |= maxWR0 & ~T0;
            add1 |= maxWR1 & ~T1;
            maxWR0 = c.s0;
            maxWR1 = c.s1;
            maxCW0 = C0;
            maxCW1 = C1;
            maxCnt = cnt;
            maxF   = f;
            maxW   = c.w;
          } else {  // :317-335
            rm0 |= C0 & ~T0;
            rm1 |= C1 & ~T1;
            add0 |= c.s0 & ~T0;
            add1 |= c.s1 & ~T1;
          }
        }
      }
      const unsigned maxNode      = maxF - 1;  // (ASM_NONE when nothing was chosen)
      const unsigned maxBaseCount = maxF ? lg8Cnt(maxW) : 0u;
      bool           stop = false, extend = false;
      if (active) {
        if (maxBaseCount < P.opt.minCoverage) {  // :343 (also "no candidate")
          stop = true;
        } else if (maxNode == cur) {  // :352-358: in an acyclic graph a walk meets its own words again only through a self loop
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
        const unsigned f  = unsigned(back >> (11 * c)) & 0x7ffu;
        const bool     ok = f != 0 && f != cur + 1 && f != maxF;  // :381, :389
        if (ok && nOther == 0) o0 = f;
        nOther += ok ? 1u : 0u;
      }
      const FRec8    ow = candRec(o0);
      const FSet     po = candPool(o0);
      uint64_t       b0, b1;
      candSup(o0, ow, po, b0, b1);
      b0 &= ~maxCW0;  // :400-414
      b1 &= ~maxCW1;
      if (wv::any(nOther > 1)) {  // more than one other neighbour (rare)
        unsigned seen = 0;
        for (unsigned c = 0; c < 4; ++c) {
          const unsigned f  = unsigned(back >> (11 * c)) & 0x7ffu;
          const bool     ok = f != 0 && f != cur + 1 && f != maxF;
          const bool     want = ok && seen >= 1;
          seen += ok ? 1u : 0u;
          if (!wv::any(want)) continue;
          const Cand x = loadCand(want ? f : 0u);
          b0 |= x.s0 & ~maxCW0;
          b1 |= x.s1 & ~maxCW1;
        }
      }
      // ---- finish this step ----
      if (extend) {
        {  // :482-484: the word leaves unusedWords when this walk is accepted
          const unsigned p = 1 + nRight + nLeft;
          logAcc |= uint64_t(maxNode) << (16 * (p & 3));
          if ((p & 3) == 3) {
            logBuf[size_t(p >> 2) * 64] = logAcc;
            logAcc         = 0;
          }
        }
        const unsigned sym = isEnd ? lg8LastBase(maxW) : lg8FirstBase(maxW);
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
        add0 |= b0;
        add1 |= b1;
        rm0 |= b0;
        rm1 |= b1;
        R0 |= add0;  // :440-442
        R1 |= add1;
        S0 |= maxWR0 & ~R0;  // :458-464
        S1 |= maxWR1 & ~R1;
        S0 &= ~rm0;  // :471-473
        S1 &= ~rm1;
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
      uint64_t* lb = lane_bits + size_t(lane) * 4;
      lb[0]        = S0;
      lb[1]        = S1;
      lb[2]        = R0;
      lb[3]        = R1;
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
  /// Returns 0 = all contigs built without a repeat hit (candidate c's walk sits in cache slot candSlotV of lane c),
  /// 1 = not for this path (a walk hit a repeat: the reference goes on to the next word length; contig too long).
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
        if (wv::readlane(flI, int(i)) != 0) {  // repeat hit (the reference moves on to the next word length) or contig too long
          bad = true;
          break;
        }
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

  // ------------------------------------------------------------------------------------------------
  // selectContigs (:722-842) + output, lane c = candidate c (see Assembler::selectAndEmit for the general form)
  // ------------------------------------------------------------------------------------------------
  WV_DEV void selectAndEmit(const unsigned locus)
  {
    uint64_t sup0 = 0, sup1 = 0, rej0 = 0, rej1 = 0;
    unsigned nLeft = 0, nRight = 0, myLen = 0;
    int      consB = 0, consE = 0;
    if (lane < nCand) {
      const uint64_t* lb = lane_bits + size_t(candSlotV) * 4;
      sup0               = lb[0];
      sup1               = lb[1];
      rej0               = lb[2];
      rej1               = lb[3];
      const int32_t* m   = lane_meta + candSlotV * 8;
      nLeft              = unsigned(m[0]);
      nRight             = unsigned(m[1]);
      consB              = m[2];
      consE              = m[3];
      myLen              = nLeft + k + nRight;
    }
    uint64_t used0 = 0, used1 = 0;  // wave-uniform
    bool     aliveL     = lane < nCand;
    unsigned finalCount = 0;
    uint64_t chosen     = 0;  // chosen candidates in order, 6 bits each (maxAssemblyCount <= 10 fits a qword; more: second word)
    uint64_t chosenHi   = 0;
    while (finalCount < P.opt.maxAssemblyCount) {
      if (!wv::any(aliveL)) break;
      const unsigned usedNormal = unsigned(wv::popc(used0)) + unsigned(wv::popc(used1));  // (no pseudo reads on this path)
      if (nNormal - usedNormal < P.opt.minUnusedReads) break;  // :750
      const unsigned nFresh = unsigned(wv::popc(sup0 & ~used0)) + unsigned(wv::popc(sup1 & ~used1));
      if (aliveL && nFresh < P.opt.minSupportReads) aliveL = false;  // :779-788
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
      used0 |= wv::readlane(sup0, selected);
      used1 |= wv::readlane(sup1, selected);
      finalCount++;
    }
    auto chosenAt = [&](const unsigned f) { return unsigned(((f < 10) ? (chosen >> (6 * f)) : (chosenHi >> (6 * (f - 10)))) & 63u); };

    AsmLocusOut out;
    out.status            = ASM_OK;
    out.n_contigs         = finalCount;
    out.n_words           = W;
    out.n_pseudo          = 0;
    out.final_word_length = k;
    out.n_iterations      = 1;
    out.cyclic_iterations = 0;
    out.reserved          = 0;
    uint64_t seqBytes = 0;
    for (unsigned f = 0; f < finalCount; ++f) seqBytes += wv::readlane(myLen, int(chosenAt(f)));
    const uint64_t     bitsWords = uint64_t(finalCount) * 2 * W;
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
    const uint32_t* gc = gCodes();
    const uint16_t* gp = gPb();
    for (unsigned f = 0; f < finalCount; ++f) {
      const int      c   = int(chosenAt(f));
      const unsigned sl  = wv::readlane(candSlotV, c);
      const unsigned nL  = wv::readlane(nLeft, c), nR = wv::readlane(nRight, c), len = nL + k + nR;
      const unsigned seedPb = gp[slotNode[sl]];
      const uint32_t* rightBuf = lane_seq + sl;       // (interleaved: dword i of slot sl at [i * 128 + sl], the left half 64 further)
      const uint32_t* leftBuf  = lane_seq + 64 + sl;
      for (unsigned i = lane; i < len; i += 64) {  // reverse(left) + seed + right
        unsigned code;
        if (i < nL) {
          const unsigned j = nL - 1 - i;
          code             = (leftBuf[size_t(j >> 4) * 128] >> (2 * (j & 15))) & 3;
        } else if (i < nL + k) {
          const unsigned pb = seedPb + (i - nL);
          code              = (gc[pb >> 4] >> (30 - 2 * (pb & 15))) & 3u;
        } else {
          const unsigned j = i - nL - k;
          code             = (rightBuf[size_t(j >> 4) * 128] >> (2 * (j & 15))) & 3;
        }
        P.seq_arena[so + i] = uint8_t("ACGT"[code]);
      }
      const uint64_t s0 = wv::readlane(sup0, c), s1 = wv::readlane(sup1, c), r0 = wv::readlane(rej0, c), r1 = wv::readlane(rej1, c);
      if (lane < 2 * W) {
        const unsigned half = lane / W, w = lane % W;
        P.bits_arena[bo + lane] = half ? (w ? r1 : r0) : (w ? s1 : s0);
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
    if (lane == 0) P.loci[locus] = out;
  }

  /// CK_DONE: results emitted.  CK_PUNT: nothing emitted, the general path takes the locus.
  WV_DEV int run(const unsigned locus)
  {
    tMark = wv::clock();
    if (!load(locus)) return CK_PUNT;
    tick(4);
    if (!acyclic && graphHasCycle()) return CK_PUNT;  // the exact repeat search is the general path's
    tick(3);
    loadPool();
    tick(4);
    if (contigRounds() != 0) return CK_PUNT;
    selectAndEmit(locus);
    tick(7);
    LG_STAT(0, 1);
    LG_STAT(3, nCand);
    LG_STAT(6, acyclic ? 1u : 0u);
    return CK_DONE;
  }
};

/// persistent single-wave workgroups with P.lds_bytes of dynamic LDS; works through the loci of size class G.cls
/// (G.class_ids / G.class_count, filled by graph_kernel); P.counter is this launch's own work counter.
WV_KERNEL_SINGLE WV_WAVES_PER_SIMD(2) void contig_kernel(const LgArgs A)
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
    LdsContig      c(P, G, lds, ws);
    const int      rc = c.run(locus);
    wv::sync();
    if (rc != CK_DONE && wv::lane() == 0) P.punt_ids[wv::atomic_add(P.punt_count, 1u)] = locus;
    wv::sync();
  }
}

}  // namespace manta_dev
