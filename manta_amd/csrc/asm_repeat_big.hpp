// repeat_big_kernel: between graph_big_kernel and contig_big_kernel, for the big-class loci whose graph came WITHOUT a proof of
// acyclicity (a handful per thousand -- but every pile with a tandem repeat, at most of its word lengths).  One wavefront per locus,
// the compact graph read from the locus' slab, a per-wave workspace in device memory, 8 KB of LDS:
//
//   1. two-sided Kahn peel (as contig_kernel's cycle test): acyclic after all -> the slab says so, done.  What the peel cannot remove
//      is the CORE: the only words a walk can meet twice (asm_contig.hpp keeps a visited bitmap over them).
//   2. cyclic: the reference's repeat search (IterativeAssembler.cpp:555-642) EXACTLY -- its visiting order is the iteration order of
//      a std::unordered_map, and which circles count as small (:612) depends on it.  Same construction as repeat_exact.hpp
//      (insertion sequence -> std::hash -> libstdc++ node order, twice -> Tarjan with whole unbranched runs per step), on the
//      compact graph: the words are renumbered by FIRST OCCURRENCE so that an unbranched stretch has consecutive numbers again
//      (the ids of the slab are in seed order), the lexicographic ranks inside a read's group come from graph_big_kernel.
//   3. core and repeat-word bitmaps into the slab, the locus into contig_big_kernel's LDS class list (the class depends on the core's size).
#pragma once
#include "asm_lds.hpp"

namespace manta_dev {

static const unsigned RPB_LDS_BYTES = 8192;  ///< per wave: peel state (a byte per word), then the first-occurrence bitmap (65 536 bits)

struct RpbWs {
  uint64_t h, ins, seqB, pool, vOf, idOf, succ4, flagV, firstRd, byLex, queue, prefix, rdBase, total;
};
WV_HD RpbWs rpbWorkspaceLayout()
{
  const uint64_t C = LGL_MAX_NODES;
  RpbWs          L;
  uint64_t       o = 0;
  L.h       = asmPut(o, 8 * C);
  L.ins     = asmPut(o, 4 * C);
  L.seqB    = asmPut(o, 4 * C);
  L.pool    = asmPut(o, 4 * (9 * C + 128));
  L.vOf     = asmPut(o, 4 * C);
  L.idOf    = asmPut(o, 4 * C);
  L.succ4   = asmPut(o, 16 * C);
  L.flagV   = asmPut(o, 4 * C);
  L.firstRd = asmPut(o, 4 * C);
  L.byLex   = asmPut(o, 4 * C);
  L.queue   = asmPut(o, 4 * (C + 64));
  L.prefix  = asmPut(o, 4 * 2048);
  L.rdBase  = asmPut(o, 4 * 320);
  L.total   = (o + 255) & ~uint64_t(255);
  return L;
}

enum { RPB_DONE = 0, RPB_PUNT = 1 };

struct RepeatBig {
  typedef LgRec<LgL>  R;
  typedef FSetT<LgL>  Set;
  const AsmParams& P;
  const LgParams&  G;
  uint8_t*         ws;
  char*            lds;
  unsigned         lane;
  uint8_t*         slab;
  LgSlab           SL;
  unsigned         n, nFat, k, nSovf, nPovf;
  const FRec8*     gRec;
  const uint16_t * gSovf, *gPovf, *gPb, *gLex;
  const uint8_t*   gRd1;
  const Set*       gPool;
  const uint32_t*  gCodes;
  uint32_t*        gFlags;

  WV_DEV RepeatBig(const AsmParams& p, const LgParams& g, uint8_t* w, char* l) : P(p), G(g), ws(w), lds(l) { lane = unsigned(wv::lane()); }

  WV_DEV uint64_t succOf(const unsigned nd, const FRec8 w) const { return R::links(w, nd, true, gSovf, nSovf); }
  WV_DEV uint64_t predOf(const unsigned nd, const FRec8 w) const { return R::links(w, nd, false, gPovf, nPovf); }

  /// two-sided Kahn peel over the slab's records; state bytes {in:3, out:3, peeled} in LDS.  Returns the number of words removed.
  WV_DEV unsigned peel(uint32_t* queue)
  {
    uint32_t*      st   = reinterpret_cast<uint32_t*>(lds);
    const unsigned stDw = (n + 3) / 4;
    uint32_t*      qTail = queue;  // [0]: the tail; entries from [1]
    if (lane == 0) *qTail = 0;
    for (unsigned w = lane; w < stDw; w += 64) st[w] = 0;
    wv::sync();
    for (unsigned nb = 0; nb < n; nb += 64) {
      const unsigned nd = nb + lane;
      if (nd >= n) continue;
      const FRec8    w  = gRec[nd];
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
      if (src) queue[1 + wv::atomic_add(qTail, 1u)] = nd;
    }
    wv::sync();
    wv::fence_acquire();
    unsigned head = 0, removed = 0;
    while (true) {
      const unsigned tail = wv::first(wv::atomic_load(qTail));
      if (tail == head) break;
      removed += tail - head;
      for (unsigned i = head + lane; i < tail; i += 64) {
        const unsigned nd = queue[1 + i];
        const FRec8    w  = gRec[nd];
        const uint64_t sl = succOf(nd, w), pl = predOf(nd, w);
        for (unsigned c = 0; c < 4; ++c) {
          const unsigned s = R::linkId(sl, c);
          if (s != ASM_NONE && s != nd) {
            const unsigned sh  = 8 * (s & 3);
            const unsigned old = wv::atomic_sub(&st[s >> 2], 1u << sh) >> sh;
            if ((old & 0x7u) == 1u && !(wv::atomic_or(&st[s >> 2], 0x40u << sh) & (0x40u << sh))) queue[1 + wv::atomic_add(qTail, 1u)] = s;
          }
          const unsigned p = R::linkId(pl, c);
          if (p != ASM_NONE && p != nd) {
            const unsigned sh  = 8 * (p & 3);
            const unsigned old = wv::atomic_sub(&st[p >> 2], 8u << sh) >> sh;
            if ((old & 0x38u) == 8u && !(wv::atomic_or(&st[p >> 2], 0x40u << sh) & (0x40u << sh))) queue[1 + wv::atomic_add(qTail, 1u)] = p;
          }
        }
      }
      wv::sync();
      wv::fence_acquire();
      head = tail;
    }
    return removed;
  }

  /// the repeat search; flagV[v] != 0 <=> the word numbered v is a repeat word.  False: not for this path (bucket count beyond the workspace).
  WV_DEV bool exactSearch(const RpbWs& L)
  {
    const unsigned C       = LGL_MAX_NODES;
    uint64_t*      h       = reinterpret_cast<uint64_t*>(ws + L.h);
    uint32_t*      ins     = reinterpret_cast<uint32_t*>(ws + L.ins);
    uint32_t*      seqB    = reinterpret_cast<uint32_t*>(ws + L.seqB);
    uint32_t*      pool    = reinterpret_cast<uint32_t*>(ws + L.pool);
    uint32_t*      vOf     = reinterpret_cast<uint32_t*>(ws + L.vOf);
    uint32_t*      idOf    = reinterpret_cast<uint32_t*>(ws + L.idOf);
    uint32_t*      succ4   = reinterpret_cast<uint32_t*>(ws + L.succ4);
    uint32_t*      flagV   = reinterpret_cast<uint32_t*>(ws + L.flagV);
    uint32_t*      firstRd = reinterpret_cast<uint32_t*>(ws + L.firstRd);
    uint32_t*      byLex   = reinterpret_cast<uint32_t*>(ws + L.byLex);
    uint32_t*      prefix  = reinterpret_cast<uint32_t*>(ws + L.prefix);
    uint32_t*      rdBase  = reinterpret_cast<uint32_t*>(ws + L.rdBase);
    {
      unsigned nbMax = 1;
      for (unsigned s = 0; s < P.n_growth; ++s)
        if (P.growth_size[s] < n) nbMax = P.growth_buckets[s];
      if (nbMax > 3 * C + 32) return false;
    }
    // ---- the words renumbered by first occurrence: v = rank of the word's packed base index ----
    uint32_t* bits = reinterpret_cast<uint32_t*>(lds);  // 65 536 bits
    for (unsigned i = lane; i < 2048; i += 64) bits[i] = 0;
    wv::sync();
    for (unsigned id = lane; id < n; id += 64) {
      const unsigned pb = gPb[id];
      wv::atomic_or(&bits[pb >> 5], 1u << (pb & 31));
    }
    wv::sync();
    {
      unsigned carry = 0;
      for (unsigned b0 = 0; b0 < 2048; b0 += 64) {
        const unsigned c   = unsigned(wv::popc(bits[b0 + lane]));
        unsigned       inc = c;
        for (int off = 1; off < 64; off <<= 1) {
          const unsigned o = wv::shfl(inc, int(lane) - off);
          if (int(lane) >= off) inc += o;
        }
        prefix[b0 + lane] = carry + inc - c;
        carry += wv::readlane(inc, 63);
      }
    }
    wv::sync();
    for (unsigned id = lane; id < n; id += 64) {
      const unsigned pb = gPb[id];
      const unsigned v  = prefix[pb >> 5] + unsigned(wv::popc(bits[pb >> 5] & ((1u << (pb & 31)) - 1u)));
      vOf[id]           = v;
      idOf[v]           = id;
    }
    if (lane < 257) rdBase[lane] = 0;
    for (unsigned r = 64 + lane; r < 320; r += 64) rdBase[r] = 0;
    wv::sync();
    wv::fence_acquire();
    // ---- successors in the new numbering, std::hash, first read, the lexicographic list ----
    for (unsigned id = lane; id < n; id += 64) {
      const FRec8    w  = gRec[id];
      const uint64_t sl = succOf(id, w);
      const unsigned v  = vOf[id];
      for (unsigned c = 0; c < 4; ++c) {
        const unsigned s = R::linkId(sl, c);
        succ4[4 * v + c] = (s == ASM_NONE) ? ASM_NONE : vOf[s];
      }
      h[v]     = libstdcxxStringHash<2>(gCodes, gPb[id], k);
      flagV[v] = 0;
      unsigned fr = 0;
      if (id < nFat) {
        const Set st = gPool[id];
        bool      got = false;
        for (unsigned q = 0; q < LgL::SETW; ++q)
          if (!got && st.w[q]) {
            fr  = 64 * q + unsigned(wv::ctz(st.w[q]));
            got = true;
          }
      } else {
        fr = gRd1[id];
      }
      firstRd[v] = fr;
      wv::atomic_add(&rdBase[fr], 1u);
      byLex[gLex[id]] = v;
    }
    wv::sync();
    wv::fence_acquire();
    // ---- insertion sequence of wordCount (:516-548): reads in order, a read's new words in lexicographic order = a stable counting
    // sort of the lexicographic list by first read ----
    {
      unsigned carry = 0;
      for (unsigned b0 = 0; b0 < 256; b0 += 64) {
        const unsigned c   = wv::atomic_load(&rdBase[b0 + lane]);
        unsigned       inc = c;
        for (int off = 1; off < 64; off <<= 1) {
          const unsigned o = wv::shfl(inc, int(lane) - off);
          if (int(lane) >= off) inc += o;
        }
        const unsigned base = carry + inc - c;
        carry += wv::readlane(inc, 63);
        wv::sync();
        rdBase[b0 + lane] = base;
      }
    }
    wv::sync();
    for (unsigned i0 = 0; i0 < n; i0 += 64) {
      const unsigned i     = i0 + lane;
      const bool     valid = i < n;
      const unsigned v     = valid ? byLex[i] : 0u;
      const unsigned d     = valid ? firstRd[v] : 0u;
      uint64_t       peers = wv::ballot(valid);
      for (int bit = 0; bit < 8; ++bit) {
        const bool     on = (d >> bit) & 1u;
        const uint64_t m  = wv::ballot(valid && on);
        peers &= on ? m : ~m;
      }
      unsigned base = 0;
      if (valid) base = rdBase[d];
      wv::sync();
      if (valid) {
        ins[base + unsigned(wv::popc(peers & ((uint64_t(1) << lane) - 1)))] = v;
        if ((peers >> lane) == 1u) rdBase[d] = base + unsigned(wv::popc(peers));
      }
      wv::sync();
      wv::fence_acquire();
    }
    // ---- libstdc++'s node order of wordCount, then of wordIndices (filled by iterating wordCount, :631-633) ----
    const uint32_t* roots;
    {
      const unsigned nbCap = 3 * C + 32;
      uint32_t*      chain = pool;
      uint32_t*      offs  = pool + C;
      uint32_t*      spare = pool + 2 * C;
      uint32_t*      bf    = pool + 3 * size_t(C);
      uint32_t*      bh    = bf + nbCap;
      uint32_t*      seqA  = byLex;  // (the lexicographic list is done with)
      uint32_t* order1 = unorderedOrderWave(P, h, ins, seqB, spare, chain, offs, bf, bh, n);
      uint32_t* s1     = (order1 == seqB) ? spare : seqB;
      uint32_t* order2 = unorderedOrderWave(P, h, order1, seqA, s1, chain, offs, bf, bh, n);
      if (order2 != seqA && order2 != seqB) {  // the search below reuses the pool: park the root order where it survives
        for (unsigned i = lane; i < n; i += 64) seqB[i] = order2[i];
        wv::sync();
        order2 = seqB;
      }
      roots = order2;
    }
    // ---- the search (:555-625), successors in alphabet order; see repeat_exact.hpp for the run arithmetic ----
    {
      const uint32_t ONSTACK = 0x80000000u, INF = 0x7fffffffu, RUNFRAME = 0x80000000u;
      uint32_t*      idxA    = pool;
      uint32_t*      lowA    = pool + C;
      uint32_t*      stackA  = pool + 2 * size_t(C);
      uint32_t*      runNext = pool + 3 * size_t(C);
      uint32_t*      frLo    = pool + 4 * size_t(C);
      uint32_t*      frHi    = pool + 5 * size_t(C);
      for (unsigned nd = lane; nd < n; nd += 64) {
        idxA[nd] = 0;
        lowA[nd] = 0;
        unsigned only = ASM_NONE, cnt = 0;
        bool     self = false;
        for (unsigned c = 0; c < 4; ++c) {
          const unsigned sx = succ4[4 * nd + c];
          if (sx == ASM_NONE) continue;
          if (sx == nd) self = true;
          only = sx;
          ++cnt;
        }
        runNext[nd] = (cnt == 1 && !self) ? only : ASM_NONE;
      }
      wv::sync();
      wv::fence_acquire();
      enum { REQ_NONE = 0, REQ_NEXTROOT = 1, REQ_DESCEND = 2, REQ_POP = 3 };
      unsigned fp = 0, sp = 0, nextIndex = 1, rootCursor = 0;
      unsigned returning = 0, retLow = INF;
      unsigned afterRun = ASM_NONE;
      unsigned req = REQ_NEXTROOT, reqA = 0, reqB = 0, reqC = 0, reqD = 0;
      bool     done = false;
      while (!done) {
        if (req == REQ_NEXTROOT) {
          unsigned root = ASM_NONE;
          while (rootCursor < n) {
            const unsigned ri = rootCursor + lane;
            const unsigned r  = (ri < n) ? roots[ri] : 0u;
            const bool     un = (ri < n) && (idxA[r] == 0);
            const uint64_t m  = wv::ballot(un);
            if (m) {
              const int l = wv::ctz(m);
              root        = wv::readlane(r, l);
              rootCursor += unsigned(l) + 1;
              break;
            }
            rootCursor += 64;
          }
          if (root == ASM_NONE) {
            done = true;
            continue;
          }
          if (runNext[root] != ASM_NONE) {
            req  = REQ_DESCEND;
            reqA = root;
          } else {
            if (lane == 0) {
              idxA[root] = nextIndex;
              lowA[root] = nextIndex | ONSTACK;
              stackA[sp] = root;
              frLo[fp]   = root << 3;
              frHi[fp]   = sp;
            }
            nextIndex++;
            sp++;
            fp++;
            req = REQ_NONE;
            wv::sync();
          }
          returning = 0;
          continue;
        }
        if (req == REQ_DESCEND) {
          unsigned       c     = reqA;
          const unsigned p0    = sp;
          unsigned       len   = 0;
          unsigned       lastN = c;
          while (true) {
            const unsigned x      = c + lane;
            const bool     inR    = x < n;
            const unsigned rn     = inR ? runNext[x] : ASM_NONE;
            const bool     okSelf = inR && rn != ASM_NONE && idxA[x] == 0;
            const unsigned contig = (rn == x + 1) ? 1u : 0u;
            const unsigned prevContig = wv::shr1(contig, 1u);
            const uint64_t good = wv::ballot(okSelf && prevContig != 0);
            const unsigned take = (~good == 0) ? 64u : unsigned(wv::ctz(~good));
            if (lane < take) {
              idxA[x]           = nextIndex + lane;
              lowA[x]           = (nextIndex + lane) | ONSTACK;
              stackA[sp + lane] = x;
            }
            nextIndex += take;
            sp += take;
            len += take;
            lastN                 = c + take - 1;
            const unsigned lastRn = wv::readlane(rn, int(take - 1));
            wv::sync();
            if (runNext[lastRn] != ASM_NONE && idxA[lastRn] == 0) {
              c = lastRn;
              continue;
            }
            break;
          }
          if (lane == 0) {
            frLo[fp] = len;
            frHi[fp] = p0 | RUNFRAME;
          }
          fp++;
          afterRun = lastN;
          req      = REQ_NONE;
          wv::sync();
          continue;
        }
        if (req == REQ_POP) {
          for (unsigned i = reqA + lane; i < reqB; i += 64) {
            const unsigned w = stackA[i];
            lowA[w] &= ~ONSTACK;
            if (reqD && i >= reqC) flagV[w] = 1;
          }
          req = REQ_NONE;
          wv::sync();
          continue;
        }
        if (lane == 0) {
          while (req == REQ_NONE) {
            if (afterRun != ASM_NONE) {
              const unsigned y = runNext[afterRun];
              afterRun         = ASM_NONE;
              if (idxA[y] == 0) {
                idxA[y] = nextIndex;
                lowA[y] = nextIndex | ONSTACK;
                nextIndex++;
                stackA[sp] = y;
                frLo[fp]   = y << 3;
                frHi[fp]   = sp;
                sp++;
                fp++;
                returning = 0;
              } else {
                retLow    = (lowA[y] & ONSTACK) ? idxA[y] : INF;
                returning = 1;
              }
              continue;
            }
            if (fp == 0) {
              req = REQ_NEXTROOT;
              break;
            }
            const unsigned hi = frHi[fp - 1];
            if (hi & RUNFRAME) {
              const unsigned p0       = hi & ~RUNFRAME;
              const unsigned len      = frLo[fp - 1];
              const unsigned firstIdx = idxA[stackA[p0]];
              const unsigned Lw       = retLow;
              fp--;
              returning = 1;
              if (Lw < firstIdx) {
                retLow = Lw;
                continue;
              }
              unsigned flagFrom = sp, small = 0;
              if (Lw < firstIdx + len) {
                flagFrom = p0 + (Lw - firstIdx);
                small    = ((idxA[stackA[sp - 1]] - Lw) <= 50) ? 1u : 0u;
                if (sp - flagFrom == 1) small = 0;
              }
              retLow = firstIdx;
              if (sp - p0 <= 4) {
                for (unsigned i = p0; i < sp; ++i) {
                  const unsigned w = stackA[i];
                  lowA[w] &= ~ONSTACK;
                  if (small && i >= flagFrom) flagV[w] = 1;
                }
                sp = p0;
              } else {
                req  = REQ_POP;
                reqA = p0;
                reqB = sp;
                reqC = flagFrom;
                reqD = small;
                sp   = p0;
              }
              continue;
            }
            const unsigned f   = frLo[fp - 1];
            const unsigned nd  = f >> 3;
            const unsigned sym = f & 7;
            if (returning) {
              const unsigned lp = lowA[nd] & ~ONSTACK;
              if (retLow < lp) lowA[nd] = retLow | (lowA[nd] & ONSTACK);
              returning = 0;
            }
            if (sym < 4) {
              frLo[fp - 1]      = f + 1;
              const unsigned sx = succ4[4 * nd + sym];
              if (sx == nd) {  // homopolymer (:574-577)
                flagV[nd] = 1;
                continue;
              }
              if (sx == ASM_NONE) continue;
              if (idxA[sx] == 0) {
                if (runNext[sx] != ASM_NONE) {
                  req  = REQ_DESCEND;
                  reqA = sx;
                } else {
                  idxA[sx] = nextIndex;
                  lowA[sx] = nextIndex | ONSTACK;
                  nextIndex++;
                  stackA[sp] = sx;
                  frLo[fp]   = sx << 3;
                  frHi[fp]   = sp;
                  sp++;
                  fp++;
                }
              } else if (lowA[sx] & ONSTACK) {
                const unsigned l = lowA[nd] & ~ONSTACK;
                if (idxA[sx] < l) lowA[nd] = idxA[sx] | ONSTACK;
              }
              continue;
            }
            const unsigned myLow = lowA[nd] & ~ONSTACK;
            const unsigned myPos = hi;
            if (myLow == idxA[nd]) {
              if (sp - myPos == 1) {
                lowA[nd] &= ~ONSTACK;
                sp = myPos;
              } else {
                const unsigned small = ((idxA[stackA[sp - 1]] - idxA[nd]) <= 50) ? 1u : 0u;
                if (sp - myPos <= 4) {
                  for (unsigned i = myPos; i < sp; ++i) {
                    const unsigned w = stackA[i];
                    lowA[w] &= ~ONSTACK;
                    if (small) flagV[w] = 1;
                  }
                } else {
                  req  = REQ_POP;
                  reqA = myPos;
                  reqB = sp;
                  reqC = myPos;
                  reqD = small;
                }
                sp = myPos;
              }
            }
            fp--;
            retLow    = myLow;
            returning = 1;
          }
        }
        wv::sync();
        fp        = wv::first(fp);
        sp        = wv::first(sp);
        nextIndex = wv::first(nextIndex);
        returning = wv::first(returning);
        retLow    = wv::first(retLow);
        afterRun  = wv::first(afterRun);
        req       = wv::first(req);
        reqA      = wv::first(reqA);
        reqB      = wv::first(reqB);
        reqC      = wv::first(reqC);
        reqD      = wv::first(reqD);
      }
    }
    wv::sync();
    wv::fence_acquire();
    return true;
  }

  WV_DEV int run(const unsigned locus)
  {
    slab        = G.arena + G.slab_off[locus];
    LgHdr* gh   = reinterpret_cast<LgHdr*>(slab);
    n           = wv::first(gh->nNodes);
    nFat        = wv::first(gh->nFat);
    k           = wv::first(gh->k);
    nSovf       = wv::first(gh->nSovf);
    nPovf       = wv::first(gh->nPovf);
    const unsigned codeWords = wv::first(gh->codeWords);
    if (n == 0 || n > LGL_MAX_NODES) return RPB_PUNT;
    SL     = lgSlabL(n, nFat, codeWords);
    gRec   = reinterpret_cast<const FRec8*>(slab + SL.recs);
    gPool  = reinterpret_cast<const Set*>(slab + SL.pool);
    gSovf  = reinterpret_cast<const uint16_t*>(slab + SL.sovf);
    gPovf  = reinterpret_cast<const uint16_t*>(slab + SL.povf);
    gRd1   = slab + SL.rd1;
    gPb    = reinterpret_cast<const uint16_t*>(slab + SL.pb);
    gCodes = reinterpret_cast<const uint32_t*>(slab + SL.codes);
    gLex   = reinterpret_cast<const uint16_t*>(slab + SL.lex);
    gFlags = reinterpret_cast<uint32_t*>(slab + SL.flags);
    const RpbWs    L       = rpbWorkspaceLayout();
    uint32_t*      queue   = reinterpret_cast<uint32_t*>(ws + L.queue);
    const unsigned removed = peel(queue);
    unsigned       need, nCore = 0;
    bool           cyclic = removed != n;
    if (!cyclic) {
      need = ckNeedOf<LgL>(n, nFat, true);
    } else {
      nCore = n - removed;
      if (nCore > LGL_CORE_CAP) return RPB_PUNT;
      // the core bitmap (before the LDS is reused)
      const uint32_t* st = reinterpret_cast<const uint32_t*>(lds);
      uint32_t        mine[LgL::UNUSED_DW / 64];
      for (unsigned u = 0; u < LgL::UNUSED_DW / 64; ++u) {
        const unsigned d = lane + 64 * u;
        uint32_t       b = 0;
        for (unsigned j = 0; j < 32; ++j) {
          const unsigned nd = 32 * d + j;
          if (nd < n && !((st[nd >> 2] >> (8 * (nd & 3))) & 0x40u)) b |= 1u << j;
        }
        mine[u] = b;
      }
      wv::sync();
      for (unsigned u = 0; u < LgL::UNUSED_DW / 64; ++u) gFlags[lane + 64 * u] = mine[u];
      if (!exactSearch(L)) return RPB_PUNT;
      const uint32_t* vOf   = reinterpret_cast<const uint32_t*>(ws + L.vOf);
      const uint32_t* flagV = reinterpret_cast<const uint32_t*>(ws + L.flagV);
      for (unsigned u = 0; u < LgL::UNUSED_DW / 64; ++u) {
        const unsigned d = lane + 64 * u;
        uint32_t       b = 0;
        for (unsigned j = 0; j < 32; ++j) {
          const unsigned nd = 32 * d + j;
          if (nd < n && flagV[vOf[nd]]) b |= 1u << j;
        }
        gFlags[LgL::UNUSED_DW + d] = b;
      }
      need = ckNeedCyclic<LgL>(n, nFat, nCore);
    }
    unsigned cls = LG_CLASSES;
    for (unsigned c = LG_CLASSES; c-- > 0;)
      if (G.class_bytes[c] && need <= G.class_bytes[c]) cls = c;
    if (cls == LG_CLASSES) return RPB_PUNT;
    wv::sync();
    if (lane == 0) {
      gh->acyclic = cyclic ? 0u : 1u;
      gh->cyclic  = cyclic ? 1u : 0u;
      gh->nCore   = nCore;
      gh->need    = need;
      G.class_ids[size_t(cls) * G.class_stride + wv::atomic_add(&G.class_count[cls], 1u)] = locus;
    }
    return RPB_DONE;
  }
};

/// persistent wavefronts over graph_big_kernel's list of graphs without a proof (G.cyc_ids / G.cyc_count); P.counter: this launch's own
/// work counter; P.lds_bytes = RPB_LDS_BYTES per wave; G.rws / G.rws_stride: a workspace per wave
WV_KERNEL_OCC(4) void repeat_big_kernel(const LgArgs A)
{
  const AsmParams& P     = A.P;
  const LgParams&  G     = A.G;
  uint8_t*         ws    = G.rws + uint64_t(wv::block()) * G.rws_stride;
  char*            lds   = wv::lds(RPB_LDS_BYTES);
  const unsigned   nLoci = wv::first(wv::atomic_load(G.cyc_count));
  while (true) {
    unsigned slot = 0;
    if (wv::lane() == 0) slot = wv::atomic_add(P.counter, 1u);
    slot = wv::first(slot);
    if (slot >= nLoci) break;
    const unsigned locus = G.cyc_ids[slot];
    RepeatBig      r(P, G, ws, lds);
    const int      rc = r.run(locus);
    wv::sync();
    if (rc != RPB_DONE && wv::lane() == 0) P.punt_ids[wv::atomic_add(P.punt_count, 1u)] = locus;
    wv::sync();
  }
}

}  // namespace manta_dev
