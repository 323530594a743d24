// repeat_big_kernel: between graph_big_kernel and contig_big_kernel, for the big-class loci whose graph came WITHOUT a proof of
// acyclicity that its peel could not empty (every pile with a tandem repeat, at most of its word lengths).  One wavefront per locus,
// the compact graph read from the locus' slab, a per-wave workspace in device memory, 8 KB of LDS:
//
//   1. the strongly connected components, in any order and only among the words graph_big_kernel's two-sided peel left (every cycle lies
//      in that core; a graph the peel empties never comes here): which words lie on a cycle -- the only ones a walk can meet twice
//      (asm_contig.hpp keeps a visited bitmap over them) -- and how large the smallest component is.
//   2. only a component of at most 51 words can pass the reference's small-circle rule (IterativeAssembler.cpp:612), and only then does the
//      visiting order matter: the reference's repeat search (:555-642) EXACTLY -- its order is the iteration order of a std::unordered_map.
//      Same construction as repeat_exact.hpp (insertion sequence -> std::hash -> libstdc++ node order, twice -> Tarjan with whole
//      unbranched runs per step), on the compact graph: the words are renumbered by FIRST OCCURRENCE so that an unbranched stretch has
//      consecutive numbers again (the ids of the slab are in seed order), the lexicographic ranks inside a read's group come from graph_big_kernel.
//   3. core and repeat-word bitmaps into the slab, the locus into contig_big_kernel's LDS class list (the class depends on the core's size).
#pragma once
#include "asm_lds.hpp"

namespace manta_dev {

static const unsigned RPB_LDS_BYTES = 8192;  ///< per wave: peel state (a byte per word), then the first-occurrence bitmap (65 536 bits)

struct RpbWs {
  uint64_t h, ins, seqB, pool, vOf, idOf, succ4, flagV, firstRd, byLex, queue, prefix, rdBase, coreV, total;
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
  L.coreV   = asmPut(o, 4 * C);
  L.total   = (o + 255) & ~uint64_t(255);
  return L;
}

enum { RPB_DONE = 0, RPB_PUNT = 1 };

/// std::hash<std::string> of the word at packed base index pb of a 2-bit pile (libstdcxxStringHash<2>, repeat_exact.hpp, with the
/// eight characters of a step taken from two dwords instead of one load per character)
WV_DEV uint64_t rpbWordHash(const uint32_t* codes, const unsigned pb, const unsigned len)
{
  auto chunk = [&](const unsigned i, const unsigned n) -> uint64_t {
    const unsigned p  = pb + i;
    const unsigned wi = p >> 4, sh = (p & 15) * 2;
    const uint64_t two = (uint64_t(codes[wi]) << 32) | codes[wi + 1];  // (every read of the pile is followed by a padding dword)
    const uint32_t v   = uint32_t((two << sh) >> 48);                  // eight bases, first base in the top bits
    uint64_t       data = 0;
    for (unsigned b = 0; b < n; ++b) {
      const unsigned c = (v >> (14 - 2 * b)) & 3u;
      data |= uint64_t((0x54474341u >> (8 * c)) & 0xffu) << (8 * b);   // "ACGT"[c]
    }
    return data;
  };
  const uint64_t mul  = (uint64_t(0xc6a4a793UL) << 32) + uint64_t(0x5bd1e995UL);
  uint64_t       hash = uint64_t(0xc70f6907UL) ^ (uint64_t(len) * mul);
  const unsigned lenAligned = len & ~7u;
  for (unsigned i = 0; i < lenAligned; i += 8) {
    const uint64_t data = murmurShiftMix(chunk(i, 8) * mul) * mul;
    hash ^= data;
    hash *= mul;
  }
  if (len & 7u) {
    hash ^= chunk(lenAligned, len & 7u);
    hash *= mul;
  }
  hash = murmurShiftMix(hash) * mul;
  hash = murmurShiftMix(hash);
  return hash;
}

/// unorderedOrderWave (repeat_exact.hpp) for this kernel, built for latency: one wave, arrays in device memory, so a pass is a chain of
/// dependent round trips per element -- here every lane carries FOUR elements through a pass (four loads / atomics in flight), and the two
/// passes over the buckets (run lengths, emission) became passes over the elements: an element finds its bucket's first time and its own rank
/// in the bucket (members with a later time) by walking the bucket's chain, 1-2 links on average.  chain[t] = {next time : 16 (0xffff:
/// none), bucket : 16}.  Same result as unorderedOrderWave: buckets by first time, latest first; inside a bucket by time, latest first.
WV_DEV uint32_t* unorderedOrderWave4(
    const AsmParams& P, const uint64_t* h, const uint32_t* ins, uint32_t* a, uint32_t* b, uint32_t* chain, uint32_t* offs, uint32_t* bf,
    uint32_t* bh, const unsigned n)
{
  const uint32_t NIL   = 0xffffffffu;
  const unsigned NIL16 = 0xffffu;
  const unsigned lane  = unsigned(wv::lane());
  uint32_t*      cur   = a;
  uint32_t*      out   = b;
  unsigned       m0    = 0;
  unsigned       sched = 0;
  unsigned       nb    = 1;
  while (m0 < n) {
    while (sched < P.n_growth && P.growth_size[sched] <= m0) nb = P.growth_buckets[sched++];
    unsigned m1 = n;
    if (sched < P.n_growth && P.growth_size[sched] < m1) m1 = P.growth_size[sched];
    for (unsigned i = lane; i < nb; i += 64) {
      bf[i] = NIL;
      bh[i] = NIL;
    }
    wv::sync();
    // A. every element into its bucket's chain; the bucket's first time
    for (unsigned t0 = lane; t0 < m1; t0 += 256) {
      unsigned tt[4], bk[4];
      uint32_t node[4], old[4];
      uint64_t hv[4];
      bool     ok[4];
      for (unsigned u = 0; u < 4; ++u) {
        tt[u]   = t0 + 64 * u;
        ok[u]   = tt[u] < m1;
        node[u] = ok[u] ? ((tt[u] < m0) ? cur[tt[u]] : ins[tt[u]]) : 0u;
      }
      for (unsigned u = 0; u < 4; ++u) hv[u] = ok[u] ? h[node[u]] : uint64_t(0);
      for (unsigned u = 0; u < 4; ++u)
        if (ok[u] && tt[u] >= m0) cur[tt[u]] = node[u];
      for (unsigned u = 0; u < 4; ++u) bk[u] = unsigned(hv[u] % nb);
      for (unsigned u = 0; u < 4; ++u) old[u] = ok[u] ? wv::atomic_exch(&bh[bk[u]], tt[u]) : NIL;
      for (unsigned u = 0; u < 4; ++u)
        if (ok[u]) wv::atomic_min(&bf[bk[u]], tt[u]);
      for (unsigned u = 0; u < 4; ++u)
        if (ok[u]) chain[tt[u]] = (old[u] & 0xffffu) | (bk[u] << 16);
    }
    wv::sync();
    wv::fence_acquire();
    // B. run lengths at the buckets' first times
    for (unsigned t0 = lane; t0 < m1; t0 += 256) {
      unsigned tt[4], x[4], cnt[4];
      bool     ok[4], isFirst[4];
      for (unsigned u = 0; u < 4; ++u) {
        tt[u] = t0 + 64 * u;
        ok[u] = tt[u] < m1;
      }
      for (unsigned u = 0; u < 4; ++u) {
        const unsigned bkt   = ok[u] ? (chain[tt[u]] >> 16) : 0u;
        const uint32_t first = ok[u] ? wv::atomic_load(&bf[bkt]) : NIL;
        isFirst[u]           = ok[u] && first == tt[u];
        x[u]                 = isFirst[u] ? (wv::atomic_load(&bh[bkt]) & 0xffffu) : NIL16;
        cnt[u]               = 0;
      }
      while (x[0] != NIL16 || x[1] != NIL16 || x[2] != NIL16 || x[3] != NIL16) {
        for (unsigned u = 0; u < 4; ++u)
          if (x[u] != NIL16) {
            cnt[u]++;
            x[u] = chain[x[u]] & 0xffffu;
          }
      }
      for (unsigned u = 0; u < 4; ++u)
        if (ok[u]) offs[tt[u]] = isFirst[u] ? cnt[u] : 0u;
    }
    wv::sync();
    // exclusive suffix sum over time, top chunk first (as unorderedOrderWave)
    {
      unsigned carry = 0;
      for (unsigned top = ((m1 + 63) / 64) * 64; top > 0; top -= 64) {
        const unsigned t   = top - 64 + (63 - lane);
        const unsigned w   = (t < m1) ? offs[t] : 0u;
        unsigned       inc = w;
        for (int off = 1; off < 64; off <<= 1) {
          const unsigned o = wv::shfl(inc, int(lane) - off);
          if (int(lane) >= off) inc += o;
        }
        if (t < m1) offs[t] = carry + inc - w;
        carry += wv::shfl(inc, 63);
      }
    }
    wv::sync();
    // C. every element to its place: its bucket's run starts at offs[first time], inside the run latest time first
    for (unsigned t0 = lane; t0 < m1; t0 += 256) {
      unsigned tt[4], x[4], rank[4], base[4];
      uint32_t node[4];
      bool     ok[4];
      for (unsigned u = 0; u < 4; ++u) {
        tt[u] = t0 + 64 * u;
        ok[u] = tt[u] < m1;
      }
      for (unsigned u = 0; u < 4; ++u) {
        const unsigned bkt   = ok[u] ? (chain[tt[u]] >> 16) : 0u;
        const uint32_t first = ok[u] ? wv::atomic_load(&bf[bkt]) : 0u;
        base[u]              = ok[u] ? offs[first] : 0u;
        x[u]                 = ok[u] ? (wv::atomic_load(&bh[bkt]) & 0xffffu) : NIL16;
        node[u]              = ok[u] ? cur[tt[u]] : 0u;
        rank[u]              = 0;
      }
      while (x[0] != NIL16 || x[1] != NIL16 || x[2] != NIL16 || x[3] != NIL16) {
        for (unsigned u = 0; u < 4; ++u)
          if (x[u] != NIL16) {
            if (x[u] > tt[u]) rank[u]++;
            x[u] = chain[x[u]] & 0xffffu;
          }
      }
      for (unsigned u = 0; u < 4; ++u)
        if (ok[u]) out[base[u] + rank[u]] = node[u];
    }
    wv::sync();
    wv::fence_acquire();
    uint32_t* tmpPtr = cur;
    cur              = out;
    out              = tmpPtr;
    m0               = m1;
  }
  return cur;
}

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

  uint64_t tMark;
  /// phases: 0 renumbering + successors, 1 the components (search in any order), 2 hash + insertion sequence, 3 node order of wordCount,
  /// 4 of wordIndices, 5 the search in the reference's order, 6 bitmaps + class
  WV_DEV void tick(const int phase)
  {
    const uint64_t now = wv::clock();
    if (G.rprof && lane == 0) wv::atomic_add(&G.rprof[phase], (unsigned long long)(now - tMark));
    tMark = now;
  }
  // the per-wave workspace
  uint64_t* h;
  uint32_t *ins, *seqB, *pool, *vOf, *idOf, *succ4, *flagV, *firstRd, *byLex, *prefix, *rdBase, *sccV, *coreV;
  WV_DEV void bind(const RpbWs& L)
  {
    h       = reinterpret_cast<uint64_t*>(ws + L.h);
    ins     = reinterpret_cast<uint32_t*>(ws + L.ins);
    seqB    = reinterpret_cast<uint32_t*>(ws + L.seqB);
    pool    = reinterpret_cast<uint32_t*>(ws + L.pool);
    vOf     = reinterpret_cast<uint32_t*>(ws + L.vOf);
    idOf    = reinterpret_cast<uint32_t*>(ws + L.idOf);
    succ4   = reinterpret_cast<uint32_t*>(ws + L.succ4);
    flagV   = reinterpret_cast<uint32_t*>(ws + L.flagV);
    firstRd = reinterpret_cast<uint32_t*>(ws + L.firstRd);
    byLex   = reinterpret_cast<uint32_t*>(ws + L.byLex);
    prefix  = reinterpret_cast<uint32_t*>(ws + L.prefix);
    rdBase  = reinterpret_cast<uint32_t*>(ws + L.rdBase);
    sccV    = reinterpret_cast<uint32_t*>(ws + L.queue);
    coreV   = reinterpret_cast<uint32_t*>(ws + L.coreV);
  }
  WV_DEV uint64_t succOf(const unsigned nd, const FRec8 w) const { return R::links(w, nd, true, gSovf, nSovf); }
  WV_DEV uint64_t predOf(const unsigned nd, const FRec8 w) const { return R::links(w, nd, false, gPovf, nPovf); }

  /// the words renumbered by first occurrence (v = rank of the word's packed base index: an unbranched stretch has consecutive numbers
  /// again -- the slab's ids are in seed order), the successors in that numbering (alphabet order)
  WV_DEV void renumber()
  {
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
    for (unsigned id = lane; id < n; id += 64) {
      const FRec8    w  = gRec[id];
      const uint64_t sl = succOf(id, w);
      const unsigned v  = vOf[id];
      for (unsigned c = 0; c < 4; ++c) {
        const unsigned s = R::linkId(sl, c);
        succ4[4 * v + c] = (s == ASM_NONE) ? ASM_NONE : vOf[s];
      }
      flagV[v] = 0;
      sccV[v]  = 0;
      coreV[v] = (gFlags[id >> 5] >> (id & 31)) & 1u;  // graph_big_kernel's peel left the word
    }
    wv::sync();
    wv::fence_acquire();
  }

  /// the reference's root order: iteration order of wordIndices (:627-642).  False: bucket count beyond the workspace.
  WV_DEV bool rootOrder(const uint32_t*& roots)
  {
    const unsigned C = LGL_MAX_NODES;
    {
      unsigned nbMax = 1;
      for (unsigned s = 0; s < P.n_growth; ++s)
        if (P.growth_size[s] < n) nbMax = P.growth_buckets[s];
      if (nbMax > 3 * C + 32) return false;
    }
    // std::hash, first read, the lexicographic list
    for (unsigned id = lane; id < n; id += 64) {
      const unsigned v = vOf[id];
      h[v]             = rpbWordHash(gCodes, gPb[id], k);
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
    tick(2);
    // ---- libstdc++'s node order of wordCount, then of wordIndices (filled by iterating wordCount, :631-633) ----
    {
      const unsigned nbCap = 3 * C + 32;
      uint32_t*      chain = pool;
      uint32_t*      offs  = pool + C;
      uint32_t*      spare = pool + 2 * C;
      uint32_t*      bf    = pool + 3 * size_t(C);
      uint32_t*      bh    = bf + nbCap;
      uint32_t*      seqA  = byLex;  // (the lexicographic list is done with)
      uint32_t* order1 = unorderedOrderWave4(P, h, ins, seqB, spare, chain, offs, bf, bh, n);
      tick(3);
      uint32_t* s1     = (order1 == seqB) ? spare : seqB;
      uint32_t* order2 = unorderedOrderWave4(P, h, order1, seqA, s1, chain, offs, bf, bh, n);
      if (order2 != seqA && order2 != seqB) {  // the search below reuses the pool: park the root order where it survives
        for (unsigned i = lane; i < n; i += 64) seqB[i] = order2[i];
        wv::sync();
        order2 = seqB;
      }
      roots = order2;
    }
    tick(4);
    return true;
  }

  /// The search (:555-625), successors in alphabet order; see repeat_exact.hpp for the run arithmetic.  roots == nullptr: roots in
  /// numbering order and only inside the core graph_big_kernel's peel left (every cycle lies in it) -- the strongly connected components
  /// do not depend on the order, so this pass marks the words that lie on a cycle (sccV: the only ones a walk can meet twice) and finds the smallest component: only a
  /// component of at most 51 words can pass the reference's small-circle test (:612: index span <= 50), and only then does the order
  /// matter.  roots != nullptr: the reference's order; flagV[v] != 0 <=> word v is a repeat word.  Returns the smallest component's size.
  WV_DEV unsigned search(const uint32_t* roots)
  {
    const unsigned C      = LGL_MAX_NODES;
    const bool     exact  = roots != nullptr;
    unsigned       minScc = 0x7fffffffu;
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
          if (!exact && sx != nd && !coreV[sx]) continue;  // (the component scan stays inside the core)
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
            const unsigned r  = (ri < n) ? (exact ? roots[ri] : ri) : 0u;
            const bool     un = (ri < n) && (idxA[r] == 0) && (exact || coreV[r] != 0);
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
            if ((reqD & 1u) && i >= reqC) flagV[w] = 1;
            if ((reqD & 2u) && i >= reqC) sccV[w] = 1;
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
                small    = (exact && (idxA[stackA[sp - 1]] - Lw) <= 50) ? 1u : 0u;
                if (sp - flagFrom == 1) {
                  small = 0;
                } else {
                  small |= 2u;  // a component of more than one word
                  if (sp - flagFrom < minScc) minScc = sp - flagFrom;
                }
              }
              retLow = firstIdx;
              if (sp - p0 <= 4) {
                for (unsigned i = p0; i < sp; ++i) {
                  const unsigned w = stackA[i];
                  lowA[w] &= ~ONSTACK;
                  if ((small & 1u) && i >= flagFrom) flagV[w] = 1;
                  if ((small & 2u) && i >= flagFrom) sccV[w] = 1;
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
              if (!exact && !coreV[sx]) continue;
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
                const unsigned small = ((exact && (idxA[stackA[sp - 1]] - idxA[nd]) <= 50) ? 1u : 0u) | 2u;
                if (sp - myPos < minScc) minScc = sp - myPos;
                if (sp - myPos <= 4) {
                  for (unsigned i = myPos; i < sp; ++i) {
                    const unsigned w = stackA[i];
                    lowA[w] &= ~ONSTACK;
                    if (small & 1u) flagV[w] = 1;
                    sccV[w] = 1;
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
        minScc    = wv::first(minScc);
      }
    }
    wv::sync();
    wv::fence_acquire();
    wv::sync();
    wv::fence_acquire();
    return minScc;
  }

  WV_DEV int run(const unsigned locus)
  {
    tMark       = wv::clock();
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
    bind(rpbWorkspaceLayout());
    renumber();
    tick(0);
    const unsigned minScc = search(nullptr);
    tick(1);
    const bool     cyclic = minScc != 0x7fffffffu;  // (self loops alone: not a cycle for this test -- as the peel of the other paths)
    unsigned       need, nCore = 0;
    if (!cyclic) {
      need = ckNeedOf<LgL>(n, nFat, true);
    } else {
      if (minScc <= 51) {  // :612 can hold for some visiting order: the reference's
        const uint32_t* roots = nullptr;
        if (!rootOrder(roots)) return RPB_PUNT;
        for (unsigned v = lane; v < n; v += 64) flagV[v] = 0;
        wv::sync();
        search(roots);
        tick(5);
      }
      // the words on a cycle / the repeat words by id
      for (unsigned u = 0; u < LgL::UNUSED_DW / 64; ++u) {
        const unsigned d = lane + 64 * u;
        uint32_t       b = 0, c = 0;
        for (unsigned j = 0; j < 32; ++j) {
          const unsigned nd = 32 * d + j;
          if (nd < n) {
            const unsigned v = vOf[nd];
            if (flagV[v]) b |= 1u << j;
            if (sccV[v]) c |= 1u << j;
          }
        }
        gFlags[d]                   = c;
        gFlags[LgL::UNUSED_DW + d] = b;
        unsigned cc = unsigned(wv::popc(c));
        for (int off = 1; off < 64; off <<= 1) cc += wv::shfl(cc, wv::lane() ^ off);
        nCore += cc;
      }
      if (nCore > LGL_CORE_CAP) return RPB_PUNT;
      need = ckNeedCyclic<LgL>(n, nFat, nCore);
    }
    unsigned cls = LG_CLASSES;
    for (unsigned c = LG_CLASSES; c-- > 0;)
      if (G.class_bytes[c] && need <= G.class_bytes[c]) cls = c;
    if (cls == LG_CLASSES) return RPB_PUNT;
    tick(6);
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

// waves per SIMD the register allocator must leave room for (4: 128 VGPRs, 264 of them spilled; 2: 256 VGPRs -- the kernel is a chain of
// dependent accesses per graph and a round rarely has more graphs than 8 waves per CU take)
#ifndef MANTA_RPB_OCC
#define MANTA_RPB_OCC 2
#endif
/// persistent wavefronts over graph_big_kernel's list of graphs without a proof (G.cyc_ids / G.cyc_count); P.counter: this launch's own
/// work counter; P.lds_bytes = RPB_LDS_BYTES per wave; G.rws / G.rws_stride: a workspace per wave
#if !MANTA_TU_DEFINES(MANTA_TU_REPEAT)
WV_KERNEL_OCC(MANTA_RPB_OCC) void repeat_big_kernel(const LgArgs A);
#else
WV_KERNEL_OCC(MANTA_RPB_OCC) void repeat_big_kernel(const LgArgs A)
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
    if (rc != RPB_DONE && wv::lane() == 0) {
      P.punt_ids[wv::atomic_add(P.punt_count, 1u)] = locus;
      if (G.stats) wv::atomic_add(&G.stats[6], 1u);
    }
    wv::sync();
  }
}
#endif

}  // namespace manta_dev
