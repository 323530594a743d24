// Speculative lane-per-contig walks.
//
// The reference builds up to 2*maxAssemblyCount contigs one after another (assembly/IterativeAssembler.cpp:685-713):
// seed = best still-unused word, walk, mark the walked words used, repeat.  At config-2 sizes that is ~20 serial walks
// of ~270 dependent steps each -- a latency chain no amount of occupancy hides.  But a walk (:149-501) READS only
// immutable graph data (links, supports, counts, the seed's repeat flag) plus its OWN wordsInContig set; the only
// coupling between contigs is which seeds are still unused.  So:
//   1. take the next T still-unused words in the reference's seed order (count desc, k-mer asc) as TENTATIVE seeds,
//   2. walk all of them at once, one lane each, with lane-private state (read sets of WQ qwords in registers,
//      a private visited bitmap = wordsInContig = the words this walk would erase from unusedWords),
//   3. replay the reference's sequential seed selection over the tentative list: a tentative seed is a real seed
//      iff no previously ACCEPTED walk visited it; accepted walks become contigs in order, the rest are discarded.
// The accepted sequence is exactly the reference's (the first tentative seed is always valid, so every round makes
// progress); a round costs one walk's latency instead of T.
#pragma once

namespace manta_dev {

/// Fills tent_sorted[0..nT) with the next <= T unused words in exact seed order; returns nT (0: none left).
/// Order = count descending, k-mer ascending (the reference's scan of the ordered `unusedWords` set with a strict
/// '>' on the count, :689-696).  Top-T selection without a sort: compact the unused words once, find the count level
/// and then the 16-base-prefix threshold that cut off T words by two binary searches over coalesced arrays, gather
/// the (about T) survivors and order only those exactly.
template <int SB>
WV_DEV unsigned AssemblerT<SB>::selectTentative(const unsigned T)
{
  static const int KW = GEN_KW;
  const unsigned lane = unsigned(wv::lane());
  if (T == 1) {
    const unsigned s = selectSeed();
    if (s == ASM_NONE) return 0;
    if (lane == 0) tent_sorted[0] = s;
    wv::sync();
    return 1;
  }
  // compact list of unused words: node id, count, 16-base prefix (scratch of the exact repeat search, free by now)
  uint32_t* uNode = exact_ws + 64;
  uint32_t* uCnt  = uNode + P.cap_nodes;
  uint32_t* uK32  = uCnt + P.cap_nodes;
  unsigned  U = 0, myMax = 0;
  for (unsigned base = 0; base < nNodes; base += 64) {
    const unsigned nd  = base + lane;
    const bool     sel = (nd < nNodes) && isUnused(nd);
    const uint64_t m   = wv::ballot(sel);
    if (sel) {
      const unsigned pos = U + unsigned(wv::popc(m & ((uint64_t(1) << lane) - 1)));
      const unsigned c   = node_cnt[nd];
      uNode[pos]         = nd;
      uCnt[pos]          = c;
      uK32[pos]          = node_k32[nd];
      myMax              = (c > myMax) ? c : myMax;
    }
    U += unsigned(wv::popc(m));
  }
  wv::sync();
  if (U == 0) return 0;
  unsigned cStar = 1, pStar = 0xffffffffu;
  if (U > T) {
    // Both thresholds come from histograms in LDS (one pass over the compacted arrays each) instead of binary searches
    // that re-read the arrays once per probe.
    // (every launch of the general kernel carries ASM_LDS_BYTES of LDS per wave: 2 * HBINS bins fit)
    uint32_t*      hist = reinterpret_cast<uint32_t*>(wv::lds(P.lds_bytes));
    const unsigned cmax = waveMax(myMax);
    // ---- count level: largest c with #{cnt >= c} >= T.  Counts above HBINS-1 share the top bin (they are all taken
    // when the cut falls below it; if the cut falls inside the top bin the binary search below resolves it). ----
    static const unsigned HBINS = 1024;
    const unsigned        nb    = (cmax + 1 < HBINS) ? (cmax + 1) : HBINS;
    for (unsigned i = lane; i < nb; i += 64) hist[i] = 0;
    wv::sync();
    for (unsigned i = lane; i < U; i += 64) {
      const unsigned c = uCnt[i];
      wv::atomic_add(&hist[(c < nb) ? c : (nb - 1)], 1u);
    }
    wv::sync();
    {
      // walk the bins from the top in chunks of 64, lane 0 = highest bin of the chunk
      unsigned carry = 0, found = 0;
      for (unsigned top = ((nb + 63) / 64) * 64; top > 0 && !found; top -= 64) {
        const unsigned bin = top - 1 - lane;
        const unsigned h   = (bin < nb) ? hist[bin] : 0u;
        unsigned       inc = h;
        for (int off = 1; off < 64; off <<= 1) {
          const unsigned o = wv::shfl(inc, int(lane) - off);
          if (int(lane) >= off) inc += o;
        }
        const uint64_t reach = wv::ballot(carry + inc >= T && bin < nb);
        if (reach) {
          cStar = top - 1 - unsigned(wv::ctz(reach));
          found = 1;
        }
        carry += wv::shfl(inc, 63);
      }
      if (!found) cStar = 1;
    }
    wv::sync();
    if (cStar >= nb - 1 && cmax + 1 > HBINS) {  // the cut lies among the very large counts: exact search there (rare)
      unsigned lo = nb - 1, hi = cmax;
      while (lo < hi) {
        const unsigned mid = lo + (hi - lo + 1) / 2;
        unsigned       n   = 0;
        for (unsigned i = lane; i < U; i += 64) n += (uCnt[i] >= mid) ? 1u : 0u;
        if (waveSum(n) >= T)
          lo = mid;
        else
          hi = mid - 1;
      }
      cStar = lo;
    }
    unsigned above = 0;
    for (unsigned i = lane; i < U; i += 64) above += (uCnt[i] > cStar) ? 1u : 0u;
    unsigned need = T - waveSum(above);  // >= 1 words wanted from the tie level
    // ---- tie level: smallest prefix p with #{cnt == cStar, k32 <= p} >= need, by radix select (8 bits per pass) ----
    unsigned prefix = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
      for (unsigned i = lane; i < 256; i += 64) hist[i] = 0;
      wv::sync();
      for (unsigned i = lane; i < U; i += 64) {
        if (uCnt[i] != cStar) continue;
        const unsigned p = uK32[i];
        if (shift < 24 && (p >> (shift + 8)) != (prefix >> (shift + 8))) continue;
        wv::atomic_add(&hist[(p >> shift) & 255u], 1u);
      }
      wv::sync();
      // lane l owns bins 4l .. 4l+3
      const unsigned h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
      const unsigned mine = h0 + h1 + h2 + h3;
      unsigned       inc  = mine;
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = wv::shfl(inc, int(lane) - off);
        if (int(lane) >= off) inc += o;
      }
      const uint64_t reach = wv::ballot(inc >= need);
      const int      ln    = wv::ctz(reach);  // never empty: the level holds at least `need` words
      const unsigned before = wv::shfl(inc - mine, ln);
      const unsigned q0 = wv::shfl(h0, ln), q1 = wv::shfl(h1, ln), q2 = wv::shfl(h2, ln);
      unsigned       digit = 4u * unsigned(ln), acc = before;
      if (acc + q0 < need) {
        acc += q0;
        digit++;
        if (acc + q1 < need) {
          acc += q1;
          digit++;
          if (acc + q2 < need) {
            acc += q2;
            digit++;
          }
        }
      }
      prefix |= digit << shift;
      need -= acc;
      wv::sync();
    }
    pStar = prefix;
  }
  // gather (every word outside this list sorts after every word inside it)
  unsigned total = 0;
  for (unsigned base = 0; base < U; base += 64) {
    const unsigned i   = base + lane;
    bool           sel = false;
    if (i < U) {
      const unsigned c = uCnt[i];
      sel              = (U <= T) || (c > cStar) || (c == cStar && uK32[i] <= pStar);
    }
    const uint64_t m   = wv::ballot(sel);
    const unsigned pos = total + unsigned(wv::popc(m & ((uint64_t(1) << lane) - 1)));
    if (sel && pos < TENT_CAP) tent_raw[pos] = uNode[i];
    total += unsigned(wv::popc(m));
  }
  wv::sync();
  if (total > TENT_CAP) {
    // pathological tie group (hundreds of words sharing a 16-base prefix): take one exact seed the slow way
    const unsigned s = selectSeed();
    if (s == ASM_NONE) return 0;
    if (lane == 0) tent_sorted[0] = s;
    wv::sync();
    return 1;
  }
  // exact order inside the list: rank by counting (the list is ~T long); count, then 16-base prefix, then -- only
  // on a prefix tie -- the full k-mer
  for (unsigned i = lane; i < total; i += 64) {
    const unsigned x    = tent_raw[i];
    const unsigned cx   = node_cnt[x];
    const unsigned px   = node_k32[x];
    unsigned       rank = 0;
    for (unsigned j = 0; j < total; ++j) {
      if (j == i) continue;
      const unsigned y  = tent_raw[j];
      const unsigned cy = node_cnt[y];
      bool           before = (cy > cx);
      if (cy == cx) {
        const unsigned py = node_k32[y];
        before            = (py < px) || (py == px && keyLess(keyAt<KW>(node_key[y]), keyAt<KW>(node_key[x])));
      }
      if (before) rank++;
    }
    tent_sorted[rank] = x;
  }
  wv::sync();
  return (total < T) ? total : T;
}

/// The (<= 4) existing candidate words of a walk step, compacted in alphabet order, with everything the step needs:
/// per candidate three 16-byte blocks of its 64-byte record (links in walking direction + count, links against it,
/// read support) and the lane's visited-bitmap word.
/// candidate slots fetched ahead (further candidates, rare, are fetched on demand).  Wide read sets keep one slot: the
/// second one's support words pushed the step loop into scratch (63 spill accesses per step at WQ = 4).
template <int WQ>
struct WalkPre {
  static const unsigned N = (WQ <= 2) ? 2 : 1;
};
template <int WQ>
struct StepData {
  unsigned m;        // number of candidates
  unsigned node[4];  // compacted, alphabet order
  unsigned syms;     // their symbols, 2 bits each
  static const unsigned NPRE = WalkPre<WQ>::N;
  uint64_t flo[NPRE], fhi[NPRE], blo[NPRE], bhi[NPRE];
  uint64_t sup[NPRE][WQ];
  unsigned visWord[NPRE];
};

/// One round: lane t < nT walks tent_sorted[t] with private state (assembly/IterativeAssembler.cpp:149-501).
///
/// The per-CU vector-memory pipe, not HBM, bounds this loop (a gather instruction costs the same whether 1 or 64
/// lanes are live), so the step is built to issue as few memory instructions as possible:
///   * most words have ONE successor: the existing candidates are compacted per lane and candidate slot i is only
///     fetched if some lane has more than i candidates;
///   * a candidate costs three 16-byte loads from its single 64-byte record (packed links carry the count);
///   * the step is software-pipelined: right after the choice, the backward-check supports (:377-427) and the next
///     step's candidates are requested together, so a step is about one memory round trip;
///   * the lane-private visited bitmaps live in LDS when they fit.
/// All lanes execute the same instruction stream (finished lanes are predicated off), so the wave-level votes that
/// skip empty candidate slots are convergent.
template <int SB>
template <int WQ>
WV_DEV void AssemblerT<SB>::walkLanes(const unsigned nT)
{
  static const int KW = GEN_KW;
  const unsigned lane     = unsigned(wv::lane());
  const unsigned visWords = (P.cap_nodes + 31) / 32;
  const unsigned useWords = (nNodes + 31) / 32;
  const bool     visInLds = (nT * useWords * 4 <= P.lds_bytes);
  uint32_t*      visBase  = visInLds ? reinterpret_cast<uint32_t*>(wv::lds(P.lds_bytes)) : lane_vis;
  const unsigned visStride = visInLds ? useWords : visWords;
  for (unsigned i = lane; i < nT * useWords; i += 64) visBase[size_t(i / useWords) * visStride + (i % useWords)] = 0;
  wv::sync();

  const bool     has  = lane < nT;
  const unsigned seed = has ? tent_sorted[lane] : 0u;
  uint32_t*      vis  = visBase + size_t(has ? lane : 0) * visStride;
  // appended / prepended bases are collected 2 bits each and stored one dword per 16 bases
  const unsigned seqWords = P.max_contig_len / 16 + 2;
  uint32_t*      rightBuf = reinterpret_cast<uint32_t*>(lane_seq) + size_t(lane) * 2 * seqWords;
  uint32_t*      leftBuf  = rightBuf + seqWords;
  uint32_t       accR = 0, accL = 0;
  uint64_t       S[WQ], Rj[WQ];
  for (int w = 0; w < WQ; ++w) {
    S[w]  = (has && unsigned(w) < W) ? recSup(seed)[w] : 0;
    Rj[w] = 0;
  }
  bool     active = has, rep = false, tooLong = false, seedRepeat = false;
  unsigned mode = 0, cur = seed, consOffset = 0, nLeft = 0, nRight = 0;
  int      consEnd = 0, consBegin = 0;
  if (has) {
    vis[seed >> 5] |= (1u << (seed & 31));
    if (node_flag[seed] & NF_REPEAT) {  // :172-179
      seedRepeat = true;
      rep        = true;
      active     = false;
    } else {  // unselected siblings of the seed reject the contig (:185-210)
      const unsigned seedPb   = node_key[seed];
      const Key<KW>  key      = keyAt<KW>(seedPb);
      const unsigned lastBase = indexOfSym(symAt(seedPb + k - 1));
      for (unsigned c = 0; c < 4; ++c) {
        if (c == lastBase) continue;
        Key<KW> sib = key;
        keySetBase(sib, k - 1, c);
        const unsigned n = lookup<KW>(sib);
        if (n != ASM_NONE)
          for (int w = 0; w < WQ; ++w)
            if (unsigned(w) < W) Rj[w] |= recSup(n)[w];
      }
    }
  }

  // Fetch the step data for the candidates named by the packed link block (lo,hi); `on` predicates the lane.
  // fdir16 / bdir16: byte offset of the links in / against the walking direction inside a record (0 or 16).
  auto fetch = [&](const bool on, const uint64_t lo, const uint64_t hi, const unsigned fdir16, const unsigned bdir16,
                   StepData<WQ>& d) {
    unsigned ids[4], cntUnused;
    unpackLinks(lo, hi, ids, cntUnused);
    d.m    = 0;
    d.syms = 0;
    for (unsigned i = 0; i < 4; ++i) d.node[i] = ASM_NONE;
    if (on) {
      for (unsigned c = 0; c < 4; ++c) {  // stable compaction keeps the reference's A,C,G,T evaluation order
        if (ids[c] == ASM_NONE) continue;
        for (unsigned i = 0; i < 4; ++i)
          if (i == d.m) d.node[i] = ids[c];
        d.syms |= c << (2 * d.m);
        d.m++;
      }
    }
    for (unsigned i = 0; i < WalkPre<WQ>::N; ++i) {
      d.flo[i] = d.fhi[i] = d.blo[i] = d.bhi[i] = 0;
      d.visWord[i] = 0;
      for (int w = 0; w < WQ; ++w) d.sup[i][w] = 0;
      if (!wv::any(i < d.m)) continue;  // nobody has an (i+1)-th candidate: no memory instructions at all
      if (i < d.m) {
        const unsigned  n  = d.node[i];
        const uint64_t* f  = recPacked(n, fdir16);
        const uint64_t* b  = recPacked(n, bdir16);
        const uint64_t* sp = recSup(n);
        d.flo[i] = f[0];
        d.fhi[i] = f[1];
        d.blo[i] = b[0];
        d.bhi[i] = b[1];
        for (int w = 0; w < WQ; ++w)
          if (unsigned(w) < W) d.sup[i][w] = sp[w];
        d.visWord[i] = vis[n >> 5];
      }
    }
  };

  StepData<WQ> D;
  {
    uint64_t lo = 0, hi = 0;
    if (active) {
      lo = recPacked(cur, 0)[0];
      hi = recPacked(cur, 0)[1];
    }
    fetch(active, lo, hi, 0u, 16u, D);
  }

  while (wv::any(active)) {
    const bool     isEnd  = (mode == 0);
    const unsigned fdir16 = isEnd ? 0u : 16u, bdir16 = isEnd ? 16u : 0u;
    // ---- choose the extension among the candidates (:241-336) ----
    unsigned maxBaseCount = 0, maxCnt = 0, maxNode = ASM_NONE, maxSym = 0, maxVis = 0;
    uint64_t maxFlo = 0, maxFhi = 0, maxBlo = 0, maxBhi = 0;
    uint64_t maxWR[WQ], maxCW[WQ], rm[WQ], add[WQ];
    for (int w = 0; w < WQ; ++w) maxWR[w] = maxCW[w] = rm[w] = add[w] = 0;
    for (unsigned i = 0; i < 4; ++i) {
      const bool live = active && i < D.m;
      uint64_t   cflo, cfhi, cblo, cbhi, csup[WQ];
      unsigned   cvis;
      if (i < WalkPre<WQ>::N) {
        cflo = D.flo[i];
        cfhi = D.fhi[i];
        cblo = D.blo[i];
        cbhi = D.bhi[i];
        cvis = D.visWord[i];
        for (int w = 0; w < WQ; ++w) csup[w] = D.sup[i][w];
      } else {
        cflo = cfhi = cblo = cbhi = 0;
        cvis = 0;
        for (int w = 0; w < WQ; ++w) csup[w] = 0;
        if (!wv::any(live)) continue;  // a third / fourth candidate is rare: fetched only when some lane has one
        if (live) {
          const unsigned  n  = D.node[i];
          const uint64_t* f  = recPacked(n, fdir16);
          const uint64_t* b  = recPacked(n, bdir16);
          const uint64_t* sp = recSup(n);
          cflo = f[0];
          cfhi = f[1];
          cblo = b[0];
          cbhi = b[1];
          for (int w = 0; w < WQ; ++w)
            if (unsigned(w) < W) csup[w] = sp[w];
          cvis = vis[n >> 5];
        }
      }
      if (!live) continue;
      unsigned cnt = 0;
      for (int w = 0; w < WQ; ++w) cnt += unsigned(wv::popc(S[w] & csup[w]));
      if (cnt == 0) continue;  // :280
      if (cnt > maxCnt) {      // :283-316
        for (int w = 0; w < WQ; ++w) {
          const uint64_t SH = maxCW[w] & csup[w];
          rm[w] |= maxCW[w] & ~SH;
          add[w] |= maxWR[w] & ~SH;
          maxWR[w] = csup[w];
          maxCW[w] = S[w] & csup[w];
        }
        maxCnt  = cnt;
        maxSym  = (D.syms >> (2 * i)) & 3;
        maxNode = D.node[i];
        maxFlo  = cflo;
        maxFhi  = cfhi;
        maxBlo  = cblo;
        maxBhi  = cbhi;
        maxVis  = cvis;
      } else {  // :317-335
        for (int w = 0; w < WQ; ++w) {
          const uint64_t SH = maxCW[w] & csup[w];
          rm[w] |= (S[w] & csup[w]) & ~SH;
          add[w] |= csup[w] & ~SH;
        }
      }
    }
    if (maxNode != ASM_NONE) {
      maxBaseCount = unsigned(maxFhi >> 21) & LINK_CNT_MAX;
      if (maxBaseCount == LINK_CNT_MAX) maxBaseCount = node_cnt[maxNode];  // saturated: the exact count is in the SoA copy
    }
    bool stop = false, extend = false;
    if (active) {
      if (maxBaseCount < P.opt.minCoverage) {  // :343 (also "no candidate")
        stop = true;
      } else if (maxVis & (1u << (maxNode & 31))) {  // :352-358
        rep  = true;
        stop = true;
      } else if (k + nRight + nLeft + 1 >= P.max_contig_len) {
        tooLong = true;
        active  = false;
      } else {
        extend = true;
      }
    }
    // ---- requests: backward-check supports of the chosen word (:377-427) ... ----
    unsigned bIds[4], bCntUnused;
    unpackLinks(maxBlo, maxBhi, bIds, bCntUnused);
    unsigned bn[3] = {ASM_NONE, ASM_NONE, ASM_NONE};
    unsigned bm    = 0;
    if (extend) {
      for (unsigned c = 0; c < 4; ++c) {
        const unsigned n = bIds[c];
        if (n == cur || n == maxNode || n == ASM_NONE) continue;  // :381, :389
        for (unsigned i = 0; i < 3; ++i)
          if (i == bm) bn[i] = n;
        bm++;
      }
    }
    uint64_t bsup[3][WQ];
    for (unsigned i = 0; i < 3; ++i) {
      for (int w = 0; w < WQ; ++w) bsup[i][w] = 0;
      if (!wv::any(i < bm)) continue;
      if (i < bm) {
        const uint64_t* sp = recSup(bn[i]);
        for (int w = 0; w < WQ; ++w)
          if (unsigned(w) < W) bsup[i][w] = sp[w];
      }
    }
    // ---- ... and the next step's candidates (or, on a direction switch, the seed's predecessors) ----
    if (extend) vis[maxNode >> 5] = maxVis | (1u << (maxNode & 31));  // :482-484, before the next fetch reads the bitmap
    const bool toLeft = stop && (mode == 0);  // :488-491
    uint64_t   nlo = maxFlo, nhi = maxFhi;
    unsigned   nf = fdir16, nb = bdir16;
    if (toLeft) {
      nlo = recPacked(seed, 16)[0];
      nhi = recPacked(seed, 16)[1];
      nf  = 16u;
      nb  = 0u;
    }
    StepData<WQ> N;
    fetch(extend || toLeft, nlo, nhi, nf, nb, N);

    // ---- finish this step ----
    if (extend) {
      if (isEnd) {  // :363
        accR |= maxSym << (2 * (nRight & 15));
        if ((nRight & 15) == 15) {
          rightBuf[nRight >> 4] = accR;
          accR                  = 0;
        }
        nRight++;
      } else {
        accL |= maxSym << (2 * (nLeft & 15));
        if ((nLeft & 15) == 15) {
          leftBuf[nLeft >> 4] = accL;
          accL                = 0;
        }
        nLeft++;
      }
      if ((consOffset != 0) || (maxBaseCount < P.opt.minConservativeCoverage)) consOffset += 1;  // :368-369
      for (unsigned i = 0; i < 3; ++i) {
        for (int w = 0; w < WQ; ++w) {
          const uint64_t upd = bsup[i][w] & ~maxCW[w];  // :400-414
          add[w] |= upd;
          rm[w] |= upd;
        }
      }
      for (int w = 0; w < WQ; ++w) {
        Rj[w] |= add[w];             // :440-442
        S[w] |= maxWR[w] & ~Rj[w];   // :458-464
        S[w] &= ~rm[w];              // :471-473
      }
      cur = maxNode;
    }
    if (stop) {
      if (mode == 0) {
        consEnd    = int(consOffset);
        mode       = 1;
        cur        = seed;
        consOffset = 0;
      } else {
        consBegin = int(consOffset);
        active    = false;
      }
    }
    D = N;
  }

  if (has) {
    for (int w = 0; w < WQ; ++w) {
      lane_bits[size_t(lane) * 2 * WQ_MAX + w]          = S[w];
      lane_bits[size_t(lane) * 2 * WQ_MAX + WQ_MAX + w] = Rj[w];
    }
    if (nRight & 15) rightBuf[nRight >> 4] = accR;
    if (nLeft & 15) leftBuf[nLeft >> 4] = accL;
    int32_t* m = lane_meta + lane * 8;
    m[0]       = int(nLeft);
    m[1]       = int(nRight);
    m[2]       = consBegin;
    m[3]       = consEnd;
    m[4]       = rep ? 1 : 0;
    m[5]       = tooLong ? 1 : 0;
    m[6]       = seedRepeat ? 1 : 0;
    // (the visited bitmaps stay where they are -- LDS or lane_vis -- and are read there during acceptance)
  }
  wv::sync();
}

/// buildContigs' contig loop (:685-713) by speculative rounds.  Returns isAssemblySuccess.
template <int SB>
template <int WQ>
WV_DEV bool AssemblerT<SB>::contigRounds()
{
  const unsigned lane     = unsigned(wv::lane());
  const unsigned capCand  = 2 * P.opt.maxAssemblyCount;
  const unsigned visWords = (P.cap_nodes + 31) / 32;
  const unsigned useWords = (nNodes + 31) / 32;
  bool           success  = true;
  nCand                   = 0;
  while (nCand < capCand) {
    // first round: the top seed alone (its walk consumes the main path, which would invalidate most of a wide
    // first round -- measured: a 32-wide first round is 3 % slower)
    // later rounds: as many tentative seeds as keep the lane-private visited bitmaps in LDS (idle lanes cost nothing: a
    // gather is priced per instruction, not per live lane), and at least what is still needed
    unsigned T = 1;
    if (nCand != 0) {
      const unsigned fit = P.lds_bytes / (useWords * 4 + 4);
      T                  = (fit > (capCand - nCand) + 2) ? fit : (capCand - nCand) + 2;
    }
    if (T > 64) T = 64;
    const unsigned nT = selectTentative(T);
    tick(5);
    if (nT == 0) break;
    walkLanes<WQ>(nT);
    tick(6);
    for (unsigned t = 0; t < nT && nCand < capCand; ++t) {
      const unsigned seed = tent_sorted[t];
      unsigned seedFree = 0;  // read by ONE lane before this iteration's updates, then broadcast
      if (lane == 0) seedFree = isUnused(seed) ? 1u : 0u;
      seedFree = wv::readlane(seedFree, 0);
      if (!seedFree) continue;  // consumed by an accepted walk: not a seed for the reference either
      const int32_t* m = lane_meta + t * 8;
      if (m[5]) {
        status = ASM_E_CONTIG_TOO_LONG;
        return true;
      }
      const unsigned nLeft = unsigned(m[0]), nRight = unsigned(m[1]);
      const unsigned len    = nLeft + k + nRight;
      const unsigned seedPb = node_key[seed];
      uint8_t*       outSeq = cand_seq + size_t(nCand) * P.max_contig_len;
      const unsigned  seqWords = P.max_contig_len / 16 + 2;
      const uint32_t* rightBuf = reinterpret_cast<const uint32_t*>(lane_seq) + size_t(t) * 2 * seqWords;
      const uint32_t* leftBuf  = rightBuf + seqWords;
      for (unsigned i = lane; i < len; i += 64) {
        uint8_t ch;  // (the walked bases are alphabet symbols, 2 bits each; the seed's text comes from the pile)
        if (i < nLeft) {
          const unsigned j = nLeft - 1 - i;
          ch               = uint8_t("ACGT"[(leftBuf[j >> 4] >> (2 * (j & 15))) & 3]);
        } else if (i < nLeft + k) {
          ch = charOfSym(symAt(seedPb + (i - nLeft)));
        } else {
          const unsigned j = i - nLeft - k;
          ch               = uint8_t("ACGT"[(rightBuf[j >> 4] >> (2 * (j & 15))) & 3]);
        }
        outSeq[i] = ch;
      }
      if (lane < 2 * W) {
        const unsigned half = lane / W, w = lane % W;
        cand_bits[size_t(nCand) * 2 * W + lane] = (w < unsigned(WQ)) ? lane_bits[size_t(t) * 2 * WQ_MAX + half * WQ_MAX + w] : 0;
      }
      if (lane == 0) {
        int32_t* meta = cand_meta + nCand * 4;
        meta[0]       = int(len);
        if (m[6]) {  // seed is a repeat word (:172-179)
          meta[1] = 0;
          meta[2] = int(k);
        } else {
          meta[1] = m[2];
          meta[2] = int(len) - m[3];  // :498
        }
      }
      // unusedWords.erase for every word of the accepted walk (:170,482)
      // the walks' visited bitmaps are still where walkLanes kept them (LDS if they fitted)
      const bool      visInLds = (nT * useWords * 4 <= P.lds_bytes);
      const uint32_t* vis      = visInLds ? (reinterpret_cast<const uint32_t*>(wv::lds(P.lds_bytes)) + size_t(t) * useWords)
                                          : (lane_vis + size_t(t) * visWords);
      for (unsigned w = lane; w < useWords; w += 64) unused_bits[w] &= ~vis[w];
      wv::sync();
      if (m[4]) success = false;
      nCand++;
    }
    tick(7);
  }
  return success;
}

}  // namespace manta_dev
