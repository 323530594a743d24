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
template <int KW>
WV_DEV unsigned Assembler::selectTentative(const unsigned T)
{
  const unsigned lane = unsigned(wv::lane());
  // unused population and its largest count
  unsigned myU = 0, myMax = 0;
  for (unsigned nd = lane; nd < nNodes; nd += 64) {
    if (node_flag[nd] & NF_UNUSED) {
      myU++;
      const unsigned c = node_cnt[nd];
      myMax            = (c > myMax) ? c : myMax;
    }
  }
  const unsigned U = waveSum(myU);
  if (U == 0) return 0;
  unsigned cStar = 1, pStar = 0xffffffffu;
  if (U > T) {
    const unsigned cmax = waveMax(myMax);
    // largest c with #{unused, cnt >= c} >= T
    unsigned lo = 1, hi = cmax;
    while (lo < hi) {
      const unsigned mid = lo + (hi - lo + 1) / 2;
      unsigned       n   = 0;
      for (unsigned nd = lane; nd < nNodes; nd += 64)
        if ((node_flag[nd] & NF_UNUSED) && node_cnt[nd] >= mid) n++;
      if (waveSum(n) >= T)
        lo = mid;
      else
        hi = mid - 1;
    }
    cStar = lo;
    unsigned above = 0;
    for (unsigned nd = lane; nd < nNodes; nd += 64)
      if ((node_flag[nd] & NF_UNUSED) && node_cnt[nd] > cStar) above++;
    const unsigned need = T - waveSum(above);  // >= 1 words wanted from the tie level
    // smallest 16-base prefix p with #{unused, cnt == cStar, k32 <= p} >= need
    unsigned plo = 0, phi = 0xffffffffu;
    while (plo < phi) {
      const unsigned mid = plo + (phi - plo) / 2;
      unsigned       n   = 0;
      for (unsigned nd = lane; nd < nNodes; nd += 64)
        if ((node_flag[nd] & NF_UNUSED) && node_cnt[nd] == cStar && node_k32[nd] <= mid) n++;
      if (waveSum(n) >= need)
        phi = mid;
      else
        plo = mid + 1;
    }
    pStar = plo;
  }
  // gather (every word outside this list sorts after every word inside it)
  unsigned total = 0;
  for (unsigned base = 0; base < nNodes; base += 64) {
    const unsigned nd  = base + lane;
    bool           sel = false;
    if (nd < nNodes && (node_flag[nd] & NF_UNUSED)) {
      const unsigned c = node_cnt[nd];
      sel              = (U <= T) || (c > cStar) || (c == cStar && node_k32[nd] <= pStar);
    }
    const uint64_t m   = wv::ballot(sel);
    const unsigned pos = total + unsigned(wv::popc(m & ((uint64_t(1) << lane) - 1)));
    if (sel && pos < TENT_CAP) tent_raw[pos] = nd;
    total += unsigned(wv::popc(m));
  }
  wv::sync();
  if (total > TENT_CAP) {
    // pathological tie group (hundreds of words sharing a 16-base prefix): take one exact seed the slow way
    const unsigned s = selectSeed<KW>();
    if (s == ASM_NONE) return 0;
    if (lane == 0) tent_sorted[0] = s;
    wv::sync();
    return 1;
  }
  // exact order inside the list: rank by counting (the list is ~T long)
  for (unsigned i = lane; i < total; i += 64) {
    const unsigned x   = tent_raw[i];
    const unsigned cx  = node_cnt[x];
    const Key<KW>  kx  = keyAt<KW>(node_key[x]);
    unsigned       rank = 0;
    for (unsigned j = 0; j < total; ++j) {
      if (j == i) continue;
      const unsigned y  = tent_raw[j];
      const unsigned cy = node_cnt[y];
      if (cy > cx || (cy == cx && keyLess(keyAt<KW>(node_key[y]), kx))) rank++;
    }
    tent_sorted[rank] = x;
  }
  wv::sync();
  return (total < T) ? total : T;
}

/// One round: lane t < nT walks tent_sorted[t] with private state (assembly/IterativeAssembler.cpp:149-501).
template <int KW, int WQ>
WV_DEV void Assembler::walkLanes(const unsigned nT)
{
  const unsigned lane     = unsigned(wv::lane());
  const unsigned visWords = (P.cap_nodes + 31) / 32;
  const unsigned useWords = (nNodes + 31) / 32;
  for (unsigned i = lane; i < nT * useWords; i += 64) lane_vis[size_t(i / useWords) * visWords + (i % useWords)] = 0;
  wv::sync();

  const bool     has  = lane < nT;
  const unsigned seed = has ? tent_sorted[lane] : 0u;
  uint32_t*      vis  = lane_vis + size_t(lane) * visWords;
  uint8_t*       rightBuf = lane_seq + size_t(lane) * 2 * P.max_contig_len;
  uint8_t*       leftBuf  = rightBuf + P.max_contig_len;
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
      const unsigned lastBase = (codes[(seedPb + k - 1) >> 4] >> (30 - 2 * ((seedPb + k - 1) & 15))) & 3;
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

  while (wv::any(active)) {
    if (!active) continue;
    const bool     isEnd  = (mode == 0);
    const unsigned fwdOff = isEnd ? 0u : 4u, bwdOff = isEnd ? 4u : 0u;  // succ[4] | pred[4] inside a node record
    // the four candidate words: one 16-byte load of the current word's links, then per candidate its
    // {count} and {support} 16-byte blocks -- all inside that candidate's single 64-byte record (W <= 2)
    const u32x4    links = *reinterpret_cast<const u32x4*>(recSucc(cur) + fwdOff);
    const unsigned cand[4] = {links.x, links.y, links.z, links.w};
    uint64_t       cw[4][WQ];
    unsigned       ccount[4], cnt[4];
    for (unsigned c = 0; c < 4; ++c) {
      ccount[c] = 0;
      cnt[c]    = 0;
      for (int w = 0; w < WQ; ++w) cw[c][w] = 0;
      if (cand[c] != ASM_NONE) {
        ccount[c]          = recCnt(cand[c]);
        const uint64_t* sp = recSup(cand[c]);
        for (int w = 0; w < WQ; ++w) {
          if (unsigned(w) < W) {
            cw[c][w] = sp[w];
            cnt[c] += unsigned(wv::popc(S[w] & cw[c][w]));
          }
        }
      }
    }
    unsigned maxBaseCount = 0, maxCnt = 0, maxNode = ASM_NONE, maxSym = 0;
    uint64_t maxWR[WQ], maxCW[WQ], rm[WQ], add[WQ];
    for (int w = 0; w < WQ; ++w) maxWR[w] = maxCW[w] = rm[w] = add[w] = 0;
    for (unsigned c = 0; c < 4; ++c) {  // :241-336
      if (cand[c] == ASM_NONE || cnt[c] == 0) continue;
      if (cnt[c] > maxCnt) {
        for (int w = 0; w < WQ; ++w) {
          const uint64_t SH = maxCW[w] & cw[c][w];
          rm[w] |= maxCW[w] & ~SH;
          add[w] |= maxWR[w] & ~SH;
          maxWR[w] = cw[c][w];
          maxCW[w] = S[w] & cw[c][w];
        }
        maxCnt       = cnt[c];
        maxBaseCount = ccount[c];
        maxSym       = c;
        maxNode      = cand[c];
      } else {
        for (int w = 0; w < WQ; ++w) {
          const uint64_t SH = maxCW[w] & cw[c][w];
          rm[w] |= (S[w] & cw[c][w]) & ~SH;
          add[w] |= cw[c][w] & ~SH;
        }
      }
    }
    bool stop = false;
    if (maxBaseCount < P.opt.minCoverage) {  // :343
      stop = true;
    } else if (vis[maxNode >> 5] & (1u << (maxNode & 31))) {  // :352-358
      rep  = true;
      stop = true;
    } else if (k + nRight + nLeft + 1 >= P.max_contig_len) {
      tooLong = true;
      active  = false;
      continue;
    } else {
      if (isEnd)
        rightBuf[nRight++] = uint8_t("ACGT"[maxSym]);  // :363
      else
        leftBuf[nLeft++] = uint8_t("ACGT"[maxSym]);
      if ((consOffset != 0) || (maxBaseCount < P.opt.minConservativeCoverage)) consOffset += 1;  // :368-369
      const u32x4    blinks = *reinterpret_cast<const u32x4*>(recSucc(maxNode) + bwdOff);
      const unsigned bnode[4] = {blinks.x, blinks.y, blinks.z, blinks.w};
      for (unsigned c = 0; c < 4; ++c) {  // one step backwards at the branching point (:377-427)
        const unsigned n = bnode[c];
        if (n == cur || n == maxNode || n == ASM_NONE) continue;
        const uint64_t* sp = recSup(n);
        for (int w = 0; w < WQ; ++w) {
          if (unsigned(w) < W) {
            const uint64_t upd = sp[w] & ~maxCW[w];
            add[w] |= upd;
            rm[w] |= upd;
          }
        }
      }
      for (int w = 0; w < WQ; ++w) {
        Rj[w] |= add[w];             // :440-442
        S[w] |= maxWR[w] & ~Rj[w];   // :458-464
        S[w] &= ~rm[w];              // :471-473
      }
      vis[maxNode >> 5] |= (1u << (maxNode & 31));  // :482-484
      cur = maxNode;
    }
    if (stop) {
      if (mode == 0) {  // :488-491
        consEnd    = int(consOffset);
        mode       = 1;
        cur        = seed;
        consOffset = 0;
      } else {
        consBegin = int(consOffset);
        active    = false;
      }
    }
  }

  if (has) {
    for (int w = 0; w < WQ; ++w) {
      lane_bits[size_t(lane) * 2 * WQ_MAX + w]          = S[w];
      lane_bits[size_t(lane) * 2 * WQ_MAX + WQ_MAX + w] = Rj[w];
    }
    int32_t* m = lane_meta + lane * 8;
    m[0]       = int(nLeft);
    m[1]       = int(nRight);
    m[2]       = consBegin;
    m[3]       = consEnd;
    m[4]       = rep ? 1 : 0;
    m[5]       = tooLong ? 1 : 0;
    m[6]       = seedRepeat ? 1 : 0;
  }
  wv::sync();
}

/// buildContigs' contig loop (:685-713) by speculative rounds.  Returns isAssemblySuccess.
template <int KW, int WQ>
WV_DEV bool Assembler::contigRounds()
{
  const unsigned lane     = unsigned(wv::lane());
  const unsigned capCand  = 2 * P.opt.maxAssemblyCount;
  const unsigned visWords = (P.cap_nodes + 31) / 32;
  const unsigned useWords = (nNodes + 31) / 32;
  bool           success  = true;
  nCand                   = 0;
  while (nCand < capCand) {
    // first round: the top seed alone (its walk consumes the main path, which would invalidate most of a wide
    // first round); later rounds: what is still needed plus a margin for invalidated tentative seeds
    unsigned T = (nCand == 0) ? 1u : (capCand - nCand) + (capCand - nCand) / 4 + 2;
    if (T > 64) T = 64;
    const unsigned nT = selectTentative<KW>(T);
    tick(5);
    if (nT == 0) break;
    walkLanes<KW, WQ>(nT);
    tick(6);
    for (unsigned t = 0; t < nT && nCand < capCand; ++t) {
      const unsigned seed = tent_sorted[t];
      unsigned seedFlag = 0;  // read by ONE lane before this iteration's flag updates, then broadcast
      if (lane == 0) seedFlag = node_flag[seed];
      seedFlag = wv::readlane(seedFlag, 0);
      if (!(seedFlag & NF_UNUSED)) continue;  // consumed by an accepted walk: not a seed for the reference either
      const int32_t* m = lane_meta + t * 8;
      if (m[5]) {
        status = ASM_E_CONTIG_TOO_LONG;
        return true;
      }
      const unsigned nLeft = unsigned(m[0]), nRight = unsigned(m[1]);
      const unsigned len    = nLeft + k + nRight;
      const unsigned seedPb = node_key[seed];
      uint8_t*       outSeq = cand_seq + size_t(nCand) * P.max_contig_len;
      const uint8_t* rightBuf = lane_seq + size_t(t) * 2 * P.max_contig_len;
      const uint8_t* leftBuf  = rightBuf + P.max_contig_len;
      for (unsigned i = lane; i < len; i += 64) {
        uint8_t ch;
        if (i < nLeft) {
          ch = leftBuf[nLeft - 1 - i];
        } else if (i < nLeft + k) {
          const unsigned pb = seedPb + (i - nLeft);
          ch                = uint8_t("ACGT"[(codes[pb >> 4] >> (30 - 2 * (pb & 15))) & 3]);
        } else {
          ch = rightBuf[i - nLeft - k];
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
      const uint32_t* vis = lane_vis + size_t(t) * visWords;
      for (unsigned w = lane; w < useWords; w += 64) {
        uint32_t bits = vis[w];
        while (bits) {
          const unsigned b = unsigned(wv::ctz(uint64_t(bits)));
          bits &= bits - 1;
          node_flag[w * 32 + b] &= ~NF_UNUSED;
        }
      }
      wv::sync();
      if (m[4]) success = false;
      nCand++;
    }
    tick(7);
  }
  return success;
}

}  // namespace manta_dev
