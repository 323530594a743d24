// HIP device code (gfx950) for Manta's SmallAssembler
//   runSmallAssembler   assembly/SmallAssembler.cpp:622-685   (iterations over the still unused reads)
//     buildContigs      :465-620   (k-mer pass, most frequent words as seeds, longest walk wins, reads -> used)
//     getKmerCounts     :396-455   (per-read word sets; a word twice in one read = "repeat read")
//     walk              :143-391   (greedy extension with supporting / rejecting read sets)
// (paths relative to /root/reference/src/c++/lib).  The reference keeps this assembler next to IterativeAssembler but has no
// production caller for it; it is built here because BASELINE.json's north_star names it.
//
// Mapping: one wavefront per read pile, on the same per-wave slab as assemble_kernel.  The k-mer pass is the
// Assembler's fused table pass (buildGraph<KW, SMALL = true>): it skips used reads and reports repeat reads; what it leaves
// behind -- dense node ids, per-node read sets (the reference's wordSupportReads), counts (wordCount) and the
// successor / predecessor links -- is everything the rest needs:
//   * "trunk + symbol" lookups of the walk are successor (right walk) / predecessor (left walk) links;
//   * the backward step at a branch ("symbol + trunk", :322-343) reads the OTHER predecessors (successors) of the word
//     that was just added;
//   * read sets live one qword per lane (lane w < W holds reads 64w .. 64w+63), set algebra is lane-wise;
//   * seenVertexBefore (a set of (k-1)-mers, :196) is a small hash set of packed-base positions in the slab, compared by
//     content; seenEdgeBefore / the candidate seed set are bitmaps over node ids;
//   * std::set<std::string> order of the seeds (:520-555) = lexicographic order of the 2-bit keys.
// Output: the Assembler's records (AsmLocusOut / AsmContigOut, contig text + support + reject bitsets); the reads the
// reference marks isFiltered (repeat reads at the last word length, :496-503) travel as one extra record after the
// contigs (seq_len 0, reserved = 0xffffffff) whose support bitset holds them.
#pragma once
#include "assemble_kernels.hpp"

namespace manta_dev {

static const unsigned SMALL_FILTERED_MARK = 0xffffffffu;

struct SmallAsm {
  static const int KW = ASM_MAX_KW;  // generic key width outside the table pass (nothing here is hot)
  Assembler&       A;
  const AsmParams& P;
  unsigned         W = 1;
  unsigned         lane;
  uint64_t         usedW = 0, filteredW = 0;  // this lane's qword of the read sets (lane < W)
  unsigned         unusedReads = 0, nContigs = 0;
  uint32_t *       aliveBits, *edgeBits, *trunkSet;
  unsigned         trunkMask = 0;

  WV_DEV SmallAsm(Assembler& a) : A(a), P(a.P), lane(unsigned(wv::lane()))
  {
    aliveBits = A.lane_vis;                                  // candidate seeds (maxWords, :520-535)
    edgeBits  = A.lane_vis + (P.cap_nodes + 31) / 32 + 1;    // seenEdgeBefore of the current walk (:187)
    trunkSet  = A.exact_ws;                                  // seenVertexBefore (:196)
  }

  WV_DEV unsigned popSum(const uint64_t v) const { return Assembler::waveSum(lane < W ? unsigned(wv::popc(v)) : 0u); }
  WV_DEV bool     anyBit(const uint64_t v) const { return wv::any(lane < W && v != 0); }

  WV_DEV bool testBit(const uint32_t* bits, const unsigned i) const { return (bits[i >> 5] >> (i & 31)) & 1u; }

  // ---- (k-1)-mers by content -------------------------------------------------------------------
  /// the `len` bases at packed position pb
  WV_DEV Key<KW> keyAtLen(const unsigned pb, const unsigned len) const
  {
    Key<KW>        key;
    const unsigned kw = (len + 15) >> 4;
    const unsigned wi = pb >> 4, sh = (pb & 15) * 2;
    for (int i = 0; i < KW; ++i) {
      uint32_t v = 0;
      if (unsigned(i) < kw) {
        const uint32_t a = A.codes[wi + i], b = A.codes[wi + i + 1];
        v                = sh ? ((a << sh) | (b >> (32 - sh))) : a;
        const unsigned have = len - 16u * unsigned(i);
        if (have < 16) v &= ~((1u << (32 - 2 * have)) - 1u);
      }
      key.w[i] = v;
    }
    return key;
  }
  WV_DEV static uint32_t hashKey(const Key<KW>& key)
  {
    uint32_t h = 0x811C9DC5u;
    for (int i = 0; i < KW; ++i) h = hashMix(h, key.w[i]);
    return h ^ (h >> 13);
  }
  /// seenVertexBefore.count(trunk) / insert(trunk); the trunk is the (k-1)-mer at packed position pb.  Wave-uniform.
  WV_DEV bool trunkSeenOrInsert(const unsigned pb)
  {
    const Key<KW> key = keyAtLen(pb, A.k - 1);
    unsigned      s   = hashKey(key) & trunkMask;
    while (true) {
      const uint32_t cur = wv::first(trunkSet[s]);  // (every lane has read the slot before lane 0 may write it)
      if (cur == ASM_NONE) {
        if (lane == 0) trunkSet[s] = pb;
        wv::sync();
        return false;
      }
      if (Assembler::keyEq(keyAtLen(cur, A.k - 1), key)) return true;
      s = (s + 1) & trunkMask;
    }
  }

  WV_DEV unsigned baseAt(const unsigned pb) const { return (A.codes[pb >> 4] >> (30 - 2 * (pb & 15))) & 3u; }

  /// node a's word < node b's word (std::string order == order of the 2-bit keys, A<C<G<T)
  WV_DEV bool wordLess(const unsigned a, const unsigned b) const
  {
    return Assembler::keyLess(A.keyAt<KW>(A.node_key[a]), A.keyAt<KW>(A.node_key[b]));
  }

  // ---- walk (:143-391) ---------------------------------------------------------------------------
  /// extends `seed` in both directions into candidate slot `slotIdx`; returns false on a workspace limit (status set)
  WV_DEV_COLD bool walk(const unsigned seed, const unsigned slotIdx)
  {
    const unsigned k       = A.k;
    uint8_t*       outSeq  = A.cand_seq + size_t(slotIdx) * P.max_contig_len;
    uint64_t*      outBits = A.cand_bits + size_t(slotIdx) * 2 * W;
    int32_t*       meta    = A.cand_meta + slotIdx * 4;
    const unsigned seedPb  = A.node_key[seed];

    uint64_t S  = A.supWord(seed);  // contig.supportReads (:158)
    uint64_t Rj = 0;                // contig.rejectReads
    // reads of the seed's siblings (same first k-1 bases, other last base) reject it (:162-185)
    {
      const Key<KW>  key      = A.keyAt<KW>(seedPb);
      const unsigned lastBase = baseAt(seedPb + k - 1);
      for (unsigned c = 0; c < 4; ++c) {
        if (c == lastBase) continue;
        Key<KW> sib = key;
        A.keySetBase(sib, k - 1, c);
        Rj |= A.supWord(A.lookup<KW>(sib));
      }
    }
    // seenEdgeBefore = {seed}; seenVertexBefore = {}
    const unsigned nEdgeWords = (A.nNodes + 31) / 32;
    for (unsigned i = lane; i < nEdgeWords; i += 64) edgeBits[i] = 0;
    for (unsigned i = lane; i <= trunkMask; i += 64) trunkSet[i] = ASM_NONE;
    wv::sync();
    if (lane == 0) edgeBits[seed >> 5] |= 1u << (seed & 31);
    wv::sync();

    unsigned nLeft = 0, nRight = 0;
    int      consEnd = 0, consBegin = 0;
    for (unsigned mode = 0; mode < 2; ++mode) {
      const bool     isEnd  = (mode == 0);
      const unsigned fwdOff = isEnd ? 0u : 4u, bwdOff = isEnd ? 4u : 0u;  // succ[4] | pred[4] of a node
      unsigned       consOffset = 0;
      unsigned       cur        = seed;  // the word at the growing end
      while (true) {
        // trunk = the k-1 bases at the growing end: suffix of `cur` walking right, prefix walking left (:201-202)
        const unsigned trunkPb = A.node_key[cur] + (isEnd ? 1u : 0u);
        if (trunkSeenOrInsert(trunkPb)) break;  // :212-219

        unsigned maxBaseCount = 0, maxSharedCount = 0, maxNode = ASM_NONE, maxSym = 0;
        uint64_t maxWordReads = 0, maxShared = 0, remove2 = 0, rejectAdd = 0;
        for (unsigned c = 0; c < 4; ++c) {  // :230-282
          const unsigned n = A.recSucc(cur)[fwdOff + c];
          if (n == ASM_NONE) continue;
          const uint64_t wr     = A.supWord(n);
          const uint64_t shared = S & wr;
          const unsigned cnt    = popSum(shared);
          if (cnt == 0) continue;  // :259
          if (cnt > maxSharedCount) {
            remove2 |= maxShared;        // :265-266
            rejectAdd |= maxWordReads;   // :269
            maxWordReads   = wr;
            maxSharedCount = cnt;
            maxShared      = shared;
            maxBaseCount   = A.node_cnt[n];
            maxSym         = c;
            maxNode        = n;
          } else {
            remove2 |= shared;  // :277-278
            rejectAdd |= wr;
          }
        }
        if (maxBaseCount < P.opt.minCoverage) break;  // :289
        if (maxBaseCount == 0) break;                 // :298
        if (lane == 0) edgeBits[maxNode >> 5] |= 1u << (maxNode & 31);  // seenEdgeBefore.insert (:301-303)
        if (k + nLeft + nRight >= P.max_contig_len) {
          A.status = ASM_E_CONTIG_TOO_LONG;
          return false;
        }
        if (isEnd) {
          if (lane == 0) A.walk_right[nRight] = uint8_t("ACGT"[maxSym]);
          nRight++;
        } else {
          if (lane == 0) A.walk_left[nLeft] = uint8_t("ACGT"[maxSym]);
          nLeft++;
        }
        if ((consOffset != 0) || (maxBaseCount < P.opt.minConservativeCoverage)) consOffset += 1;  // :309-311

        // one step backwards at the branching point (:319-345).  previousWordReads is declared inside the loop body
        // (:228), so it is empty at the test and the step is taken whenever a word was chosen.
        if (anyBit(maxWordReads)) {
          for (unsigned c = 0; c < 4; ++c) {
            const unsigned n = A.recSucc(maxNode)[bwdOff + c];
            if (n == cur) continue;  // the selected branch (same trunk, the symbol of previousWord, :324)
            if (n == ASM_NONE) continue;
            rejectAdd |= A.supWord(n);
          }
        }
        Rj |= rejectAdd;            // :359-361
        S |= maxWordReads & ~Rj;    // :374-379
        S &= ~remove2;              // :387-389
        cur = maxNode;
      }
      if (isEnd)
        consEnd = int(consOffset);  // :393-397
      else
        consBegin = int(consOffset);
    }
    wv::sync();
    const unsigned len = nLeft + k + nRight;
    for (unsigned i = lane; i < len; i += 64) {
      uint8_t ch;
      if (i < nLeft)
        ch = A.walk_left[nLeft - 1 - i];
      else if (i < nLeft + k)
        ch = uint8_t("ACGT"[baseAt(seedPb + (i - nLeft))]);
      else
        ch = A.walk_right[i - nLeft - k];
      outSeq[i] = ch;
    }
    if (lane < W) {
      outBits[lane]     = S;
      outBits[W + lane] = Rj;
    }
    if (lane == 0) {
      meta[0] = int(len);
      meta[1] = consBegin;
      meta[2] = int(len) - consEnd;  // :403
      meta[3] = 0;
    }
    wv::sync();
    return true;
  }

  WV_DEV void graphForK()
  {
    const unsigned kw = (A.k + 15) >> 4;
    if (kw <= 2)
      A.buildGraph<2, true>();
    else if (kw <= 4)
      A.buildGraph<4, true>();
    else
      A.buildGraph<8, true>();
  }

  // ---- buildContigs (:465-620) -------------------------------------------------------------------
  WV_DEV_COLD bool buildContigs(const bool isLastWord)
  {
    const uint64_t validW = (lane < W) ? A.normalMask(lane) : 0;
    if (lane < ASM_MAX_W) A.small_active[lane] = validW & ~usedW;
    wv::sync();
    graphForK();
    if (A.status != ASM_OK) return false;
    const uint64_t repeatW = (lane < W) ? A.small_repeat[lane] : 0;
    if (anyBit(repeatW)) {  // !isGoodKmerCount (:494-505)
      if (isLastWord) {
        filteredW |= repeatW;
        usedW |= repeatW;
        unusedReads -= popSum(repeatW);
      }
      return false;
    }
    const unsigned nNodes = A.nNodes;
    // the most frequent words (:508-535)
    unsigned maxCount = 0;
    for (unsigned nd = lane; nd < nNodes; nd += 64) {
      const unsigned c = A.node_cnt[nd];
      maxCount         = (c > maxCount) ? c : maxCount;
    }
    maxCount = Assembler::waveMax(maxCount);
    if (maxCount < P.opt.minCoverage) return false;
    unsigned nAlive = 0;
    for (unsigned base = 0; base < nNodes; base += 64) {
      const unsigned nd = base + lane;
      const uint64_t m  = wv::ballot(nd < nNodes && A.node_cnt[nd] == maxCount);
      if (lane < 2) aliveBits[(base >> 5) + lane] = uint32_t(m >> (32 * lane));
      nAlive += unsigned(wv::popc(m));
    }
    trunkMask = 64;
    while (trunkMask < 4 * nNodes + 8) trunkMask <<= 1;
    trunkMask -= 1;
    wv::sync();

    const unsigned bestSlot = nContigs, newSlot = nContigs + 1;
    unsigned       bestLen  = 0;
    bool           haveSeed = false;
    if (lane == 0) {
      A.cand_meta[bestSlot * 4 + 0] = 0;
      A.cand_meta[bestSlot * 4 + 1] = 0;
      A.cand_meta[bestSlot * 4 + 2] = 0;
    }
    if (lane < 2 * W) A.cand_bits[size_t(bestSlot) * 2 * W + lane] = 0;
    while (nAlive > 0) {  // :545-560
      // *maxWords.begin(): the smallest word still in the set
      unsigned best = ASM_NONE;
      for (unsigned nd = lane; nd < nNodes; nd += 64)
        if (testBit(aliveBits, nd) && (best == ASM_NONE || wordLess(nd, best))) best = nd;
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = wv::shfl(best, int(lane) ^ off);
        if (o != ASM_NONE && (best == ASM_NONE || wordLess(o, best))) best = o;
      }
      best = wv::first(best);
      if (lane == 0) aliveBits[best >> 5] &= ~(1u << (best & 31));
      wv::sync();
      haveSeed = true;
      if (!walk(best, newSlot)) return false;
      const unsigned newLen = unsigned(A.cand_meta[newSlot * 4 + 0]);
      if (newLen > bestLen) {  // :551-553
        const uint8_t* src = A.cand_seq + size_t(newSlot) * P.max_contig_len;
        uint8_t*       dst = A.cand_seq + size_t(bestSlot) * P.max_contig_len;
        for (unsigned i = lane; i < newLen; i += 64) dst[i] = src[i];
        if (lane < 2 * W) A.cand_bits[size_t(bestSlot) * 2 * W + lane] = A.cand_bits[size_t(newSlot) * 2 * W + lane];
        if (lane < 4) A.cand_meta[bestSlot * 4 + lane] = A.cand_meta[newSlot * 4 + lane];
        bestLen = newLen;
      }
      // maxWords -= seenEdgeBefore (:556)
      nAlive = 0;
      for (unsigned i = lane; i < (nNodes + 31) / 32; i += 64) {
        const uint32_t v = aliveBits[i] & ~edgeBits[i];
        aliveBits[i]     = v;
        nAlive += unsigned(wv::popc(unsigned(v)));
      }
      nAlive = Assembler::waveSum(nAlive);
      wv::sync();
    }
    // seedReadCount = reads holding the LAST seed tried (:577-580): every seed has the maximal count
    const unsigned seedReadCount = haveSeed ? maxCount : 0u;
    if (seedReadCount < P.small_min_seed_reads) return false;  // :586-591
    if (lane == 0) A.cand_meta[bestSlot * 4 + 3] = int(seedReadCount);
    // reads of the contig become used (:594-606); its support holds unused reads only (used ones were not counted)
    const uint64_t sup = (lane < W) ? A.cand_bits[size_t(bestSlot) * 2 * W + lane] : 0;
    const uint64_t fresh = sup & ~usedW;
    usedW |= fresh;
    unusedReads -= popSum(fresh);
    nContigs++;
    wv::sync();
    return true;
  }

  // ---- runSmallAssembler (:622-685) ---------------------------------------------------------------
  WV_DEV void run(const unsigned locus)
  {
    for (int i = 0; i < 8; ++i) A.tPhase[i] = 0;
    A.tMark       = wv::clock();
    A.cyclicIters = 0;
    A.nCand       = 0;
    const unsigned minWL = P.opt.minWordLength, maxWL = P.opt.maxWordLength;
    A.k                  = minWL;
    for (unsigned i = lane; i < A.maskWordCap() + 2; i += 64) A.nmask[i] = 0;
    wv::sync();
    A.packNormalReads(locus);
    wv::sync();
    wv::fence_acquire();
    // bytes outside {A,C,G,T,N}: here a word that holds one matters whenever it repeats inside its read or ties the maximal
    // count (:432, :508-535) -- no cheap exact rule, so the pile is reported
    if (A.countJunkReads() > 0 && A.status == ASM_OK) A.status = ASM_E_ALPHABET;
    W = (A.nNormal + 63) / 64;
    if (W == 0) W = 1;
    A.W         = W;
    A.recStride = asmRecStride(W);
    A.nReads    = A.nNormal;
    if (A.status == ASM_OK && (maxWL > 16u * ASM_MAX_KW || minWL < 2 || P.opt.wordStepSize == 0)) A.status = ASM_E_WORD_TOO_LONG;
    if (A.status == ASM_OK && P.small_max_iterations + 1 > P.opt.maxAssemblyCount) A.status = ASM_E_INTERNAL;
    unusedReads     = A.nNormal;
    unsigned nIter  = 0, lastWL = 0;
    if (A.status == ASM_OK) {
      for (unsigned it = 0; it < P.small_max_iterations; ++it) {
        if (unusedReads < P.small_min_seed_reads) break;  // :642
        const unsigned before = unusedReads;
        for (unsigned wl = minWL; wl <= maxWL; wl += P.opt.wordStepSize) {
          const bool isLastWord = (wl + P.opt.wordStepSize > maxWL);
          A.k                   = wl;
          lastWL                = wl;
          const bool ok         = buildContigs(isLastWord);
          if (A.status != ASM_OK || ok) break;
        }
        nIter++;
        if (A.status != ASM_OK) break;
        if (unusedReads == before) break;  // :657
      }
    }
    emit(locus, nIter, lastWL);
  }

  WV_DEV_COLD void emit(const unsigned locus, const unsigned nIter, const unsigned lastWL)
  {
    AsmLocusOut out;
    out.status            = A.status;
    out.n_contigs         = 0;
    out.n_words           = W;
    out.n_pseudo          = 0;
    out.pseudo_off        = 0;
    out.pseudo_len_off    = 0;
    out.final_word_length = lastWL;
    out.n_iterations      = nIter;
    out.cyclic_iterations = 0;
    out.reserved          = 0;
    if (A.status != ASM_OK) {
      if (lane == 0) P.loci[locus] = out;
      return;
    }
    uint64_t seqBytes = 0;
    for (unsigned f = 0; f < nContigs; ++f) seqBytes += unsigned(A.cand_meta[f * 4 + 0]);
    const uint64_t     bitsWords = uint64_t(nContigs + 1) * 2 * W;
    unsigned long long seqBase = 0, bitsBase = 0;
    if (lane == 0) {
      seqBase  = wv::atomic_add(P.seq_used, (unsigned long long)(seqBytes));
      bitsBase = wv::atomic_add(P.bits_used, (unsigned long long)(bitsWords));
    }
    seqBase  = wv::readlane(uint64_t(seqBase), 0);
    bitsBase = wv::readlane(uint64_t(bitsBase), 0);
    if (seqBase + seqBytes > P.seq_cap || bitsBase + bitsWords > P.bits_cap) {
      out.status = ASM_E_OUT_CAPACITY;
      if (lane == 0) P.loci[locus] = out;
      return;
    }
    uint64_t so = seqBase, bo = bitsBase;
    for (unsigned f = 0; f <= nContigs; ++f) {
      const bool     mark = (f == nContigs);  // the isFiltered reads
      const unsigned len  = mark ? 0u : unsigned(A.cand_meta[f * 4 + 0]);
      const uint8_t* src  = A.cand_seq + size_t(f) * P.max_contig_len;
      for (unsigned i = lane; i < len; i += 64) P.seq_arena[so + i] = src[i];
      if (mark) {
        if (lane < W) P.bits_arena[bo + lane] = filteredW;
        if (lane >= W && lane < 2 * W) P.bits_arena[bo + lane] = 0;
      } else if (lane < 2 * W) {
        P.bits_arena[bo + lane] = A.cand_bits[size_t(f) * 2 * W + lane];
      }
      if (lane == 0) {
        AsmContigOut c;
        c.seq_off    = so;
        c.bits_off   = bo;
        c.seq_len    = len;
        c.cons_begin = mark ? 0 : A.cand_meta[f * 4 + 1];
        c.cons_end   = mark ? 0 : A.cand_meta[f * 4 + 2];
        c.reserved   = mark ? SMALL_FILTERED_MARK : uint32_t(A.cand_meta[f * 4 + 3]);  // contig.seedReadCount
        P.contigs[size_t(locus) * P.opt.maxAssemblyCount + f] = c;
      }
      so += len;
      bo += 2 * W;
    }
    out.n_contigs = nContigs + 1;
    if (lane == 0) P.loci[locus] = out;
  }
};

/// persistent waves over the pile queue, as assemble_kernel; P.opt.maxAssemblyCount = maxAssemblyIterations + 1 record slots
#if !MANTA_TU_DEFINES(MANTA_TU_ASM)
WV_KERNEL void small_assemble_kernel(const AsmParams P);
#else
WV_KERNEL void small_assemble_kernel(const AsmParams P)
{
  uint8_t* wsBase = P.ws + uint64_t(wv::block()) * P.ws_stride;
  while (true) {
    unsigned slot = 0;
    if (wv::lane() == 0) slot = wv::atomic_add(P.counter, 1u);
    slot = wv::first(slot);
    if (slot >= P.n_loci) break;
    const unsigned locus = P.locus_ids ? P.locus_ids[slot] : slot;
    Assembler      a(P, wsBase);
    SmallAsm       s(a);
    s.run(locus);
    wv::sync();
  }
}
#endif

}  // namespace manta_dev
