// Exact repeat-k-mer search for CYCLIC k-mer graphs.
//
// The reference marks "repeat words" with a Tarjan-style DFS whose ROOT ORDER is the iteration order of a
// std::unordered_map<std::string,...> (assembly/IterativeAssembler.cpp:627-642), and whose "small circle"
// rule (:612-613, DFS-index span <= 50) depends on that order.  To stay bit-identical on cyclic graphs the
// order is re-derived here from first principles:
//   1. insertion sequence of `wordCount`: reads in order, each read's NEW distinct k-mers in lexicographic
//      order (:516-548)                                       -> firstRead + in-group rank (wave-parallel)
//   2. std::hash<std::string> of every k-mer (libstdc++ _Hash_bytes, Murmur-style, seed 0xc70f6907)
//                                                             -> one lane per node
//   3. libstdc++ node order after those insertions (front-of-bucket insertion, whole-list re-insertion on
//      every rehash, bucket growth schedule recorded by the host from the live library), applied twice
//      because `wordIndices` is filled by iterating `wordCount` (:631-633)      -> lane 0, serial
//   4. the DFS itself (:555-625), recursion unrolled, successors in alphabet order -> lane 0, serial
// Steps 3-4 are inherently sequential in the reference as well; they only run for loci whose k-mer graph has
// a cycle (tandem repeats), which the wave-parallel Kahn peel in assemble_kernels.hpp detects.
// The same construction is restated on the CPU in oracle/manta_oracle.cpp (unorderedMapOrder / repeatNodes)
// and checked there against the real std::unordered_map.
#pragma once

namespace manta_dev {

WV_DEV uint64_t murmurShiftMix(const uint64_t v)
{
  return v ^ (v >> 47);
}

/// 64-bit little-endian word of ASCII bases i..i+n-1 (n <= 8) of the k-mer starting at packed base index pb
WV_DEV uint64_t asciiChunk(const uint32_t* codes, const unsigned pb, const unsigned i, const unsigned n)
{
  uint64_t data = 0;
  for (unsigned b = 0; b < n; ++b) {
    const unsigned p = pb + i + b;
    const unsigned c = (codes[p >> 4] >> (30 - 2 * (p & 15))) & 3;
    data |= uint64_t(uint8_t("ACGT"[c])) << (8 * b);
  }
  return data;
}

/// std::hash<std::string> of libstdc++ (libsupc++ hash_bytes.cc, 64-bit): checked against the live library
/// by the host at context creation and by tests/test_oracle_vs_ref.py
WV_DEV uint64_t libstdcxxStringHash(const uint32_t* codes, const unsigned pb, const unsigned len)
{
  const uint64_t mul  = (uint64_t(0xc6a4a793UL) << 32) + uint64_t(0x5bd1e995UL);
  uint64_t       hash = uint64_t(0xc70f6907UL) ^ (uint64_t(len) * mul);
  const unsigned lenAligned = len & ~7u;
  for (unsigned i = 0; i < lenAligned; i += 8) {
    const uint64_t data = murmurShiftMix(asciiChunk(codes, pb, i, 8) * mul) * mul;
    hash ^= data;
    hash *= mul;
  }
  if (len & 7u) {
    hash ^= asciiChunk(codes, pb, lenAligned, len & 7u);
    hash *= mul;
  }
  hash = murmurShiftMix(hash) * mul;
  hash = murmurShiftMix(hash);
  return hash;
}

/// serial (one lane) emulation of libstdc++'s node order: seqIn[0..n) = keys (node ids) in insertion order,
/// seqOut[0..n) = the same ids in iteration order
WV_DEV void unorderedOrderSerial(
    const AsmParams& P, const uint64_t* h, const uint32_t* seqIn, uint32_t* seqOut, uint32_t* next, uint32_t* before,
    uint32_t* tmp, const unsigned n)
{
  const uint32_t NIL = 0xffffffffu, EMPTY = 0xfffffffeu, HEAD = 0xfffffffdu;
  unsigned       nb = 1;
  before[0]         = EMPTY;
  uint32_t head     = NIL;
  unsigned schedPos = 0;
  for (unsigned i = 0; i < n; ++i) {
    if (schedPos < P.n_growth && P.growth_size[schedPos] == i) {
      // rehash: re-insert every node in current list order (bits/hashtable.h _M_rehash_aux)
      nb = P.growth_buckets[schedPos];
      ++schedPos;
      unsigned m = 0;
      for (uint32_t p = head; p != NIL; p = next[p]) tmp[m++] = p;
      for (unsigned b = 0; b < nb; ++b) before[b] = EMPTY;
      head = NIL;
      for (unsigned j = 0; j < m; ++j) {
        const uint32_t node = tmp[j];
        const unsigned b    = unsigned(h[node] % nb);
        if (before[b] != EMPTY) {
          if (before[b] == HEAD) {
            next[node] = head;
            head       = node;
          } else {
            next[node]       = next[before[b]];
            next[before[b]] = node;
          }
        } else {
          next[node] = head;
          head       = node;
          if (next[node] != NIL) before[unsigned(h[next[node]] % nb)] = node;
          before[b] = HEAD;
        }
      }
    }
    const uint32_t node = seqIn[i];
    const unsigned b    = unsigned(h[node] % nb);
    if (before[b] != EMPTY) {  // bits/hashtable.h _M_insert_bucket_begin
      if (before[b] == HEAD) {
        next[node] = head;
        head       = node;
      } else {
        next[node]       = next[before[b]];
        next[before[b]] = node;
      }
    } else {
      next[node] = head;
      head       = node;
      if (next[node] != NIL) before[unsigned(h[next[node]] % nb)] = node;
      before[b] = HEAD;
    }
  }
  unsigned m = 0;
  for (uint32_t p = head; p != NIL; p = next[p]) seqOut[m++] = p;
}

WV_DEV_COLD void Assembler::exactRepeatSearch()
{
  static const int KW = ASM_MAX_KW;
  const unsigned lane = unsigned(wv::lane());
  const unsigned n    = nNodes;
  // carve (u32 units) out of the `exact` workspace region: 14 * cap_nodes + 64 words
  uint32_t* base    = exact_ws + 16;
  uint64_t* h       = reinterpret_cast<uint64_t*>(base);              // 2n
  uint32_t* seqA    = base + 2 * size_t(P.cap_nodes);                 // n
  uint32_t* seqB    = seqA + P.cap_nodes;                             // n
  uint32_t* next    = seqB + P.cap_nodes;                             // n
  uint32_t* tmp     = next + P.cap_nodes;                             // n
  uint32_t* idx     = tmp + P.cap_nodes;                              // n
  uint32_t* low     = idx + P.cap_nodes;                              // n
  uint32_t* stack   = low + P.cap_nodes;                              // n
  uint32_t* frames  = stack + P.cap_nodes;                            // n  (node << 3 | next symbol)
  uint32_t* before  = frames + P.cap_nodes;                           // 3n + 64
  uint32_t* firstRd = idx;                                            // reused before the DFS
  uint32_t* grpCnt  = low;                                            // per-read counters (n >= reads not guaranteed -> rd_len scratch below)
  uint32_t* grp     = stack;

  // the largest bucket count the schedule can ask for must fit `before`
  {
    unsigned nbMax = 1;
    for (unsigned s = 0; s < P.n_growth; ++s)
      if (P.growth_size[s] < n) nbMax = P.growth_buckets[s];
    if (nbMax > 3 * P.cap_nodes + 32 || nReads + 2 > P.cap_nodes) {
      status = ASM_E_TABLE_FULL;
      return;
    }
  }

  // ---- 1. insertion sequence -------------------------------------------------------------------
  for (unsigned r = lane; r <= nReads; r += 64) grpCnt[r] = 0;
  wv::sync();
  for (unsigned nd = lane; nd < n; nd += 64) {
    unsigned fr = 0;
    for (unsigned w = 0; w < W; ++w) {
      const uint64_t s = recSup(nd)[w];
      if (s) {
        fr = w * 64 + unsigned(wv::ctz(s));
        break;
      }
    }
    firstRd[nd] = fr;
    wv::atomic_add(&grpCnt[fr], 1u);
    h[nd] = libstdcxxStringHash(codes, node_key[nd], k);
  }
  wv::sync();
  wv::fence_acquire();
  // exclusive prefix over reads (lane 0; <= ~1000 reads) into tmp[r]; tmp[nReads] = n
  if (lane == 0) {
    unsigned acc = 0;
    for (unsigned r = 0; r < nReads; ++r) {
      tmp[r] = acc;
      acc += grpCnt[r];
      grpCnt[r] = 0;
    }
    tmp[nReads] = acc;
  }
  wv::sync();
  for (unsigned nd = lane; nd < n; nd += 64) {
    const unsigned fr = firstRd[nd];
    grp[tmp[fr] + wv::atomic_add(&grpCnt[fr], 1u)] = nd;
  }
  wv::sync();
  wv::fence_acquire();
  // rank inside each group by lexicographic k-mer order (keys are distinct)
  for (unsigned r = 0; r < nReads; ++r) {
    const unsigned g0 = tmp[r], g1 = tmp[r + 1];
    for (unsigned i = g0 + lane; i < g1; i += 64) {
      const unsigned x    = grp[i];
      const Key<KW>  kx   = keyAt<KW>(node_key[x]);
      unsigned       rank = 0;
      for (unsigned j = g0; j < g1; ++j) {
        if (j == i) continue;
        if (keyLess(keyAt<KW>(node_key[grp[j]]), kx)) rank++;
      }
      seqA[g0 + rank] = x;
    }
  }
  wv::sync();

  // ---- 2./3. unordered_map order, twice; 4. DFS -------------------------------------------------
  if (lane == 0) {
    unorderedOrderSerial(P, h, seqA, seqB, next, before, tmp, n);  // iteration order of wordCount
    unorderedOrderSerial(P, h, seqB, seqA, next, before, tmp, n);  // iteration order of wordIndices
    for (unsigned i = 0; i < n; ++i) {
      idx[i] = 0;
      low[i] = 0;
    }
    const uint32_t ONSTACK = 0x80000000u;  // kept in the top bit of low[]
    unsigned       sp = 0, fp = 0, nextIndex = 1;
    for (unsigned ri = 0; ri < n; ++ri) {
      const unsigned root = seqA[ri];
      if (idx[root] != 0) continue;
      idx[root]    = nextIndex;
      low[root]    = nextIndex | ONSTACK;
      nextIndex++;
      stack[sp++]  = root;
      frames[fp++] = root << 3;
      while (fp > 0) {
        const unsigned f   = frames[fp - 1];
        const unsigned nd  = f >> 3;
        const unsigned sym = f & 7;
        if (sym < 4) {
          frames[fp - 1]   = f + 1;
          const unsigned s = recSucc(nd)[sym];
          if (s == nd) {  // homopolymer (:574-577)
            node_flag[nd] |= NF_REPEAT;
            continue;
          }
          if (s == ASM_NONE) continue;  // :580
          if (idx[s] == 0) {            // :583-590
            idx[s]       = nextIndex;
            low[s]       = nextIndex | ONSTACK;
            nextIndex++;
            stack[sp++]  = s;
            frames[fp++] = s << 3;
          } else if (low[s] & ONSTACK) {  // :592-598
            const unsigned l = low[nd] & ~ONSTACK;
            if (idx[s] < l) low[nd] = idx[s] | ONSTACK;
          }
          continue;
        }
        // all successors done (:603-622)
        const unsigned myLow = low[nd] & ~ONSTACK;
        if (myLow == idx[nd]) {
          const unsigned last = stack[sp - 1];
          if (last == nd) {
            sp--;
            low[nd] &= ~ONSTACK;
          } else {
            const bool isSmallCircle = (idx[last] - idx[nd]) <= 50;
            while (true) {
              const unsigned w = stack[--sp];
              if (isSmallCircle) node_flag[w] |= NF_REPEAT;
              low[w] &= ~ONSTACK;
              if (w == nd) break;
            }
          }
        }
        fp--;
        if (fp > 0) {  // caller's lowlink update after the recursive call returns (:588-590)
          const unsigned p  = frames[fp - 1] >> 3;
          const unsigned lp = low[p] & ~ONSTACK, ln = low[nd] & ~ONSTACK;
          if (ln < lp) low[p] = ln | (low[p] & ONSTACK);
        }
      }
    }
  }
  wv::sync();
  wv::fence_acquire();
}

}  // namespace manta_dev
