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
//      because `wordIndices` is filled by iterating `wordCount` (:631-633)      -> a sort per rehash stage, wave-parallel
//   4. the DFS itself (:555-625), successors in alphabet order -> runs of single-successor nodes are discovered and
//      unwound by the whole wave, lane 0 keeps the Tarjan bookkeeping of the junction nodes
// This only runs for loci whose k-mer graph has a cycle (tandem repeats), which the wave-parallel Kahn peel in
// assemble_kernels.hpp detects.
// The same construction is restated on the CPU in oracle/manta_oracle.cpp (unorderedMapOrder / repeatNodes)
// and checked there against the real std::unordered_map.
#pragma once

namespace manta_dev {

WV_DEV uint64_t murmurShiftMix(const uint64_t v)
{
  return v ^ (v >> 47);
}

/// 64-bit little-endian word of the characters i..i+n-1 (n <= 8) of the k-mer starting at packed base index pb
/// (SB = 2: base codes, 16 per dword; SB = 8: the bytes themselves, 4 per dword)
template <int SB>
WV_DEV uint64_t asciiChunk(const uint32_t* codes, const unsigned pb, const unsigned i, const unsigned n)
{
  const unsigned spd  = 32u / SB;
  uint64_t       data = 0;
  for (unsigned b = 0; b < n; ++b) {
    const unsigned p = pb + i + b;
    const unsigned c = (codes[p / spd] >> (32 - SB - SB * (p % spd))) & ((1u << SB) - 1u);
    data |= uint64_t((SB == 2) ? uint8_t("ACGT"[c & 3u]) : uint8_t(c)) << (8 * b);
  }
  return data;
}

/// std::hash<std::string> of libstdc++ (libsupc++ hash_bytes.cc, 64-bit): checked against the live library
/// by the host at context creation and by tests/test_oracle_vs_ref.py
template <int SB>
WV_DEV uint64_t libstdcxxStringHash(const uint32_t* codes, const unsigned pb, const unsigned len)
{
  const uint64_t mul  = (uint64_t(0xc6a4a793UL) << 32) + uint64_t(0x5bd1e995UL);
  uint64_t       hash = uint64_t(0xc70f6907UL) ^ (uint64_t(len) * mul);
  const unsigned lenAligned = len & ~7u;
  for (unsigned i = 0; i < lenAligned; i += 8) {
    const uint64_t data = murmurShiftMix(asciiChunk<SB>(codes, pb, i, 8) * mul) * mul;
    hash ^= data;
    hash *= mul;
  }
  if (len & 7u) {
    hash ^= asciiChunk<SB>(codes, pb, lenAligned, len & 7u);
    hash *= mul;
  }
  hash = murmurShiftMix(hash) * mul;
  hash = murmurShiftMix(hash);
  return hash;
}

/// libstdc++'s node order after inserting `ins` (iteration order of the unordered_map).  The insertion rule (front of the bucket's run if the bucket is
/// non-empty, else front of the whole list) makes the iteration order after inserting a sequence S into `nb` buckets a
/// pure sort: buckets by the time of their FIRST element, latest first; inside a bucket by time, latest first.  A rehash
/// (bits/hashtable.h _M_rehash_aux, unique keys) re-inserts the current list in list order under the new bucket count,
/// so the whole history is a chain of such stages -- one per entry of the recorded growth schedule -- whose input is
/// (order after the previous stage) ++ (the insertions up to the next growth).  Each stage here:
///   1. bucket arrays bf[b] = first time (atomic min), bh[b] = head of an UNORDERED chain of the bucket's times (atomic exch)
///   2. lane per non-empty bucket: chain length -> w[first time]
///   3. exclusive suffix sum of w over time = output position of every bucket's run
///   4. lane per non-empty bucket: emit the chain's times in descending order
/// ins[0..n): insertion sequence; a, b: two scratch arrays of n (the result is returned in one of them);
/// chain, offs: n each; bf, bh: nbMax each.  All lanes must call.
WV_DEV uint32_t* unorderedOrderWave(
    const AsmParams& P, const uint64_t* h, const uint32_t* ins, uint32_t* a, uint32_t* b, uint32_t* chain, uint32_t* offs, uint32_t* bf,
    uint32_t* bh, const unsigned n)
{
  const uint32_t NIL  = 0xffffffffu;
  const unsigned lane = unsigned(wv::lane());
  uint32_t*      cur  = a;
  uint32_t*      out  = b;
  unsigned       m0   = 0;  // elements already ordered in cur[0..m0)
  unsigned       sched = 0;
  unsigned       nb    = 1;  // bucket count before the first recorded growth step
  while (m0 < n) {
    // a stage starts where the recorded schedule re-buckets (insertion index == growth_size[s]) and runs up to the next one
    while (sched < P.n_growth && P.growth_size[sched] <= m0) nb = P.growth_buckets[sched++];
    unsigned m1 = n;
    if (sched < P.n_growth && P.growth_size[sched] < m1) m1 = P.growth_size[sched];
    // 1. buckets
    for (unsigned i = lane; i < nb; i += 64) {
      bf[i] = NIL;
      bh[i] = NIL;
    }
    for (unsigned t = lane; t < m1; t += 64) offs[t] = 0;
    wv::sync();
    for (unsigned t = lane; t < m1; t += 64) {
      uint32_t node;
      if (t < m0) {
        node = cur[t];
      } else {
        node   = ins[t];
        cur[t] = node;
      }
      const unsigned bkt = unsigned(h[node] % nb);
      chain[t]           = wv::atomic_exch(&bh[bkt], t);
      wv::atomic_min(&bf[bkt], t);
    }
    wv::sync();
    wv::fence_acquire();
    // 2. run lengths at the buckets' first times
    for (unsigned i = lane; i < nb; i += 64) {
      const uint32_t first = wv::atomic_load(&bf[i]);
      if (first == NIL) continue;
      unsigned c = 0;
      for (uint32_t t = wv::atomic_load(&bh[i]); t != NIL; t = chain[t]) ++c;
      offs[first] = c;
    }
    wv::sync();
    // 3. exclusive suffix sum over time, top chunk first
    {
      unsigned carry = 0;
      for (unsigned top = ((m1 + 63) / 64) * 64; top > 0; top -= 64) {
        const unsigned t = top - 64 + (63 - lane);  // lane 0 takes the highest time of the chunk
        const unsigned w = (t < m1) ? offs[t] : 0u;
        unsigned       inc = w;                     // inclusive prefix over lanes (= over descending time)
        for (int off = 1; off < 64; off <<= 1) {
          const unsigned o = wv::shfl(inc, int(lane) - off);
          if (int(lane) >= off) inc += o;
        }
        if (t < m1) offs[t] = carry + inc - w;
        carry += wv::shfl(inc, 63);
      }
    }
    wv::sync();
    // 4. emit every bucket's run, latest time first
    for (unsigned i = lane; i < nb; i += 64) {
      const uint32_t first = wv::atomic_load(&bf[i]);
      if (first == NIL) continue;
      unsigned       pos  = offs[first];
      uint32_t       prev = NIL;  // times are emitted in strictly descending order: next = largest time below `prev`
      const uint32_t head = wv::atomic_load(&bh[i]);
      while (true) {
        uint32_t best = NIL;
        for (uint32_t t = head; t != NIL; t = chain[t])
          if ((prev == NIL || t < prev) && (best == NIL || t > best)) best = t;
        if (best == NIL) break;
        out[pos++] = cur[best];
        prev       = best;
      }
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

template <int SB>
WV_DEV_COLD void AssemblerT<SB>::exactRepeatSearch()
{
  static const int KW = GEN_KW;
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
  tickExact(7);
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
    h[nd] = libstdcxxStringHash<SB>(codes, node_key[nd], k);
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
  tickExact(0);
  // (the first two key dwords of every member, by group position: the quadratic loop below then reads two consecutive arrays
  // instead of chasing grp -> node_key -> pile for every pair; the full comparison is left for equal 32-symbol prefixes)
  uint32_t* const pk0 = next;
  uint32_t* const pk1 = frames;
  for (unsigned i = lane; i < n; i += 64) {
    const Key<KW> kx = keyAt<KW>(node_key[grp[i]]);
    pk0[i]           = kx.w[0];
    pk1[i]           = kx.w[1];
  }
  wv::sync();
  for (unsigned r = 0; r < nReads; ++r) {
    const unsigned g0 = tmp[r], g1 = tmp[r + 1];
    for (unsigned i = g0 + lane; i < g1; i += 64) {
      const unsigned x    = grp[i];
      const Key<KW>  kx   = keyAt<KW>(node_key[x]);
      unsigned       rank = 0;
      for (unsigned j = g0; j < g1; ++j) {
        if (j == i) continue;
        const uint32_t b0 = pk0[j], b1 = pk1[j];
        bool           less = (b0 < kx.w[0]) || (b0 == kx.w[0] && b1 < kx.w[1]);
        if (b0 == kx.w[0] && b1 == kx.w[1]) less = keyLess(keyAt<KW>(node_key[grp[j]]), kx);
        if (less) rank++;
      }
      seqA[g0 + rank] = x;
    }
  }
  wv::sync();

  tickExact(1);
  // ---- 2./3. unordered_map order, twice ------------------------------------------------------------
  // scratch: next | tmp | idx | low | stack | frames | before  =  9n + 64 words, all free until the DFS
  const uint32_t* roots;
  {
    const unsigned nbCap = 3 * P.cap_nodes + 32;
    uint32_t*      pool  = next;
    uint32_t*      chain = pool;                    // n
    uint32_t*      offs  = pool + P.cap_nodes;      // n
    uint32_t*      spare = pool + 2 * P.cap_nodes;  // n
    uint32_t*      bf    = pool + 3 * size_t(P.cap_nodes);
    uint32_t*      bh    = bf + nbCap;              // 3n + 3n + 64 more words
    // iteration order of wordCount (insertion sequence seqA), then of wordIndices (filled by iterating wordCount)
    uint32_t* order1 = unorderedOrderWave(P, h, seqA, seqB, spare, chain, offs, bf, bh, n);
    tickExact(2);
    uint32_t* s1     = (order1 == seqB) ? spare : seqB;
    uint32_t* order2 = unorderedOrderWave(P, h, order1, seqA, s1, chain, offs, bf, bh, n);
    // the DFS below reuses the pool: park the root order where it survives
    if (order2 != seqA && order2 != seqB) {
      for (unsigned i = lane; i < n; i += 64) seqB[i] = order2[i];
      wv::sync();
      order2 = seqB;
    }
    roots = order2;
  }

  tickExact(3);
  // ---- 4. DFS (:555-625) -------------------------------------------------------------------------------
  // The reference's recursion visits every node once; on these graphs almost every node has exactly one successor
  // ("run" nodes), so the traversal is mostly forced.  Lane 0 drives the Tarjan bookkeeping of the junction nodes;
  // whole runs are discovered, and whole stack segments popped, by all 64 lanes at once:
  //   * discovery of a run: lanes test nodes c, c+1, .. c+63 (node ids follow read order, so a run's ids are mostly
  //     consecutive) for "unvisited run node whose predecessor on the chain is the previous lane's node" and take the
  //     leading stretch in one step -- DFS indices, low links, stack entries are written by 64 lanes;
  //   * unwinding a run whose DFS indices are [f, f+len): with L = low link coming back from below,
  //       L <  f      : no node of the run is an SCC root, the link passes through unchanged;
  //       f <= L < f+len : the node with index L is the root of an SCC = everything above it on the stack ("small
  //                     circle" test :612 against the stack top), every run node below it is a singleton SCC;
  //       L >= f+len  : every run node is a singleton SCC;
  //   which is exactly what the node-by-node unwinding computes (low[v_i] = min(idx[v_i], low[v_i+1]), root test :603).
  {
    const uint32_t ONSTACK = 0x80000000u, INF = 0x7fffffffu, RUNFRAME = 0x80000000u;
    uint32_t*      pool    = next;
    uint32_t*      idxA    = pool;                       // DFS index, 0 = unvisited
    uint32_t*      lowA    = pool + P.cap_nodes;         // low link | ONSTACK
    uint32_t*      stackA  = pool + 2 * size_t(P.cap_nodes);
    uint32_t*      runNext = pool + 3 * size_t(P.cap_nodes);  // the single successor of a run node, else ASM_NONE
    uint32_t*      frLo    = pool + 4 * size_t(P.cap_nodes);  // frames: junction (node << 3 | next symbol) or run length
    uint32_t*      frHi    = pool + 5 * size_t(P.cap_nodes);  //         stack position at discovery | RUNFRAME
    for (unsigned nd = lane; nd < n; nd += 64) {
      idxA[nd] = 0;
      lowA[nd] = 0;
      unsigned only = ASM_NONE, cnt = 0;
      bool     self = false;
      for (unsigned c = 0; c < 4; ++c) {
        const unsigned sx = recSucc(nd)[c];
        if (sx == ASM_NONE) continue;
        if (sx == nd) self = true;
        only = sx;
        ++cnt;
      }
      runNext[nd] = (cnt == 1 && !self) ? only : ASM_NONE;
    }
    wv::sync();
    tickExact(5);

    enum { REQ_NONE = 0, REQ_NEXTROOT = 1, REQ_DESCEND = 2, REQ_POP = 3 };
    unsigned fp = 0, sp = 0, nextIndex = 1, rootCursor = 0;
    unsigned returning = 0, retLow = INF;
    unsigned afterRun = ASM_NONE;  // last node of a run that was just discovered: its successor is examined next
    unsigned req = REQ_NEXTROOT, reqA = 0, reqB = 0, reqC = 0, reqD = 0;
    bool     done = false;
    while (!done) {
      // ---------------- wave-wide requests ----------------
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
            idxA[root]  = nextIndex;
            lowA[root]  = nextIndex | ONSTACK;
            stackA[sp]  = root;
            frLo[fp]    = root << 3;
            frHi[fp]    = sp;
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
        // discover the run starting at reqA (unvisited run node)
        unsigned       c     = reqA;
        const unsigned p0    = sp;
        unsigned       len   = 0;
        unsigned       lastN = c;
        while (true) {
          const unsigned x     = c + lane;
          const bool     inR   = x < n;
          const unsigned rn    = inR ? runNext[x] : ASM_NONE;
          const bool     okSelf = inR && rn != ASM_NONE && idxA[x] == 0;
          const unsigned contig = (rn == x + 1) ? 1u : 0u;            // this lane's node continues into the next lane's node
          const unsigned prevContig = wv::shr1(contig, 1u);           // lane 0 is the entry point
          const uint64_t good  = wv::ballot(okSelf && prevContig != 0);
          const unsigned take  = (~good == 0) ? 64u : unsigned(wv::ctz(~good));  // leading stretch (>= 1: the requester checked the entry)
          if (lane < take) {
            idxA[x]             = nextIndex + lane;
            lowA[x]             = (nextIndex + lane) | ONSTACK;
            stackA[sp + lane]   = x;
          }
          nextIndex += take;
          sp += take;
          len += take;
          lastN               = c + take - 1;
          const unsigned lastRn = wv::readlane(rn, int(take - 1));
          wv::sync();
          // continue the same run at its (non-adjacent or beyond-the-chunk) successor if that is an unvisited run node
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
        // stack positions [reqA, reqB) leave the stack; those at or above reqC are flagged as repeat words if reqD
        for (unsigned i = reqA + lane; i < reqB; i += 64) {
          const unsigned w = stackA[i];
          lowA[w] &= ~ONSTACK;
          if (reqD && i >= reqC) node_flag[w] |= NF_REPEAT;
        }
        req = REQ_NONE;
        wv::sync();
        continue;
      }

      // ---------------- lane 0: junction bookkeeping until the next wave-wide request ----------------
      if (lane == 0) {
        while (req == REQ_NONE) {
          if (afterRun != ASM_NONE) {
            const unsigned y = runNext[afterRun];
            afterRun         = ASM_NONE;
            if (idxA[y] == 0) {  // an unvisited junction (an unvisited run node would have been taken by the descent)
              idxA[y]    = nextIndex;
              lowA[y]    = nextIndex | ONSTACK;
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
            // unwind a whole run with the low link `retLow` coming back from below
            const unsigned p0       = hi & ~RUNFRAME;
            const unsigned len      = frLo[fp - 1];
            const unsigned firstIdx = idxA[stackA[p0]];
            const unsigned L        = retLow;
            fp--;
            returning = 1;
            if (L < firstIdx) {
              retLow = L;  // passes through, nothing leaves the stack
              continue;
            }
            unsigned flagFrom = sp, small = 0;
            if (L < firstIdx + len) {
              flagFrom = p0 + (L - firstIdx);  // the SCC rooted inside the run: everything from here to the stack top
              small    = ((idxA[stackA[sp - 1]] - L) <= 50) ? 1u : 0u;
              if (sp - flagFrom == 1) small = 0;  // a root alone on top of the stack is a singleton (:605-607)
            }
            retLow = firstIdx;
            if (sp - p0 <= 4) {
              for (unsigned i = p0; i < sp; ++i) {
                const unsigned w = stackA[i];
                lowA[w] &= ~ONSTACK;
                if (small && i >= flagFrom) node_flag[w] |= NF_REPEAT;
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
          // junction frame
          const unsigned f   = frLo[fp - 1];
          const unsigned nd  = f >> 3;
          const unsigned sym = f & 7;
          if (returning) {  // :588-590
            const unsigned lp = lowA[nd] & ~ONSTACK;
            if (retLow < lp) lowA[nd] = retLow | (lowA[nd] & ONSTACK);
            returning = 0;
          }
          if (sym < 4) {
            frLo[fp - 1]     = f + 1;
            const unsigned sx = recSucc(nd)[sym];
            if (sx == nd) {  // homopolymer (:574-577)
              node_flag[nd] |= NF_REPEAT;
              continue;
            }
            if (sx == ASM_NONE) continue;  // :580
            if (idxA[sx] == 0) {           // :583-590
              if (runNext[sx] != ASM_NONE) {
                req  = REQ_DESCEND;
                reqA = sx;
              } else {
                idxA[sx]   = nextIndex;
                lowA[sx]   = nextIndex | ONSTACK;
                nextIndex++;
                stackA[sp] = sx;
                frLo[fp]   = sx << 3;
                frHi[fp]   = sp;
                sp++;
                fp++;
              }
            } else if (lowA[sx] & ONSTACK) {  // :592-598
              const unsigned l = lowA[nd] & ~ONSTACK;
              if (idxA[sx] < l) lowA[nd] = idxA[sx] | ONSTACK;
            }
            continue;
          }
          // all successors done (:603-622)
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
                  if (small) node_flag[w] |= NF_REPEAT;
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
      // lane 0's state becomes the wave's
      wv::sync();
      fp         = wv::first(fp);
      sp         = wv::first(sp);
      nextIndex  = wv::first(nextIndex);
      returning  = wv::first(returning);
      retLow     = wv::first(retLow);
      afterRun   = wv::first(afterRun);
      req        = wv::first(req);
      reqA       = wv::first(reqA);
      reqB       = wv::first(reqB);
      reqC       = wv::first(reqC);
      reqD       = wv::first(reqD);
    }
  }
  wv::sync();
  wv::fence_acquire();
}

}  // namespace manta_dev
