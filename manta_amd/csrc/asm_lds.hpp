// LDS-resident fast path of the iterative assembler (assembly/IterativeAssembler.cpp:844-931, first word length).
//
// assemble_kernel (assemble_kernels.hpp) keeps a locus' k-mer table, node records and links in a per-wave HBM slab;
// with thousands of resident waves that working set is far past L2 / Infinity Cache and every table probe and every walk
// step is a scattered HBM access (round 1: 140x the algorithmic bytes, 73 % of wave cycles waiting).  A config-2 sized
// locus, however, is ~1.1 k distinct words: packed tightly its whole graph fits in a third of a CU's 160 KB LDS.
//
//   one single-wave workgroup per locus, MANTA_LDS_BUDGET bytes of LDS (3 workgroups per CU), everything between the
//   packed read pile and the finished candidate contigs in LDS:
//     pile      2-bit codes, 16 bases / dword, MSB first (+ N bitmap)                        ~3.5 + 1.9 KB
//     table     2048 slots in 512 buckets of 4 (one ds_read_b128 per probe), slot = {first occurrence : 15, node id :
//               11, hash tag : 6} -- a mismatching tag settles a probe without touching the key               8 KB
//     nodes     16 B per distinct word: successor links (4 x 11 bit, id+1, 0 = none) + count + first occurrence,
//               predecessor links + support reference                                                      ~17 KB
//     supports  a word seen in ONE read (most error k-mers) carries the read index in its record; only words with
//               >= 2 reads own a 16-byte bitset in a pool that grows down from the top                      ~6 KB
//     scratch   what is left between the node array and the pool: cycle-peel state, histograms, visited bitmaps
//   HBM sees the read bases once and the results once.
//
// Scope: the FIRST word length of a locus whose graph is acyclic and whose reads fit W <= 2 set words (<= 108 reads).
// Anything else -- a cycle (the exact libstdc++-order repeat search), a repeat hit that asks for the next word length
// (pseudo reads), more words than the LDS holds -- is re-run from scratch by the general path (Assembler::run) in the
// same wave; nothing is approximated.  The arithmetic is the same as in assemble_kernels.hpp / walk_lanes.hpp, whose
// comments carry the reference line numbers; only the storage differs.
#pragma once
#include "assemble_kernels.hpp"

namespace manta_dev {

#ifndef MANTA_LDS_BUDGET
#define MANTA_LDS_BUDGET 53248
#endif
static const unsigned LN_BUDGET      = MANTA_LDS_BUDGET;  // bytes of LDS per locus (3 x 52 KB <= 160 KB per CU)
static const unsigned LN_SLOTS       = 2048;
static const unsigned LN_BUCKETS     = LN_SLOTS / 4;
static const unsigned LN_MAX_NODES   = 1843;              // 0.9 x slots; node ids are stored +1 in 11-bit link fields
static const unsigned LN_MAX_READS   = 128;               // W <= 2
static const unsigned LN_MAX_PILE    = 2046;              // code dwords: a packed base index must fit 15 bits
static const unsigned LN_EMPTY       = 0xffffffffu;
static const unsigned LN_ID_PENDING  = 0x7ffu;            // slot claimed, node id not assigned yet
static const unsigned LN_FAT         = 0x8000u;           // support reference: index into the bitset pool (else: a read)

// fixed part of the LDS map (bytes)
static const unsigned LN_OFF_SLOTS  = 0;
static const unsigned LN_OFF_UNUSED = LN_OFF_SLOTS + 4 * LN_SLOTS;  // "unusedWords" bitmap, 64 dwords
static const unsigned LN_OFF_REPEAT = LN_OFF_UNUSED + 256;          // repeatWords bitmap (self loops), 64 dwords
static const unsigned LN_OFF_VARS   = LN_OFF_REPEAT + 256;          // [0] bitset-pool size
static const unsigned LN_OFF_RD     = LN_OFF_VARS + 64;             // read descriptors {code dword offset : 11, length : 16, has N : 1}
static const unsigned LN_OFF_DYN    = LN_OFF_RD + 4 * LN_MAX_READS; // codes, N bitmap, nodes ... pool

struct alignas(16) LRec {
  uint64_t w0;  ///< successor links 4 x 11 | count << 44 (8 bit) | first occurrence, low 12 bits << 52
  uint64_t w1;  ///< predecessor links 4 x 11 | support reference << 44 (16 bit) | first occurrence, high 3 bits << 60
};
struct alignas(16) LSet {
  uint64_t w[2];
};
struct alignas(16) LBucket {
  uint32_t s[4];
};

enum { LN_DONE = 0, LN_PUNT = 1 };

struct LdsAssembler {
  Assembler&       A;  // HBM workspace views (candidate / lane outputs), parameters, selectAndEmit
  const AsmParams& P;
  char*            lds;
  uint32_t *       slots, *unused_bits, *repeat_bits, *vars, *rd, *codes, *nmask;
  LRec*            nodes;
  unsigned         nNormal, W, k, nNodes, nCodeWords, nMaskWords, nodesOff, nCand;
  uint64_t         tMark;  // per-phase shader clocks (only with -DMANTA_ASM_PROFILE)

  /// optional phase profile into P.phase_cycles (same slots as Assembler::tick: pack, table+links, -, cycle-check, -, seed, walk, select+emit)
  WV_DEV void tick(const int phase)
  {
#ifdef MANTA_ASM_PROFILE
    const uint64_t now = wv::clock();
    if (P.phase_cycles && wv::lane() == 0) wv::atomic_add(&P.phase_cycles[phase], (unsigned long long)(now - tMark));
    tMark = now;
#else
    (void)phase;
#endif
  }

  WV_DEV LdsAssembler(Assembler& a, char* ldsBase) : A(a), P(a.P), lds(ldsBase)
  {
    slots       = reinterpret_cast<uint32_t*>(lds + LN_OFF_SLOTS);
    unused_bits = reinterpret_cast<uint32_t*>(lds + LN_OFF_UNUSED);
    repeat_bits = reinterpret_cast<uint32_t*>(lds + LN_OFF_REPEAT);
    vars        = reinterpret_cast<uint32_t*>(lds + LN_OFF_VARS);
    rd          = reinterpret_cast<uint32_t*>(lds + LN_OFF_RD);
    codes       = reinterpret_cast<uint32_t*>(lds + LN_OFF_DYN);
    nmask       = codes;
    nodes       = nullptr;
  }

  // ---- record fields ----
  WV_DEV static unsigned recCnt(const uint64_t w0) { return unsigned(w0 >> 44) & 0xffu; }
  WV_DEV static unsigned recPb(const uint64_t w0, const uint64_t w1) { return (unsigned(w0 >> 52) & 0xfffu) | ((unsigned(w1 >> 60) & 7u) << 12); }
  WV_DEV static unsigned recSupRef(const uint64_t w1) { return unsigned(w1 >> 44) & 0xffffu; }
  /// link field c of a packed link word: node id or ASM_NONE
  WV_DEV static unsigned linkId(const uint64_t w, const unsigned c)
  {
    const unsigned f = unsigned(w >> (11 * c)) & 0x7ffu;
    return f ? f - 1 : ASM_NONE;
  }
  WV_DEV LSet* pool(const unsigned idx) const { return reinterpret_cast<LSet*>(lds + LN_BUDGET) - (idx + 1); }
  WV_DEV unsigned poolBytes() const { return 16u * vars[0]; }

  /// read support of a node as two set words
  WV_DEV void supOf(const uint64_t w1, uint64_t& s0, uint64_t& s1) const
  {
    const unsigned ref = recSupRef(w1);
    if (ref & LN_FAT) {
      const LSet v = *pool(ref & 0x7fffu);
      s0           = v.w[0];
      s1           = v.w[1];
    } else {
      s0 = (ref < 64) ? (uint64_t(1) << ref) : 0;
      s1 = (ref >= 64) ? (uint64_t(1) << (ref - 64)) : 0;
    }
  }

  // ---- packed pile ----
  template <int KW>
  WV_DEV Key<KW> keyAt(const unsigned pb) const
  {
    Key<KW>        key;
    const unsigned kw = (k + 15) >> 4;
    const unsigned wi = pb >> 4, sh = (pb & 15) * 2;
    uint32_t       raw[KW + 1];
    for (int i = 0; i <= KW; ++i) raw[i] = (unsigned(i) <= kw) ? codes[wi + i] : 0u;
    for (int i = 0; i < KW; ++i) {
      uint32_t v = 0;
      if (unsigned(i) < kw) {
        v                   = uint32_t((((uint64_t(raw[i]) << 32) | raw[i + 1]) << sh) >> 32);
        const unsigned have = k - 16u * unsigned(i);
        if (have < 16) v &= ~((1u << (32 - 2 * have)) - 1u);
      }
      key.w[i] = v;
    }
    return key;
  }
  WV_DEV uint32_t codes16(const unsigned pb) const
  {
    const unsigned wi = pb >> 4, sh = (pb & 15) * 2;
    const uint32_t a  = codes[wi];
    if (sh == 0) return a;
    return (a << sh) | (codes[wi + 1] >> (32 - sh));
  }
  WV_DEV unsigned baseAt(const unsigned pb) const { return (codes[pb >> 4] >> (30 - 2 * (pb & 15))) & 3u; }

  WV_DEV bool windowHasN(const unsigned maskWordBase, const unsigned j) const
  {
    unsigned pos = j, left = k;
    while (left > 0) {
      const unsigned wi = pos >> 5, bit = pos & 31;
      const unsigned take = (32 - bit < left) ? (32 - bit) : left;
      uint32_t       m    = nmask[maskWordBase + wi] >> bit;
      if (take < 32) m &= (1u << take) - 1u;
      if (m) return true;
      pos += take;
      left -= take;
    }
    return false;
  }

  /// hash of a key: bucket from the low bits, 6-bit tag from the high bits
  template <int KW>
  WV_DEV uint32_t keyHash(const Key<KW>& key) const
  {
    const unsigned kw = (k + 15) >> 4;
    uint32_t       h  = 0x811C9DC5u;
    for (int i = 0; i < KW; ++i)
      if (unsigned(i) < kw) h = hashMix(h, key.w[i]);
    h ^= h >> 13;
    h *= 0x85EBCA6Bu;
    h ^= h >> 16;
    return h;
  }

  /// node id of `key` or ASM_NONE.  One 16-byte read per probed bucket; an empty slot ends the search (slots of a bucket
  /// fill in order and are never freed), a tag mismatch skips the slot without a key compare.
  template <int KW>
  WV_DEV unsigned lookup(const Key<KW>& key) const
  {
    const uint32_t h   = keyHash(key);
    const unsigned tag = h >> 26;
    unsigned       b   = h & (LN_BUCKETS - 1);
    for (unsigned probe = 0; probe < LN_BUCKETS; ++probe) {
      const LBucket bk = *reinterpret_cast<const LBucket*>(slots + 4 * b);
      for (int i = 0; i < 4; ++i) {
        const uint32_t s = bk.s[i];
        if (s == LN_EMPTY) return ASM_NONE;
        if ((s >> 26) == tag && Assembler::keyEq(keyAt<KW>(s & 0x7fffu), key)) return (s >> 15) & 0x7ffu;
      }
      b = (b + 1) & (LN_BUCKETS - 1);
    }
    return ASM_NONE;
  }

  // ------------------------------------------------------------------------------------------------
  // stage 0: bytes -> 2 bit + N bitmap, straight into LDS.  Returns false if the locus does not fit this path.
  // ------------------------------------------------------------------------------------------------
  WV_DEV bool pack(const unsigned locus)
  {
    const unsigned lane   = unsigned(wv::lane());
    const unsigned rBegin = P.locus_read_begin[locus], rEnd = P.locus_read_begin[locus + 1];
    nNormal               = rEnd - rBegin;
    if (nNormal + 2 * P.opt.maxAssemblyCount > LN_MAX_READS) return false;
    W = (nNormal + 2 * P.opt.maxAssemblyCount + 63) / 64;
    if (W == 0) W = 1;
    unsigned cw = 0, mw = 0;
    bool     tooLong = false;
    for (unsigned base = 0; base < nNormal; base += 64) {
      const unsigned r   = base + lane;
      unsigned       len = 0;
      if (r < nNormal) len = unsigned(P.read_off[rBegin + r + 1] - P.read_off[rBegin + r]);
      if (len > 0xffffu) tooLong = true;
      const unsigned myC = (r < nNormal) ? (len + 15) / 16 + 1 : 0u;
      const unsigned myM = (r < nNormal) ? (len + 31) / 32 + 1 : 0u;
      unsigned       sc = myC, sm = myM;
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned oc = wv::shfl(sc, wv::lane() - off), om = wv::shfl(sm, wv::lane() - off);
        if (wv::lane() >= off) {
          sc += oc;
          sm += om;
        }
      }
      const unsigned cwo = cw + sc - myC;
      if (r < nNormal && cwo <= 0x7ffu) rd[r] = cwo | ((len & 0xffffu) << 11);
      // the N-bitmap offset of a read is a function of its code offset only when every read has the same rounding; keep
      // it explicit: stash it in the HBM descriptor array of the general path (read back only for reads that hold an N)
      if (r < nNormal) A.rd_mw[r] = mw + sm - myM;
      cw += wv::readlane(sc, 63);
      mw += wv::readlane(sm, 63);
    }
    if (wv::any(tooLong) || cw + 2 > LN_MAX_PILE) return false;
    nCodeWords = cw;
    nMaskWords = mw;
    nmask      = codes + ((cw + 2 + 3) & ~3u);
    nodesOff   = LN_OFF_DYN + 4 * (((cw + 2 + 3) & ~3u) + ((mw + 2 + 3) & ~3u));
    if (nodesOff + 4096 > LN_BUDGET) return false;
    nodes = reinterpret_cast<LRec*>(lds + nodesOff);
    for (unsigned i = lane; i < mw + 2; i += 64) nmask[i] = 0;
    wv::sync();
    bool bad = false;
    // 8 lanes per read, 8 reads per pass (see Assembler::packNormalReads)
    for (unsigned base = 0; base < nNormal; base += 8) {
      const unsigned r = base + (lane >> 3);
      if (r >= nNormal) continue;
      const uint8_t* src = P.bases + P.read_off[rBegin + r];
      const unsigned d = rd[r], cwo = d & 0x7ffu, len = (d >> 11) & 0xffffu;
      const unsigned mwo = A.rd_mw[r];
      const unsigned nCw = (len + 15) / 16 + 1;
      bool           sawN = false;
      for (unsigned wi = (lane & 7); wi < nCw; wi += 8) {
        uint32_t code = 0, nbits = 0;
        if (wi * 16 < len) {
          const uintptr_t addr  = reinterpret_cast<uintptr_t>(src + wi * 16);
          const uint32_t* ap    = reinterpret_cast<const uint32_t*>(addr & ~uintptr_t(3));
          const unsigned  shift = unsigned(addr & 3) * 8;
          uint32_t        dw[5];
          for (int q = 0; q < 5; ++q) dw[q] = ap[q];
          for (unsigned q = 0; q < 4; ++q) {
            const uint32_t four = shift ? ((dw[q] >> shift) | (dw[q + 1] << (32 - shift))) : dw[q];
            for (unsigned b4 = 0; b4 < 4; ++b4) {
              const unsigned b = q * 4 + b4;
              const unsigned i = wi * 16 + b;
              unsigned       c = 0;
              if (i < len) {
                c = baseCode(uint8_t(four >> (8 * b4)));
                if (c == 5) bad = true;
                if (c >= 4) {
                  nbits |= (1u << b);
                  c = 0;
                }
              }
              code |= c << (30 - 2 * b);
            }
          }
        }
        codes[cwo + wi] = code;
        if (nbits) {
          wv::atomic_or(&nmask[mwo + (wi >> 1)], (wi & 1) ? (nbits << 16) : nbits);
          sawN = true;
        }
      }
      if (sawN) wv::atomic_or(&rd[r], 1u << 27);
    }
    if (wv::any(bad)) return false;  // the general path reports ASM_E_ALPHABET
    wv::sync();
    return true;
  }

  WV_DEV uint64_t normalMask(const unsigned w) const
  {
    const unsigned lo = w * 64;
    if (nNormal >= lo + 64) return ~uint64_t(0);
    if (nNormal <= lo) return 0;
    return (uint64_t(1) << (nNormal - lo)) - 1;
  }

  // ------------------------------------------------------------------------------------------------
  // k-mer graph (getKmerCounts :506-550 + successor / predecessor links)
  // ------------------------------------------------------------------------------------------------
  template <int KW>
  WV_DEV bool buildGraph()
  {
    const unsigned lane = unsigned(wv::lane());
    for (unsigned s = lane; s < LN_SLOTS; s += 64) slots[s] = LN_EMPTY;
    if (lane == 0) vars[0] = 0;
    nNodes = 0;
    wv::sync();
    bool fail = false;
    for (unsigned r = 0; r < nNormal && !fail; ++r) {
      const unsigned d = rd[r], cwo = d & 0x7ffu, len = (d >> 11) & 0xffffu;
      if (len < k) continue;
      const bool     rdHasN = (d >> 27) & 1u;
      const unsigned mwo    = rdHasN ? A.rd_mw[r] : 0u;
      for (unsigned j0 = 0; j0 + k <= len && !fail; j0 += 64) {
        // room for this step's worst case: 64 new records below, 64 new bitsets above
        if (nodesOff + 16 * (nNodes + 64) + 16 * 64 + poolBytes() > LN_BUDGET || nNodes + 64 > LN_MAX_NODES) {
          fail = true;
          break;
        }
        const unsigned j    = j0 + lane;
        const unsigned pb   = cwo * 16 + j;
        bool           have = false, won = false;
        unsigned       slot = 0, foundId = LN_ID_PENDING;
        if (j + k <= len && !(rdHasN && windowHasN(mwo, j))) {
          const Key<KW>  key = keyAt<KW>(pb);
          const uint32_t h   = keyHash(key);
          const unsigned tag = h >> 26;
          unsigned       b   = h & (LN_BUCKETS - 1);
          for (unsigned probe = 0; probe < LN_BUCKETS && !have; ++probe) {
            LBucket bk = *reinterpret_cast<const LBucket*>(slots + 4 * b);
            for (int i = 0; i < 4 && !have; ++i) {
              uint32_t s = bk.s[i];
              if (s == LN_EMPTY) {
                const uint32_t mine = pb | (LN_ID_PENDING << 15) | (tag << 26);
                s                   = wv::atomic_cas(&slots[4 * b + i], LN_EMPTY, mine);
                if (s == LN_EMPTY) {
                  have = won = true;
                  slot       = 4 * b + i;
                  break;
                }
              }
              if ((s >> 26) == tag && Assembler::keyEq(keyAt<KW>(s & 0x7fffu), key)) {
                have    = true;
                slot    = 4 * b + i;
                foundId = (s >> 15) & 0x7ffu;
              }
            }
            b = (b + 1) & (LN_BUCKETS - 1);
          }
          if (!have) fail = true;  // table full (cannot happen below LN_MAX_NODES)
        }
        const uint64_t m  = wv::ballot(won);
        const unsigned id = nNodes + unsigned(wv::popc(m & ((uint64_t(1) << lane) - 1)));
        nNodes += unsigned(wv::popc(m));
        if (won) {
          slots[slot] = (slots[slot] & ~(0x7ffu << 15)) | (id << 15);
          LRec rec;
          rec.w0    = uint64_t(pb & 0xfffu) << 52;
          rec.w1    = (uint64_t(r) << 44) | (uint64_t(pb >> 12) << 60);  // support = {r}
          nodes[id] = rec;
        }
        if (m) wv::sync();
        if (have && !won) {
          if (foundId == LN_ID_PENDING) foundId = (slots[slot] >> 15) & 0x7ffu;
          // add read r to the word's support
          uint32_t* hi = reinterpret_cast<uint32_t*>(&nodes[foundId].w1) + 1;
          while (true) {
            const uint32_t cur = wv::atomic_load(hi);
            const unsigned ref = (cur >> 12) & 0xffffu;
            if (ref & LN_FAT) {
              unsigned long long* w = reinterpret_cast<unsigned long long*>(&pool(ref & 0x7fffu)->w[r >> 6]);
              wv::atomic_or(w, (unsigned long long)(uint64_t(1) << (r & 63)));
              break;
            }
            if (ref == r) break;
            // second read of a so far single-read word: give it a bitset
            const unsigned f = wv::atomic_add(&vars[0], 1u);
            LSet           v;
            v.w[0] = ((ref < 64) ? (uint64_t(1) << ref) : 0) | ((r < 64) ? (uint64_t(1) << r) : 0);
            v.w[1] = ((ref >= 64) ? (uint64_t(1) << (ref - 64)) : 0) | ((r >= 64) ? (uint64_t(1) << (r - 64)) : 0);
            *pool(f) = v;
            const uint32_t want = (cur & ~(0xffffu << 12)) | ((LN_FAT | f) << 12);
            if (wv::atomic_cas(hi, cur, want) == cur) break;
            // (a twin of this k-mer in the same read got there first: its bitset already holds r; ours is abandoned)
          }
        }
        if (wv::any(fail)) fail = true;
      }
    }
    wv::sync();
    if (fail) return false;

    // counts, successor lookups, predecessor scatter
    for (unsigned nb = 0; nb < nNodes; nb += 64) {
      const unsigned nd = nb + lane;
      if (nd < nNodes) {
        const LRec     rec = nodes[nd];
        uint64_t       s0, s1;
        supOf(rec.w1, s0, s1);
        const unsigned cnt = unsigned(wv::popc(s0 & normalMask(0))) + unsigned(wv::popc(s1 & normalMask(1)));
        const Key<KW>  key = keyAt<KW>(recPb(rec.w0, rec.w1));
        uint64_t       w0  = (rec.w0 & (uint64_t(0xfff) << 52)) | (uint64_t(cnt > 255 ? 255 : cnt) << 44);
        const unsigned firstBase = key.w[0] >> 30;
        bool           selfLoop  = false;
        for (unsigned c = 0; c < 4; ++c) {
          const unsigned s = lookup<KW>(A.template keyShiftAppend<KW>(key, c));
          if (s == ASM_NONE) continue;
          w0 |= uint64_t(s + 1) << (11 * c);
          if (s == nd) selfLoop = true;
          wv::atomic_or(reinterpret_cast<unsigned long long*>(&nodes[s].w1), (unsigned long long)(uint64_t(nd + 1) << (11 * firstBase)));
        }
        nodes[nd].w0 = w0;
        if (selfLoop) wv::atomic_or(&repeat_bits[nd >> 5], 1u << (nd & 31));
      }
    }
    wv::sync();
    // seed eligibility (:679-682)
    for (unsigned nb = 0; nb < ((nNodes + 63) & ~63u); nb += 64) {
      const unsigned nd = nb + lane;
      const uint64_t m  = wv::ballot(nd < nNodes && recCnt(nodes[nd < nNodes ? nd : 0].w0) >= P.opt.minCoverage);
      if (lane < 2) unused_bits[(nb >> 5) + lane] = uint32_t(m >> (32 * lane));
    }
    wv::sync();
    return true;
  }

  WV_DEV bool isUnused(const unsigned nd) const { return (unused_bits[nd >> 5] >> (nd & 31)) & 1u; }
  WV_DEV bool isRepeat(const unsigned nd) const { return (repeat_bits[nd >> 5] >> (nd & 31)) & 1u; }

  WV_DEV char*    scratch() const { return lds + nodesOff + 16 * nNodes; }
  WV_DEV unsigned scratchBytes() const { return LN_BUDGET - poolBytes() - (nodesOff + 16 * nNodes); }

  // ------------------------------------------------------------------------------------------------
  // cycle test (see Assembler::graphHasCycle): two-sided Kahn peel, per-node state byte {in:3, out:3, peeled, simple},
  // one append-only queue.  Returns 0 acyclic, 1 cyclic, 2 scratch too small.
  // ------------------------------------------------------------------------------------------------
  WV_DEV int graphHasCycle()
  {
    const unsigned lane   = unsigned(wv::lane());
    const unsigned stDw   = (nNodes + 3) / 4;
    const unsigned need   = 4 * stDw + 2 * nNodes + 16;
    if (need > scratchBytes()) return 2;
    uint32_t* st    = reinterpret_cast<uint32_t*>(scratch());
    uint16_t* queue = reinterpret_cast<uint16_t*>(st + stDw);
    uint32_t* qTail = &vars[1];
    if (lane == 0) *qTail = 0;
    for (unsigned w = lane; w < stDw; w += 64) st[w] = 0;
    wv::sync();
    auto degrees = [&](const LRec& rec, const unsigned nd, unsigned& id, unsigned& od, unsigned& only) {
      id = od = 0;
      only    = ASM_NONE;
      for (unsigned c = 0; c < 4; ++c) {
        const unsigned s = linkId(rec.w0, c), p = linkId(rec.w1, c);
        if (s != ASM_NONE && s != nd) {
          od++;
          only = s;
        }
        if (p != ASM_NONE && p != nd) id++;
      }
    };
    for (unsigned nb = 0; nb < nNodes; nb += 64) {
      const unsigned nd = nb + lane;
      if (nd >= nNodes) continue;
      const LRec rec = nodes[nd];
      unsigned   id, od, only;
      degrees(rec, nd, id, od, only);
      const bool src = (id == 0 || od == 0);
      bool       simple = false;
      if (od == 1 && only == nd + 1 && nd + 1 < nNodes) {
        unsigned id2, od2, only2;
        degrees(nodes[nd + 1], nd + 1, id2, od2, only2);
        simple = (id2 == 1);
      }
      const unsigned v = id | (od << 3) | (src ? 0x40u : 0u) | (simple ? 0x80u : 0u);
      wv::atomic_or(&st[nd >> 2], v << (8 * (nd & 3)));
      if (src) queue[wv::atomic_add(qTail, 1u)] = uint16_t(nd);
    }
    wv::sync();
    auto stateOf = [&](const unsigned n) { return (wv::atomic_load(&st[n >> 2]) >> (8 * (n & 3))) & 0xffu; };
    unsigned head = 0, removed = 0;
    while (true) {
      const unsigned tail = wv::first(wv::atomic_load(qTail));
      if (tail == head) break;
      removed += tail - head;
      for (unsigned i = head + lane; i < tail; i += 64) {
        const unsigned nd  = queue[i];
        const LRec     rec = nodes[nd];
        for (unsigned c = 0; c < 4; ++c) {
          const unsigned s = linkId(rec.w0, c);
          if (s != ASM_NONE && s != nd) {
            const unsigned sh  = 8 * (s & 3);
            const unsigned old = wv::atomic_sub(&st[s >> 2], 1u << sh) >> sh;
            if ((old & 0x7u) == 1u && !(wv::atomic_or(&st[s >> 2], 0x40u << sh) & (0x40u << sh))) queue[wv::atomic_add(qTail, 1u)] = uint16_t(s);
          }
          const unsigned p = linkId(rec.w1, c);
          if (p != ASM_NONE && p != nd) {
            const unsigned sh  = 8 * (p & 3);
            const unsigned old = wv::atomic_sub(&st[p >> 2], 8u << sh) >> sh;
            if ((old & 0x38u) == 8u && !(wv::atomic_or(&st[p >> 2], 0x40u << sh) & (0x40u << sh))) queue[wv::atomic_add(qTail, 1u)] = uint16_t(p);
          }
        }
      }
      wv::sync();
      // stretch peel (see Assembler::graphHasCycle): runs of simple edges leave 64 nodes at a time
      const unsigned newTail = wv::first(wv::atomic_load(qTail));
      if (newTail - tail > 0 && newTail - tail <= ASM_STRETCH_MAX) {
        for (unsigned qi = tail; qi < newTail; ++qi) {
          const unsigned f  = wv::first(unsigned(queue[qi]));
          const unsigned sf = wv::first(stateOf(f));
          for (int dir = 0; dir < 2; ++dir) {
            if (dir == 0 ? ((sf & 0x7u) != 0) : ((sf & 0x38u) != 0)) continue;
            unsigned c = f, total = 0;
            while (true) {
              bool ok = false;
              if (dir == 0) {
                const unsigned a = c + lane;
                if (a + 1 < nNodes) ok = (stateOf(a) & 0x80u) && !(stateOf(a + 1) & 0x40u);
              } else if (c >= lane + 1) {
                const unsigned b = c - lane - 1;
                ok               = (stateOf(b) & 0x80u) && !(stateOf(b) & 0x40u);
              }
              const uint64_t good = wv::ballot(ok);
              const unsigned take = (~good == 0) ? 64u : unsigned(wv::ctz(~good));
              if (take == 0) break;
              if (lane < take) {
                const unsigned n = (dir == 0) ? (c + lane + 1) : (c - lane - 1);
                wv::atomic_or(&st[n >> 2], 0x40u << (8 * (n & 3)));
              }
              total += take;
              c = (dir == 0) ? (c + take) : (c - take);
              wv::sync();
              if (take < 64) break;
            }
            if (total > 0) {
              if (lane == 0) queue[wv::atomic_add(qTail, 1u)] = uint16_t(c);
              removed += total - 1;
              wv::sync();
            }
          }
        }
      }
      head = tail;
    }
    return (removed != nNodes) ? 1 : 0;
  }

  // ------------------------------------------------------------------------------------------------
  // seed order (:686-696): count descending, k-mer ascending
  // ------------------------------------------------------------------------------------------------
  typedef Key<ASM_MAX_KW> GKey;

  WV_DEV unsigned selectSeed()
  {
    const unsigned lane = unsigned(wv::lane());
    unsigned       best = 0;
    for (unsigned nd = lane; nd < nNodes; nd += 64)
      if (isUnused(nd)) {
        const unsigned c = recCnt(nodes[nd].w0);
        best             = (c > best) ? c : best;
      }
    best = Assembler::waveMax(best);
    if (best == 0) return ASM_NONE;
    unsigned mine = ASM_NONE;
    GKey     mineKey;
    for (int i = 0; i < ASM_MAX_KW; ++i) mineKey.w[i] = 0xffffffffu;
    for (unsigned nd = lane; nd < nNodes; nd += 64) {
      const LRec rec = nodes[nd];
      if (isUnused(nd) && recCnt(rec.w0) == best) {
        const GKey key = keyAt<ASM_MAX_KW>(recPb(rec.w0, rec.w1));
        if (mine == ASM_NONE || Assembler::keyLess(key, mineKey)) {
          mine    = nd;
          mineKey = key;
        }
      }
    }
    for (int off = 1; off < 64; off <<= 1) {
      const int      src = wv::lane() ^ off;
      const unsigned on  = wv::shfl(mine, src);
      GKey           ok;
      for (int i = 0; i < ASM_MAX_KW; ++i) ok.w[i] = wv::shfl(mineKey.w[i], src);
      if (on != ASM_NONE && (mine == ASM_NONE || Assembler::keyLess(ok, mineKey))) {
        mine    = on;
        mineKey = ok;
      }
    }
    return mine;
  }

  /// next <= T unused words in exact seed order into tent[0..nT) (u16 node ids, LDS); see Assembler::selectTentative.
  /// `area`/`areaBytes`: scratch for the histogram (1 KB) and the raw candidate list.
  WV_DEV unsigned selectTentative(const unsigned T, uint16_t* tent, char* area, const unsigned areaBytes)
  {
    const unsigned lane = unsigned(wv::lane());
    if (T == 1 || areaBytes < 1024 + 2 * TENT_CAP) {
      const unsigned s = selectSeed();
      if (s == ASM_NONE) return 0;
      if (lane == 0) tent[0] = uint16_t(s);
      wv::sync();
      return 1;
    }
    uint32_t* hist = reinterpret_cast<uint32_t*>(area);
    uint16_t* raw  = reinterpret_cast<uint16_t*>(area + 1024);
    // count level: counts are <= 255 here (no pseudo reads) -> one 256-bin histogram over the unused words
    for (unsigned i = lane; i < 256; i += 64) hist[i] = 0;
    wv::sync();
    unsigned U = 0;
    for (unsigned nb = 0; nb < nNodes; nb += 64) {
      const unsigned nd = nb + lane;
      if (nd < nNodes && isUnused(nd)) {
        wv::atomic_add(&hist[recCnt(nodes[nd].w0)], 1u);
        U++;
      }
    }
    U = Assembler::waveSum(U);
    wv::sync();
    if (U == 0) return 0;
    unsigned cStar = 1, pStar = 0xffffffffu;
    if (U > T) {
      unsigned carry = 0, found = 0;
      for (unsigned top = 256; top > 0 && !found; top -= 64) {
        const unsigned bin = top - 1 - lane;
        unsigned       inc = hist[bin];
        for (int off = 1; off < 64; off <<= 1) {
          const unsigned o = wv::shfl(inc, int(lane) - off);
          if (int(lane) >= off) inc += o;
        }
        const uint64_t reach = wv::ballot(carry + inc >= T);
        if (reach) {
          cStar = top - 1 - unsigned(wv::ctz(reach));
          found = 1;
        }
        carry += wv::shfl(inc, 63);
      }
      if (!found || cStar == 0) cStar = 1;
      wv::sync();
      unsigned above = 0;
      for (unsigned c = cStar + 1 + lane; c < 256; c += 64) above += hist[c];
      unsigned need = T - Assembler::waveSum(above);
      wv::sync();
      // tie level: radix select on the 16-base prefix among the words with count == cStar
      unsigned prefix = 0;
      for (int shift = 24; shift >= 0; shift -= 8) {
        for (unsigned i = lane; i < 256; i += 64) hist[i] = 0;
        wv::sync();
        for (unsigned nb = 0; nb < nNodes; nb += 64) {
          const unsigned nd = nb + lane;
          if (nd >= nNodes || !isUnused(nd)) continue;
          const LRec rec = nodes[nd];
          if (recCnt(rec.w0) != cStar) continue;
          const unsigned p = codes16(recPb(rec.w0, rec.w1));
          if (shift < 24 && (p >> (shift + 8)) != (prefix >> (shift + 8))) continue;
          wv::atomic_add(&hist[(p >> shift) & 255u], 1u);
        }
        wv::sync();
        const unsigned h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
        const unsigned mine = h0 + h1 + h2 + h3;
        unsigned       inc  = mine;
        for (int off = 1; off < 64; off <<= 1) {
          const unsigned o = wv::shfl(inc, int(lane) - off);
          if (int(lane) >= off) inc += o;
        }
        const uint64_t reach  = wv::ballot(inc >= need);
        const int      ln     = wv::ctz(reach);
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
    // gather the survivors
    unsigned total = 0;
    for (unsigned nb = 0; nb < ((nNodes + 63) & ~63u); nb += 64) {
      const unsigned nd  = nb + lane;
      bool           sel = false;
      if (nd < nNodes && isUnused(nd)) {
        const LRec     rec = nodes[nd];
        const unsigned c   = recCnt(rec.w0);
        sel                = (U <= T) || (c > cStar) || (c == cStar && codes16(recPb(rec.w0, rec.w1)) <= pStar);
      }
      const uint64_t m   = wv::ballot(sel);
      const unsigned pos = total + unsigned(wv::popc(m & ((uint64_t(1) << lane) - 1)));
      if (sel && pos < TENT_CAP) raw[pos] = uint16_t(nd);
      total += unsigned(wv::popc(m));
    }
    wv::sync();
    if (total > TENT_CAP) {
      const unsigned s = selectSeed();
      if (s == ASM_NONE) return 0;
      if (lane == 0) tent[0] = uint16_t(s);
      wv::sync();
      return 1;
    }
    // exact rank inside the list
    const unsigned keep = (total < T) ? total : T;
    for (unsigned i = lane; i < total; i += 64) {
      const unsigned x  = raw[i];
      const LRec     rx = nodes[x];
      const unsigned cx = recCnt(rx.w0), pbx = recPb(rx.w0, rx.w1), px = codes16(pbx);
      unsigned       rank = 0;
      for (unsigned j = 0; j < total; ++j) {
        if (j == i) continue;
        const unsigned y  = raw[j];
        const LRec     ry = nodes[y];
        const unsigned cy = recCnt(ry.w0);
        bool           before = (cy > cx);
        if (cy == cx) {
          const unsigned pby = recPb(ry.w0, ry.w1), py = codes16(pby);
          before             = (py < px) || (py == px && Assembler::keyLess(keyAt<ASM_MAX_KW>(pby), keyAt<ASM_MAX_KW>(pbx)));
        }
        if (before) rank++;
      }
      if (rank < keep) tent[rank] = uint16_t(x);
    }
    wv::sync();
    return keep;
  }

  // ------------------------------------------------------------------------------------------------
  // one speculative round: lane t < nT walks tent[t] (see walk_lanes.hpp for the scheme and the reference lines)
  // ------------------------------------------------------------------------------------------------
  struct Cand {
    uint64_t w0, w1, s0, s1;
    unsigned vis;
  };

  WV_DEV void walkLanes(const unsigned nT, const uint16_t* tent, uint32_t* visBase, const unsigned useWords)
  {
    const unsigned lane = unsigned(wv::lane());
    for (unsigned i = lane; i < nT * useWords; i += 64) visBase[i] = 0;
    wv::sync();
    const bool     has  = lane < nT;
    const unsigned seed = has ? unsigned(tent[lane]) : 0u;
    uint32_t*      vis  = visBase + size_t(has ? lane : 0) * useWords;
    const unsigned seqWords = P.max_contig_len / 16 + 2;
    uint32_t*      rightBuf = reinterpret_cast<uint32_t*>(A.lane_seq) + size_t(lane) * 2 * seqWords;
    uint32_t*      leftBuf  = rightBuf + seqWords;
    uint32_t       accR = 0, accL = 0;
    uint64_t       S0 = 0, S1 = 0, R0 = 0, R1 = 0;
    bool           active = has, rep = false, tooLong = false, seedRepeat = false;
    unsigned       mode = 0, cur = seed, consOffset = 0, nLeft = 0, nRight = 0;
    int            consEnd = 0, consBegin = 0;
    LRec           seedRec = {0, 0};
    if (has) {
      seedRec = nodes[seed];
      supOf(seedRec.w1, S0, S1);
      vis[seed >> 5] |= (1u << (seed & 31));
      if (isRepeat(seed)) {  // :172-179
        seedRepeat = true;
        rep        = true;
        active     = false;
      } else {  // unselected siblings of the seed reject the contig (:185-210)
        const unsigned seedPb   = recPb(seedRec.w0, seedRec.w1);
        const GKey     key      = keyAt<ASM_MAX_KW>(seedPb);
        const unsigned lastBase = baseAt(seedPb + k - 1);
        for (unsigned c = 0; c < 4; ++c) {
          if (c == lastBase) continue;
          GKey sib = key;
          A.keySetBase(sib, k - 1, c);
          const unsigned n = lookup<ASM_MAX_KW>(sib);
          if (n != ASM_NONE) {
            uint64_t a, b;
            supOf(nodes[n].w1, a, b);
            R0 |= a;
            R1 |= b;
          }
        }
      }
    }
    // the (<= 4) existing candidates behind a packed link word, compacted in alphabet order
    unsigned dM = 0, dSyms = 0, dNode[4];
    Cand     dC[2];
    auto fetch = [&](const bool on, const uint64_t linkWord, unsigned& m, unsigned& syms, unsigned (&node)[4], Cand (&pre)[2]) {
      m = syms = 0;
      for (unsigned i = 0; i < 4; ++i) node[i] = ASM_NONE;
      if (on) {
        for (unsigned c = 0; c < 4; ++c) {
          const unsigned id = linkId(linkWord, c);
          if (id == ASM_NONE) continue;
          for (unsigned i = 0; i < 4; ++i)
            if (i == m) node[i] = id;
          syms |= c << (2 * m);
          m++;
        }
      }
      for (unsigned i = 0; i < 2; ++i) {
        pre[i].w0 = pre[i].w1 = pre[i].s0 = pre[i].s1 = 0;
        pre[i].vis                                    = 0;
        if (!wv::any(i < m)) continue;
        if (i < m) {
          const LRec r = nodes[node[i]];
          pre[i].w0    = r.w0;
          pre[i].w1    = r.w1;
          supOf(r.w1, pre[i].s0, pre[i].s1);
          pre[i].vis = vis[node[i] >> 5];
        }
      }
    };
    fetch(active, seedRec.w0, dM, dSyms, dNode, dC);

    while (wv::any(active)) {
      const bool isEnd = (mode == 0);
      unsigned   maxBaseCount = 0, maxCnt = 0, maxNode = ASM_NONE, maxSym = 0, maxVis = 0;
      uint64_t   maxW0 = 0, maxW1 = 0;
      uint64_t   maxWR0 = 0, maxWR1 = 0, maxCW0 = 0, maxCW1 = 0, rm0 = 0, rm1 = 0, add0 = 0, add1 = 0;
      for (unsigned i = 0; i < 4; ++i) {
        const bool live = active && i < dM;
        Cand       c;
        if (i < 2) {
          c = dC[i];
        } else {
          c.w0 = c.w1 = c.s0 = c.s1 = 0;
          c.vis                     = 0;
          if (!wv::any(live)) continue;
          if (live) {
            const LRec r = nodes[dNode[i]];
            c.w0         = r.w0;
            c.w1         = r.w1;
            supOf(r.w1, c.s0, c.s1);
            c.vis = vis[dNode[i] >> 5];
          }
        }
        if (!live) continue;
        const unsigned cnt = unsigned(wv::popc(S0 & c.s0)) + unsigned(wv::popc(S1 & c.s1));
        if (cnt == 0) continue;  // :280
        const uint64_t SH0 = maxCW0 & c.s0, SH1 = maxCW1 & c.s1;
        if (cnt > maxCnt) {  // :283-316
          rm0 |= maxCW0 & ~SH0;
          rm1 |= maxCW1 & ~SH1;
          add0 |= maxWR0 & ~SH0;
          add1 |= maxWR1 & ~SH1;
          maxWR0  = c.s0;
          maxWR1  = c.s1;
          maxCW0  = S0 & c.s0;
          maxCW1  = S1 & c.s1;
          maxCnt  = cnt;
          maxSym  = (dSyms >> (2 * i)) & 3;
          maxNode = dNode[i];
          maxW0   = c.w0;
          maxW1   = c.w1;
          maxVis  = c.vis;
        } else {  // :317-335
          rm0 |= (S0 & c.s0) & ~SH0;
          rm1 |= (S1 & c.s1) & ~SH1;
          add0 |= c.s0 & ~SH0;
          add1 |= c.s1 & ~SH1;
        }
      }
      if (maxNode != ASM_NONE) maxBaseCount = recCnt(maxW0);
      bool stop = false, extend = false;
      if (active) {
        if (maxBaseCount < P.opt.minCoverage) {  // :343
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
      // backward-check supports of the chosen word (:377-427): its neighbours against the walking direction
      const uint64_t backWord = isEnd ? maxW1 : maxW0;
      uint64_t       b0 = 0, b1 = 0;
      if (wv::any(extend)) {
        for (unsigned c = 0; c < 4; ++c) {
          const unsigned n    = extend ? linkId(backWord, c) : ASM_NONE;
          const bool     want = (n != ASM_NONE && n != cur && n != maxNode);  // :381, :389
          if (!wv::any(want)) continue;
          if (want) {
            uint64_t a, b;
            supOf(nodes[n].w1, a, b);
            b0 |= a & ~maxCW0;  // :400-414
            b1 |= b & ~maxCW1;
          }
        }
      }
      if (extend) vis[maxNode >> 5] = maxVis | (1u << (maxNode & 31));  // :482-484, before the next fetch reads the bitmap
      const bool toLeft = stop && (mode == 0);  // :488-491
      unsigned   nM, nSyms, nNode[4];
      Cand       nC[2];
      fetch(extend || toLeft, toLeft ? seedRec.w1 : (isEnd ? maxW0 : maxW1), nM, nSyms, nNode, nC);

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
      dM    = nM;
      dSyms = nSyms;
      for (unsigned i = 0; i < 4; ++i) dNode[i] = nNode[i];
      dC[0] = nC[0];
      dC[1] = nC[1];
    }

    if (has) {
      uint64_t* lb = A.lane_bits + size_t(lane) * 2 * WQ_MAX;
      lb[0]          = S0;
      lb[1]          = S1;
      lb[WQ_MAX]     = R0;
      lb[WQ_MAX + 1] = R1;
      if (nRight & 15) rightBuf[nRight >> 4] = accR;
      if (nLeft & 15) leftBuf[nLeft >> 4] = accL;
      int32_t* m = A.lane_meta + lane * 8;
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

  /// buildContigs' contig loop (:685-713) by speculative rounds.  Returns 0 = all contigs built without a repeat hit,
  /// 1 = some walk hit a repeat (the reference goes on to the next word length), 2 = does not fit.
  WV_DEV int contigRounds()
  {
    const unsigned lane     = unsigned(wv::lane());
    const unsigned capCand  = 2 * P.opt.maxAssemblyCount;
    const unsigned useWords = (nNodes + 31) / 32;
    nCand                   = 0;
    if (nNodes == 0) return 0;  // no word at this length (:522): no contig, no repeat
    // scratch map: [tent u16 x 64][visited bitmaps T x useWords dwords]; the selection's histogram and raw list share
    // the bitmap area (they are dead before the walks start)
    char*          sc      = scratch();
    const unsigned scBytes = scratchBytes();
    if (scBytes < 128 + 4 * useWords + 1024 + 2 * TENT_CAP) return 2;
    uint16_t* tent    = reinterpret_cast<uint16_t*>(sc);
    uint32_t* visBase = reinterpret_cast<uint32_t*>(sc + 128);
    const unsigned fit = (scBytes - 128) / (4 * useWords);
    bool           success = true;
    nCand                  = 0;
    while (nCand < capCand) {
      unsigned T = 1;
      if (nCand != 0) {
        T = (capCand - nCand) + 2;
        if (fit > T) T = fit;
        if (T > fit) T = fit;
      }
      if (T > 64) T = 64;
      if (T == 0) T = 1;
      const unsigned nT = selectTentative(T, tent, sc + 128, scBytes - 128);
      tick(5);
      if (nT == 0) break;
      walkLanes(nT, tent, visBase, useWords);
      tick(6);
      for (unsigned t = 0; t < nT && nCand < capCand; ++t) {
        const unsigned seed = tent[t];
        unsigned       seedFree = 0;
        if (lane == 0) seedFree = isUnused(seed) ? 1u : 0u;
        seedFree = wv::readlane(seedFree, 0);
        if (!seedFree) continue;
        const int32_t* m = A.lane_meta + t * 8;
        if (m[5]) return 2;  // contig longer than the workspace allows: the general path reports it
        const unsigned nLeft = unsigned(m[0]), nRight = unsigned(m[1]);
        const unsigned len    = nLeft + k + nRight;
        const LRec     sr     = nodes[seed];
        const unsigned seedPb = recPb(sr.w0, sr.w1);
        uint8_t*       outSeq = A.cand_seq + size_t(nCand) * P.max_contig_len;
        const unsigned  seqWords = P.max_contig_len / 16 + 2;
        const uint32_t* rightBuf = reinterpret_cast<const uint32_t*>(A.lane_seq) + size_t(t) * 2 * seqWords;
        const uint32_t* leftBuf  = rightBuf + seqWords;
        for (unsigned i = lane; i < len; i += 64) {
          unsigned code;
          if (i < nLeft) {
            const unsigned j = nLeft - 1 - i;
            code             = (leftBuf[j >> 4] >> (2 * (j & 15))) & 3;
          } else if (i < nLeft + k) {
            code = baseAt(seedPb + (i - nLeft));
          } else {
            const unsigned j = i - nLeft - k;
            code             = (rightBuf[j >> 4] >> (2 * (j & 15))) & 3;
          }
          outSeq[i] = uint8_t("ACGT"[code]);
        }
        if (lane < 2 * W) {
          const unsigned half = lane / W, w = lane % W;
          A.cand_bits[size_t(nCand) * 2 * W + lane] = A.lane_bits[size_t(t) * 2 * WQ_MAX + half * WQ_MAX + w];
        }
        if (lane == 0) {
          int32_t* meta = A.cand_meta + nCand * 4;
          meta[0]       = int(len);
          if (m[6]) {
            meta[1] = 0;
            meta[2] = int(k);
          } else {
            meta[1] = m[2];
            meta[2] = int(len) - m[3];
          }
        }
        const uint32_t* vis = visBase + size_t(t) * useWords;
        for (unsigned w = lane; w < useWords; w += 64) unused_bits[w] &= ~vis[w];
        wv::sync();
        if (m[4]) success = false;
        nCand++;
      }
    }
    return success ? 0 : 1;
  }

  /// the whole fast path for one locus.  LN_DONE: results emitted.  LN_PUNT: nothing emitted, run the general path.
  WV_DEV int run(const unsigned locus)
  {
    const unsigned minWL = P.locus_min_wl ? P.locus_min_wl[locus] : P.opt.minWordLength;
    const unsigned maxWL = P.locus_max_wl ? P.locus_max_wl[locus] : P.opt.maxWordLength;
    if (minWL == 0 || maxWL > 16u * ASM_MAX_KW || minWL > maxWL || 2 * P.opt.maxAssemblyCount > ASM_MAX_CAND) return LN_PUNT;
    if (P.opt.minCoverage > 255) return LN_PUNT;
    k     = minWL;
    A.k   = k;  // the key helpers of the general path read it
    tMark = wv::clock();
    if (!pack(locus)) return LN_PUNT;
    tick(0);
    for (unsigned i = unsigned(wv::lane()); i < 64; i += 64) {
      unused_bits[i] = 0;
      repeat_bits[i] = 0;
    }
    wv::sync();
    const unsigned kw = (k + 15) >> 4;
    bool           ok;
    if (kw <= 2)
      ok = buildGraph<2>();
    else if (kw <= 4)
      ok = buildGraph<4>();
    else
      ok = buildGraph<8>();
    if (!ok) return LN_PUNT;
    tick(1);
    if (graphHasCycle() != 0) return LN_PUNT;  // cyclic (exact repeat search) or no room: general path
    tick(3);
    const int rc = contigRounds();
    if (rc != 0) return LN_PUNT;  // a repeat hit asks for the next word length (pseudo reads), or no room
    // selectContigs + output through the general path's emitter (candidates sit in the HBM workspace in its layout)
    A.nNormal     = nNormal;
    A.W           = W;
    A.k           = k;
    A.nCand       = nCand;
    A.status      = ASM_OK;
    A.cyclicIters = 0;
    A.selectAndEmit(locus, 0, 1);
    tick(7);
    return LN_DONE;
  }
};

#ifndef MANTA_ASM_LDS_KERNEL_WAVES
#define MANTA_ASM_LDS_KERNEL_WAVES 1
#endif

/// persistent single-wave workgroups, LN_BUDGET bytes of dynamic LDS each; params as assemble_kernel.
/// P.counter[1] counts the loci the fast path handed to the general path (statistics).
WV_KERNEL_SINGLE void assemble_lds_kernel(const AsmParams P)
{
  uint8_t* wsBase = P.ws + uint64_t(wv::block_single()) * P.ws_stride;
  char*    lds    = wv::lds_single();
  while (true) {
    unsigned slot = 0;
    if (wv::lane() == 0) slot = wv::atomic_add(P.counter, 1u);
    slot = wv::first(slot);
    if (slot >= P.n_loci) break;
    const unsigned locus = P.locus_ids ? P.locus_ids[slot] : slot;
    Assembler      a(P, wsBase);
    int            rc = LN_PUNT;
    if (!(P.flags & ASM_FLAG_NO_LDS_PATH)) {
      LdsAssembler f(a, lds);
      rc = f.run(locus);
    }
    wv::sync();
    if (rc != LN_DONE) {
      if (wv::lane() == 0) wv::atomic_add(P.counter + 1, 1u);
      a.run(locus);
    }
    wv::sync();
  }
}

}  // namespace manta_dev
