// The iterative assembler's common case as an LDS pipeline of TWO kernels (assembly/IterativeAssembler.cpp:844-931; first word
// length of a locus whose k-mer graph is acyclic and small enough):
//
//   graph_kernel   one workgroup of LG_WAVES wavefronts per locus, LG_BUDGET bytes of LDS (two workgroups per CU).  Everything
//                  that is parallel over k-mer instances or over words: pack -> table pass (getKmerCounts :506-550) -> counts ->
//                  SORT of the words into the reference's seed order (:686-696: count descending, k-mer ascending) -> node
//                  records with successor / predecessor links in that numbering -> the compact graph (16 bytes per word + 16 per
//                  word with more than one read) goes to a per-locus slab in global memory.
//   contig_kernel  one single-wave workgroup per locus, only the compact graph in LDS (~22 KB for a config-2 locus: six per CU
//                  instead of three): cycle test, the contig loop (:685-713) on speculative lane-per-seed walks (:149-501),
//                  selectContigs (:722-842), output.
//
// Why two kernels.  The fused predecessor of this file (assemble_fast_kernel, rounds 2-3) held a locus' 52 KB of LDS from the
// first byte to the last contig: three loci per CU, and for two thirds of a locus' time only one wave of the workgroup had work
// (the walks are one dependent chain per lane).  The parallel half wants many waves and room for simple dense structures; the
// serial half wants as many loci in flight as possible and needs neither the pile nor the hash table.  Splitting at the graph
// gives each half its own occupancy, register budget and LDS map; the hand-over costs ~20 KB per locus, written and read once,
// coalesced.
//
// What the sort buys: with node ids in seed order, "the next unused seed" is the lowest set bit of a bitmap, the two lowest
// count tiers are an id range, the words with more than one read are the ids below nFat (their bitsets need no reference), and
// the contig kernel never looks at a key -- no pile, no table, no key compares in LDS.
//
// Anything this path does not cover -- a cycle, a repeat hit that asks for the next word length, > 108 reads, a graph that does
// not fit, bytes outside {A,C,G,T,N} -- goes onto the punt list; assemble_kernel, launched behind, picks it up from device
// memory.  Nothing is approximated.
#pragma once
#include "assemble_kernels.hpp"

namespace manta_dev {

static const unsigned LG_SLOTS     = 2048;
static const unsigned LG_BUCKETS   = LG_SLOTS / 4;
static const unsigned LG_MAX_NODES = 1843;   // 0.9 x slots; node ids are stored +1 in 11-bit link fields
static const unsigned LG_MAX_READS = 128;    // read sets of two qwords
static const unsigned LG_MAX_PILE  = 2046;   // code dwords: a packed base index must fit 15 bits
static const unsigned LG_EMPTY     = 0xffffffffu;
static const unsigned LG_FAT       = 0x800u; // support reference (12 bits): index into the bitset pool (else: a read)
static const unsigned LG_NO_SLOT   = 0xffffu;
static const unsigned LG_WAVES     = 8;      // wavefronts of a graph_kernel workgroup
static const unsigned LG_BUDGET    = 81920;  // its LDS: two workgroups per CU
static const unsigned LG_SIB_CAP   = 64;     // words without a predecessor that have siblings (side table)
static const unsigned LG_CLASSES   = 4;      // LDS size classes of contig_kernel (one launch each)

// graph_kernel LDS map (bytes)
static const unsigned LG_OFF_HDR   = 0;                          // u32[64]
static const unsigned LG_OFF_RD    = 256;                        // u32[128] read descriptors {code dword offset : 11, length : 16, has N : 1}
static const unsigned LG_OFF_RDM   = LG_OFF_RD + 512;            // u16[128] N-bitmap dword offset of a read
static const unsigned LG_OFF_DBASE = LG_OFF_RDM + 256;           // u32[256] digit bases of a sort pass
static const unsigned LG_OFF_WHIST = LG_OFF_DBASE + 1024;        // u32[LG_WAVES][256] per-wave digit counts / running offsets
static const unsigned LG_OFF_SLOTS = LG_OFF_WHIST + 1024 * LG_WAVES;  // u32[2048] {first occurrence : 15, tag : 17}
static const unsigned LG_OFF_SETS  = LG_OFF_SLOTS + 4 * LG_SLOTS;     // FSet[2048] read sets by slot; the node records later
static const unsigned LG_OFF_SORTA = LG_OFF_SETS + 16 * LG_SLOTS;     // u16[2048]
static const unsigned LG_OFF_SORTB = LG_OFF_SORTA + 2 * LG_SLOTS;     // u16[2048]
static const unsigned LG_OFF_KEYS  = LG_OFF_SORTB + 2 * LG_SLOTS;     // u32[2048] first 16 bases by slot; u16[2048] node id by slot after the sort
static const unsigned LG_OFF_CNT   = LG_OFF_KEYS + 4 * LG_SLOTS;      // u8[2048] count by slot (0x80 | read: the word's only read)
static const unsigned LG_OFF_DYN   = LG_OFF_CNT + LG_SLOTS;           // codes, N bitmap
static_assert(LG_OFF_DYN + 12288 <= LG_BUDGET, "graph_kernel LDS map");
// (dead after the sort: the per-wave histograms hold the sibling table and the chain labels of the speculation list)
static const unsigned LG_OFF_SIB   = LG_OFF_WHIST;                    // u16[LG_SIB_CAP][4]
static const unsigned LG_OFF_CHAIN = LG_OFF_WHIST + 8 * LG_SIB_CAP;   // u16[128] label, u16[128] distance

// header words (graph_kernel LDS)
enum {
  LG_H_SLOT = 0,   ///< the queue slot of the workgroup's current locus
  LG_H_FLAG = 1,   ///< pack: a byte outside the alphabet was seen; table: full
  LG_H_N    = 2,   ///< sort: number of words
  LG_H_NSIB = 3,
  LG_H_OFF_LO = 4, ///< slab offset of this locus in the arena (bytes)
  LG_H_OFF_HI = 5,
  LG_H_PUNT = 6,
  LG_H_TOT  = 8    ///< [4] scan: totals of the four digit quarters
};

struct alignas(16) FRec {
  uint64_t w0;  ///< successor links 4 x 11 (id+1; packed from field 0 up in A,C,G,T order, 0 ends the list) | count << 44 (8 bit) | first occurrence, low 12 bits << 52
  uint64_t w1;  ///< predecessor links 4 x 11 (same) | support reference << 44 (12 bit) | first base << 56 | last base << 58 | first occurrence, high 3 bits << 60 | self loop << 63
};
struct alignas(16) FSet {
  uint64_t w[2];
};
struct alignas(16) FBucket {
  uint32_t s[4];
};

/// what graph_kernel leaves in a locus' slab: this header, then FRec[nNodes], FSet[nFat], u16[64] speculation list,
/// u16[LG_SIB_CAP][4] sibling table, u32[codeWords] 2-bit pile (for the seeds' text)
struct alignas(16) LgHdr {
  uint32_t nNodes, nFat, k, nNormal;
  uint32_t nEligible;     ///< ids below it are seeds (:678-682)
  uint32_t nSpec;         ///< entries of the round-0 walk list (entry 0 = the first seed)
  uint32_t nSib, codeWords;
  uint32_t W, need, reserved0, reserved1;
  uint32_t pad[4];
};
static const unsigned LG_SLAB_FIXED = sizeof(LgHdr) + 128 + 8 * LG_SIB_CAP;
WV_HD uint64_t lgSlabBytes(const unsigned nNodes, const unsigned nFat, const unsigned codeWords)
{
  return LG_SLAB_FIXED + 16ull * nNodes + 16ull * nFat + 4ull * ((codeWords + 3) & ~3u);
}

// contig_kernel LDS map
static const unsigned CK_OFF_HDR    = 0;     // LgHdr
static const unsigned CK_OFF_UNUSED = 64;    // u32[64] "unusedWords" bitmap over node ids
static const unsigned CK_OFF_TENT   = 320;   // u16[128] seed list of the round
static const unsigned CK_OFF_SLOTND = 576;   // u16[64] word walked by cache slot s
static const unsigned CK_OFF_TBL    = 704;   // u8[64]
static const unsigned CK_OFF_SIB    = 768;   // u16[LG_SIB_CAP][4]
static const unsigned CK_OFF_RECS   = CK_OFF_SIB + 8 * LG_SIB_CAP;
/// LDS the contig kernel needs for a graph: records + the larger of {bitset pool, cycle-test state}
WV_HD unsigned ckNeed(const unsigned nNodes, const unsigned nFat)
{
  const unsigned kahn = 4 * ((nNodes + 3) / 4) + 2 * nNodes + 32;
  const unsigned pool = 16 * nFat;
  return CK_OFF_RECS + 16 * nNodes + ((pool > kahn) ? pool : ((kahn + 15) & ~15u));
}

/// parameters of the pipeline beyond AsmParams (both kernels take the pair)
struct LgParams;
struct LgArgs;
struct LgParams {
  uint8_t*            arena;       ///< slabs
  uint64_t            arena_cap;
  unsigned long long* arena_used;
  uint64_t*           slab_off;    ///< [n_loci]
  uint32_t*           class_ids;   ///< [LG_CLASSES][class_stride]
  uint32_t*           class_count; ///< [LG_CLASSES]
  uint32_t            class_stride;
  uint32_t            class_bytes[LG_CLASSES];  ///< ascending LDS budgets; 0 = unused class
  uint32_t            cls;         ///< contig_kernel: the class this launch runs
  uint32_t            reserved;
  uint8_t*            cws;         ///< contig_kernel workspaces
  uint64_t            cws_stride;
};

/// one kernel argument: the assembler's parameters and the pipeline's
struct LgArgs {
  AsmParams P;
  LgParams  G;
};

#ifdef MANTA_WAVE_EMU
/// test-build statistics of the speculation (tests/emu only): loci done, walk rounds, walks, accepted candidates, cache evictions
inline unsigned long long* fastStats()
{
  static unsigned long long v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  return v;
}
#define LG_STAT(i, n) do { if (wv::lane() == 0) fastStats()[i] += (n); } while (0)
#else
#define LG_STAT(i, n) do { } while (0)
#endif

// record fields
WV_DEV unsigned lgCnt(const uint64_t w0) { return unsigned(w0 >> 44) & 0xffu; }
WV_DEV unsigned lgPb(const uint64_t w0, const uint64_t w1) { return (unsigned(w0 >> 52) & 0xfffu) | ((unsigned(w1 >> 60) & 7u) << 12); }
WV_DEV unsigned lgSupRef(const uint64_t w1) { return unsigned(w1 >> 44) & 0xfffu; }
WV_DEV unsigned lgFirstBase(const uint64_t w1) { return unsigned(w1 >> 56) & 3u; }
WV_DEV unsigned lgLastBase(const uint64_t w1) { return unsigned(w1 >> 58) & 3u; }
WV_DEV bool     lgSelfLoop(const uint64_t w1) { return (w1 >> 63) != 0; }
WV_DEV unsigned lgLinkId(const uint64_t w, const unsigned c)
{
  const unsigned f = unsigned(w >> (11 * c)) & 0x7ffu;
  return f ? f - 1 : ASM_NONE;
}
static const uint64_t LG_M44 = (uint64_t(1) << 44) - 1;

// ====================================================================================================================
// graph_kernel
// ====================================================================================================================
struct LdsGraph {
  const AsmParams& P;
  const LgParams&  G;
  char*            lds;
  unsigned         tw, tn, lane;
  uint32_t *       hdr, *rd, *dbase, *whist, *slots, *keyArr, *codes, *nmask;
  uint16_t *       rdm, *sortA, *sortB, *slotId;
  uint8_t*         cntArr;
  FSet*            sets;
  FRec*            nodes;
  unsigned         nNormal, W, k, nNodes, nFat, nEligible, lowTier, codeWords;
  uint64_t         tMark;

  WV_DEV LdsGraph(const AsmParams& p, const LgParams& g, char* base) : P(p), G(g), lds(base)
  {
    tw     = unsigned(wv::wave_in_wg());
    tn     = unsigned(wv::wg_waves());
    lane   = unsigned(wv::lane());
    hdr    = reinterpret_cast<uint32_t*>(lds + LG_OFF_HDR);
    rd     = reinterpret_cast<uint32_t*>(lds + LG_OFF_RD);
    rdm    = reinterpret_cast<uint16_t*>(lds + LG_OFF_RDM);
    dbase  = reinterpret_cast<uint32_t*>(lds + LG_OFF_DBASE);
    whist  = reinterpret_cast<uint32_t*>(lds + LG_OFF_WHIST);
    slots  = reinterpret_cast<uint32_t*>(lds + LG_OFF_SLOTS);
    sets   = reinterpret_cast<FSet*>(lds + LG_OFF_SETS);
    nodes  = reinterpret_cast<FRec*>(lds + LG_OFF_SETS);
    sortA  = reinterpret_cast<uint16_t*>(lds + LG_OFF_SORTA);
    sortB  = reinterpret_cast<uint16_t*>(lds + LG_OFF_SORTB);
    keyArr = reinterpret_cast<uint32_t*>(lds + LG_OFF_KEYS);
    slotId = reinterpret_cast<uint16_t*>(lds + LG_OFF_KEYS);
    cntArr = reinterpret_cast<uint8_t*>(lds + LG_OFF_CNT);
    codes  = reinterpret_cast<uint32_t*>(lds + LG_OFF_DYN);
    nmask  = codes;
  }

  /// per-phase shader clocks of the workgroup's first wave (-DMANTA_ASM_PROFILE).  -DMANTA_LG_PROFILE_GRAPH: the eight counters
  /// split graph_kernel alone (`fine`: 0 pack, 1 table, 2 counts + slot list, 3 radix passes, 4 ties + ids, 5 slab + bitsets +
  /// record init, 6 links + predecessors + siblings, 7 speculation list + slab write); contig_kernel counts nothing then
  WV_DEV void tick(const int phase, const int fine)
  {
#ifdef MANTA_ASM_PROFILE
    const uint64_t now = wv::clock();
#ifdef MANTA_LG_PROFILE_GRAPH
    const int slot = fine;
#else
    const int slot = phase;
#endif
    if (P.phase_cycles && tw == 0 && lane == 0) wv::atomic_add(&P.phase_cycles[slot], (unsigned long long)(now - tMark));
    tMark = now;
#else
    (void)phase;
    (void)fine;
#endif
  }

  /// every wave of the workgroup has written what the others read next
  WV_DEV void teamSync() const
  {
    wv::sync();
    wv::wg_barrier();
  }
  WV_DEV unsigned tid() const { return 64 * tw + lane; }
  WV_DEV unsigned nThreads() const { return 64 * tn; }

  // ---- keys (2-bit codes, 16 bases per dword, MSB first: dword order == base order) ----
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
  template <int KW>
  WV_DEV static bool keyEq(const Key<KW>& a, const Key<KW>& b)
  {
    bool eq = true;
    for (int i = 0; i < KW; ++i) eq = eq && (a.w[i] == b.w[i]);
    return eq;
  }
  template <int KW>
  WV_DEV static bool keyLess(const Key<KW>& a, const Key<KW>& b)
  {
    for (int i = 0; i < KW; ++i)
      if (a.w[i] != b.w[i]) return a.w[i] < b.w[i];
    return false;
  }
  template <int KW>
  WV_DEV void keySetBase(Key<KW>& key, const unsigned i, const unsigned c) const
  {
    const unsigned sh = 30 - 2 * (i & 15);
    for (int w = 0; w < KW; ++w)
      if (unsigned(w) == (i >> 4)) key.w[w] = (key.w[w] & ~(3u << sh)) | (c << sh);
  }
  /// word[1..k-1] + c
  template <int KW>
  WV_DEV Key<KW> keyShiftAppend(const Key<KW>& key, const unsigned c) const
  {
    Key<KW> r;
    for (int w = 0; w < KW; ++w) r.w[w] = (key.w[w] << 2) | ((w + 1 < KW) ? (key.w[w + 1] >> 30) : 0u);
    keySetBase(r, k - 1, c);
    return r;
  }
  /// hash of a key: bucket from the low bits, 17-bit tag from the high bits
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
  WV_DEV uint32_t prefix32(const unsigned pb) const  // the word's first 16 bases (fewer: zero padded)
  {
    const unsigned wi = pb >> 4, sh = (pb & 15) * 2;
    uint32_t       v  = codes[wi];
    if (sh) v = (v << sh) | (codes[wi + 1] >> (32 - sh));
    if (k < 16) v &= ~((1u << (32 - 2 * k)) - 1u);
    return v;
  }

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

  /// slot of `key` or ASM_NONE.  One 16-byte read per probed bucket; an empty slot ends the search (the slots of a bucket fill
  /// in order and are never freed), a tag mismatch (17 bits) skips the slot without a key compare.
  template <int KW>
  WV_DEV unsigned lookupSlot(const Key<KW>& key) const
  {
    const uint32_t h   = keyHash(key);
    const unsigned tag = h >> 15;
    unsigned       b   = h & (LG_BUCKETS - 1);
    for (unsigned probe = 0; probe < LG_BUCKETS; ++probe) {
      const FBucket bk = *reinterpret_cast<const FBucket*>(slots + 4 * b);
      for (int i = 0; i < 4; ++i) {
        const uint32_t s = bk.s[i];
        if (s == LG_EMPTY) return ASM_NONE;
        if ((s >> 15) == tag && keyEq(keyAt<KW>(s & 0x7fffu), key)) return 4 * b + unsigned(i);
      }
      b = (b + 1) & (LG_BUCKETS - 1);
    }
    return ASM_NONE;
  }

  WV_DEV uint64_t plShift(const unsigned locus, const unsigned i) const
  {
    return P.pl_chunk_shift ? P.pl_chunk_shift[3 * size_t(locus / P.chunk_loci) + i] : uint64_t(0);
  }

  // ------------------------------------------------------------------------------------------------
  // stage 0: the locus' reads -> 2 bit + N bitmap in LDS, from any of the three input forms (1 byte per base, the same
  // arriving chunk by chunk behind the running kernel, packed piles).  False: the locus does not fit this path.
  // (every wave computes the same offsets; the reads are dealt out eight at a time)
  // ------------------------------------------------------------------------------------------------
  WV_DEV bool pack(const unsigned locus)
  {
    const unsigned rBegin = P.locus_read_begin[locus], rEnd = P.locus_read_begin[locus + 1];
    nNormal               = rEnd - rBegin;
    if (nNormal + 2 * P.opt.maxAssemblyCount > LG_MAX_READS) return false;
    W = (nNormal + 2 * P.opt.maxAssemblyCount + 63) / 64;
    if (W == 0) W = 1;
    const uint64_t plR = plShift(locus, 0), plC = plShift(locus, 1), plM = plShift(locus, 2);
    unsigned       cw = 0, mw = 0;
    bool           tooLong = false;
    for (unsigned base = 0; base < nNormal; base += 64) {
      const unsigned r   = base + lane;
      unsigned       len = 0;
      if (r < nNormal) len = P.pl_codes ? P.pl_read_len[rBegin + r + plR] : unsigned(P.read_off[rBegin + r + 1] - P.read_off[rBegin + r]);
      if (len > 0xffffu) tooLong = true;
      const unsigned myC = (r < nNormal) ? (len + 15) / 16 + 1 : 0u;  // +1 padding dword so key fetches may read one past
      const unsigned myM = (r < nNormal) ? (len + 31) / 32 + 1 : 0u;
      unsigned       sc = myC, sm = myM;
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned oc = wv::shfl(sc, wv::lane() - off), om = wv::shfl(sm, wv::lane() - off);
        if (wv::lane() >= off) {
          sc += oc;
          sm += om;
        }
      }
      const unsigned cwo = cw + sc - myC, mwo = mw + sm - myM;
      if (tw == 0 && r < nNormal && cwo <= 0x7ffu) {
        rd[r]  = cwo | ((len & 0xffffu) << 11);
        rdm[r] = uint16_t(mwo);
      }
      cw += wv::readlane(sc, 63);
      mw += wv::readlane(sm, 63);
    }
    if (wv::any(tooLong) || cw + 2 > LG_MAX_PILE) return false;
    const unsigned cwPad = (cw + 2 + 3) & ~3u, mwPad = (mw + 2 + 3) & ~3u;
    if (LG_OFF_DYN + 4 * (cwPad + mwPad) > LG_BUDGET) return false;
    codeWords = cw + 2;
    nmask     = codes + cwPad;
    for (unsigned i = tid(); i < mw + 2; i += nThreads()) nmask[i] = 0;
    if (tid() == 0) hdr[LG_H_FLAG] = 0;
    teamSync();
    if (P.pl_codes) {  // packed piles arrive in this layout: copy, 8 lanes per read
      for (unsigned base = 8 * tw; base < nNormal; base += 8 * tn) {
        const unsigned r = base + (lane >> 3);
        if (r >= nNormal) continue;
        const unsigned  d = rd[r], cwo = d & 0x7ffu, len = (d >> 11) & 0xffffu, mwo = rdm[r];
        const unsigned  nCw = (len + 15) / 16, nMw = (len + 31) / 32;
        const uint32_t* sc  = P.pl_codes + (P.pl_code_off[rBegin + r + plR] + plC);
        const uint32_t* sm  = P.pl_nmask + (P.pl_mask_off[rBegin + r + plR] + plM);
        for (unsigned wi = (lane & 7); wi <= nCw; wi += 8) codes[cwo + wi] = (wi < nCw) ? sc[wi] : 0u;
        bool sawN = false;
        for (unsigned wi = (lane & 7); wi < nMw; wi += 8) {
          const uint32_t m = sm[wi];
          nmask[mwo + wi]  = m;
          sawN             = sawN || (m != 0);
        }
        if (sawN) wv::atomic_or(&rd[r], 1u << 27);
      }
      if (tid() < 2) codes[cw + tid()] = 0;
      teamSync();
      return true;
    }
    bool           bad   = false;
    const uint32_t shift = P.chunk_shift ? P.chunk_shift[locus / P.chunk_loci] : 0u;
    // 8 lanes per read, 8 reads per pass: lane (g, i) converts code dwords i, i+8, ... of read (base + g)
    for (unsigned base = 8 * tw; base < nNormal; base += 8 * tn) {
      const unsigned r = base + (lane >> 3);
      if (r >= nNormal) continue;
      const uint8_t* src = P.bases + P.read_off[rBegin + r] + shift;
      const unsigned d = rd[r], cwo = d & 0x7ffu, len = (d >> 11) & 0xffffu, mwo = rdm[r];
      const unsigned nCw = (len + 15) / 16 + 1;
      bool           sawN = false;
      for (unsigned wi = (lane & 7); wi < nCw; wi += 8) {
        uint32_t code = 0, nbits = 0;
        if (wi * 16 < len) {
          // 16 bases = five aligned dword loads + a byte funnel (the input arena is padded)
          const uintptr_t addr = reinterpret_cast<uintptr_t>(src + wi * 16);
          const uint32_t* ap   = reinterpret_cast<const uint32_t*>(addr & ~uintptr_t(3));
          const unsigned  sh   = unsigned(addr & 3) * 8;
          uint32_t        dw[5];
          for (int q = 0; q < 5; ++q) dw[q] = ap[q];
          for (unsigned q = 0; q < 4; ++q) {
            const uint32_t four = sh ? ((dw[q] >> sh) | (dw[q + 1] << (32 - sh))) : dw[q];
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
    if (tid() < 2) codes[cw + tid()] = 0;
    if (wv::any(bad) && lane == 0) wv::atomic_or(&hdr[LG_H_FLAG], 1u);  // bytes outside {A,C,G,T,N}: the general path decides what is exact
    teamSync();
    return wv::atomic_load(&hdr[LG_H_FLAG]) == 0;
  }

  // ------------------------------------------------------------------------------------------------
  // table pass (getKmerCounts :506-550).  A word IS its slot until the sort numbers the words: an instance claims an empty slot
  // with one compare-and-swap or finds its word there, then ORs its read into the slot's set -- no ids, no allocation, no
  // waiting for another lane's record in the loop.
  // ------------------------------------------------------------------------------------------------
  template <int KW>
  WV_DEV bool tablePass()
  {
    for (unsigned s = tid(); s < LG_SLOTS; s += nThreads()) {
      slots[s]      = LG_EMPTY;
      sets[s].w[0] = 0;
      sets[s].w[1] = 0;
    }
    teamSync();
    bool fail = false;
    for (unsigned rBase = 0; rBase < nNormal; rBase += 64) {
      // descriptors of up to 64 reads in lane registers; v_readlane hands them out per read
      const unsigned rMine = rBase + lane;
      const unsigned dV    = (rMine < nNormal) ? rd[rMine] : 0u;
      const unsigned mV    = (rMine < nNormal) ? unsigned(rdm[rMine]) : 0u;
      const unsigned rEnd  = (nNormal - rBase < 64) ? (nNormal - rBase) : 64u;
      for (unsigned ri = 0; ri < rEnd; ++ri) {
        const unsigned r = rBase + ri;
        if (r % tn != tw) continue;
        const unsigned d = wv::readlane(dV, int(ri)), cwo = d & 0x7ffu, len = (d >> 11) & 0xffffu;
        if (len < k) continue;  // :522
        const bool     rdHasN = (d >> 27) & 1u;
        const unsigned mwo    = wv::readlane(mV, int(ri));
        unsigned long long* const setWord = reinterpret_cast<unsigned long long*>(&sets[0].w[r >> 6]);
        const unsigned long long  setBit  = (unsigned long long)1 << (r & 63);
        for (unsigned j0 = 0; j0 + k <= len; j0 += 64) {
          const unsigned j  = j0 + lane;
          const unsigned pb = cwo * 16 + j;
          bool           todo = (j + k <= len) && !(rdHasN && windowHasN(mwo, j));  // :531
          Key<KW>        key;
          uint32_t       mine = 0;
          unsigned       tag = 0, b = 0, slot = 0, skip = 0;
          if (todo) {
            key              = keyAt<KW>(pb);
            const uint32_t h = keyHash(key);
            tag              = h >> 15;
            b                = h & (LG_BUCKETS - 1);
            mine             = pb | (tag << 15);
          } else {
            for (int w = 0; w < KW; ++w) key.w[w] = 0;
          }
          unsigned probes = 0;
          while (wv::any(todo)) {
            if (todo) {
              const FBucket bk = *reinterpret_cast<const FBucket*>(slots + 4 * b);
              // first slot of the bucket (past `skip`) that is empty or carries the tag
              unsigned at = 4;
              for (int i = 3; i >= 0; --i)
                if (unsigned(i) >= skip && (bk.s[i] == LG_EMPTY || (bk.s[i] >> 15) == tag)) at = unsigned(i);
              if (at == 4) {
                b    = (b + 1) & (LG_BUCKETS - 1);
                skip = 0;
                if (++probes >= LG_BUCKETS) {
                  fail = true;
                  todo = false;
                }
              } else {
                uint32_t s = (at == 0) ? bk.s[0] : (at == 1) ? bk.s[1] : (at == 2) ? bk.s[2] : bk.s[3];
                if (s == LG_EMPTY) s = wv::atomic_cas(&slots[4 * b + at], LG_EMPTY, mine);
                if (s == LG_EMPTY) {  // claimed
                  slot = 4 * b + at;
                  todo = false;
                } else if ((s >> 15) == tag && keyEq(keyAt<KW>(s & 0x7fffu), key)) {
                  slot = 4 * b + at;
                  todo = false;
                } else {
                  skip = at + 1;  // another word (taken under this lane's eyes, or a tag collision): next slot of the bucket
                  if (skip == 4) {
                    b    = (b + 1) & (LG_BUCKETS - 1);
                    skip = 0;
                    if (++probes >= LG_BUCKETS) {
                      fail = true;
                      todo = false;
                    }
                  }
                }
              }
            }
          }
          if ((j + k <= len) && !fail && !(rdHasN && windowHasN(mwo, j))) wv::atomic_or(setWord + 2 * slot, setBit);
        }
      }
    }
    if (wv::any(fail) && lane == 0) wv::atomic_or(&hdr[LG_H_FLAG], 2u);
    teamSync();
    return wv::atomic_load(&hdr[LG_H_FLAG]) == 0;
  }

  // ------------------------------------------------------------------------------------------------
  // the words in seed order (:686-696: count descending, k-mer ascending).  Stable LSD radix sort of the occupied slots over
  // {first 16 bases (four passes), 255 - count}; what still ties (same count, same first 16 bases) is ranked by full key compares
  // inside its (short) run.  Result: sortA[id] = slot, slotId[slot] = id, nFat / nEligible / lowTier from the count digits.
  // ------------------------------------------------------------------------------------------------
  WV_DEV unsigned cntOf(const unsigned slot) const
  {
    const unsigned v = cntArr[slot];
    return (v & 0x80u) ? 1u : v;
  }

  template <int KW>
  WV_DEV bool sortWords()
  {
    // counts, prefixes, the list of occupied slots (any order)
    if (tid() == 0) hdr[LG_H_N] = 0;
    teamSync();
    for (unsigned sb = 64 * tw; sb < LG_SLOTS; sb += 64 * tn) {
      const unsigned s   = sb + lane;
      const uint32_t v   = slots[s];
      const bool     occ = v != LG_EMPTY;
      if (occ) {
        const FSet     st = sets[s];
        const unsigned c  = unsigned(wv::popc(st.w[0])) + unsigned(wv::popc(st.w[1]));
        unsigned       enc = c;
        if (c == 1) enc = 0x80u | (st.w[0] ? unsigned(wv::ctz(st.w[0])) : 64u + unsigned(wv::ctz(st.w[1])));
        cntArr[s] = uint8_t(enc);
        keyArr[s] = prefix32(v & 0x7fffu);
      }
      const uint64_t m = wv::ballot(occ);
      if (m) {
        unsigned at = 0;
        if (lane == 0) at = wv::atomic_add(&hdr[LG_H_N], unsigned(wv::popc(m)));
        at = wv::first(at);
        if (occ) sortA[at + unsigned(wv::popc(m & ((uint64_t(1) << lane) - 1)))] = uint16_t(s);
      }
    }
    teamSync();
    const unsigned n = wv::atomic_load(&hdr[LG_H_N]);
    nNodes           = n;
    if (n > LG_MAX_NODES) return false;
    if (n == 0) {
      nFat = nEligible = lowTier = 0;
      return true;
    }
    tick(2, 2);
    const unsigned chunk = (((n + tn - 1) / tn) + 63) & ~63u;  // elements of one wave, in order
    const unsigned c0 = chunk * tw, c1 = (c0 + chunk < n) ? (c0 + chunk) : n;
    uint16_t *     src = sortA, *dst = sortB;
    uint32_t*      myHist = whist + 256 * tw;
    for (int pass = 0; pass < 5; ++pass) {
      for (unsigned i = tid(); i < 256 * tn; i += nThreads()) whist[i] = 0;
      teamSync();
      auto digitOf = [&](const unsigned slot) -> unsigned {
        return (pass < 4) ? ((keyArr[slot] >> (8 * pass)) & 255u) : (255u - cntOf(slot));
      };
      for (unsigned i0 = c0; i0 < c1; i0 += 64) {
        const unsigned i = i0 + lane;
        if (i < c1) wv::atomic_add(&myHist[digitOf(src[i])], 1u);
      }
      teamSync();
      // per digit: running offsets over the waves, totals; then the digit bases (four quarters of 64 digits, dealt out to the waves)
      for (unsigned q = tw; q < 4; q += tn) {
        const unsigned d   = 64 * q + lane;
        unsigned       run = 0;
        for (unsigned w = 0; w < tn; ++w) {
          const unsigned c   = whist[256 * w + d];
          whist[256 * w + d] = run;
          run += c;
        }
        unsigned inc = run;
        for (int off = 1; off < 64; off <<= 1) {
          const unsigned o = wv::shfl(inc, int(lane) - off);
          if (int(lane) >= off) inc += o;
        }
        dbase[d] = inc - run;  // exclusive inside the quarter
        if (lane == 63) hdr[LG_H_TOT + q] = inc;
      }
      teamSync();
      for (unsigned q = tw; q < 4; q += tn) {
        unsigned before = 0;
        for (unsigned p = 0; p < q; ++p) before += hdr[LG_H_TOT + p];
        dbase[64 * q + lane] += before;
      }
      teamSync();
      for (unsigned i0 = c0; i0 < c1; i0 += 64) {
        const unsigned i     = i0 + lane;
        const bool     valid = i < c1;
        const unsigned slot  = valid ? unsigned(src[i]) : 0u;
        const unsigned d     = valid ? digitOf(slot) : 0u;
        uint64_t       peers = wv::ballot(valid);
        for (int bit = 0; bit < 8; ++bit) {
          const bool     on = (d >> bit) & 1u;
          const uint64_t m  = wv::ballot(valid && on);
          peers &= on ? m : ~m;
        }
        unsigned base = 0;
        if (valid) base = dbase[d] + myHist[d];
        wv::sync();
        if (valid) {
          dst[base + unsigned(wv::popc(peers & ((uint64_t(1) << lane) - 1)))] = uint16_t(slot);
          if ((peers >> lane) == 1u) myHist[d] += unsigned(wv::popc(peers));  // the highest lane of the digit's group
        }
        wv::sync();
      }
      teamSync();
      if (pass == 4 && tid() == 0) {
        // ids below dbase[d] have counts above 255 - d
        const unsigned minCov = P.opt.minCoverage;
        hdr[LG_H_TOT + 4] = dbase[254];                                                      // count >= 2
        hdr[LG_H_TOT + 5] = (minCov <= 1) ? n : ((minCov > 255) ? 0u : dbase[256 - minCov]);   // count >= minCoverage
        hdr[LG_H_TOT + 6] = (minCov + 2 > 255) ? 0u : dbase[255 - (minCov + 1)];               // first id with count <= minCoverage + 1
      }
      uint16_t* t = src;
      src         = dst;
      dst         = t;
    }
    teamSync();
    tick(2, 3);
    nFat      = hdr[LG_H_TOT + 4];
    nEligible = hdr[LG_H_TOT + 5];
    lowTier   = hdr[LG_H_TOT + 6];
    // five passes: the sorted list sits in sortB (= src); runs of equal {count, first 16 bases} -> exact order into sortA
    for (unsigned i = tid(); i < n; i += nThreads()) {
      const unsigned slot = src[i];
      const unsigned c = cntOf(slot), p = keyArr[slot];
      unsigned       lo = i, hi = i;
      while (lo > 0) {
        const unsigned o = src[lo - 1];
        if (cntOf(o) != c || keyArr[o] != p) break;
        lo--;
      }
      while (hi + 1 < n) {
        const unsigned o = src[hi + 1];
        if (cntOf(o) != c || keyArr[o] != p) break;
        hi++;
      }
      unsigned rank = 0;
      if (hi > lo) {
        const Key<KW> mineKey = keyAt<KW>(slots[slot] & 0x7fffu);
        for (unsigned j = lo; j <= hi; ++j) {
          if (j == i) continue;
          if (keyLess(keyAt<KW>(slots[src[j]] & 0x7fffu), mineKey)) rank++;
        }
      }
      dst[lo + rank] = uint16_t(slot);
    }
    teamSync();
    // (the prefixes are dead: the same bytes take the slot -> id map)
    for (unsigned i = tid(); i < n; i += nThreads()) slotId[sortA[i]] = uint16_t(i);
    teamSync();
    tick(2, 4);
    return true;
  }

  // ------------------------------------------------------------------------------------------------
  // node records in the new numbering, links (8 table lookups per word), sibling table, speculation list; see FRec
  // ------------------------------------------------------------------------------------------------
  template <int KW>
  WV_DEV bool buildRecords(uint8_t* slab)
  {
    // bitsets of the words with more than one read: ids below nFat, straight into the slab
    FSet* gPool = reinterpret_cast<FSet*>(slab + sizeof(LgHdr) + 16ull * nNodes);
    for (unsigned i = tid(); i < nFat; i += nThreads()) gPool[i] = sets[sortA[i]];
    teamSync();  // (the sets are dead: the records take their place)
    for (unsigned i = tid(); i < nNodes; i += nThreads()) {
      const unsigned slot = sortA[i], pb = slots[slot] & 0x7fffu, enc = cntArr[slot];
      const unsigned cnt = (enc & 0x80u) ? 1u : enc;
      const unsigned sup = (enc & 0x80u) ? (enc & 0x7fu) : (LG_FAT | i);
      const Key<KW>  key = keyAt<KW>(pb);
      unsigned       last = 0;
      for (int w = 0; w < KW; ++w)
        if (unsigned(w) == ((k - 1) >> 4)) last = (key.w[w] >> (30 - 2 * ((k - 1) & 15))) & 3u;
      FRec rec;
      rec.w0   = (uint64_t(pb & 0xfffu) << 52) | (uint64_t(cnt) << 44);
      rec.w1   = (uint64_t(sup) << 44) | (uint64_t(key.w[0] >> 30) << 56) | (uint64_t(last) << 58) | (uint64_t(pb >> 12) << 60);
      nodes[i] = rec;
    }
    if (tid() == 0) hdr[LG_H_NSIB] = 0;
    teamSync();
    tick(2, 5);
    // successor lookups, predecessor scatter (by symbol position; packed below)
    for (unsigned nb = 64 * tw; nb < nNodes; nb += 64 * tn) {
      const unsigned nd = nb + lane;
      if (nd < nNodes) {
        const uint64_t w0in = nodes[nd].w0;
        const unsigned pb   = slots[sortA[nd]] & 0x7fffu;
        const Key<KW>  key  = keyAt<KW>(pb);
        const unsigned firstBase = key.w[0] >> 30;
        uint64_t       w0 = w0in;
        bool           selfLoop = false;
        unsigned       m = 0;
        for (unsigned c = 0; c < 4; ++c) {
          const unsigned ss = lookupSlot<KW>(keyShiftAppend<KW>(key, c));
          if (ss == ASM_NONE) continue;
          const unsigned s = slotId[ss];
          w0 |= uint64_t(s + 1) << (11 * m);
          m++;
          if (s == nd) selfLoop = true;
          wv::atomic_or(reinterpret_cast<unsigned long long*>(&nodes[s].w1), (unsigned long long)(uint64_t(nd + 1) << (11 * firstBase)));
        }
        nodes[nd].w0 = w0;
        if (selfLoop) wv::atomic_or(reinterpret_cast<unsigned long long*>(&nodes[nd].w1), (unsigned long long)(uint64_t(1) << 63));
      }
    }
    teamSync();
    // predecessor lists packed like the successor lists; words without a predecessor: their siblings (the words that differ in
    // the last base only, :185-210) cannot be found through a predecessor's successor list -> side table
    uint16_t* sib = reinterpret_cast<uint16_t*>(lds + LG_OFF_SIB);
    for (unsigned nd = tid(); nd < nNodes; nd += nThreads()) {
      const uint64_t w1 = nodes[nd].w1;
      uint64_t       pk = 0;
      unsigned       m  = 0;
      for (unsigned c = 0; c < 4; ++c) {
        const uint64_t f = (w1 >> (11 * c)) & 0x7ffu;
        if (f == 0) continue;
        pk |= f << (11 * m);
        m++;
      }
      nodes[nd].w1 = (w1 & ~LG_M44) | pk;
      if (m == 0) {
        const uint64_t w0  = nodes[nd].w0;
        const unsigned pb  = lgPb(w0, w1);
        const Key<KW>  key = keyAt<KW>(pb);
        const unsigned lastBase = lgLastBase(w1);
        unsigned       found[3] = {LG_NO_SLOT, LG_NO_SLOT, LG_NO_SLOT};
        unsigned       nf = 0;
        for (unsigned c = 0; c < 4; ++c) {
          if (c == lastBase) continue;
          Key<KW> s2 = key;
          keySetBase(s2, k - 1, c);
          const unsigned ss = lookupSlot<KW>(s2);
          if (ss != ASM_NONE) found[nf++] = slotId[ss];
        }
        if (nf) {
          const unsigned at = wv::atomic_add(&hdr[LG_H_NSIB], 1u);
          if (at < LG_SIB_CAP) {
            sib[4 * at + 0] = uint16_t(nd);
            sib[4 * at + 1] = uint16_t(found[0]);
            sib[4 * at + 2] = uint16_t(found[1]);
            sib[4 * at + 3] = uint16_t(found[2]);
          }
        }
      }
    }
    teamSync();
    tick(2, 6);
    if (wv::atomic_load(&hdr[LG_H_NSIB]) > LG_SIB_CAP) return false;
    return true;
  }

  /// Round 0 of the contig loop walks, beside the first seed, the words most likely to be the next seeds: the two lowest count
  /// tiers in seed order (error branches) -- an id range now -- thinned per unbranched stretch: a walk always runs from its seed
  /// to the far end of the seed's stretch (the seed's reads carry it), so a word with an earlier-ranked word of its stretch
  /// upstream is consumed by that word's walk.  (A heuristic like the rest of the speculation: what it drops or keeps wrongly
  /// costs a later walk round, never a result.)  Writes the list (entry 0 = first seed) to `spec`, returns its length.
  WV_DEV unsigned speculationList(uint16_t* spec)
  {
    uint16_t* label = reinterpret_cast<uint16_t*>(lds + LG_OFF_CHAIN);
    uint16_t* dist  = label + 128;
    const unsigned e0 = lowTier, e1 = (nEligible < lowTier + 128) ? nEligible : (lowTier + 128);
    const unsigned nE = (e1 > e0) ? (e1 - e0) : 0u;
    auto outOnly = [&](const FRec& rec, const unsigned nd, unsigned& od) -> unsigned {
      unsigned only = ASM_NONE;
      od            = 0;
      for (unsigned c = 0; c < 4; ++c) {
        const unsigned s = lgLinkId(rec.w0, c);
        if (s != ASM_NONE && s != nd) {
          od++;
          only = s;
        }
      }
      return only;
    };
    auto inDeg = [&](const FRec& rec, const unsigned nd) -> unsigned {
      unsigned id = 0;
      for (unsigned c = 0; c < 4; ++c) {
        const unsigned p = lgLinkId(rec.w1, c);
        if (p != ASM_NONE && p != nd) id++;
      }
      return id;
    };
    if (tid() < nE) {
      unsigned cur = e0 + tid(), steps = 0;
      while (steps < 192) {
        unsigned       od;
        const unsigned nx = outOnly(nodes[cur], cur, od);
        if (od != 1 || inDeg(nodes[nx], nx) != 1) break;
        cur = nx;
        steps++;
      }
      label[tid()] = uint16_t(cur);
      dist[tid()]  = uint16_t(steps);
    }
    teamSync();
    unsigned n0 = 0;
    if (tw == 0) {
      if (nEligible > 0) {
        bool     dup[2] = {false, false};
        unsigned eL[2], eD[2];
        for (unsigned h = 0; h < 2; ++h) {
          const unsigned i = lane + 64 * h;
          eL[h]            = (i < nE) ? unsigned(label[i]) : ASM_NONE;
          eD[h]            = (i < nE) ? unsigned(dist[i]) : 0u;
        }
        for (unsigned j = 0; j < nE; ++j) {
          const unsigned lj = label[j], dj = dist[j];
          for (unsigned h = 0; h < 2; ++h)
            if (j < lane + 64 * h && lj == eL[h] && dj > eD[h]) dup[h] = true;
        }
        n0 = 1;
        for (unsigned h = 0; h < 2; ++h) {
          const unsigned i    = lane + 64 * h;
          const bool     keep = (i < nE) && !dup[h] && (e0 + i) != 0u;
          const uint64_t mk   = wv::ballot(keep);
          const unsigned pos  = n0 + unsigned(wv::popc(mk & ((uint64_t(1) << lane) - 1)));
          if (keep && pos < 64) spec[pos] = uint16_t(e0 + i);
          n0 += unsigned(wv::popc(mk));
        }
        if (n0 > 64) n0 = 64;
        if (lane == 0) spec[0] = 0;  // the first seed: highest count, smallest word = id 0
      }
      if (lane == 0) hdr[LG_H_TOT + 7] = n0;
    }
    teamSync();
    return wv::atomic_load(&hdr[LG_H_TOT + 7]);
  }

  template <int KW>
  WV_DEV bool runK(const unsigned locus)
  {
    if (!tablePass<KW>()) return false;
    tick(1, 1);
    if (!sortWords<KW>()) return false;
    // slab for this locus
    const uint64_t bytes = lgSlabBytes(nNodes, nFat, codeWords);
    if (tid() == 0) {
      const unsigned long long off = wv::atomic_add(G.arena_used, (unsigned long long)bytes);
      hdr[LG_H_OFF_LO] = uint32_t(off);
      hdr[LG_H_OFF_HI] = uint32_t(off >> 32);
    }
    teamSync();
    const uint64_t off = (uint64_t(wv::atomic_load(&hdr[LG_H_OFF_HI])) << 32) | wv::atomic_load(&hdr[LG_H_OFF_LO]);
    if (off + bytes > G.arena_cap) return false;
    const unsigned need = ckNeed(nNodes, nFat);
    unsigned       cls  = LG_CLASSES;
    for (unsigned c = LG_CLASSES; c-- > 0;)
      if (G.class_bytes[c] && need <= G.class_bytes[c]) cls = c;
    if (cls == LG_CLASSES) return false;
    uint8_t* slab = G.arena + off;
    if (!buildRecords<KW>(slab)) return false;
    uint16_t*      gSpec = reinterpret_cast<uint16_t*>(slab + sizeof(LgHdr) + 16ull * nNodes + 16ull * nFat);
    const unsigned nSpec = speculationList(gSpec);
    // the rest of the slab
    FRec* gRec = reinterpret_cast<FRec*>(slab + sizeof(LgHdr));
    for (unsigned i = tid(); i < nNodes; i += nThreads()) gRec[i] = nodes[i];
    uint16_t*       gSib  = gSpec + 64;
    const uint16_t* sib   = reinterpret_cast<const uint16_t*>(lds + LG_OFF_SIB);
    const unsigned  nSib  = wv::atomic_load(&hdr[LG_H_NSIB]);
    for (unsigned i = tid(); i < 4 * nSib; i += nThreads()) gSib[i] = sib[i];
    uint32_t* gCodes = reinterpret_cast<uint32_t*>(gSib + 4 * LG_SIB_CAP);
    for (unsigned i = tid(); i < codeWords; i += nThreads()) gCodes[i] = codes[i];
    if (tid() == 0) {
      LgHdr h;
      h.nNodes    = nNodes;
      h.nFat      = nFat;
      h.k         = k;
      h.nNormal   = nNormal;
      h.nEligible = nEligible;
      h.nSpec     = nSpec;
      h.nSib      = nSib;
      h.codeWords = codeWords;
      h.W         = W;
      h.need      = need;
      h.reserved0 = h.reserved1 = 0;
      for (int i = 0; i < 4; ++i) h.pad[i] = 0;
      *reinterpret_cast<LgHdr*>(slab) = h;
      G.slab_off[locus]               = off;
      G.class_ids[size_t(cls) * G.class_stride + wv::atomic_add(&G.class_count[cls], 1u)] = locus;
    }
    tick(4, 7);
    return true;
  }

  /// false: the general path takes the locus
  WV_DEV bool run(const unsigned locus)
  {
    const unsigned minWL = P.locus_min_wl ? P.locus_min_wl[locus] : P.opt.minWordLength;
    const unsigned maxWL = P.locus_max_wl ? P.locus_max_wl[locus] : P.opt.maxWordLength;
    if (minWL == 0 || maxWL > 16u * ASM_MAX_KW || minWL > maxWL || 2 * P.opt.maxAssemblyCount > ASM_MAX_CAND) return false;
    if (P.opt.minCoverage > 250 || P.opt.maxAssemblyCount > 20) return false;
    k     = minWL;
    tMark = wv::clock();
    if (!pack(locus)) return false;
    tick(0, 0);
    const unsigned kw = (k + 15) >> 4;
    if (kw <= 2) return runK<2>(locus);
    if (kw <= 4) return runK<4>(locus);
    return runK<8>(locus);
  }
};

/// persistent workgroups of LG_WAVES wavefronts, LG_BUDGET bytes of dynamic LDS each; params as assemble_kernel plus the
/// pipeline's own.  Loci this path does not cover are appended to P.punt_ids (P.punt_count counts them).
#ifndef MANTA_LG_WAVES_PER_SIMD
#define MANTA_LG_WAVES_PER_SIMD 4
#endif
WV_KERNEL_WG(LG_WAVES) WV_WAVES_PER_SIMD(MANTA_LG_WAVES_PER_SIMD) void graph_kernel(const LgArgs A)
{
  const AsmParams& P = A.P;
  const LgParams&  G = A.G;
  char*          lds = wv::lds_single();
  uint32_t*      hdr = reinterpret_cast<uint32_t*>(lds + LG_OFF_HDR);
  const unsigned tw  = unsigned(wv::wave_in_wg());
  while (true) {
    if (tw == 0 && wv::lane() == 0) hdr[LG_H_SLOT] = wv::atomic_add(P.counter, 1u);
    wv::sync();
    wv::wg_barrier();
    const unsigned slot = wv::first(wv::atomic_load(&hdr[LG_H_SLOT]));
    if (slot >= P.n_loci) break;
    const unsigned locus   = P.locus_ids ? P.locus_ids[slot] : slot;
    const bool     arrived = !P.upload_chunks_done || asmWaitUploaded(P, locus);
    bool           ok      = false;
    if (arrived) {
      LdsGraph g(P, G, lds);
      ok = g.run(locus);
    }
    wv::sync();
    if (tw == 0 && !ok && wv::lane() == 0) P.punt_ids[wv::atomic_add(P.punt_count, 1u)] = locus;
    wv::sync();
    wv::wg_barrier();  // (the slot word is rewritten next)
  }
}

}  // namespace manta_dev

#include "asm_contig.hpp"
