// graph_big_kernel: the LDS assembler pipeline's graph stage for BIG piles -- up to 256 reads, up to LGL_MAX_NODES (7 168) words,
// a 2-bit pile of up to 3 600 dwords: the config-4/5 shape (200 reads x 250 bases, several thousand words of which nine in ten are
// held by a single read).  Same job as graph_kernel (asm_lds.hpp; assembly/IterativeAssembler.cpp:506-550, 644-720 up to the contig
// loop): pack -> table pass -> sort into seed order -> 8-byte records with links -> proof of acyclicity -> compact graph to the
// locus' slab.  contig_big_kernel (asm_contig.hpp, the same template as contig_kernel) does the serial half.
//
// One workgroup of LGL_WAVES = 16 wavefronts owns a CU's whole LDS (160 KB).  What does not scale from the small class and is done
// differently here:
//   * READ SETS.  A set per table slot (8 192 x 32 bytes) does not fit, and it is not needed: nine words in ten have one read.  A
//     slot starts as a single-read word whose read is implied by its first occurrence; the first instance that comes from ANOTHER
//     read allocates a 4-qword set from a pool (one LDS counter) and publishes its index inside the slot word with a compare-and-
//     swap; whoever loses that race reads the winner's index and leaks one pool entry.  Slot word while the table is built:
//     {first occurrence : 16, pool index + 1 : 11, hash tag : 5}.
//   * NO SLOT -> ID MAP.  After the sort the slot words are rewritten to {id : 13, hash tag : 19} and the id -> slot list becomes
//     id -> first occurrence, so a lookup's key compare goes slot -> id -> text.
//   * PREDECESSORS BY LOOKUP.  The small class scatters predecessor links from the successor lookups into a second 8-byte array;
//     here every word looks its four possible predecessors up as well (the presence filter answers most of them).
//   * Records: LgL (asm_lds.hpp) -- 13-bit links, overflow entries addressed by index, the single read of a word in a byte array.
//
// Anything outside this envelope, a cyclic graph or a repeat hit (-> next word length) is punted to assemble_kernel as before.
#pragma once
#include "asm_lds.hpp"

namespace manta_dev {

// graph_big_kernel LDS map (bytes)
static const unsigned LGL_OFF_HDR   = 0;                               // u32[64] header words
static const unsigned LGL_OFF_PAR   = 256;                             // u8[256] parent read of a read's anchor (readOffsets)
static const unsigned LGL_OFF_RD    = 512;                             // u32[256] read descriptors {code dword offset : 12, length : 16, has N : 1}
static const unsigned LGL_OFF_RDM   = LGL_OFF_RD + 1024;               // u16[256] N-bitmap dword offset of a read; the reads' offsets later
static const unsigned LGL_OFF_RST   = LGL_OFF_RDM + 512;               // u16[256] byte offset of a read in the staged pile
static const unsigned LGL_OFF_ANCH  = LGL_OFF_RST + 512;               // u32[256] anchors, i32[256] offsets
static const unsigned LGL_OFF_DBASE = LGL_OFF_ANCH + 2048;             // u32[256] digit bases of a sort pass
static const unsigned LGL_OFF_SIB   = LGL_OFF_DBASE + 1024;            // u16[LG_SIB_CAP][4]
static const unsigned LGL_OFF_SOVF  = LGL_OFF_SIB + 8 * LG_SIB_CAP;    // u16[LGL_OVF_CAP][4]
static const unsigned LGL_OFF_POVF  = LGL_OFF_SOVF + 8 * LGL_OVF_CAP;  // u16[LGL_OVF_CAP][4]
static const unsigned LGL_OFF_CHAIN = LGL_OFF_POVF + 8 * LGL_OVF_CAP;  // u16[128] label, u16[128] distance, u32[4] duplicate bits
static const unsigned LGL_OFF_WHIST = (LGL_OFF_CHAIN + 528 + 255) & ~255u;   // u32[LGL_WAVES][256] per-wave digit counts
static const unsigned LGL_OFF_POOL  = LGL_OFF_WHIST + 1024 * LGL_WAVES;      // FSetT<LgL>[LGL_POOL_CAP]; histograms + pool = the records later
static const unsigned LGL_OFF_SLOTS = LGL_OFF_POOL + 32 * LGL_POOL_CAP;      // u32[8192]; pool + slots = the staged bytes of the pile during the pack
static const unsigned LGL_OFF_SORTA = LGL_OFF_SLOTS + 4 * LGL_SLOTS;         // u16[8192] id -> slot, then id -> first occurrence
static const unsigned LGL_OFF_SORTB = LGL_OFF_SORTA + 2 * LGL_SLOTS;         // u16[8192]; the words' potentials later
static const unsigned LGL_OFF_CNT   = LGL_OFF_SORTB + 2 * LGL_SLOTS;         // u8[8192] count by slot; the presence filter later
static const unsigned LGL_OFF_DYN   = LGL_OFF_CNT + LGL_SLOTS;               // codes, N bitmap (count by id after the sort)
static const unsigned LGL_FILTER_BITS = 8 * LGL_SLOTS;
static const unsigned LGL_STAGE_BYTES = 32 * LGL_POOL_CAP + 4 * LGL_SLOTS;
static const unsigned LGL_DYN_DWORDS  = (LGL_BUDGET - LGL_OFF_DYN) / 4;
static_assert(8 * LGL_MAX_NODES <= 1024 * LGL_WAVES + 32 * LGL_POOL_CAP, "records inside histograms + pool");
static_assert(LGL_OFF_DYN + 4 * (LGL_MAX_PILE + 2 + 2) + LGL_MAX_NODES + 64 <= LGL_BUDGET, "graph_big_kernel LDS map");
static_assert(LGL_MAX_NODES < (1u << 13) - 1, "ids + 1 in 13 bits");

enum {
  LGL_H_SLOT = 0, LGL_H_FLAG = 1, LGL_H_N = 2, LGL_H_NSIB = 3, LGL_H_OFF_LO = 4, LGL_H_OFF_HI = 5, LGL_H_POOLN = 6, LGL_H_CYC = 7,
  LGL_H_TOT = 8,  ///< [8]
  LGL_H_NSOVF = 16, LGL_H_NPOVF = 17, LGL_H_NSPEC = 18
};

struct LdsGraphL {
  typedef FSetT<LgL> Set;
  typedef LgRec<LgL> R;
  static const unsigned RPL = LGL_MAX_READS / 64;  ///< reads per lane where one wave handles all reads
  const AsmParams& P;
  const LgParams&  G;
  char*            lds;
  unsigned         tw, tn, lane;
  uint32_t *       hdr, *rd, *dbase, *whist, *slots, *codes, *nmask, *filter;
  uint16_t *       rdm, *rstart, *sortA, *sortB, *idPb;
  uint8_t *        cntArr, *cntId;
  Set*             pool;
  Set*             poolOvf;  ///< sets LGL_POOL_CAP .. of the table pass: device memory (nullptr: none)
  uint16_t *       lexBySlot, *lexById;  ///< the words' lexicographic ranks (device memory, rounds only): sortWords leaves them, lexOrder copies them
  FRec8*           nodes;
  int16_t *        phi, *roff;
  unsigned         nNormal, W, k, nNodes, nFat, nEligible, lowTier, codeWords;
  unsigned         nReads, nPseudo, pseudoPb0;  ///< later word lengths: the previous length's contigs as reads nNormal .. nReads - 1 (packed bases from pseudoPb0 on)
  const LgIter*    it;
  uint64_t         tMark;

  WV_DEV LdsGraphL(const AsmParams& p, const LgParams& g, char* base) : P(p), G(g), lds(base)
  {
    tw     = unsigned(wv::wave_in_wg());
    tn     = unsigned(wv::wg_waves());
    lane   = unsigned(wv::lane());
    hdr    = reinterpret_cast<uint32_t*>(lds + LGL_OFF_HDR);
    rd     = reinterpret_cast<uint32_t*>(lds + LGL_OFF_RD);
    rdm    = reinterpret_cast<uint16_t*>(lds + LGL_OFF_RDM);
    roff   = reinterpret_cast<int16_t*>(lds + LGL_OFF_RDM);
    rstart = reinterpret_cast<uint16_t*>(lds + LGL_OFF_RST);
    dbase  = reinterpret_cast<uint32_t*>(lds + LGL_OFF_DBASE);
    whist  = reinterpret_cast<uint32_t*>(lds + LGL_OFF_WHIST);
    pool   = reinterpret_cast<Set*>(lds + LGL_OFF_POOL);
    poolOvf = g.gws ? reinterpret_cast<Set*>(g.gws + size_t(wv::block_single()) * LGL_GWS_BYTES) : nullptr;
    lexBySlot = g.gws ? reinterpret_cast<uint16_t*>(g.gws + size_t(wv::block_single()) * LGL_GWS_BYTES + LGL_GWS_SETS) : nullptr;
    lexById   = lexBySlot ? lexBySlot + 8192 : nullptr;
    nodes  = reinterpret_cast<FRec8*>(lds + LGL_OFF_WHIST);
    slots  = reinterpret_cast<uint32_t*>(lds + LGL_OFF_SLOTS);
    sortA  = reinterpret_cast<uint16_t*>(lds + LGL_OFF_SORTA);
    idPb   = sortA;
    sortB  = reinterpret_cast<uint16_t*>(lds + LGL_OFF_SORTB);
    phi    = reinterpret_cast<int16_t*>(lds + LGL_OFF_SORTB);
    cntArr = reinterpret_cast<uint8_t*>(lds + LGL_OFF_CNT);
    filter = reinterpret_cast<uint32_t*>(lds + LGL_OFF_CNT);
    codes  = reinterpret_cast<uint32_t*>(lds + LGL_OFF_DYN);
    nmask  = codes;
    cntId  = nullptr;
    nReads = nPseudo = 0;
    pseudoPb0 = 0x10000u;
    it        = nullptr;
  }

  /// per-phase shader clocks of the workgroup's first wave (-DMANTA_ASM_PROFILE; the slots of graph_kernel's coarse profile)
  /// coarse: 0 pack, 1 table + offsets, 2 sort + records, 4 slab.  -DMANTA_LG_PROFILE_GRAPH: the eight counters split this kernel alone
  /// (`fine`: 0 pack, 1 table, 2 reads' offsets, 3 sort, 4 sets to the slab + slot rewrite + filter, 5 links, 6 speculation list, 7 slab write)
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
  /// why a locus was handed back (G.stats[2..9], read by the host's debug line): 2 envelope / alphabet, 3 table or set pool full,
  /// 4 too many words / side tables / no contig class, 5 slab arena full, 6 repeat_big_kernel, 7 contig kernel (LDS / contig length),
  /// 8 pseudo arena full, 9 more word lengths than rounds
  WV_DEV void why(const unsigned code) const
  {
    if (G.stats && tid() == 0) wv::atomic_add(&G.stats[code], 1u);
  }
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
  /// c + word[0..k-2]
  template <int KW>
  WV_DEV Key<KW> keyShiftPrepend(const Key<KW>& key, const unsigned c) const
  {
    Key<KW> r;
    for (int w = 0; w < KW; ++w) r.w[w] = (key.w[w] >> 2) | ((w > 0) ? (key.w[w - 1] << 30) : (c << 30));
    // drop the base that moved to position k
    const unsigned kw = (k + 15) >> 4;
    for (int w = 0; w < KW; ++w) {
      if (unsigned(w) >= kw) {
        r.w[w] = 0;
      } else if (unsigned(w) == kw - 1) {
        const unsigned have = k - 16u * unsigned(w);
        if (have < 16) r.w[w] &= ~((1u << (32 - 2 * have)) - 1u);
      }
    }
    return r;
  }
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
  /// byte `b` of the word at pile position pb: its bases [4 b, 4 b + 4) as eight bits, bases beyond the word length read as zero.  Comparing
  /// words byte by byte from b = 0 is comparing their text (2-bit codes, A < C < G < T): the digits of the radix sorts below.
  WV_DEV unsigned keyByte(const unsigned pb, const unsigned b) const
  {
    const unsigned at = pb + 4 * b, wi = at >> 4, sh = (at & 15) * 2;
    uint32_t       v  = codes[wi] << sh;
    if (sh > 24) v |= codes[wi + 1] >> (32 - sh);
    v >>= 24;
    const unsigned have = k - 4 * b;  // (callers pass b < ceil(k / 4))
    if (have < 4) v &= ~((1u << (8 - 2 * have)) - 1u);
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

  /// set `pi` (1-based, as in a slot word) of the table pass: LDS, or -- beyond LGL_POOL_CAP -- the workgroup's device-memory workspace
  /// (two explicit paths: one pointer for both would make every access a flat one)
  WV_DEV Set setLoad(const unsigned pi) const
  {
    if (pi <= LGL_POOL_CAP) return pool[pi - 1];
    return poolOvf[pi - 1 - LGL_POOL_CAP];
  }
  WV_DEV void setAtomicOr(const unsigned pi, const unsigned q, const unsigned long long bit) const
  {
    if (pi <= LGL_POOL_CAP)
      wv::atomic_or(reinterpret_cast<unsigned long long*>(&pool[pi - 1].w[q]), bit);
    else
      wv::atomic_or(reinterpret_cast<unsigned long long*>(&poolOvf[pi - 1 - LGL_POOL_CAP].w[q]), bit);
  }
  WV_DEV void setPlainOr(const unsigned pi, const unsigned q, const uint64_t bit) const
  {
    if (pi <= LGL_POOL_CAP)
      pool[pi - 1].w[q] |= bit;
    else
      poolOvf[pi - 1 - LGL_POOL_CAP].w[q] |= bit;
  }

  // slot words.  Table pass: {first occurrence : 16, pool index + 1 : 11, tag : 5}; after the sort: {id : 13, tag : 19}
  WV_DEV static unsigned aPb(const uint32_t v) { return v & 0xffffu; }
  WV_DEV static unsigned aPool(const uint32_t v) { return (v >> 16) & 0x7ffu; }

  /// table-pass form: slot of `key` or ASM_NONE
  template <int KW>
  WV_DEV unsigned lookupSlotA(const Key<KW>& key) const
  {
    const uint32_t h   = keyHash(key);
    const unsigned tag = h >> 27;
    unsigned       b   = h & (LGL_BUCKETS - 1);
    for (unsigned probe = 0; probe < LGL_BUCKETS; ++probe) {
      const FBucket bk = *reinterpret_cast<const FBucket*>(slots + 4 * b);
      for (int i = 0; i < 4; ++i) {
        const uint32_t s = bk.s[i];
        if (s == LG_EMPTY) return ASM_NONE;
        if ((s >> 27) == tag && keyEq(keyAt<KW>(aPb(s)), key)) return 4 * b + unsigned(i);
      }
      b = (b + 1) & (LGL_BUCKETS - 1);
    }
    return ASM_NONE;
  }
  /// final form, behind the presence filter: id of `key` or ASM_NONE
  template <int KW>
  WV_DEV unsigned lookupId(const Key<KW>& key) const
  {
    const uint32_t h = keyHash(key);
    const unsigned t = (h >> 13) & (LGL_FILTER_BITS - 1);
    if (!((filter[t >> 5] >> (t & 31)) & 1u)) return ASM_NONE;
    const unsigned tag = h >> 13;
    unsigned       b   = h & (LGL_BUCKETS - 1);
    for (unsigned probe = 0; probe < LGL_BUCKETS; ++probe) {
      const FBucket bk = *reinterpret_cast<const FBucket*>(slots + 4 * b);
      for (int i = 0; i < 4; ++i) {
        const uint32_t s = bk.s[i];
        if (s == LG_EMPTY) return ASM_NONE;
        if ((s >> 13) == tag && keyEq(keyAt<KW>(idPb[s & 0x1fffu]), key)) return s & 0x1fffu;
      }
      b = (b + 1) & (LGL_BUCKETS - 1);
    }
    return ASM_NONE;
  }

  WV_DEV uint64_t plShift(const unsigned locus, const unsigned i) const
  {
    return P.pl_chunk_shift ? P.pl_chunk_shift[3 * size_t(locus / P.chunk_loci) + i] : uint64_t(0);
  }

  // ------------------------------------------------------------------------------------------------
  // stage 0: the locus' reads -> 2 bit + N bitmap in LDS (as LdsGraph::pack; the raw bytes are staged in pool + slots)
  // ------------------------------------------------------------------------------------------------
  WV_DEV bool pack(const unsigned locus)
  {
    const unsigned rBegin = P.locus_read_begin[locus], rEnd = P.locus_read_begin[locus + 1];
    nNormal               = rEnd - rBegin;
    if (nNormal + 2 * P.opt.maxAssemblyCount > LGL_MAX_READS) return false;
    W = (nNormal + 2 * P.opt.maxAssemblyCount + 63) / 64;
    if (W == 0) W = 1;
    const uint64_t plR = plShift(locus, 0), plC = plShift(locus, 1), plM = plShift(locus, 2);
    unsigned       cw = 0, mw = 0, nb = 0;  // code dwords, N-bitmap dwords, bases so far
    bool           tooLong = false;
    for (unsigned base = 0; base < nNormal; base += 64) {
      const unsigned r   = base + lane;
      unsigned       len = 0;
      if (r < nNormal) len = P.pl_codes ? P.pl_read_len[rBegin + r + plR] : unsigned(P.read_off[rBegin + r + 1] - P.read_off[rBegin + r]);
      if (len > 0xffffu) tooLong = true;
      const unsigned myC = (r < nNormal) ? (len + 15) / 16 + 1 : 0u;  // +1 padding dword so key fetches may read one past
      const unsigned myM = (r < nNormal) ? (len + 31) / 32 + 1 : 0u;
      unsigned       sc = myC, sm = myM, sb = (r < nNormal) ? len : 0u;
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned oc = wv::shfl(sc, wv::lane() - off), om = wv::shfl(sm, wv::lane() - off), ob = wv::shfl(sb, wv::lane() - off);
        if (wv::lane() >= off) {
          sc += oc;
          sm += om;
          sb += ob;
        }
      }
      const unsigned cwo = cw + sc - myC, mwo = mw + sm - myM;
      if (tw == 0 && r < nNormal && cwo <= 0xfffu && nb + sb - len <= 0xffffu) {
        rd[r]     = cwo | ((len & 0xffffu) << 12);
        rdm[r]    = uint16_t(mwo);
        rstart[r] = uint16_t(nb + sb - len);
      }
      cw += wv::readlane(sc, 63);
      mw += wv::readlane(sm, 63);
      nb += wv::readlane(sb, 63);
    }
    if (wv::any(tooLong) || cw + 2 > LGL_MAX_PILE + 2) return false;
    // the previous word length's contigs behind the locus' own reads (:898-905): codes only, a contig holds no N
    unsigned pcw = 0;
    nReads       = nNormal + nPseudo;
    pseudoPb0    = nPseudo ? 16u * cw : 0x10000u;
    if (nPseudo) {
      const unsigned len = (lane < nPseudo) ? unsigned(it->len[lane]) : 0u;
      const unsigned myC = (lane < nPseudo) ? (len + 15) / 16 + 1 : 0u;
      unsigned       sc  = myC;
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned oc = wv::shfl(sc, wv::lane() - off);
        if (wv::lane() >= off) sc += oc;
      }
      pcw = wv::readlane(sc, 63);
      if (pcw != it->codeWords || cw + pcw + 2 > LGL_MAX_PILE_ALL + 2) return false;
      if (tw == 0 && lane < nPseudo) {
        rd[nNormal + lane]     = (cw + sc - myC) | ((len & 0xffffu) << 12);
        rdm[nNormal + lane]    = 0;
        rstart[nNormal + lane] = 0;
      }
    }
    const unsigned cwN = cw;  // the locus' own reads end here
    cw += pcw;
    const unsigned cwPad = (cw + 2 + 3) & ~3u, mwPad = (mw + 2 + 3) & ~3u;
    if (cwPad + mwPad > LGL_DYN_DWORDS || 4 * cwPad + LGL_MAX_NODES + 16 > 4 * LGL_DYN_DWORDS) return false;
    if (!P.pl_codes && nb + 64 > LGL_STAGE_BYTES) return false;
    codeWords = cw + 2;
    nmask     = codes + cwPad;
    for (unsigned i = tid(); i < pcw; i += nThreads()) codes[cwN + i] = G.parena[it->off + i];
    cntId     = reinterpret_cast<uint8_t*>(codes + cwPad);  // (the N bitmap is dead once the reads' offsets are known)
    for (unsigned i = tid(); i < mw + 2; i += nThreads()) nmask[i] = 0;
    if (tid() == 0) hdr[LGL_H_FLAG] = 0;
    teamSync();
    if (P.pl_codes) {  // packed piles arrive in this layout: copy, 8 lanes per read
      for (unsigned base = 8 * tw; base < nNormal; base += 8 * tn) {
        const unsigned r = base + (lane >> 3);
        if (r >= nNormal) continue;
        const unsigned  d = rd[r], cwo = d & 0xfffu, len = (d >> 12) & 0xffffu, mwo = rdm[r];
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
        if (sawN) wv::atomic_or(&rd[r], 1u << 28);
      }
      if (tid() < 2) codes[cw + tid()] = 0;
      teamSync();
      return true;
    }
    bool           bad   = false;
    const uint32_t shift = P.chunk_shift ? P.chunk_shift[locus / P.chunk_loci] : 0u;
    // the locus' bases are one contiguous run of the input arena: 16-byte pieces into LDS, all loads in flight at once
    char* const stage = lds + LGL_OFF_POOL;
    unsigned    lead  = 0;
    {
      const uintptr_t g0 = reinterpret_cast<uintptr_t>(P.bases + P.read_off[rBegin] + shift);
      lead               = unsigned(g0 & 15);
      const u32x4*   gsrc    = reinterpret_cast<const u32x4*>(g0 - lead);
      const unsigned nChunks = (lead + nb + 15) / 16 + 1;  // (+1: the byte funnel reads up to four bytes past a read's last dword; the arena is padded)
      for (unsigned c = tid(); c < nChunks; c += nThreads()) reinterpret_cast<u32x4*>(stage)[c] = gsrc[c];
    }
    teamSync();
    // 8 lanes per read: lane (g, i) converts code dwords i, i+8, ... of read (base + g)
    for (unsigned base = 8 * tw; base < nNormal; base += 8 * tn) {
      const unsigned r = base + (lane >> 3);
      if (r >= nNormal) continue;
      const char*    src = stage + lead + rstart[r];
      const unsigned d = rd[r], cwo = d & 0xfffu, len = (d >> 12) & 0xffffu, mwo = rdm[r];
      const unsigned nCw = (len + 15) / 16 + 1;
      bool           sawN = false;
      for (unsigned wi = (lane & 7); wi < nCw; wi += 8) {
        uint32_t code = 0, nbits = 0;
        if (wi * 16 < len) {
          // 16 bases = five aligned dword reads + a byte funnel
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
      if (sawN) wv::atomic_or(&rd[r], 1u << 28);
    }
    if (tid() < 2) codes[cw + tid()] = 0;
    if (wv::any(bad) && lane == 0) wv::atomic_or(&hdr[LGL_H_FLAG], 1u);  // bytes outside {A,C,G,T,N}: the general path decides what is exact
    teamSync();
    return wv::atomic_load(&hdr[LGL_H_FLAG]) == 0;
  }

  // ------------------------------------------------------------------------------------------------
  // table pass (getKmerCounts :506-550): see the head of the file for the slot protocol
  // ------------------------------------------------------------------------------------------------
  template <int KW>
  WV_DEV bool tablePass()
  {
    if (tid() < LGL_MAX_READS) reinterpret_cast<uint32_t*>(lds + LGL_OFF_ANCH)[tid()] = LG_NO_ANCHOR;
    for (unsigned s = tid(); s < LGL_SLOTS; s += nThreads()) slots[s] = LG_EMPTY;
    {
      uint64_t* pz = reinterpret_cast<uint64_t*>(pool);
      for (unsigned i = tid(); i < LGL_POOL_CAP * 4; i += nThreads()) pz[i] = 0;
      if (poolOvf) {  // (22 KB per locus; needed by a few piles in a thousand -- and by most later word lengths: pseudo reads share their words)
        uint64_t* oz = reinterpret_cast<uint64_t*>(poolOvf);
        for (unsigned i = tid(); i < LGL_POOL_OVF * 4; i += nThreads()) oz[i] = 0;
      }
    }
    if (tid() == 0) hdr[LGL_H_POOLN] = 0;
    teamSync();
    bool fail = false;
    for (unsigned rBase = 0; rBase < nReads; rBase += 64) {
      const unsigned rMine = rBase + lane;
      const unsigned dV    = (rMine < nReads) ? rd[rMine] : 0u;
      const unsigned mV    = (rMine < nReads) ? unsigned(rdm[rMine]) : 0u;
      const unsigned rEnd  = (nReads - rBase < 64) ? (nReads - rBase) : 64u;
      for (unsigned ri = 0; ri < rEnd; ++ri) {
        const unsigned r = rBase + ri;
        if (r % tn != tw) continue;
        const unsigned d = wv::readlane(dV, int(ri)), cwo = d & 0xfffu, len = (d >> 12) & 0xffffu;
        if (len < k) continue;  // :522
        const bool     rdHasN = (d >> 28) & 1u;
        const unsigned mwo    = wv::readlane(mV, int(ri));
        const unsigned myLo = cwo * 16, myHi = cwo * 16 + len;  // this read's bases in the pile
        const unsigned setQ = r >> 6;
        const unsigned long long setBit = (unsigned long long)1 << (r & 63);
        bool           haveAnchor = false;  // (see readOffsets)
        for (unsigned j0 = 0; j0 + k <= len; j0 += 64) {
          const unsigned j  = j0 + lane;
          const unsigned pb = cwo * 16 + j;
          bool           todo = (j + k <= len) && !(rdHasN && windowHasN(mwo, j));  // :531
          Key<KW>        key;
          uint32_t       mine = 0;
          unsigned       tag = 0, b = 0, skip = 0, foundPb = 0x10000u;
          if (todo) {
            key              = keyAt<KW>(pb);
            const uint32_t h = keyHash(key);
            tag              = h >> 27;
            b                = h & (LGL_BUCKETS - 1);
            mine             = pb | (tag << 27);
          } else {
            for (int w = 0; w < KW; ++w) key.w[w] = 0;
          }
          unsigned probes = 0;
          // One probe round: read the bucket; the first slot (past `skip`) that is empty or carries the tag decides -- an empty slot is
          // claimed with a compare-and-swap, a tagged one has its word fetched and compared.  A lane that loses a swap goes round again.
          while (wv::any(todo)) {
            const FBucket bk = *reinterpret_cast<const FBucket*>(slots + 4 * b);
            unsigned      at = 4;
            uint32_t      sv = LG_EMPTY;
            for (int i = 3; i >= 0; --i) {
              const bool hit = unsigned(i) >= skip && (bk.s[i] == LG_EMPTY || (bk.s[i] >> 27) == tag);
              at             = hit ? unsigned(i) : at;
              sv             = hit ? bk.s[i] : sv;
            }
            const bool tryClaim = todo && at < 4 && sv == LG_EMPTY;
            const bool tryMatch = todo && at < 4 && sv != LG_EMPTY;
            uint32_t   casOld   = 0;
            if (tryClaim) casOld = wv::atomic_cas(&slots[4 * b + at], LG_EMPTY, mine);
            const Key<KW> got  = keyAt<KW>(tryMatch ? aPb(sv) : 0u);
            const bool    same = keyEq(got, key);
            if (tryClaim) {
              if (casOld == LG_EMPTY) todo = false;  // a new word: one read so far, implied by the first occurrence
            } else if (tryMatch && same) {
              foundPb = aPb(sv);
              todo    = false;
              if (foundPb < myLo || foundPb >= myHi) {
                // the word came from another read: it needs a set.  The owner's bit is added after the pass (ownerBits).
                unsigned pi = aPool(sv);
                if (pi == 0) {
                  const unsigned n = wv::atomic_add(&hdr[LGL_H_POOLN], 1u) + 1u;
                  if (n > LGL_POOL_CAP + (poolOvf ? LGL_POOL_OVF : 0u)) {
                    fail = true;
                  } else {
                    const uint32_t old = wv::atomic_cas(&slots[4 * b + at], sv, sv | (n << 16));
                    pi                 = (old == sv) ? n : aPool(old);  // (lost: the winner's set; entry n stays empty)
                  }
                }
                if (pi != 0) setAtomicOr(pi, setQ, setBit);
              }
            } else if (todo) {
              // another word under the tag: next slot of the bucket; a full bucket without the word: next bucket
              skip = tryMatch ? at + 1 : 4u;
              if (skip == 4) {
                b    = (b + 1) & (LGL_BUCKETS - 1);
                skip = 0;
                if (++probes >= LGL_BUCKETS) {
                  fail = true;
                  todo = false;
                }
              }
            }
          }
          if (!haveAnchor) {
            // the read's anchor: its first word that an EARLIER read had brought before (see LdsGraph::tablePass)
            const bool     cand = foundPb < myLo;
            const uint64_t mc   = wv::ballot(cand);
            if (mc) {
              const int      l   = wv::ctz(mc);
              const unsigned apb = wv::readlane(foundPb, l);
              if (lane == 0) reinterpret_cast<uint32_t*>(lds + LGL_OFF_ANCH)[r] = apb | ((j0 + unsigned(l)) << 16);
              haveAnchor = true;
            }
          }
        }
      }
    }
    if (wv::any(fail) && lane == 0) wv::atomic_or(&hdr[LGL_H_FLAG], 2u);
    teamSync();
    if (wv::atomic_load(&hdr[LGL_H_FLAG]) != 0) return false;
    wv::fence_acquire();  // (overflow sets: device memory updated by atomics, read plainly from here on)
    // the owners' bits: the read a word's first occurrence lies in
    for (unsigned s = tid(); s < LGL_SLOTS; s += nThreads()) {
      const uint32_t v = slots[s];
      if (v == LG_EMPTY || aPool(v) == 0) continue;
      const unsigned o = readOfPb(aPb(v));
      setPlainOr(aPool(v), o >> 6, uint64_t(1) << (o & 63));
    }
    teamSync();
    return true;
  }

  /// the locus' own reads among the bits of set qword q (the others: pseudo reads)
  WV_DEV uint64_t normalMask(const unsigned q) const
  {
    const unsigned lo = 64 * q;
    return (nNormal >= lo + 64) ? ~uint64_t(0) : ((nNormal > lo) ? ((uint64_t(1) << (nNormal - lo)) - 1) : uint64_t(0));
  }
  /// read that owns packed base index pb (the reads' code offsets ascend)
  WV_DEV unsigned readOfPb(const unsigned pb) const
  {
    const unsigned cwd = pb >> 4;
    unsigned       lo = 0, hi = nReads;  // rd[lo].cwo <= cwd < rd[hi].cwo
    while (hi - lo > 1) {
      const unsigned mid = (lo + hi) >> 1;
      if ((rd[mid] & 0xfffu) <= cwd) lo = mid; else hi = mid;
    }
    return lo;
  }

  // ------------------------------------------------------------------------------------------------
  // the reads' offsets on a common axis (LdsGraph::readOffsets, four reads per lane): anchors -> forest -> pointer doubling ->
  // the trees tied together through shared words.  Only ever costs the proof, never a result.
  // ------------------------------------------------------------------------------------------------
  template <int KW>
  WV_DEV void readOffsets()
  {
    if (tw == 0) {
      const uint32_t* anch = reinterpret_cast<const uint32_t*>(lds + LGL_OFF_ANCH);
      int32_t*        off  = reinterpret_cast<int32_t*>(lds + LGL_OFF_ANCH) + LGL_MAX_READS;
      uint8_t*        par  = reinterpret_cast<uint8_t*>(lds + LGL_OFF_PAR);
      int32_t         myOff[RPL];
      unsigned        myPar[RPL], myRoot[RPL];
      for (unsigned h = 0; h < RPL; ++h) {
        const unsigned r = lane + 64 * h;
        myOff[h]  = 0;
        myPar[h]  = r;
        myRoot[h] = r;
        if (r < nReads) {
          const uint32_t a = anch[r];
          if (a != LG_NO_ANCHOR) {
            const unsigned apb = a & 0xffffu, j = a >> 16;
            const unsigned r0  = readOfPb(apb);
            myPar[h] = r0;
            myOff[h] = int32_t(apb - 16u * (rd[r0] & 0xfffu)) - int32_t(j);
          }
        }
      }
      wv::sync();
      for (unsigned h = 0; h < RPL; ++h) {
        off[lane + 64 * h] = myOff[h];
        par[lane + 64 * h] = uint8_t(myPar[h]);
      }
      wv::sync();
      for (int round = 0; round < 8; ++round) {
        int32_t  po[RPL];
        unsigned pp[RPL];
        for (unsigned h = 0; h < RPL; ++h) {
          po[h] = off[myPar[h]];
          pp[h] = par[myPar[h]];
        }
        wv::sync();
        for (unsigned h = 0; h < RPL; ++h) {
          const unsigned r = lane + 64 * h;
          if (myPar[h] != r) {  // (a root keeps its offset)
            myOff[h] += po[h];
            if (pp[h] == myPar[h]) {  // the parent is a root: done after this addition
              off[r]    = myOff[h];
              par[r]    = uint8_t(r);
              myRoot[h] = myPar[h];
              myPar[h]  = r;
            } else {
              off[r]   = myOff[h];
              par[r]   = uint8_t(pp[h]);
              myPar[h] = pp[h];
            }
          }
        }
        wv::sync();
      }
      // (myRoot is the read the offset was completed through -- the root for the root's children, a finished inner read for the
      // reads below: the roots proper by pointer jumping)
      uint8_t* rootOf = reinterpret_cast<uint8_t*>(lds + LGL_OFF_WHIST);  // (the histograms are not in use before the sort)
      for (unsigned h = 0; h < RPL; ++h) rootOf[lane + 64 * h] = uint8_t(myRoot[h]);
      wv::sync();
      for (int round = 0; round < 8; ++round) {
        unsigned up[RPL];
        for (unsigned h = 0; h < RPL; ++h) up[h] = rootOf[rootOf[lane + 64 * h]];
        wv::sync();
        for (unsigned h = 0; h < RPL; ++h) rootOf[lane + 64 * h] = uint8_t(up[h]);
        wv::sync();
      }
    }
    teamSync();
    // ---- the trees tied together.  Sixteen waves insert reads at once, so which of two overlapping reads "was first" is a race and a
    // locus ends with a handful of trees, each at its own origin: one stray tree and the proof is lost.  Now that the table is complete
    // every tree looks for a word it shares with ANY other tree -- a word of one of its reads whose first occurrence lies in a read of
    // another tree gives the two trees' relative offset -- the trees in parallel over the waves (read-only), then wave 0 applies the
    // links one after the other (each with the offsets as they are by then).  Any phi that rises along every edge proves acyclicity,
    // so nothing here can make the proof unsound.
    if (!(G.flags & LG_FLAG_NO_RESCUE)) {
      uint8_t*  rootOf = reinterpret_cast<uint8_t*>(lds + LGL_OFF_WHIST);
      uint8_t*  roots  = rootOf + LGL_MAX_READS;                                   // [256] the roots, ascending
      uint32_t* link   = reinterpret_cast<uint32_t*>(rootOf + 2 * LGL_MAX_READS);  // [256][2] {read x | position j << 16, first occurrence} per root
      int32_t*  off    = reinterpret_cast<int32_t*>(lds + LGL_OFF_ANCH) + LGL_MAX_READS;
      for (unsigned round = 0; round < 3; ++round) {
        // the roots (every wave computes the same list; wave 0 writes it)
        unsigned nRoots = 0;
        for (unsigned h = 0; h < RPL; ++h) {
          const unsigned r    = lane + 64 * h;
          const bool     isR  = r < nReads && unsigned(rootOf[r]) == r;
          const uint64_t m    = wv::ballot(isR);
          if (tw == 0 && isR) roots[nRoots + unsigned(wv::popc(m & ((uint64_t(1) << lane) - 1)))] = uint8_t(r);
          nRoots += unsigned(wv::popc(m));
        }
        teamSync();
        if (nRoots <= 1) break;
        for (unsigned t = tw; t < nRoots; t += tn) {
          const unsigned s = roots[t];
          uint64_t       mem[RPL], memAll[RPL];
          for (unsigned h = 0; h < RPL; ++h) memAll[h] = mem[h] = wv::ballot(lane + 64 * h < nReads && unsigned(rootOf[lane + 64 * h]) == s);
          bool     tied = false;
          uint32_t l0 = 0xffffffffu, l1 = 0;
          for (unsigned tries = 0; tries < 12 && !tied; ++tries) {
            unsigned x = ASM_NONE;
            for (unsigned g = 0; g < RPL && x == ASM_NONE; ++g)
              if (mem[g]) {
                x = 64 * g + unsigned(wv::ctz(mem[g]));
                mem[g] &= mem[g] - 1;
              }
            if (x == ASM_NONE) break;
            const unsigned d = rd[x], cwo = d & 0xfffu, len = (d >> 12) & 0xffffu, mwo = rdm[x];
            const bool     rdHasN = (d >> 28) & 1u;
            for (unsigned j0 = 0; j0 + k <= len && !tied; j0 += 128) {
              const unsigned j     = j0 + 2 * lane;
              const bool     valid = (j + k <= len) && !(rdHasN && windowHasN(mwo, j));
              unsigned       slot  = ASM_NONE;
              if (valid) slot = lookupSlotA<KW>(keyAt<KW>(cwo * 16 + j));
              // (a) the word's first occurrence lies in a read of another tree; (b) it lies in this tree, but the word's set holds a
              // read of another tree (this tree's reads were first all along: only the sets know who else holds its words)
              bool     okA = false, okB = false;
              unsigned fpb = 0, qB = 0;
              if (slot != ASM_NONE) {
                const uint32_t v = slots[slot];
                fpb              = aPb(v);
                okA              = unsigned(rootOf[readOfPb(fpb)]) != s;
                if (!okA && aPool(v)) {
                  const Set st = setLoad(aPool(v));
                  for (unsigned h = 0; h < RPL; ++h) {
                    const uint64_t m = st.w[h] & ~memAll[h];
                    if (m && !okB) {
                      okB = true;
                      qB  = 64 * h + unsigned(wv::ctz(m));
                    }
                  }
                }
              }
              const uint64_t mA = wv::ballot(okA);
              if (mA) {
                const int l = wv::ctz(mA);
                tied        = true;
                l0          = x | ((j0 + 2 * unsigned(l)) << 16);
                l1          = wv::readlane(fpb, l);
                continue;
              }
              const uint64_t mB = wv::ballot(okB);
              if (mB) {
                // where does read q hold the word ?  (all lanes compare the word against the positions of q)
                const int      l  = wv::ctz(mB);
                const unsigned q  = wv::readlane(qB, l), jx = j0 + 2 * unsigned(l);
                const Key<KW>  kx = keyAt<KW>(cwo * 16 + jx);
                const unsigned dq = rd[q], cwq = dq & 0xfffu, lenq = (dq >> 12) & 0xffffu, mwq = rdm[q];
                const bool     qHasN = (dq >> 28) & 1u;
                for (unsigned i0 = 0; i0 + k <= lenq && !tied; i0 += 64) {
                  const unsigned jj    = i0 + lane;
                  const bool     match = (jj + k <= lenq) && !(qHasN && windowHasN(mwq, jj)) && keyEq(keyAt<KW>(cwq * 16 + jj), kx);
                  const uint64_t mm    = wv::ballot(match);
                  if (mm) {
                    tied = true;
                    l0   = x | (jx << 16);
                    l1   = cwq * 16 + i0 + unsigned(wv::ctz(mm));
                  }
                }
              }
            }
          }
          if (lane == 0) {
            link[2 * t]     = l0;
            link[2 * t + 1] = l1;
          }
        }
        teamSync();
        bool any = false;
        if (tw == 0) {
          for (unsigned t = 0; t < nRoots; ++t) {
            const uint32_t l0 = link[2 * t], l1 = link[2 * t + 1];
            if (l0 == 0xffffffffu) continue;
            const unsigned x = l0 & 0xffffu, j = l0 >> 16, o = readOfPb(l1);
            const unsigned rs = rootOf[x], ro = rootOf[o];
            if (rs == ro) continue;  // (an earlier link of this round has joined the two already)
            const int32_t delta = (off[o] + int32_t(l1 - 16u * (rd[o] & 0xfffu))) - (off[x] + int32_t(j));
#ifdef MANTA_WAVE_EMU
            if (std::getenv("MANTA_EMU_PROOF_TRACE") && lane == 0)
              std::fprintf(stderr, "  round %u: tree %u joins tree %u through read %u pos %u = read %u pos %u, delta %d (%u roots)\n", round, rs, ro, x, j, o, l1 - 16u * (rd[o] & 0xfffu), int(delta), nRoots);
#endif
            wv::sync();
            for (unsigned h = 0; h < RPL; ++h) {  // the whole tree of x moves onto o's axis
              const unsigned r = lane + 64 * h;
              if (r < nReads && unsigned(rootOf[r]) == rs) {
                off[r] += delta;
                rootOf[r] = uint8_t(ro);
              }
            }
            any = true;
            if (lane == 0 && G.stats) wv::atomic_add(&G.stats[1], 1u);
            wv::sync();
          }
          if (lane == 0) hdr[LGL_H_NSPEC] = any ? 1u : 0u;
        }
        teamSync();
        if (wv::atomic_load(&hdr[LGL_H_NSPEC]) == 0) break;
      }
    }
    // (rdm is dead after the table pass: its bytes take the offsets; one that does not fit 16 bits only loses the proof)
    if (tw == 0) {
      const int32_t* off = reinterpret_cast<const int32_t*>(lds + LGL_OFF_ANCH) + LGL_MAX_READS;
      for (unsigned h = 0; h < RPL; ++h) {
        const int32_t v     = off[lane + 64 * h];
        roff[lane + 64 * h] = int16_t((v > 30000) ? 30000 : ((v < -30000) ? -30000 : v));
      }
    }
    teamSync();
  }

  // ------------------------------------------------------------------------------------------------
  // the words in seed order (:686-696: count descending, k-mer ascending): an LSD radix sort over the WHOLE word -- ceil(k / 4) byte passes
  // from the last four bases to the first -- and then over 255 - count.  Result: sortA[id] = slot.
  // (Up to round 5: three passes over {first 8 bases, count} and the ties ranked by full key compares inside their run, as the small class
  // does.  A config-5 pile holds ~17-33 single-read error words per start position that share count and first bases, a tandem pile several
  // times that: the compare loop was four fifths of this phase -- 20 % of the kernel -- on plain piles and, with lexOrder's twin of it,
  // three quarters of the kernel on tandem piles; `tools/perf_big_rounds.py`.)
  // ------------------------------------------------------------------------------------------------
  template <int KW>
  WV_DEV bool sortWords()
  {
    if (tid() == 0) hdr[LGL_H_N] = 0;
    teamSync();
    for (unsigned sb = 64 * tw; sb < LGL_SLOTS; sb += 64 * tn) {
      const unsigned s   = sb + lane;
      const uint32_t v   = slots[s];
      const bool     occ = v != LG_EMPTY;
      if (occ) {
        // :537-548: a read adds one to a word's count, a pseudo read minCoverage
        unsigned c = (aPb(v) >= pseudoPb0) ? P.opt.minCoverage : 1u;
        if (aPool(v)) {
          const Set st = setLoad(aPool(v));
          c            = 0;
          for (unsigned q = 0; q < LgL::SETW; ++q) {
            const uint64_t nm = normalMask(q);
            c += unsigned(wv::popc(st.w[q] & nm)) + P.opt.minCoverage * unsigned(wv::popc(st.w[q] & ~nm));
          }
        }
        cntArr[s] = uint8_t(c);
      }
      const uint64_t m = wv::ballot(occ);
      if (m) {
        unsigned at = 0;
        if (lane == 0) at = wv::atomic_add(&hdr[LGL_H_N], unsigned(wv::popc(m)));
        at = wv::first(at);
        if (occ) sortA[at + unsigned(wv::popc(m & ((uint64_t(1) << lane) - 1)))] = uint16_t(s);
      }
    }
    teamSync();
    const unsigned n = wv::atomic_load(&hdr[LGL_H_N]);
    nNodes           = n;
    if (n > LGL_MAX_NODES) return false;
    if (n == 0) {
      nFat = nEligible = lowTier = 0;
      return true;
    }
    const unsigned chunk = (((n + tn - 1) / tn) + 63) & ~63u;  // elements of one wave, in order
    const unsigned c0 = chunk * tw, c1 = (c0 + chunk < n) ? (c0 + chunk) : n;
    uint16_t *     src = sortA, *dst = sortB;
    uint32_t*      myHist = whist + 256 * tw;
    const int      nb = int((k + 3) >> 2);  // bytes of a word
    static const unsigned SORT_IT = (LGL_MAX_NODES / LGL_WAVES + 63) / 64;  // 64-element steps of a wave's share
    static_assert(64 * SORT_IT * LGL_WAVES >= LGL_MAX_NODES, "a wave's share of the words in SORT_IT steps");
    for (int pass = 0; pass <= nb; ++pass) {
      // after the byte passes `src` is the words in lexicographic order: the repeat search of a cyclic graph wants exactly these ranks
      // (lexOrder) -- kept by table slot in the workgroup's device-memory workspace instead of sorting the words a second time
      if (pass == nb && lexBySlot)
        for (unsigned i = tid(); i < n; i += nThreads()) lexBySlot[src[i]] = uint16_t(i);
      auto digitOf = [&](const unsigned slot) -> unsigned {
        return (pass < nb) ? keyByte(aPb(slots[slot]), unsigned(nb - 1 - pass)) : (255u - unsigned(cntArr[slot]));
      };
      // this lane's (at most SORT_IT) elements and their digits, fetched ONCE per pass -- independent LDS chains in flight -- for the count
      // below and the scatter further down; the wave's own histogram needs no workgroup barrier to be cleared (nobody reads it before the next one)
      unsigned sv[SORT_IT], dv[SORT_IT];
#pragma unroll
      for (unsigned j = 0; j < SORT_IT; ++j) {
        const unsigned i = c0 + 64 * j + lane;
        sv[j]            = (i < c1) ? unsigned(src[i]) : 0xffffffffu;
      }
#pragma unroll
      for (unsigned j = 0; j < SORT_IT; ++j) dv[j] = (sv[j] != 0xffffffffu) ? digitOf(sv[j]) : 0u;
      for (unsigned q = lane; q < 256; q += 64) myHist[q] = 0;
      wv::sync();
#pragma unroll
      for (unsigned j = 0; j < SORT_IT; ++j)
        if (sv[j] != 0xffffffffu) wv::atomic_add(&myHist[dv[j]], 1u);
      teamSync();
      // per digit: running offsets over the waves, totals; then the digit bases (four quarters of 64 digits, one wave each)
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
        if (lane == 63) hdr[LGL_H_TOT + q] = inc;
      }
      teamSync();
      for (unsigned q = tw; q < 4; q += tn) {
        unsigned before = 0;
        for (unsigned p = 0; p < q; ++p) before += hdr[LGL_H_TOT + p];
        dbase[64 * q + lane] += before;
      }
      teamSync();
#pragma unroll
      for (unsigned j = 0; j < SORT_IT; ++j) {
        if (c0 + 64 * j >= c1) break;  // (wave-uniform)
        const bool     valid = sv[j] != 0xffffffffu;
        const unsigned slot  = valid ? sv[j] : 0u;
        const unsigned d     = dv[j];
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
      if (pass == nb && tid() == 0) {
        // ids below dbase[d] have counts above 255 - d
        const unsigned minCov = P.opt.minCoverage;
        hdr[LGL_H_TOT + 4] = dbase[254];                                                      // count >= 2
        hdr[LGL_H_TOT + 5] = (minCov <= 1) ? n : ((minCov > 255) ? 0u : dbase[256 - minCov]);   // count >= minCoverage
        hdr[LGL_H_TOT + 6] = (minCov + 4 > 255) ? 0u : dbase[255 - (minCov + 3)];               // first id with count <= minCoverage + 3 (+ 5: the same number of walk rounds)
      }
      uint16_t* t = src;
      src         = dst;
      dst         = t;
    }
    teamSync();
    nFat      = hdr[LGL_H_TOT + 4];
    nEligible = hdr[LGL_H_TOT + 5];
    lowTier   = hdr[LGL_H_TOT + 6];
    // nb + 1 passes: the sorted list sits in `src`; the callers read sortA
    if (src != sortA)
      for (unsigned i = tid(); i < n; i += nThreads()) sortA[i] = src[i];
    teamSync();
    if (lexById)  // ... and by id, while sortA still says which slot an id is (buildRecords turns it into id -> first occurrence)
      for (unsigned i = tid(); i < n; i += nThreads()) lexById[i] = lexBySlot[sortA[i]];
    teamSync();
    return true;
  }

  // ------------------------------------------------------------------------------------------------
  // the compact graph: sets and first occurrences to the slab, slot words -> {id, tag}, records with links (8 lookups per word),
  // side tables, the potential along every edge
  // ------------------------------------------------------------------------------------------------
  template <int KW>
  WV_DEV bool buildRecords(uint8_t* slab, const LgSlab& SL)
  {
    Set*      gPool = reinterpret_cast<Set*>(slab + SL.pool);
    uint16_t* gPb   = reinterpret_cast<uint16_t*>(slab + SL.pb);
    uint8_t*  gRd1  = slab + SL.rd1;
    // pass 1 (reads the pool and the slot-indexed counts; writes nothing they overlap with)
    for (unsigned i = tid(); i < nNodes; i += nThreads()) {
      const unsigned slot = sortA[i];
      const uint32_t v    = slots[slot];
      const unsigned pb   = aPb(v);
      const unsigned r = readOfPb(pb);
      if (i < nFat) {
        // count >= 2: the word has a set -- or it is held by ONE pseudo read that counts minCoverage >= 2: its set is made up here
        Set st;
        if (aPool(v)) {
          st = setLoad(aPool(v));
        } else {
          for (unsigned q = 0; q < LgL::SETW; ++q) st.w[q] = ((r >> 6) == q) ? (uint64_t(1) << (r & 63)) : 0;
        }
        gPool[i] = st;
      }
      gRd1[i]          = uint8_t(r);
      gPb[i]           = uint16_t(pb);
      idPb[i]          = uint16_t(pb);  // (sortA's bytes: id -> first occurrence; entry i is read by this thread only)
      cntId[i]         = cntArr[slot];
      phi[i]           = int16_t(int(roff[r]) + int(pb - 16u * (rd[r] & 0xfffu)));
      const uint32_t h = keyHash(keyAt<KW>(pb));
      slots[slot]      = i | ((h >> 13) << 13);
    }
    teamSync();
    for (unsigned i = tid(); i < LGL_FILTER_BITS / 32; i += nThreads()) filter[i] = 0;
    if (tid() == 0) {
      hdr[LGL_H_NSIB]  = 0;
      hdr[LGL_H_NSOVF] = 0;
      hdr[LGL_H_NPOVF] = 0;
      hdr[LGL_H_CYC]   = 0;
    }
    teamSync();
    for (unsigned s = tid(); s < LGL_SLOTS; s += nThreads()) {
      const uint32_t v = slots[s];
      if (v != LG_EMPTY) {
        const unsigned t = (v >> 13) & (LGL_FILTER_BITS - 1);
        wv::atomic_or(&filter[t >> 5], 1u << (t & 31));
      }
    }
    teamSync();
    tick(2, 4);
    uint16_t* sib  = reinterpret_cast<uint16_t*>(lds + LGL_OFF_SIB);
    uint16_t* sovf = reinterpret_cast<uint16_t*>(lds + LGL_OFF_SOVF);
    uint16_t* povf = reinterpret_cast<uint16_t*>(lds + LGL_OFF_POVF);
    bool      against = false;
    for (unsigned nb = 64 * tw; nb < nNodes; nb += 64 * tn) {
      const unsigned nd = nb + lane;
      if (nd >= nNodes) continue;
      const unsigned pb    = idPb[nd];
      const Key<KW>  key   = keyAt<KW>(pb);
      const int      myPhi = phi[nd];
      const unsigned cnt   = cntId[nd];
      unsigned       last  = 0;
      for (int w = 0; w < KW; ++w)
        if (unsigned(w) == ((k - 1) >> 4)) last = (key.w[w] >> (30 - 2 * ((k - 1) & 15))) & 3u;
      uint64_t rec = (uint64_t(cnt > 15 ? 15 : cnt) << LgL::CNT_SH) | (uint64_t(key.w[0] >> 30) << LgL::FIRST_SH) | (uint64_t(last) << LgL::LAST_SH);
      unsigned sId[4], pId[4];
      for (unsigned c = 0; c < 4; ++c) sId[c] = lookupId<KW>(keyShiftAppend<KW>(key, c));
      for (unsigned c = 0; c < 4; ++c) pId[c] = lookupId<KW>(keyShiftPrepend<KW>(key, c));
      unsigned sf[4], pf[4], ns = 0, np = 0;
      for (unsigned c = 0; c < 4; ++c) {
        if (sId[c] != ASM_NONE) {
          sf[ns++] = sId[c] + 1;
          if (sId[c] == nd)
            rec |= uint64_t(1) << LgL::SELF_SH;
          else if (int(phi[sId[c]]) <= myPhi) {
            against = true;
#ifdef MANTA_WAVE_EMU
            if (std::getenv("MANTA_EMU_PROOF_TRACE")) {
              const unsigned pbs = idPb[sId[c]];
              std::fprintf(stderr, "  against: word %u (read %u pos %u phi %d cnt %u) -> word %u (read %u pos %u phi %d cnt %u)\n", nd, readOfPb(pb), pb - 16 * (rd[readOfPb(pb)] & 0xfffu), myPhi, cnt,
                           sId[c], readOfPb(pbs), pbs - 16 * (rd[readOfPb(pbs)] & 0xfffu), int(phi[sId[c]]), unsigned(cntId[sId[c]]));
            }
#endif
          }
        }
        if (pId[c] != ASM_NONE) pf[np++] = pId[c] + 1;
      }
      if (ns > 0) rec |= uint64_t(sf[0]);
      if (ns == 2) rec |= uint64_t(sf[1]) << 13;
      if (ns > 2) {
        const unsigned at = wv::atomic_add(&hdr[LGL_H_NSOVF], 1u);
        rec |= uint64_t(1) << LgL::SOVF_SH;
        if (at < LGL_OVF_CAP) {
          rec |= uint64_t(at) << 13;
          sovf[4 * at + 0] = uint16_t(sf[1]);
          sovf[4 * at + 1] = uint16_t(sf[2]);
          sovf[4 * at + 2] = uint16_t(ns > 3 ? sf[3] : 0u);
          sovf[4 * at + 3] = 0;
        }
      }
      if (np > 0) rec |= uint64_t(pf[0]) << 26;
      if (np == 2) rec |= uint64_t(pf[1]) << 39;
      if (np > 2) {
        const unsigned at = wv::atomic_add(&hdr[LGL_H_NPOVF], 1u);
        rec |= uint64_t(1) << LgL::POVF_SH;
        if (at < LGL_OVF_CAP) {
          rec |= uint64_t(at) << 39;
          povf[4 * at + 0] = uint16_t(pf[1]);
          povf[4 * at + 1] = uint16_t(pf[2]);
          povf[4 * at + 2] = uint16_t(np > 3 ? pf[3] : 0u);
          povf[4 * at + 3] = 0;
        }
      }
      nodes[nd] = rec;
      if (np == 0) {
        // a word without a predecessor: its siblings (the words that differ in the last base only, :185-210) cannot be found
        // through a predecessor's successor list -> side table
        unsigned found[3] = {LG_NO_SLOT, LG_NO_SLOT, LG_NO_SLOT};
        unsigned nf = 0;
        for (unsigned c = 0; c < 4; ++c) {
          if (c == last) continue;
          Key<KW> s2 = key;
          keySetBase(s2, k - 1, c);
          const unsigned ss = lookupId<KW>(s2);
          if (ss != ASM_NONE) found[nf++] = ss;
        }
        if (nf) {
          const unsigned at = wv::atomic_add(&hdr[LGL_H_NSIB], 1u);
          if (at < LG_SIB_CAP) {
            sib[4 * at + 0] = uint16_t(nd);
            sib[4 * at + 1] = uint16_t(found[0]);
            sib[4 * at + 2] = uint16_t(found[1]);
            sib[4 * at + 3] = uint16_t(found[2]);
          }
        }
      }
    }
    if (wv::any(against) && lane == 0) wv::atomic_or(&hdr[LGL_H_CYC], 1u);
    teamSync();
    tick(2, 5);
    if (wv::atomic_load(&hdr[LGL_H_NSIB]) > LG_SIB_CAP || wv::atomic_load(&hdr[LGL_H_NSOVF]) > LGL_OVF_CAP ||
        wv::atomic_load(&hdr[LGL_H_NPOVF]) > LGL_OVF_CAP)
      return false;
    return true;
  }

  // ------------------------------------------------------------------------------------------------
  // A graph without a proof of acyclicity may need the reference's repeat search (repeat_big_kernel), whose visiting order starts from
  // the insertion order of the words: reads in order, a read's new words in LEXICOGRAPHIC order (:516-548).  The words' lexicographic
  // ranks go to the slab: stable byte passes over the ids, the whole word, last byte first.
  // (After buildRecords: lists in the slot table's bytes, histograms in the potentials' bytes -- the records sit where sortWords had them.)
  // ------------------------------------------------------------------------------------------------
  WV_DEV void radixPassIds(const uint16_t* src, uint16_t* dst, const unsigned n, const unsigned byteIdx, uint32_t* hist)
  {
    const unsigned chunk = (((n + tn - 1) / tn) + 63) & ~63u;
    const unsigned c0 = chunk * tw, c1 = (c0 + chunk < n) ? (c0 + chunk) : n;
    uint32_t*      myHist = hist + 256 * tw;
    for (unsigned i = tid(); i < 256 * tn; i += nThreads()) hist[i] = 0;
    teamSync();
    auto digitOf = [&](const unsigned id) -> unsigned { return keyByte(idPb[id], byteIdx); };
    for (unsigned i0 = c0; i0 < c1; i0 += 64) {
      const unsigned i = i0 + lane;
      if (i < c1) wv::atomic_add(&myHist[digitOf(src[i])], 1u);
    }
    teamSync();
    for (unsigned q = tw; q < 4; q += tn) {
      const unsigned d   = 64 * q + lane;
      unsigned       run = 0;
      for (unsigned w = 0; w < tn; ++w) {
        const unsigned c  = hist[256 * w + d];
        hist[256 * w + d] = run;
        run += c;
      }
      unsigned inc = run;
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = wv::shfl(inc, int(lane) - off);
        if (int(lane) >= off) inc += o;
      }
      dbase[d] = inc - run;
      if (lane == 63) hdr[LGL_H_TOT + q] = inc;
    }
    teamSync();
    for (unsigned q = tw; q < 4; q += tn) {
      unsigned before = 0;
      for (unsigned p = 0; p < q; ++p) before += hdr[LGL_H_TOT + p];
      dbase[64 * q + lane] += before;
    }
    teamSync();
    for (unsigned i0 = c0; i0 < c1; i0 += 64) {
      const unsigned i     = i0 + lane;
      const bool     valid = i < c1;
      const unsigned id    = valid ? unsigned(src[i]) : 0u;
      const unsigned d     = valid ? digitOf(id) : 0u;
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
        dst[base + unsigned(wv::popc(peers & ((uint64_t(1) << lane) - 1)))] = uint16_t(id);
        if ((peers >> lane) == 1u) myHist[d] += unsigned(wv::popc(peers));
      }
      wv::sync();
    }
    teamSync();
  }
  // (as a real call -- one graph in seven takes it -- the kernel spills more, not less: 144 VGPRs and 1 KB of scratch against 109 / 180 B)
  template <int KW>
  WV_DEV void lexOrder(uint16_t* gLex)
  {
    const unsigned n = nNodes;
    if (lexById) {  // sortWords left them
      for (unsigned i = tid(); i < n; i += nThreads()) gLex[i] = lexById[i];
      teamSync();
      return;
    }
    uint16_t* bufA = reinterpret_cast<uint16_t*>(lds + LGL_OFF_SLOTS);
    uint16_t* bufB = bufA + LGL_SLOTS;
    uint32_t* hist = reinterpret_cast<uint32_t*>(lds + LGL_OFF_SORTB);
    for (unsigned i = tid(); i < n; i += nThreads()) bufA[i] = uint16_t(i);
    teamSync();
    // LSD over the whole word, last byte first (see sortWords: ranking the runs of an 8-base prefix by full key compares was the larger
    // half of this kernel's time on tandem piles)
    uint16_t *src = bufA, *dst = bufB;
    for (int b = int((k + 3) >> 2) - 1; b >= 0; --b) {
      radixPassIds(src, dst, n, unsigned(b), hist);
      uint16_t* t = src;
      src         = dst;
      dst         = t;
    }
    for (unsigned i = tid(); i < n; i += nThreads()) gLex[src[i]] = uint16_t(i);
    teamSync();
  }

  // ------------------------------------------------------------------------------------------------
  // A graph without a proof: the two-sided Kahn peel (contig_kernel's cycle test, here by sixteen waves on the records in LDS; self loops
  // aside as there).  Everything peeled: the graph is acyclic after all.  What is left is the CORE -- every cycle lies inside it -- and
  // goes to the slab as a bitmap: repeat_big_kernel looks for the strongly connected components among those words only.
  // (After buildRecords: state bytes in the presence filter's bytes, the queue in the slot table's.)  Returns the size of the core.
  // ------------------------------------------------------------------------------------------------
  WV_DEV unsigned peelCore(uint32_t* gCore)
  {
    uint32_t*       st    = reinterpret_cast<uint32_t*>(lds + LGL_OFF_CNT);
    uint16_t*       queue = reinterpret_cast<uint16_t*>(lds + LGL_OFF_SLOTS);
    uint16_t*       jf    = queue + LGL_SLOTS;                                   // second half of the slot table's bytes
    uint16_t*       jb    = reinterpret_cast<uint16_t*>(lds + LGL_OFF_SORTB);    // (the potentials are done with)
    const uint16_t* sovf  = reinterpret_cast<const uint16_t*>(lds + LGL_OFF_SOVF);
    const uint16_t* povf  = reinterpret_cast<const uint16_t*>(lds + LGL_OFF_POVF);
    const unsigned  nS = wv::atomic_load(&hdr[LGL_H_NSOVF]), nP = wv::atomic_load(&hdr[LGL_H_NPOVF]);
    const unsigned  n = nNodes, stDw = (n + 3) / 4;
    // A k-mer graph is chains: a node-by-node peel takes as many levels as the longest path has words (thousands, two or three words per
    // level: 25 % of this kernel on tandem piles, `tools/perf_big_rounds.py`).  So the chains are contracted first.  An INNER word has
    // exactly one predecessor and one successor (self loops aside); pointer jumping gives every inner word the two non-inner words
    // (terminals) its chain hangs between.  A chain b -> ... -> t goes exactly when b goes (its first word loses its only predecessor, and
    // so on down to t's in-degree) or when t goes (backwards, down to b's out-degree): for the peel it is ONE edge b -> t.  The peel then
    // runs over the terminals only -- tens of levels -- on one wave (no workgroup barrier per level), and an inner word is peeled iff one of
    // its two terminals is.  A cycle of inner words alone never resolves to a terminal and stays, as it must.  The fixpoint of "remove
    // what has no predecessor or no successor" does not depend on the order, so the core is the one the node-by-node peel left.
    // State byte: in-degree : 3, out-degree : 3, 0x40 queued / peeled (terminals), 0x80 inner.
    for (unsigned w = tid(); w < stDw; w += nThreads()) st[w] = 0;
    if (tid() == 0) hdr[LGL_H_N] = 0;  // the queue's tail
    teamSync();
    auto stByte = [&](const unsigned nd) -> unsigned { return (st[nd >> 2] >> (8 * (nd & 3))) & 0xffu; };
    for (unsigned nd = tid(); nd < n; nd += nThreads()) {
      const FRec8    w  = nodes[nd];
      const uint64_t sl = R::links(w, nd, true, sovf, nS), pl = R::links(w, nd, false, povf, nP);
      unsigned       id = 0, od = 0, onlyS = nd, onlyP = nd;
      for (unsigned c = 0; c < 4; ++c) {
        const unsigned s2 = R::linkId(sl, c), p2 = R::linkId(pl, c);
        if (s2 != ASM_NONE && s2 != nd) {
          od++;
          onlyS = s2;
        }
        if (p2 != ASM_NONE && p2 != nd) {
          id++;
          onlyP = p2;
        }
      }
      const bool     src   = (id == 0 || od == 0);
      const bool     inner = (id == 1 && od == 1);
      const unsigned v     = id | (od << 3) | (src ? 0x40u : 0u) | (inner ? 0x80u : 0u);
      wv::atomic_or(&st[nd >> 2], v << (8 * (nd & 3)));
      if (src) queue[wv::atomic_add(&hdr[LGL_H_N], 1u)] = uint16_t(nd);
      jf[nd] = uint16_t(inner ? onlyS : nd);
      jb[nd] = uint16_t(inner ? onlyP : nd);
    }
    teamSync();
    {
      unsigned rounds = 1;
      while ((1u << rounds) < n) ++rounds;
      for (unsigned r = 0; r < rounds; ++r) {  // (in place: a pointer only ever moves further along its chain)
        for (unsigned nd = tid(); nd < n; nd += nThreads()) {
          if (!(stByte(nd) & 0x80u)) continue;
          jf[nd] = jf[jf[nd]];
          jb[nd] = jb[jb[nd]];
        }
        teamSync();
      }
    }
    unsigned head = 0;
    while (tw == 0) {
      const unsigned tail = wv::first(wv::atomic_load(&hdr[LGL_H_N]));
      if (tail == head) break;
      for (unsigned i = head + lane; i < tail; i += 64) {
        const unsigned nd = queue[i];
        const FRec8    w  = nodes[nd];
        const uint64_t sl = R::links(w, nd, true, sovf, nS), pl = R::links(w, nd, false, povf, nP);
        for (unsigned c = 0; c < 4; ++c) {
          unsigned s2 = R::linkId(sl, c);
          if (s2 != ASM_NONE && s2 != nd) {
            if (stByte(s2) & 0x80u) s2 = jf[s2];  // the terminal at the far end of the chain
            if (!(stByte(s2) & 0x80u)) {
              const unsigned sh  = 8 * (s2 & 3);
              const unsigned old = wv::atomic_sub(&st[s2 >> 2], 1u << sh) >> sh;
              if ((old & 0x7u) == 1u && !(wv::atomic_or(&st[s2 >> 2], 0x40u << sh) & (0x40u << sh))) queue[wv::atomic_add(&hdr[LGL_H_N], 1u)] = uint16_t(s2);
            }
          }
          unsigned p2 = R::linkId(pl, c);
          if (p2 != ASM_NONE && p2 != nd) {
            if (stByte(p2) & 0x80u) p2 = jb[p2];
            if (!(stByte(p2) & 0x80u)) {
              const unsigned sh  = 8 * (p2 & 3);
              const unsigned old = wv::atomic_sub(&st[p2 >> 2], 8u << sh) >> sh;
              if ((old & 0x38u) == 8u && !(wv::atomic_or(&st[p2 >> 2], 0x40u << sh) & (0x40u << sh))) queue[wv::atomic_add(&hdr[LGL_H_N], 1u)] = uint16_t(p2);
            }
          }
        }
      }
      wv::sync();
      head = tail;
    }
    teamSync();
    if (tid() == 0) hdr[LGL_H_N] = 0;  // now: the words left
    teamSync();
    for (unsigned d = tid(); d < LgL::UNUSED_DW; d += nThreads()) {
      uint32_t b = 0;
      for (unsigned j = 0; j < 32; ++j) {
        const unsigned nd = 32 * d + j;
        if (nd >= n) break;
        const unsigned v = stByte(nd);
        bool           gone;
        if (!(v & 0x80u)) {
          gone = (v & 0x40u) != 0;
        } else {
          const unsigned vf = stByte(jf[nd]), vb = stByte(jb[nd]);
          gone = (!(vf & 0x80u) && (vf & 0x40u)) || (!(vb & 0x80u) && (vb & 0x40u));
        }
        if (!gone) b |= 1u << j;
      }
      gCore[d] = b;
      if (b) wv::atomic_add(&hdr[LGL_H_N], unsigned(wv::popc(b)));
    }
    teamSync();
    return wv::atomic_load(&hdr[LGL_H_N]);
  }

  /// round 0's walk list: the first seed and beside it the words most likely to follow it -- the low count tiers in seed order, ONE WORD
  /// PER UNBRANCHED STRETCH (the words of a stretch come out of one walk; contig_big_kernel's stretchSeedList makes the later lists the
  /// same way).  The first contig takes the well-covered words with it; what follows starts a few counts above minCoverage (a pile's
  /// second seed typically has count 3-4: the window starts at count <= minCoverage + 3) and runs through the count-2 words into the
  /// single-read ones: 256 words are looked at.  (After lexOrder: the chain arrays lie in the potentials' bytes.)
  WV_DEV unsigned speculationList(uint16_t* spec, const bool topToo)
  {
    static const unsigned WIN = 256;
    uint16_t* label = reinterpret_cast<uint16_t*>(lds + LGL_OFF_SORTB);
    uint16_t* dist  = label + WIN;
    uint32_t* dupW  = reinterpret_cast<uint32_t*>(dist + WIN);  // [WIN / 32] duplicate bits
    const unsigned e0 = lowTier, e1 = (nEligible < lowTier + WIN) ? nEligible : (lowTier + WIN);
    const unsigned nE = (e1 > e0) ? (e1 - e0) : 0u;
    // a word's only successor / number of predecessors, self loops aside (a word with an overflow entry has three or more)
    auto outOnly = [&](const FRec8 w, const unsigned nd, unsigned& od) -> unsigned {
      unsigned only = ASM_NONE;
      od            = 0;
      if (R::sOvf(w)) {
        od = 3;
        return only;
      }
      for (unsigned c = 0; c < 2; ++c) {
        const unsigned f = R::succ(w, c);
        if (f && f - 1 != nd) {
          od++;
          only = f - 1;
        }
      }
      return only;
    };
    auto inDeg = [&](const FRec8 w, const unsigned nd) -> unsigned {
      if (R::pOvf(w)) return 3u;
      unsigned id = 0;
      for (unsigned c = 0; c < 2; ++c) {
        const unsigned f = R::pred(w, c);
        if (f && f - 1 != nd) id++;
      }
      return id;
    };
    // one window of the seed order: entries e0 .. e0 + nE, one per stretch, appended to the list up to `quota` entries in all
    unsigned n0 = (nEligible > 0) ? 1u : 0u;  // (entry 0: the first seed, id 0)
    auto addWindow = [&](const unsigned w0, const unsigned wN, const unsigned quota) {
      if (tid() < WIN / 32) dupW[tid()] = 0;
      if (tid() < wN) {
        unsigned cur = w0 + tid(), steps = 0;
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
      if (wN > 0) {
        // entry i is dropped when an earlier entry of its stretch lies upstream of it (that entry's walk comes through here first)
        unsigned eL[WIN / 64], eD[WIN / 64];
        bool     dup[WIN / 64];
        for (unsigned h = 0; h < WIN / 64; ++h) {
          const unsigned i = lane + 64 * h;
          eL[h]            = (i < wN) ? unsigned(label[i]) : ASM_NONE;
          eD[h]            = (i < wN) ? unsigned(dist[i]) : 0u;
          dup[h]           = false;
        }
        for (unsigned j = tw; j < wN; j += tn) {
          const unsigned lj = label[j], dj = dist[j];
          for (unsigned h = 0; h < WIN / 64; ++h)
            if (j < lane + 64 * h && lj == eL[h] && dj > eD[h]) dup[h] = true;
        }
        for (unsigned h = 0; h < WIN / 64; ++h) {
          const uint64_t m = wv::ballot(dup[h]);
          if (lane == 0 && m) {
            wv::atomic_or(&dupW[2 * h], uint32_t(m));
            wv::atomic_or(&dupW[2 * h + 1], uint32_t(m >> 32));
          }
        }
      }
      teamSync();
      for (unsigned h = 0; h < WIN / 64; ++h) {  // (every wave counts, wave 0 writes)
        const unsigned i    = lane + 64 * h;
        const bool     dup  = (dupW[2 * h + (lane >> 5)] >> (lane & 31)) & 1u;
        const bool     keep = (i < wN) && !dup && (w0 + i) != 0u;
        const uint64_t mk   = wv::ballot(keep);
        const unsigned pos  = n0 + unsigned(wv::popc(mk & ((uint64_t(1) << lane) - 1)));
        if (tw == 0 && keep && pos < quota) spec[pos] = uint16_t(w0 + i);
        n0 += unsigned(wv::popc(mk));
        if (n0 > quota) n0 = quota;
      }
      teamSync();
    };
    if (nEligible > 0) {
      // A graph without a proof of acyclicity is probably cyclic: its first contig ends at the repeat and the next seeds are more of the
      // well-covered words -- a third of the list goes to the top of the seed order.
      if (topToo) addWindow(1, (nEligible > 1 + 128) ? 128u : (nEligible - 1), 22);
      addWindow(e0, nE, 64);
      if (tid() == 0) spec[0] = 0;  // the first seed: highest count, smallest word = id 0
    }
    if (tid() == 0) hdr[LGL_H_NSPEC] = n0;
    teamSync();
    return wv::atomic_load(&hdr[LGL_H_NSPEC]);
  }

#ifdef MANTA_WAVE_EMU
#define LGL_TRACE(why) do { if (std::getenv("MANTA_EMU_PUNT_TRACE") && tid() == 0) std::fprintf(stderr, "  graph_big_kernel punts locus %u: %s (k %u, %u reads, %u words, %u sets handed out)\n", locus, why, k, nNormal, nNodes, hdr[LGL_H_POOLN]); } while (0)
#else
#define LGL_TRACE(why) do { } while (0)
#endif
  template <int KW>
  WV_DEV bool runK(const unsigned locus)
  {
    nNodes = 0;
    if (!tablePass<KW>()) {
      LGL_TRACE("table pass (table or set pool full)");
      why(3);
      return false;
    }
    tick(1, 1);
    readOffsets<KW>();
    tick(1, 2);
    if (!sortWords<KW>()) {
      LGL_TRACE("too many words");
      why(4);
      return false;
    }
    tick(2, 3);
    // slab for this locus
    const LgSlab   SL    = lgSlabL(nNodes, nFat, codeWords);
    uint64_t       bytes = SL.total;
    if (tid() == 0) {
      // a later word length of a locus builds in the slab of the previous one when it fits (the words of a pile grow slowly with the
      // word length: a new slab of a later round comes with a quarter of headroom)
      unsigned long long off;
      if (it && bytes <= it->slabCap) {
        off = G.slab_off[locus];
      } else {
        const uint64_t want = it ? (bytes + bytes / 4 + 15) & ~uint64_t(15) : bytes;
        off                 = wv::atomic_add(G.arena_used, (unsigned long long)want);
        if (G.iter) G.iter[locus].slabCap = (off + want <= G.arena_cap) ? uint32_t(want) : 0u;
        if (off + want > G.arena_cap) off = G.arena_cap;  // (full: reported below)
      }
      hdr[LGL_H_OFF_LO] = uint32_t(off);
      hdr[LGL_H_OFF_HI] = uint32_t(off >> 32);
    }
    teamSync();
    const uint64_t off = (uint64_t(wv::atomic_load(&hdr[LGL_H_OFF_HI])) << 32) | wv::atomic_load(&hdr[LGL_H_OFF_LO]);
    if (off + bytes > G.arena_cap) {
      LGL_TRACE("slab arena full");
      why(5);
      return false;
    }
    uint8_t* slab = G.arena + off;
    if (!buildRecords<KW>(slab, SL)) {
      LGL_TRACE("side tables full");
      why(4);
      return false;
    }
    bool           acyclic = wv::atomic_load(&hdr[LGL_H_CYC]) == 0 && !(G.flags & LG_FLAG_NO_PROOF);
    if (acyclic && tid() == 0 && G.stats) wv::atomic_add(&G.stats[0], 1u);
    // With the word-length rounds on, a graph without a proof is peeled here; if something is left it goes to repeat_big_kernel first
    // (components, repeat search, LDS class), with the words' lexicographic ranks and the core's bitmap in its slab.
    unsigned       nCorePeel = 0;
    if (!acyclic && G.iter != nullptr) {
#ifdef MANTA_LG_PROFILE_PEEL  // (one-off: what came before the peel to slot 6, the peel to slot 0, the ranks to slot 1, the speculation list to slot 6)
      tick(2, 6);
#endif
      nCorePeel = peelCore(reinterpret_cast<uint32_t*>(slab + SL.flags));
#ifdef MANTA_LG_PROFILE_PEEL
      tick(0, 0);
#endif
      if (nCorePeel == 0) acyclic = true;  // (no proof from the reads' offsets, but nothing survives the peel)
    }
    const bool     toRepeat = !acyclic && G.iter != nullptr;
    if (toRepeat) lexOrder<KW>(reinterpret_cast<uint16_t*>(slab + SL.lex));
#ifdef MANTA_LG_PROFILE_PEEL
    tick(1, 1);
#endif
    const unsigned need    = ckNeedOf<LgL>(nNodes, nFat, acyclic);
    unsigned       cls     = LG_CLASSES;
    for (unsigned c = LG_CLASSES; c-- > 0;)
      if (G.class_bytes[c] && need <= G.class_bytes[c]) cls = c;
    if (cls == LG_CLASSES && !toRepeat) {
      LGL_TRACE("graph fits no contig LDS class");
      why(4);
      return false;
    }
    uint16_t*      gSpec = reinterpret_cast<uint16_t*>(slab + SL.spec);
    const unsigned nSpec = speculationList(gSpec, !acyclic);
    tick(2, 6);
    // the rest of the slab
    FRec8* gRec = reinterpret_cast<FRec8*>(slab + SL.recs);
    for (unsigned i = tid(); i < nNodes; i += nThreads()) gRec[i] = nodes[i];
    const unsigned nSib = wv::atomic_load(&hdr[LGL_H_NSIB]), nSovf = wv::atomic_load(&hdr[LGL_H_NSOVF]), nPovf = wv::atomic_load(&hdr[LGL_H_NPOVF]);
    {
      // the three side tables lie back to back in LDS and in the slab (fixed capacities)
      const uint16_t* tsrc = reinterpret_cast<const uint16_t*>(lds + LGL_OFF_SIB);
      uint16_t*       tdst = reinterpret_cast<uint16_t*>(slab + SL.sib);
      for (unsigned i = tid(); i < 4 * (LG_SIB_CAP + 2 * LGL_OVF_CAP); i += nThreads()) tdst[i] = tsrc[i];
    }
    uint32_t* gCodes = reinterpret_cast<uint32_t*>(slab + SL.codes);
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
      h.nSovf     = nSovf;
      h.nPovf     = nPovf;
      h.acyclic   = acyclic ? 1u : 0u;
      h.nPseudo   = nPseudo;
      h.cyclic    = 0;
      h.nCore     = nCorePeel;  // (repeat_big_kernel: the peel's core; it leaves the number of words on cycles here)
      *reinterpret_cast<LgHdr*>(slab) = h;
      G.slab_off[locus]               = off;
      if (toRepeat)
        G.cyc_ids[wv::atomic_add(G.cyc_count, 1u)] = locus;
      else
        G.class_ids[size_t(cls) * G.class_stride + wv::atomic_add(&G.class_count[cls], 1u)] = locus;
    }
    tick(4, 7);
    return true;
  }

  /// false: the general path takes the locus.  MAXKW: the widest key (dwords) this instantiation carries code for
  template <int MAXKW>
  WV_DEV bool run(const unsigned locus)
  {
    const unsigned minWL = P.locus_min_wl ? P.locus_min_wl[locus] : P.opt.minWordLength;
    const unsigned maxWL = P.locus_max_wl ? P.locus_max_wl[locus] : P.opt.maxWordLength;
    if (minWL == 0 || maxWL > 16u * ASM_MAX_KW || minWL > maxWL || 2 * P.opt.maxAssemblyCount > ASM_MAX_CAND) return false;
    if (P.opt.minCoverage > 15 || P.opt.minConservativeCoverage > 15 || P.opt.maxAssemblyCount > 20) return false;  // (records keep counts up to 15)
    k      = minWL;
    nNodes = 0;
    tMark  = wv::clock();
    hdr[LGL_H_POOLN] = 0;
    if (G.iter && G.round > 0) {  // a later word length of this locus (contig_big_kernel left the state)
      it      = &G.iter[locus];
      k       = it->k;
      nPseudo = it->nPseudo;
      if (k < minWL || k > maxWL || nPseudo > LGL_MAX_PSEUDO || nPseudo > 2 * P.opt.maxAssemblyCount) return false;
    }
    if (!pack(locus)) {
      LGL_TRACE("pack (envelope / alphabet)");
      why(2);
      return false;
    }
    tick(0, 0);
    if (nNormal + nPseudo * P.opt.minCoverage > 255) {  // (counts by slot are bytes)
      why(2);
      return false;
    }
    const unsigned kw = (k + 15) >> 4;
    if (kw > unsigned(MAXKW)) {
      why(2);
      return false;
    }
    if (kw <= 2) return runK<2>(locus);
    if (MAXKW >= 4 && kw <= 4) return runK<(MAXKW >= 4 ? 4 : 2)>(locus);
    return runK<MAXKW>(locus);
  }
};

/// persistent workgroups of LGL_WAVES wavefronts, LGL_BUDGET bytes of dynamic LDS each (one per CU); parameters as graph_kernel.
/// One instantiation per key width (word lengths up to 32 / 64 / 80 / 128).
template <int MAXKW>
WV_KERNEL_WG(LGL_WAVES) WV_WAVES_PER_SIMD(4) void graph_big_kernel(const LgArgs A)
{
  const AsmParams& P = A.P;
  const LgParams&  G = A.G;
  char*          lds = wv::lds_single();
  uint32_t*      hdr = reinterpret_cast<uint32_t*>(lds + LGL_OFF_HDR);
  const unsigned tw  = unsigned(wv::wave_in_wg());
  const unsigned nLoci = P.n_loci_dev ? wv::first(wv::atomic_load(P.n_loci_dev)) : P.n_loci;  // (later rounds: the previous round's list)
  while (true) {
    if (tw == 0 && wv::lane() == 0) hdr[LGL_H_SLOT] = wv::atomic_add(P.counter, 1u);
    wv::sync();
    wv::wg_barrier();
    const unsigned slot = wv::first(wv::atomic_load(&hdr[LGL_H_SLOT]));
    if (slot >= nLoci) break;
    const unsigned locus   = P.locus_ids ? P.locus_ids[slot] : slot;
    const bool     arrived = !P.upload_chunks_done || asmWaitUploaded(P, locus);
    bool           ok      = false;
    if (arrived) {
      LdsGraphL g(P, G, lds);
      ok = g.template run<MAXKW>(locus);
    }
    wv::sync();
    if (tw == 0 && !ok && wv::lane() == 0) P.punt_ids[wv::atomic_add(P.punt_count, 1u)] = locus;
    wv::sync();
    wv::wg_barrier();  // (the slot word is rewritten next)
  }
}

#if MANTA_TU != MANTA_TU_ALL
#if MANTA_TU == MANTA_TU_GRAPH_BIG
#define MANTA_X
#else
#define MANTA_X extern
#endif
MANTA_X template __global__ void graph_big_kernel<5>(const LgArgs);
MANTA_X template __global__ void graph_big_kernel<8>(const LgArgs);
#undef MANTA_X
#endif

}  // namespace manta_dev
