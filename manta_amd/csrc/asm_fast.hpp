// assemble_fast_kernel: the iterative assembler's common case with the whole k-mer graph in LDS
// (assembly/IterativeAssembler.cpp:844-931, first word length of a locus whose graph is acyclic).
//
// Why a second kernel.  assemble_kernel (assemble_kernels.hpp) is the general path: any word length schedule, pseudo
// reads, cyclic graphs with the exact libstdc++-order repeat search, read sets of up to 1000 reads -- on a per-wave HBM
// slab.  Profiles of rounds 1-2 showed what that costs for the ordinary locus (config 2: 80 reads x 150 bp, ~1.3 k
// distinct words): 140x its algorithmic bytes in scattered 64-byte sectors, ~250 k wave instructions per locus, and a
// register allocation (all phases of all paths in one function) that spills.  This kernel keeps only the ordinary case:
//
//   * one single-wave workgroup per locus, FA_BUDGET bytes of LDS (3 workgroups per CU); pile, table, node records
//     and read-support sets live in LDS from the first byte to the selected contigs, HBM sees the reads once and the
//     results once (plus a few KB of walk bookkeeping per locus that stays in L2);
//   * anything it does not cover -- a cycle, a repeat hit that asks for the next word length, > 108 reads, a graph that
//     does not fit, bytes outside {A,C,G,T,N} -- is NOT handled here: the locus id goes onto a punt list and the general
//     kernel, launched right behind this one, picks the list up from device memory.  So this kernel's registers and
//     code are those of the fast path alone (no spills), and nothing is approximated;
//   * the contig loop (:685-713) runs on SPECULATION: together with the first seed's walk (one lane), 63 further lanes walk
//     the words most likely to be the next seeds (lowest count tier in exact seed order: the error branches a main-path
//     walk never takes).  Walks read only immutable graph data (walk_lanes.hpp), so a walk result is valid whenever its
//     seed turns out to be the reference's next seed; results are cached by node and the reference's seed sequence is
//     replayed over them.  The ordinary locus needs ONE round of ~330 dependent steps instead of three;
//   * in an acyclic graph a walk cannot meet one of its own words again except through a self loop, so the per-walk
//     visited bitmaps shrink to "chosen word == current word"; which walk touched which word is kept as one 64-bit lane
//     mask per node (no-return atomics into L2), which makes the replay a handful of register operations.
#pragma once
#include "assemble_kernels.hpp"

namespace manta_dev {

#ifndef MANTA_FAST_BUDGET
#define MANTA_FAST_BUDGET 53248
#endif
static const unsigned FA_BUDGET     = MANTA_FAST_BUDGET;  // bytes of LDS per locus (3 x 52 KB <= 160 KB per CU)
static const unsigned FA_SLOTS      = 2048;
static const unsigned FA_BUCKETS    = FA_SLOTS / 4;
static const unsigned FA_MAX_NODES  = 1843;               // 0.9 x slots; node ids are stored +1 in 11-bit link fields
static const unsigned FA_MAX_READS  = 128;                // read sets of two qwords
static const unsigned FA_MAX_PILE   = 2046;               // code dwords: a packed base index must fit 15 bits
static const unsigned FA_EMPTY      = 0xffffffffu;
static const unsigned FA_ID_PENDING = 0x7ffu;             // slot claimed, node id not assigned yet
static const unsigned FA_FAT        = 0x800u;             // support reference (12 bits): index into the bitset pool (else: a read)
static const unsigned FA_NO_SLOT    = 0xffffu;
#ifndef MANTA_FAST_TEAM
#define MANTA_FAST_TEAM 4
#endif
static const unsigned FA_TEAM       = MANTA_FAST_TEAM;    // wavefronts that share one locus (the launch may use fewer)

// fixed part of the LDS map (bytes)
static const unsigned FA_OFF_SLOTS  = 0;
static const unsigned FA_OFF_UNUSED = FA_OFF_SLOTS + 4 * FA_SLOTS;   // "unusedWords" bitmap, 64 dwords
static const unsigned FA_OFF_REPEAT = FA_OFF_UNUSED + 256;           // repeatWords bitmap (self loops), 64 dwords
static const unsigned FA_OFF_TENT   = FA_OFF_REPEAT + 256;           // u16[128]: seed list of the round, exact order
static const unsigned FA_OFF_SLOTND = FA_OFF_TENT + 256;             // u16[64]: word walked by cache slot s
static const unsigned FA_OFF_RD     = FA_OFF_SLOTND + 128;           // read descriptors {code dword offset : 11, length : 16, has N : 1}
static const unsigned FA_OFF_RDM    = FA_OFF_RD + 4 * FA_MAX_READS;  // u16[128]: N-bitmap dword offset of a read
static const unsigned FA_OFF_HDR    = FA_OFF_RDM + 2 * FA_MAX_READS; // u32[16]: what the waves of the workgroup share (FA_H_*)
static const unsigned FA_OFF_DYN    = FA_OFF_HDR + 64;              // codes, N bitmap, nodes ... pool
// header words
enum {
  FA_H_ALLOC = 0,  ///< node records in use | bitsets in use << 16 | "does not fit" << 31: one word, so that a reservation sees both ends
  FA_H_FLAG  = 1,  ///< pack: a byte outside the alphabet was seen
  FA_H_SLOT  = 2,  ///< kernel loop: the queue slot of the workgroup's current locus
  FA_H_CMD   = 3,  ///< what the first wave asks the team to do next (FA_CMD_*), with two arguments
  FA_H_ARG0  = 4,
  FA_H_ARG1  = 5,
  FA_H_ARG2  = 6,
  FA_H_ACC0  = 7,  ///< accumulators of the team operations (counts, maxima)
  FA_H_ACC1  = 8,
  FA_H_BEST  = 12  ///< [4] selectSeed: the best word of each wave of the team
};
// team operations: after the graph is built the first wave runs the contig loop; the other waves of the team wait for these
enum { FA_CMD_EXIT = 0, FA_CMD_SEED = 1, FA_CMD_TENT = 2 };
static const uint32_t FA_ALLOC_FAIL = 0x80000000u;

struct alignas(16) FRec {
  uint64_t w0;  ///< successor links 4 x 11 (id+1; packed from field 0 up in A,C,G,T order, 0 ends the list) | count << 44 (8 bit) | first occurrence, low 12 bits << 52
  uint64_t w1;  ///< predecessor links 4 x 11 (same; by symbol position until compactPreds) | support reference << 44 (12 bit) | first base << 56 | last base << 58 | first occurrence, high 3 bits << 60
};
struct alignas(16) FSet {
  uint64_t w[2];
};
struct alignas(16) FBucket {
  uint32_t s[4];
};

enum { FA_DONE = 0, FA_PUNT = 1 };

#ifdef MANTA_WAVE_EMU
/// test-build statistics of the speculation (tests/emu only): loci done, walk rounds, walks, accepted candidates, cache evictions
inline unsigned long long* fastStats()
{
  static unsigned long long v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  return v;
}
#define FA_STAT(i, n) do { if (wv::lane() == 0) fastStats()[i] += (n); } while (0)
#else
#define FA_STAT(i, n) do { } while (0)
#endif

struct FastAsm {
  Assembler&       A;  // HBM workspace views (walk results per cache slot), parameters
  const AsmParams& P;
  char*            lds;
  uint32_t *       slots, *unused_bits, *repeat_bits, *rd, *codes, *nmask, *hdr;
  unsigned         tw, tn;  // this wave's index in the workgroup's team and the team's size (waves that share the locus)
  uint16_t *       tent, *slotNode, *rdm;
  FRec*            nodes;
  unsigned         nNormal, W, k, nNodes, nodesOff, poolCount, nCand;
  unsigned         candSlotV;  // lane c: cache slot that holds candidate c's walk
  unsigned         chainBytes; // scratch bytes taken by the chain ids (0: none)
  uint64_t         tMark;

  WV_DEV void tick(const int phase)
  {
#ifdef MANTA_ASM_PROFILE
    const uint64_t now = wv::clock();
    if (P.phase_cycles && tw == 0 && wv::lane() == 0) wv::atomic_add(&P.phase_cycles[phase], (unsigned long long)(now - tMark));
    tMark = now;
#else
    (void)phase;
#endif
  }

  /// every wave of the team has written what the others read next
  WV_DEV void teamSync() const
  {
    wv::sync();
    if (tn > 1) wv::wg_barrier();
  }

  WV_DEV FastAsm(Assembler& a, char* ldsBase) : A(a), P(a.P), lds(ldsBase)
  {
    tw          = unsigned(wv::wave_in_wg());
    tn          = unsigned(wv::wg_waves());
    hdr         = reinterpret_cast<uint32_t*>(lds + FA_OFF_HDR);
    slots       = reinterpret_cast<uint32_t*>(lds + FA_OFF_SLOTS);
    unused_bits = reinterpret_cast<uint32_t*>(lds + FA_OFF_UNUSED);
    repeat_bits = reinterpret_cast<uint32_t*>(lds + FA_OFF_REPEAT);
    tent        = reinterpret_cast<uint16_t*>(lds + FA_OFF_TENT);
    slotNode    = reinterpret_cast<uint16_t*>(lds + FA_OFF_SLOTND);
    rd          = reinterpret_cast<uint32_t*>(lds + FA_OFF_RD);
    rdm         = reinterpret_cast<uint16_t*>(lds + FA_OFF_RDM);
    codes       = reinterpret_cast<uint32_t*>(lds + FA_OFF_DYN);
    nmask       = codes;
    nodes       = nullptr;
    candSlotV   = 0;
    chainBytes  = 0;
  }

  // ---- record fields ----
  WV_DEV static unsigned recCnt(const uint64_t w0) { return unsigned(w0 >> 44) & 0xffu; }
  WV_DEV static unsigned recPb(const uint64_t w0, const uint64_t w1) { return (unsigned(w0 >> 52) & 0xfffu) | ((unsigned(w1 >> 60) & 7u) << 12); }
  WV_DEV static unsigned recSupRef(const uint64_t w1) { return unsigned(w1 >> 44) & 0xfffu; }
  WV_DEV static unsigned recFirstBase(const uint64_t w1) { return unsigned(w1 >> 56) & 3u; }
  WV_DEV static unsigned recLastBase(const uint64_t w1) { return unsigned(w1 >> 58) & 3u; }
  WV_DEV static unsigned linkId(const uint64_t w, const unsigned c)
  {
    const unsigned f = unsigned(w >> (11 * c)) & 0x7ffu;
    return f ? f - 1 : ASM_NONE;
  }
  WV_DEV FSet* pool(const unsigned idx) const { return reinterpret_cast<FSet*>(lds + FA_BUDGET) - (idx + 1); }

  /// read support of a node as two set words
  WV_DEV void supOf(const uint64_t w1, uint64_t& s0, uint64_t& s1) const
  {
    const unsigned ref = recSupRef(w1);
    if (ref & FA_FAT) {
      const FSet v = *pool(ref & 0x7ffu);
      s0           = v.w[0];
      s1           = v.w[1];
    } else {
      s0 = (ref < 64) ? (uint64_t(1) << ref) : 0;
      s1 = (ref >= 64) ? (uint64_t(1) << (ref - 64)) : 0;
    }
  }

  // ---- packed pile (LDS) ----
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
    unsigned       b   = h & (FA_BUCKETS - 1);
    for (unsigned probe = 0; probe < FA_BUCKETS; ++probe) {
      const FBucket bk = *reinterpret_cast<const FBucket*>(slots + 4 * b);
      for (int i = 0; i < 4; ++i) {
        const uint32_t s = bk.s[i];
        if (s == FA_EMPTY) return ASM_NONE;
        if ((s >> 26) == tag && Assembler::keyEq(keyAt<KW>(s & 0x7fffu), key)) return (s >> 15) & 0x7ffu;
      }
      b = (b + 1) & (FA_BUCKETS - 1);
    }
    return ASM_NONE;
  }

  // ------------------------------------------------------------------------------------------------
  // stage 0: the locus' reads -> 2 bit + N bitmap in LDS, from any of the three input forms (1 byte per base, the same
  // arriving chunk by chunk behind the running kernel, packed piles).  False: the locus does not fit this path.
  // ------------------------------------------------------------------------------------------------
  WV_DEV bool pack(const unsigned locus)
  {
    const unsigned lane   = unsigned(wv::lane());
    const unsigned rBegin = P.locus_read_begin[locus], rEnd = P.locus_read_begin[locus + 1];
    nNormal               = rEnd - rBegin;
    if (nNormal + 2 * P.opt.maxAssemblyCount > FA_MAX_READS) return false;
    W = (nNormal + 2 * P.opt.maxAssemblyCount + 63) / 64;
    if (W == 0) W = 1;
    const uint64_t plR = A.plShift(locus, 0), plC = A.plShift(locus, 1), plM = A.plShift(locus, 2);
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
    if (wv::any(tooLong) || cw + 2 > FA_MAX_PILE) return false;
    nmask    = codes + ((cw + 2 + 3) & ~3u);
    nodesOff = FA_OFF_DYN + 4 * (((cw + 2 + 3) & ~3u) + ((mw + 2 + 3) & ~3u));
    if (nodesOff + 4096 > FA_BUDGET) return false;
    nodes = reinterpret_cast<FRec*>(lds + nodesOff);
    for (unsigned i = lane + 64 * tw; i < mw + 2; i += 64 * tn) nmask[i] = 0;
    if (tw == 0 && lane == 0) hdr[FA_H_FLAG] = 0;
    teamSync();
    // (the reads are dealt out to the team's waves eight at a time)
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
    if (wv::any(bad) && lane == 0) wv::atomic_or(&hdr[FA_H_FLAG], 1u);  // bytes outside {A,C,G,T,N}: the general path decides what is exact
    teamSync();
    return wv::atomic_load(&hdr[FA_H_FLAG]) == 0;
  }

  // ------------------------------------------------------------------------------------------------
  // k-mer graph (getKmerCounts :506-550 + successor / predecessor links)
  // ------------------------------------------------------------------------------------------------
  template <int KW>
  WV_DEV bool buildGraph()
  {
    const unsigned lane = unsigned(wv::lane());
    for (unsigned s = lane + 64 * tw; s < FA_SLOTS; s += 64 * tn) slots[s] = FA_EMPTY;
    if (tw == 0 && lane == 0) hdr[FA_H_ALLOC] = 0;
    teamSync();
    // The reads are dealt out to the team's waves (read r: wave r mod team size); slots, records and bitsets are shared.
    // Node records grow up from nodesOff, bitsets down from the end of the budget: both counts live in one header word, so
    // the atomic that reserves either sees where the other end stands (FA_ALLOC_FAIL set by whoever does not fit).
    const uint64_t below = (uint64_t(1) << lane) - 1;
    bool           fail  = false;
    for (unsigned rBase = 0; rBase < nNormal && !fail; rBase += 64) {
      // descriptors of up to 64 reads in lane registers; v_readlane hands them out per read
      const unsigned rMine = rBase + lane;
      const unsigned dV    = (rMine < nNormal) ? rd[rMine] : 0u;
      const unsigned mV    = (rMine < nNormal) ? unsigned(rdm[rMine]) : 0u;
      const unsigned rEnd  = (nNormal - rBase < 64) ? (nNormal - rBase) : 64u;
      for (unsigned ri = 0; ri < rEnd && !fail; ++ri) {
        const unsigned r = rBase + ri;
        if (r % tn != tw) continue;
        const unsigned d = wv::readlane(dV, int(ri)), cwo = d & 0x7ffu, len = (d >> 11) & 0xffffu;
        if (len < k) continue;  // :522
        const bool     rdHasN = (d >> 27) & 1u;
        const unsigned mwo    = wv::readlane(mV, int(ri));
        for (unsigned j0 = 0; j0 + k <= len; j0 += 64) {
          const unsigned j    = j0 + lane;
          const unsigned pb   = cwo * 16 + j;
          bool           have = false, won = false;
          unsigned       slot = 0, foundId = FA_ID_PENDING;
          uint32_t       mine = 0;
          unsigned       firstLast = 0;
          if (j + k <= len && !(rdHasN && windowHasN(mwo, j))) {  // :531
            const Key<KW>  key = keyAt<KW>(pb);
            const uint32_t h   = keyHash(key);
            const unsigned tag = h >> 26;
            unsigned       b   = h & (FA_BUCKETS - 1);
            mine               = pb | (FA_ID_PENDING << 15) | (tag << 26);
            {
              unsigned last = 0;
              for (int w = 0; w < KW; ++w)
                if (unsigned(w) == ((k - 1) >> 4)) last = (key.w[w] >> (30 - 2 * ((k - 1) & 15))) & 3u;
              firstLast = (key.w[0] >> 30) | (last << 2);
            }
            for (unsigned probe = 0; probe < FA_BUCKETS && !have; ++probe) {
              const FBucket bk = *reinterpret_cast<const FBucket*>(slots + 4 * b);
              for (int i = 0; i < 4 && !have; ++i) {
                uint32_t s = bk.s[i];
                if (s == FA_EMPTY) {
                  s = wv::atomic_cas(&slots[4 * b + i], FA_EMPTY, mine);
                  if (s == FA_EMPTY) {
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
              b = (b + 1) & (FA_BUCKETS - 1);
            }
            if (!have) fail = true;  // table full (cannot happen below FA_MAX_NODES)
          }
          // new words: records reserved for the whole step with one atomic; the record is written before the slot names it
          const uint64_t m = wv::ballot(won);
          if (m) {
            const unsigned c   = unsigned(wv::popc(m));
            unsigned       old = 0;
            if (lane == 0) old = wv::atomic_add(&hdr[FA_H_ALLOC], c);
            old               = wv::first(old);
            const unsigned n0 = old & 0x7fffu, p0 = (old >> 16) & 0x7fffu;
            if ((old & FA_ALLOC_FAIL) || nodesOff + 16 * (n0 + c) + 16 * p0 > FA_BUDGET || n0 + c > FA_MAX_NODES) {
              fail = true;
            } else if (won) {
              const unsigned id = n0 + unsigned(wv::popc(m & below));
              FRec           rec;
              rec.w0    = uint64_t(pb & 0xfffu) << 52;
              rec.w1    = (uint64_t(r) << 44) | (uint64_t(firstLast) << 56) | (uint64_t(pb >> 12) << 60);  // support = {r}
              nodes[id] = rec;
              wv::fence_wg();
              wv::atomic_store(&slots[slot], (mine & ~(0x7ffu << 15)) | (id << 15));
            }
            wv::sync();
          }
          if (wv::any(fail)) {
            fail = true;
            if (lane == 0) wv::atomic_or(&hdr[FA_H_ALLOC], FA_ALLOC_FAIL);
            break;
          }
          // add read r to the support of the words that existed already.  A slot may still carry the pending id (claimed by
          // another wave -- or a twin lane -- that has not named its record yet): wait for it.
          bool todo = have && !won;
          while (wv::any(todo && foundId == FA_ID_PENDING)) {
            if (todo && foundId == FA_ID_PENDING) foundId = (wv::atomic_load(&slots[slot]) >> 15) & 0x7ffu;
            if (wv::atomic_load(&hdr[FA_H_ALLOC]) & FA_ALLOC_FAIL) {
              fail = true;
              break;
            }
            if (wv::any(todo && foundId == FA_ID_PENDING)) wv::spin();
          }
          if (wv::any(fail)) {
            fail = true;
            break;
          }
          wv::fence_wg();
          // A word starts with its first read's index in the record; the second read gives it a bitset.  Several lanes (of
          // several waves) may find the same word at that point: each brings a bitset, one compare-and-swap wins, the others
          // find the winner's bitset on their next turn and add their read there (their own bitset stays unused).
          unsigned myF = ASM_NONE;
          while (wv::any(todo)) {
            bool      needFat = false;
            uint32_t  curHi   = 0;
            uint32_t* hi      = nullptr;
            if (todo) {
              hi                 = reinterpret_cast<uint32_t*>(&nodes[foundId].w1) + 1;
              curHi              = wv::atomic_load(hi);
              const unsigned ref = (curHi >> 12) & 0xfffu;
              if (ref & FA_FAT) {
                unsigned long long* w = reinterpret_cast<unsigned long long*>(&pool(ref & 0x7ffu)->w[r >> 6]);
                wv::atomic_or(w, (unsigned long long)(uint64_t(1) << (r & 63)));
                todo = false;
              } else if (ref == r) {
                todo = false;
              } else {
                needFat = true;
              }
            }
            const uint64_t mf = wv::ballot(needFat && myF == ASM_NONE);
            if (mf) {
              const unsigned c   = unsigned(wv::popc(mf));
              unsigned       old = 0;
              if (lane == 0) old = wv::atomic_add(&hdr[FA_H_ALLOC], c << 16);
              old               = wv::first(old);
              const unsigned n0 = old & 0x7fffu, p0 = (old >> 16) & 0x7fffu;
              if ((old & FA_ALLOC_FAIL) || nodesOff + 16 * n0 + 16 * (p0 + c) > FA_BUDGET || p0 + c > 0x7ffu) {
                fail = true;
                if (lane == 0) wv::atomic_or(&hdr[FA_H_ALLOC], FA_ALLOC_FAIL);
                break;
              }
              if (needFat && myF == ASM_NONE) myF = p0 + unsigned(wv::popc(mf & below));
            }
            if (needFat) {
              const unsigned ref = (curHi >> 12) & 0xfffu;
              FSet           v;
              v.w[0] = ((ref < 64) ? (uint64_t(1) << ref) : 0) | ((r < 64) ? (uint64_t(1) << r) : 0);
              v.w[1] = ((ref >= 64) ? (uint64_t(1) << (ref - 64)) : 0) | ((r >= 64) ? (uint64_t(1) << (r - 64)) : 0);
              *pool(myF) = v;
              wv::fence_wg();
              const uint32_t want = (curHi & ~(0xfffu << 12)) | ((FA_FAT | myF) << 12);
              if (wv::atomic_cas(hi, curHi, want) == curHi) todo = false;
            }
            wv::sync();
          }
          if (fail) break;
        }
      }
    }
    teamSync();
    {
      const uint32_t al = wv::atomic_load(&hdr[FA_H_ALLOC]);
      if (al & FA_ALLOC_FAIL) return false;
      nNodes    = al & 0x7fffu;
      poolCount = (al >> 16) & 0x7fffu;
    }
    tick(1);

    // counts, successor lookups, predecessor scatter
    for (unsigned nb = 64 * tw; nb < nNodes; nb += 64 * tn) {
      const unsigned nd = nb + lane;
      if (nd < nNodes) {
        const FRec     rec = nodes[nd];
        uint64_t       s0, s1;
        supOf(rec.w1, s0, s1);
        const unsigned cnt = unsigned(wv::popc(s0)) + unsigned(wv::popc(s1));  // (no pseudo reads at the first word length)
        const Key<KW>  key = keyAt<KW>(recPb(rec.w0, rec.w1));
        uint64_t       w0  = (rec.w0 & (uint64_t(0xfff) << 52)) | (uint64_t(cnt > 255 ? 255 : cnt) << 44);
        const unsigned firstBase = key.w[0] >> 30;
        bool           selfLoop  = false;
        unsigned       m         = 0;
        for (unsigned c = 0; c < 4; ++c) {
          const unsigned s = lookup<KW>(A.template keyShiftAppend<KW>(key, c));
          if (s == ASM_NONE) continue;
          w0 |= uint64_t(s + 1) << (11 * m);
          m++;
          if (s == nd) selfLoop = true;
          wv::atomic_or(reinterpret_cast<unsigned long long*>(&nodes[s].w1), (unsigned long long)(uint64_t(nd + 1) << (11 * firstBase)));
        }
        nodes[nd].w0 = w0;
        if (selfLoop) wv::atomic_or(&repeat_bits[nd >> 5], 1u << (nd & 31));
      }
    }
    teamSync();
    // predecessor lists packed like the successor lists (the scatter above addressed them by symbol)
    for (unsigned nd = lane + 64 * tw; nd < nNodes; nd += 64 * tn) {
      const uint64_t w1 = nodes[nd].w1;
      uint64_t       pk = 0;
      unsigned       m  = 0;
      for (unsigned c = 0; c < 4; ++c) {
        const uint64_t f = (w1 >> (11 * c)) & 0x7ffu;
        if (f == 0) continue;
        pk |= f << (11 * m);
        m++;
      }
      nodes[nd].w1 = (w1 & ~((uint64_t(1) << 44) - 1)) | pk;
    }
    // seed eligibility (:679-682; the counts are final since the barrier above)
    for (unsigned nb = 64 * tw; nb < ((nNodes + 63) & ~63u); nb += 64 * tn) {
      const unsigned nd = nb + lane;
      const uint64_t m  = wv::ballot(nd < nNodes && recCnt(nodes[nd < nNodes ? nd : 0].w0) >= P.opt.minCoverage);
      if (lane < 2) unused_bits[(nb >> 5) + lane] = uint32_t(m >> (32 * lane));
    }
    teamSync();
    tick(2);
    return true;
  }

  WV_DEV bool isUnused(const unsigned nd) const { return (unused_bits[nd >> 5] >> (nd & 31)) & 1u; }
  WV_DEV bool isRepeat(const unsigned nd) const { return (repeat_bits[nd >> 5] >> (nd & 31)) & 1u; }

  WV_DEV char*    scratch() const { return lds + nodesOff + 16 * nNodes; }
  WV_DEV unsigned scratchBytes() const { return FA_BUDGET - 16u * poolCount - (nodesOff + 16 * nNodes); }

  // ------------------------------------------------------------------------------------------------
  // cycle test (see Assembler::graphHasCycle): two-sided Kahn peel, per-node state byte {in:3, out:3, peeled, simple},
  // one append-only queue.  Returns 0 acyclic, 1 cyclic, 2 scratch too small.
  // ------------------------------------------------------------------------------------------------
  WV_DEV int graphHasCycle()
  {
    const unsigned lane = unsigned(wv::lane());
    const unsigned stDw = (nNodes + 3) / 4;
    const unsigned need = 4 * stDw + 2 * nNodes + 32;
    if (need > scratchBytes()) return 2;
    uint32_t* st    = reinterpret_cast<uint32_t*>(scratch());
    uint32_t* qTail = st + stDw;
    uint16_t* queue = reinterpret_cast<uint16_t*>(qTail + 4);
    if (lane == 0) *qTail = 0;
    for (unsigned w = lane; w < stDw; w += 64) st[w] = 0;
    wv::sync();
    auto degrees = [&](const FRec& rec, const unsigned nd, unsigned& id, unsigned& od, unsigned& only) {
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
      const FRec rec = nodes[nd];
      unsigned   id, od, only;
      degrees(rec, nd, id, od, only);
      const bool src    = (id == 0 || od == 0);
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
        const FRec     rec = nodes[nd];
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
    if (removed != nNodes) return 1;
    // Chain ids for the speculation (contigRounds): words joined by simple edges nd -> nd+1 form one unbranched stretch
    // (an error branch is one: its words were created in order by the read that holds the error); chain[nd] = id of the
    // stretch's first word.  Built from the `simple` bits of the peel state, then written over it (the state is dead).
    chainBytes = 0;
    {
      const unsigned nChunks = (nNodes + 63) / 64;  // <= 29
      const unsigned cb      = (2 * nNodes + 15) & ~15u;
      if (nChunks <= 64 && cb + 1024 + 2 * TENT_CAP + 64 <= scratchBytes()) {
        uint64_t keep = 0;  // lane c: simple bits of chunk c
        for (unsigned c = 0; c < nChunks; ++c) {
          const unsigned nd = 64 * c + lane;
          const uint64_t m  = wv::ballot(nd < nNodes && (stateOf(nd < nNodes ? nd : 0) & 0x80u));
          if (lane == c) keep = m;
        }
        wv::sync();
        uint16_t* chain     = reinterpret_cast<uint16_t*>(scratch());
        unsigned  carryId   = 0;      // chain of the last word of the previous chunk
        unsigned  carryEdge = 0;      // 1: that word has a simple edge into this chunk's first word
        for (unsigned c = 0; c < nChunks; ++c) {
          const uint64_t m      = wv::readlane(keep, int(c));
          const uint64_t starts = ~((m << 1) | uint64_t(carryEdge));  // bit l: word 64c+l starts a chain
          const uint64_t below  = starts & ((lane == 63) ? ~uint64_t(0) : ((uint64_t(2) << lane) - 1));
          const unsigned id     = below ? (64 * c + 63u - unsigned(wv::clz(below))) : carryId;
          const unsigned nd     = 64 * c + lane;
          if (nd < nNodes) chain[nd] = uint16_t(id);
          carryId   = wv::readlane(id, 63);
          carryEdge = unsigned(m >> 63) & 1u;
        }
        chainBytes = cb;
        wv::sync();
      }
    }
    return 0;
  }

  // ------------------------------------------------------------------------------------------------
  // seed order (:686-696): count descending, k-mer ascending
  // ------------------------------------------------------------------------------------------------
  /// (a team operation: every wave of the team runs it; see command())
  template <int KW>
  WV_DEV unsigned selectSeed()
  {
    const unsigned lane  = unsigned(wv::lane());
    uint32_t*      bestW = hdr + FA_H_BEST;
    if (tw == 0 && lane == 0) hdr[FA_H_ACC0] = 0;
    teamSync();
    unsigned best = 0;
    for (unsigned nd = lane + 64 * tw; nd < nNodes; nd += 64 * tn)
      if (isUnused(nd)) {
        const unsigned c = recCnt(nodes[nd].w0);
        best             = (c > best) ? c : best;
      }
    best = Assembler::waveMax(best);
    if (tn > 1) {
      if (lane == 0) wv::atomic_max(&hdr[FA_H_ACC0], best);
      teamSync();
      best = wv::atomic_load(&hdr[FA_H_ACC0]);
    }
    if (best == 0) return ASM_NONE;
    unsigned mine = ASM_NONE;
    Key<KW>  mineKey;
    for (int i = 0; i < KW; ++i) mineKey.w[i] = 0xffffffffu;
    for (unsigned nd = lane + 64 * tw; nd < nNodes; nd += 64 * tn) {
      const FRec rec = nodes[nd];
      if (isUnused(nd) && recCnt(rec.w0) == best) {
        const Key<KW> key = keyAt<KW>(recPb(rec.w0, rec.w1));
        if (mine == ASM_NONE || Assembler::keyLess(key, mineKey)) {
          mine    = nd;
          mineKey = key;
        }
      }
    }
    for (int off = 1; off < 64; off <<= 1) {
      const int      src = wv::lane() ^ off;
      const unsigned on  = wv::shfl(mine, src);
      Key<KW>        ok;
      for (int i = 0; i < KW; ++i) ok.w[i] = wv::shfl(mineKey.w[i], src);
      if (on != ASM_NONE && (mine == ASM_NONE || Assembler::keyLess(ok, mineKey))) {
        mine    = on;
        mineKey = ok;
      }
    }
    if (tn > 1) {  // the best of each wave, then the best of those
      if (lane == 0) bestW[tw] = mine;
      teamSync();
      mine = ASM_NONE;
      for (unsigned v = 0; v < tn; ++v) {
        const unsigned on = wv::atomic_load(&bestW[v]);
        if (on == ASM_NONE) continue;
        const FRec    rec = nodes[on];
        const Key<KW> ok  = keyAt<KW>(recPb(rec.w0, rec.w1));
        if (mine == ASM_NONE || Assembler::keyLess(ok, mineKey)) {
          mine    = on;
          mineKey = ok;
        }
      }
    }
    return mine;
  }

  /// next <= T unused words WITH COUNT <= maxCount in exact seed order into tent[0..nT) (u16 node ids, LDS); see
  /// Assembler::selectTentative.  `area`/`areaBytes`: scratch for the histogram (1 KB) and the raw candidate list.
  /// maxCount = 255 is the reference's order over all unused words; a smaller bound orders one tier of them (speculation).
  /// (a team operation like selectSeed: the passes over the words and the ranking are dealt out to the team's waves)
  template <int KW>
  WV_DEV unsigned selectTentative(const unsigned T, const unsigned maxCount, char* area, const unsigned areaBytes)
  {
    const unsigned lane = unsigned(wv::lane());
    if (areaBytes < 1024 + 2 * TENT_CAP) {
      if (maxCount < 255) return 0;  // (speculation only: nothing is lost)
      const unsigned s = selectSeed<KW>();
      if (s == ASM_NONE) return 0;
      if (tw == 0 && lane == 0) tent[0] = uint16_t(s);
      teamSync();
      return 1;
    }
    uint32_t* hist = reinterpret_cast<uint32_t*>(area);
    uint16_t* raw  = reinterpret_cast<uint16_t*>(area + 1024);
    // count level: counts are <= 255 here (no pseudo reads) -> one 256-bin histogram over the unused words
    for (unsigned i = lane + 64 * tw; i < 256; i += 64 * tn) hist[i] = 0;
    if (tw == 0 && lane == 0) {
      hdr[FA_H_ACC0] = 0;
      hdr[FA_H_ACC1] = 0;
    }
    teamSync();
    unsigned U = 0;
    for (unsigned nb = 64 * tw; nb < nNodes; nb += 64 * tn) {
      const unsigned nd = nb + lane;
      if (nd < nNodes && isUnused(nd)) {
        const unsigned c = recCnt(nodes[nd].w0);
        if (c <= maxCount) {
          wv::atomic_add(&hist[c], 1u);
          U++;
        }
      }
    }
    U = Assembler::waveSum(U);
    if (tn > 1) {
      if (lane == 0 && U) wv::atomic_add(&hdr[FA_H_ACC0], U);
      teamSync();
      U = wv::atomic_load(&hdr[FA_H_ACC0]);
    } else {
      wv::sync();
    }
    if (U == 0) return 0;
    unsigned cStar = 1, pStar = 0xffffffffu;
    if (U > T) {
      // (every wave reads the same histogram and arrives at the same numbers)
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
      // tie level: radix select on the 16-base prefix among the words with count == cStar
      unsigned prefix = 0;
      for (int shift = 24; shift >= 0; shift -= 8) {
        teamSync();  // (everybody is done reading the histogram)
        for (unsigned i = lane + 64 * tw; i < 256; i += 64 * tn) hist[i] = 0;
        teamSync();
        for (unsigned nb = 64 * tw; nb < nNodes; nb += 64 * tn) {
          const unsigned nd = nb + lane;
          if (nd >= nNodes || !isUnused(nd)) continue;
          const FRec rec = nodes[nd];
          if (recCnt(rec.w0) != cStar) continue;
          const unsigned p = codes16(recPb(rec.w0, rec.w1));
          if (shift < 24 && (p >> (shift + 8)) != (prefix >> (shift + 8))) continue;
          wv::atomic_add(&hist[(p >> shift) & 255u], 1u);
        }
        teamSync();
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
      }
      pStar = prefix;
    }
    // gather the survivors (in any order: the ranking below is what orders them)
    for (unsigned nb = 64 * tw; nb < ((nNodes + 63) & ~63u); nb += 64 * tn) {
      const unsigned nd  = nb + lane;
      bool           sel = false;
      if (nd < nNodes && isUnused(nd)) {
        const FRec     rec = nodes[nd];
        const unsigned c   = recCnt(rec.w0);
        sel                = (c <= maxCount) && ((U <= T) || (c > cStar) || (c == cStar && codes16(recPb(rec.w0, rec.w1)) <= pStar));
      }
      const uint64_t m = wv::ballot(sel);
      if (m == 0) continue;
      unsigned at = 0;
      if (lane == 0) at = wv::atomic_add(&hdr[FA_H_ACC1], unsigned(wv::popc(m)));
      at                 = wv::first(at);
      const unsigned pos = at + unsigned(wv::popc(m & ((uint64_t(1) << lane) - 1)));
      if (sel && pos < TENT_CAP) raw[pos] = uint16_t(nd);
    }
    teamSync();
    const unsigned total = wv::atomic_load(&hdr[FA_H_ACC1]);
    if (total > TENT_CAP) {  // pathological tie group (hundreds of words sharing a 16-base prefix)
      if (maxCount < 255) return 0;
      const unsigned s = selectSeed<KW>();
      if (s == ASM_NONE) return 0;
      if (tw == 0 && lane == 0) tent[0] = uint16_t(s);
      teamSync();
      return 1;
    }
    // exact rank inside the list
    const unsigned keep = (total < T) ? total : T;
    for (unsigned i = lane + 64 * tw; i < total; i += 64 * tn) {
      const unsigned x  = raw[i];
      const FRec     rx = nodes[x];
      const unsigned cx = recCnt(rx.w0), pbx = recPb(rx.w0, rx.w1), px = codes16(pbx);
      unsigned       rank = 0;
      for (unsigned j = 0; j < total; ++j) {
        if (j == i) continue;
        const unsigned y  = raw[j];
        const FRec     ry = nodes[y];
        const unsigned cy = recCnt(ry.w0);
        bool           before = (cy > cx);
        if (cy == cx) {
          const unsigned pby = recPb(ry.w0, ry.w1), py = codes16(pby);
          before             = (py < px) || (py == px && Assembler::keyLess(keyAt<KW>(pby), keyAt<KW>(pbx)));
        }
        if (before) rank++;
      }
      if (rank < keep) tent[rank] = uint16_t(x);
    }
    teamSync();
    return keep;
  }

  // ------------------------------------------------------------------------------------------------
  // walks (:149-501), one lane per cache slot; see walk_lanes.hpp for the scheme and the reference lines
  // ------------------------------------------------------------------------------------------------
  WV_DEV unsigned long long* visMask() const { return reinterpret_cast<unsigned long long*>(A.lane_vis); }

  struct Cand {
    uint64_t w0, w1, s0, s1;
  };

  /// Fetching the word behind a link field `f` (id + 1; 0 = no word) takes two dependent LDS reads: its record, then --
  /// if the word has one -- its bitset.  The walk issues the first reads of everything a step needs together, then the
  /// second reads, then combines with mask arithmetic (no selects on loaded values: the compiler would turn those into
  /// branches around the loads and serialise the round trips).  A word without a bitset reads pool entry 0 and masks it out.
  WV_DEV FRec candRec(const unsigned f) const { return nodes[f ? f - 1 : 0]; }
  WV_DEV FSet candPool(const uint64_t w1) const
  {
    const unsigned ref = recSupRef(w1);
    return *pool((ref & FA_FAT) ? (ref & 0x7ffu) : 0u);
  }
  WV_DEV static void candSup(const unsigned f, const uint64_t w1, const FSet& p, uint64_t& s0, uint64_t& s1)
  {
    const unsigned ref  = recSupRef(w1);
    const bool     fat  = (ref & FA_FAT) != 0;
    const uint64_t useM = (f != 0 && fat) ? ~uint64_t(0) : 0;
    const uint64_t bit  = uint64_t((f != 0 && !fat) ? 1u : 0u) << (ref & 63);
    const uint64_t hiM  = (ref & 64u) ? ~uint64_t(0) : 0;
    s0                  = (p.w[0] & useM) | (bit & ~hiM);
    s1                  = (p.w[1] & useM) | (bit & hiM);
  }
  WV_DEV Cand loadCand(const unsigned f) const
  {
    Cand       c;
    const FRec r = candRec(f);
    const FSet p = candPool(r.w1);
    c.w0         = r.w0;
    c.w1         = r.w1;
    candSup(f, r.w1, p, c.s0, c.s1);
    return c;
  }
  WV_DEV void loadSup(const unsigned f, uint64_t& s0, uint64_t& s1) const
  {
    const uint64_t w1 = nodes[f ? f - 1 : 0].w1;
    const FSet     p  = candPool(w1);
    candSup(f, w1, p, s0, s1);
  }

  /// the lanes of walkMask walk slotNode[lane]; results go to the slot's HBM records (lane_bits / lane_meta / lane_seq)
  /// and bit `lane` of visMask[word] for every word of the walk.
  ///
  /// One lane executes the instruction stream of all 64, so the step is written for the union: the first two candidates
  /// of a step (packed link lists: fields 0 and 1) are always fetched and compared branch-free, a third or fourth one
  /// (three-way branches are rare) sits behind a wave vote; likewise the backward check (:377-427) fetches one "other"
  /// neighbour of the chosen word branch-free and further ones behind a vote.  Appended bases come from the chosen
  /// word's record (first / last base), not from the link position.
  template <int KW>
  WV_DEV void walkSlots(const uint64_t walkMask)
  {
    const unsigned lane = unsigned(wv::lane());
    const bool     has  = (walkMask >> lane) & 1u;
    FA_STAT(1, 1);
    FA_STAT(2, unsigned(wv::popc(walkMask)));
    const unsigned seed = has ? unsigned(slotNode[lane]) : 0u;
    unsigned long long* vm = visMask();
    const unsigned long long laneBit = (unsigned long long)1 << lane;
    const unsigned seqWords = P.max_contig_len / 16 + 2;
    uint32_t*      rightBuf = reinterpret_cast<uint32_t*>(A.lane_seq) + size_t(lane) * 2 * seqWords;
    uint32_t*      leftBuf  = rightBuf + seqWords;
    uint32_t       accR = 0, accL = 0;
    uint64_t       S0 = 0, S1 = 0, R0 = 0, R1 = 0;
    bool           active = has, rep = false, tooLong = false;
    unsigned       mode = 0, cur = seed, consOffset = 0, nLeft = 0, nRight = 0;
    int            consEnd = 0, consBegin = 0;
    FRec           seedRec = {0, 0};
    if (has) {
      seedRec = nodes[seed];
      supOf(seedRec.w1, S0, S1);
      wv::atomic_or(&vm[seed], laneBit);
      if (isRepeat(seed)) {  // :172-179
        rep    = true;
        active = false;
      } else {  // unselected siblings of the seed reject the contig (:185-210)
        const unsigned seedPb   = recPb(seedRec.w0, seedRec.w1);
        const Key<KW>  key      = keyAt<KW>(seedPb);
        const unsigned lastBase = recLastBase(seedRec.w1);
        for (unsigned c = 0; c < 4; ++c) {
          if (c == lastBase) continue;
          Key<KW> sib = key;
          A.keySetBase(sib, k - 1, c);
          const unsigned n = lookup<KW>(sib);
          if (n != ASM_NONE) {
            uint64_t a, b;
            supOf(nodes[n].w1, a, b);
            R0 |= a;
            R1 |= b;
          }
        }
      }
    }
    const uint64_t M44  = (uint64_t(1) << 44) - 1;
    uint64_t       link = active ? (seedRec.w0 & M44) : 0;  // candidate list of the current word in walking direction
    Cand           ca = loadCand(unsigned(link) & 0x7ffu), cb = loadCand(unsigned(link >> 11) & 0x7ffu);

    while (wv::any(active)) {
      const bool isEnd = (mode == 0);
      // ---- choose the extension (:241-336): candidates a, b in alphabet order, strict '>' on the shared-read count ----
      const uint64_t A0 = S0 & ca.s0, A1 = S1 & ca.s1, B0 = S0 & cb.s0, B1 = S1 & cb.s1;
      const unsigned cntA = unsigned(wv::popc(A0)) + unsigned(wv::popc(A1)), cntB = unsigned(wv::popc(B0)) + unsigned(wv::popc(B1));
      const bool     bWins = cntB > cntA;
      const uint64_t SH0 = A0 & cb.s0, SH1 = A1 & cb.s1;
      // the loser's shared reads leave the contig, its reads reject it (an ignored candidate -- count 0 -- loses nothing)
      const bool     loserOn = bWins ? (cntA != 0) : (cntB != 0);
      uint64_t       rm0  = (bWins ? A0 : B0) & ~SH0, rm1 = (bWins ? A1 : B1) & ~SH1;
      uint64_t       add0 = loserOn ? ((bWins ? ca.s0 : cb.s0) & ~SH0) : 0, add1 = loserOn ? ((bWins ? ca.s1 : cb.s1) & ~SH1) : 0;
      uint64_t       maxWR0 = bWins ? cb.s0 : ca.s0, maxWR1 = bWins ? cb.s1 : ca.s1;
      uint64_t       maxCW0 = bWins ? B0 : A0, maxCW1 = bWins ? B1 : A1;
      uint64_t       maxW0 = bWins ? cb.w0 : ca.w0, maxW1 = bWins ? cb.w1 : ca.w1;
      unsigned       maxCnt = bWins ? cntB : cntA;
      unsigned       maxF   = bWins ? (unsigned(link >> 11) & 0x7ffu) : (unsigned(link) & 0x7ffu);  // id + 1 of the chosen word
      if (maxCnt == 0) {
        maxWR0 = maxWR1 = maxCW0 = maxCW1 = 0;
        maxF = 0;
      }
      if (wv::any(active && ((link >> 22) & 0x7ffu) != 0)) {  // a third / fourth candidate somewhere in the wave (rare)
        for (unsigned i = 2; i < 4; ++i) {
          const unsigned f = active ? (unsigned(link >> (11 * i)) & 0x7ffu) : 0u;
          if (!wv::any(f != 0)) continue;
          const Cand c = loadCand(f);
          const uint64_t C0 = S0 & c.s0, C1 = S1 & c.s1;
          const unsigned cnt = unsigned(wv::popc(C0)) + unsigned(wv::popc(C1));
          if (cnt == 0) continue;  // :280
          const uint64_t T0 = maxCW0 & c.s0, T1 = maxCW1 & c.s1;
          if (cnt > maxCnt) {  // :283-316
            rm0 |= maxCW0 & ~T0;
            rm1 |= maxCW1 & ~T1;
            add0 |= maxWR0 & ~T0;
            add1 |= maxWR1 & ~T1;
            maxWR0 = c.s0;
            maxWR1 = c.s1;
            maxCW0 = C0;
            maxCW1 = C1;
            maxCnt = cnt;
            maxF   = f;
            maxW0  = c.w0;
            maxW1  = c.w1;
          } else {  // :317-335
            rm0 |= C0 & ~T0;
            rm1 |= C1 & ~T1;
            add0 |= c.s0 & ~T0;
            add1 |= c.s1 & ~T1;
          }
        }
      }
      const unsigned maxNode      = maxF - 1;  // (ASM_NONE when nothing was chosen)
      const unsigned maxBaseCount = maxF ? recCnt(maxW0) : 0u;
      bool           stop = false, extend = false;
      if (active) {
        if (maxBaseCount < P.opt.minCoverage) {  // :343 (also "no candidate")
          stop = true;
        } else if (maxNode == cur) {  // :352-358: in an acyclic graph a walk meets its own words again only through a self loop
          rep  = true;
          stop = true;
        } else if (k + nRight + nLeft + 1 >= P.max_contig_len) {
          tooLong = true;
          active  = false;
        } else {
          extend = true;
        }
      }
      // ---- requests: the neighbours of the chosen word against the walking direction (:377-427) ... ----
      const uint64_t back = extend ? ((isEnd ? maxW1 : maxW0) & M44) : 0;
      unsigned       o0 = 0, nOther = 0;
      for (unsigned c = 0; c < 4; ++c) {
        const unsigned f  = unsigned(back >> (11 * c)) & 0x7ffu;
        const bool     ok = f != 0 && f != cur + 1 && f != maxF;  // :381, :389
        if (ok && nOther == 0) o0 = f;
        nOther += ok ? 1u : 0u;
      }
      // ---- ... and the next step's candidates (on a direction switch: the seed's predecessors) ----
      const bool     toLeft = stop && (mode == 0);  // :488-491
      const uint64_t next   = extend ? ((isEnd ? maxW0 : maxW1) & M44) : (toLeft ? (seedRec.w1 & M44) : 0);
      const unsigned fa = unsigned(next) & 0x7ffu, fb = unsigned(next >> 11) & 0x7ffu;
      const uint64_t ow1 = nodes[o0 ? o0 - 1 : 0].w1;  // first reads ...
      const FRec     ra = candRec(fa), rb = candRec(fb);
      const FSet     po = candPool(ow1), pa = candPool(ra.w1), pb = candPool(rb.w1);  // ... second reads
      uint64_t       b0, b1;
      Cand           na, nb;
      candSup(o0, ow1, po, b0, b1);
      na.w0 = ra.w0;
      na.w1 = ra.w1;
      nb.w0 = rb.w0;
      nb.w1 = rb.w1;
      candSup(fa, ra.w1, pa, na.s0, na.s1);
      candSup(fb, rb.w1, pb, nb.s0, nb.s1);
      b0 &= ~maxCW0;  // :400-414
      b1 &= ~maxCW1;
      if (wv::any(nOther > 1)) {  // more than one other neighbour (rare)
        unsigned seen = 0;
        for (unsigned c = 0; c < 4; ++c) {
          const unsigned f  = unsigned(back >> (11 * c)) & 0x7ffu;
          const bool     ok = f != 0 && f != cur + 1 && f != maxF;
          const bool     want = ok && seen >= 1;
          seen += ok ? 1u : 0u;
          if (!wv::any(want)) continue;
          uint64_t x0, x1;
          loadSup(want ? f : 0u, x0, x1);
          b0 |= x0 & ~maxCW0;
          b1 |= x1 & ~maxCW1;
        }
      }
      // ---- finish this step ----
      if (extend) {
        wv::atomic_or(&vm[maxNode], laneBit);  // :482-484
        const unsigned sym = isEnd ? recLastBase(maxW1) : recFirstBase(maxW1);
        if (isEnd) {  // :363
          accR |= sym << (2 * (nRight & 15));
          if ((nRight & 15) == 15) {
            rightBuf[nRight >> 4] = accR;
            accR                  = 0;
          }
          nRight++;
        } else {
          accL |= sym << (2 * (nLeft & 15));
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
      link = next;
      ca   = na;
      cb   = nb;
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
      m[4]       = (rep ? 1 : 0) | (tooLong ? 2 : 0);
    }
    wv::sync();
  }

  /// buildContigs' contig loop (:685-713): the reference's seed sequence replayed over cached speculative walks.
  /// Returns 0 = all contigs built without a repeat hit (candidate c's walk sits in cache slot candSlotV of lane c),
  /// 1 = not for this path (a walk hit a repeat: the reference goes on to the next word length; contig too long; no room).
  template <int KW>
  WV_DEV int contigRounds()
  {
    const unsigned lane    = unsigned(wv::lane());
    const unsigned capCand = 2 * P.opt.maxAssemblyCount;
    nCand                  = 0;
    if (nNodes == 0) return 0;  // no word at this length (:522): no contig, no repeat
    const uint16_t* chain  = reinterpret_cast<const uint16_t*>(scratch());  // (valid if chainBytes != 0)
    char*          sc      = scratch() + chainBytes;
    const unsigned scBytes = scratchBytes() - chainBytes;
    if (scBytes < 1024 + 2 * TENT_CAP) return 1;  // no room for the seed selection's histograms
    unsigned long long* vm = visMask();
    for (unsigned i = lane; i < nNodes; i += 64) vm[i] = 0;
    slotNode[lane] = uint16_t(FA_NO_SLOT);
    uint64_t cached = 0;  // cache slots in use
    uint64_t accAll = 0;  // slots that hold accepted candidates (never evicted)
    // ---- round 0: the first seed and, beside it, the two lowest count tiers in exact seed order (error branches, also
    // those two reads share; the low-coverage ends of the main path that slip in cost a lane each, nothing else) ----
    const unsigned s1 = teamSeed<KW>();
    if (s1 == ASM_NONE) return 0;
    {
      // Up to 128 words of those tiers in exact order, thinned per unbranched stretch (chain ids, graphHasCycle): a walk
      // always runs from its seed to the far end of the seed's stretch (the seed's reads carry it), so a word with an
      // earlier-ranked word of its stretch on its near side is consumed by that word's walk -- walking it too would only
      // fill the cache with walks nobody asks for.  What lies on the other side of a walked word may survive (the walk back
      // from the seed can have lost the stretch's read by then): those words stay in.
      // (A heuristic like the rest of the speculation: what it drops or keeps wrongly costs a later round, never a result.)
      const unsigned nE = teamTentative<KW>(128, P.opt.minCoverage + 1, sc, scBytes);
      unsigned eN[2], eCh[2];
      for (unsigned h = 0; h < 2; ++h) {
        const unsigned i = lane + 64 * h;
        eN[h]            = (i < nE) ? unsigned(tent[i]) : ASM_NONE;
        eCh[h]           = (i < nE) ? (chainBytes ? unsigned(chain[eN[h]]) : eN[h]) : ASM_NONE;
      }
      bool dup[2] = {false, false};
      for (unsigned j = 0; j < nE; ++j) {
        const unsigned cj = (j < 64) ? wv::readlane(eCh[0], int(j)) : wv::readlane(eCh[1], int(j - 64));
        const unsigned nj = (j < 64) ? wv::readlane(eN[0], int(j)) : wv::readlane(eN[1], int(j - 64));
        for (unsigned h = 0; h < 2; ++h)
          if (j < lane + 64 * h && cj == eCh[h] && nj < eN[h]) dup[h] = true;
      }
      wv::sync();
      unsigned n0 = 1;
      for (unsigned h = 0; h < 2; ++h) {
        const bool     keep = (lane + 64 * h < nE) && !dup[h] && eN[h] != s1;
        const uint64_t mk   = wv::ballot(keep);
        const unsigned pos  = n0 + unsigned(wv::popc(mk & ((uint64_t(1) << lane) - 1)));
        if (keep && pos < 64) slotNode[pos] = uint16_t(eN[h]);
        n0 += unsigned(wv::popc(mk));
      }
      if (n0 > 64) n0 = 64;
      if (lane == 0) slotNode[0] = uint16_t(s1);
      cached = (n0 >= 64) ? ~uint64_t(0) : ((uint64_t(1) << n0) - 1);
      wv::sync();
      tick(5);
      walkSlots<KW>(cached);
      tick(6);
    }
    bool first = true;
    while (nCand < capCand) {
      unsigned nL = 1;
      if (first) {
        if (lane == 0) tent[0] = uint16_t(s1);
        wv::sync();
      } else {
        nL = teamTentative<KW>(64, 255, sc, scBytes);
        if (nL == 0) break;
      }
      first = false;
      // cache slot of every list entry (lane i: entry i)
      const unsigned node = (lane < nL) ? unsigned(tent[lane]) : ASM_NONE;
      unsigned       slot = FA_NO_SLOT;
      auto findSlots = [&]() {
        const unsigned mineNode = unsigned(slotNode[lane]);  // lane s: the word of slot s (FA_NO_SLOT: none)
        slot                    = FA_NO_SLOT;
        for (unsigned s = 0; s < 64; ++s) {
          const unsigned v = wv::readlane(mineNode, int(s));
          if (v == node) slot = s;
        }
      };
      findSlots();
      uint64_t miss = wv::ballot(lane < nL && slot == FA_NO_SLOT);
      tick(5);
#ifdef MANTA_WAVE_EMU
      if (std::getenv("MANTA_EMU_FAST_TRACE")) {
        const unsigned c = (lane < nL) ? recCnt(nodes[node].w0) : 0;
        for (unsigned i = 0; i < nL; ++i) {
          const unsigned ci = wv::readlane(c, int(i)), si = wv::readlane(slot, int(i));
          const FRec rr = nodes[wv::readlane(node, int(i))];
          const unsigned rdx = recSupRef(rr.w1);
          if (lane == 0 && std::getenv("MANTA_EMU_FAST_TRACE")[0] == '2' && !(rdx & FA_FAT))
            std::fprintf(stderr, "%s%u:%u%s", i ? " " : "  L: ", rdx, recPb(rr.w0, rr.w1) - 16 * (rd[rdx] & 0x7ffu), si == FA_NO_SLOT ? "!" : "");
          else
          if (lane == 0) std::fprintf(stderr, "%s%u%s", i ? " " : "  L: ", ci, si == FA_NO_SLOT ? "!" : "");
        }
        if (lane == 0) std::fprintf(stderr, "   (nCand %u, nNodes %u)\n", nCand, nNodes);
      }
#endif
      // Walk only when the very next seed has no walk yet.  Otherwise replay first: most rounds end there (enough candidates,
      // or the unwalked seeds turn out consumed -- the other words of a branch whose first word was walked), and a walk
      // round is the expensive thing.
      if (!(miss & 1u)) miss = 0;
      if (miss) {
        if (unsigned(wv::popc(~cached)) < unsigned(wv::popc(miss))) {
          // short of slots: take back those whose seed has been consumed since (such a walk can never be accepted)
          const unsigned sn   = unsigned(slotNode[lane]);
          const uint64_t dead = wv::ballot(((cached & ~accAll) >> lane) & 1u && sn != FA_NO_SLOT && !isUnused(sn));
          if (dead) {
            for (unsigned i = lane; i < nNodes; i += 64) vm[i] = wv::atomic_load(&vm[i]) & ~dead;
            if ((dead >> lane) & 1u) slotNode[lane] = uint16_t(FA_NO_SLOT);
            cached &= ~dead;
            FA_STAT(5, unsigned(wv::popc(dead)));
            wv::sync();
          }
        }
        if (cached == ~uint64_t(0) && (miss & 1u)) {
          // still full and the very next seed has no walk: drop every cached walk that is not an accepted candidate
          for (unsigned i = lane; i < nNodes; i += 64) vm[i] = wv::atomic_load(&vm[i]) & accAll;
          if (!((accAll >> lane) & 1u)) slotNode[lane] = uint16_t(FA_NO_SLOT);
          cached = accAll;
          FA_STAT(4, 1);
          wv::sync();
          findSlots();
          miss = wv::ballot(lane < nL && slot == FA_NO_SLOT);
        }
        // the r-th missing entry takes the r-th free slot; entries past the free slots (and everything behind the first
        // of them) wait for the next round
        const uint64_t freeMask = ~cached;
        const unsigned nFree    = unsigned(wv::popc(freeMask));
        uint8_t*       tbl      = reinterpret_cast<uint8_t*>(sc);  // (the selection's histogram is dead by now)
        if ((freeMask >> lane) & 1u) tbl[wv::popc(freeMask & ((uint64_t(1) << lane) - 1))] = uint8_t(lane);
        wv::sync();
        const bool     isMiss = (miss >> lane) & 1u;
        const unsigned rnk    = unsigned(wv::popc(miss & ((uint64_t(1) << lane) - 1)));
        const uint64_t late   = wv::ballot(isMiss && rnk >= nFree);
        if (late) nL = unsigned(wv::ctz(late));
        uint64_t walkMask = 0;
        if (isMiss && rnk < nFree && lane < nL) {
          slot           = tbl[rnk];
          slotNode[slot] = uint16_t(node);
        }
        {
          // (bits of the slots just taken, gathered from the lanes that took them)
          unsigned mineSlot = (isMiss && rnk < nFree && lane < nL) ? slot : 64u;
          for (int off = 0; off < 64; ++off) {
            const unsigned v = wv::readlane(mineSlot, off);
            if (v < 64) walkMask |= uint64_t(1) << v;
          }
        }
        cached |= walkMask;
        wv::sync();
        if (nL == 0) return 1;  // (cannot happen: a missing first entry finds a free slot after the eviction)
        if (walkMask) {
          walkSlots<KW>(walkMask);
          tick(6);
        }
      }
      // replay: entry i is the reference's next seed iff no walk accepted before it in this round touched its word (every
      // entry was unused when the list was made)
      unsigned long long vmI = 0;
      unsigned           flI = 0;
      if (lane < nL) {
        vmI = wv::atomic_load(&vm[node]);
        if (slot != FA_NO_SLOT) flI = unsigned(A.lane_meta[slot * 8 + 4]);
      }
      uint64_t acc = 0;
      bool     bad = false;
      for (unsigned i = 0; i < nL && nCand < capCand; ++i) {
        const uint64_t v = wv::readlane(uint64_t(vmI), int(i));
#ifdef MANTA_WAVE_EMU
        if (std::getenv("MANTA_EMU_FAST_TRACE") && std::getenv("MANTA_EMU_FAST_TRACE")[0] == '3' && lane == 0)
          std::fprintf(stderr, " [%u:%s]", i, (v & acc) ? "consumed" : (wv_emu::W()->xbuf[0][0], "live"));
#endif
        if (v & acc) continue;  // consumed by an accepted walk: not a seed for the reference either
        const unsigned sl = wv::readlane(slot, int(i));
        if (sl == FA_NO_SLOT) break;  // the next seed has not been walked: next round
        if (wv::readlane(flI, int(i)) != 0) {  // repeat hit (the reference moves on to the next word length) or contig too long
          bad = true;
          break;
        }
        acc |= uint64_t(1) << sl;
        if (lane == nCand) candSlotV = sl;
        nCand++;
      }
      if (bad) return 1;
      accAll |= acc;
      // unusedWords.erase for every word of the accepted walks (:170,482)
      for (unsigned nb = 0; nb < ((nNodes + 63) & ~63u); nb += 64) {
        const unsigned nd = nb + lane;
        const uint64_t m  = wv::ballot(nd < nNodes && (wv::atomic_load(&vm[nd < nNodes ? nd : 0]) & acc) != 0);
        if (lane < 2) unused_bits[(nb >> 5) + lane] &= ~uint32_t(m >> (32 * lane));
      }
      wv::sync();
      tick(7);
    }
    return 0;
  }

  // ------------------------------------------------------------------------------------------------
  // selectContigs (:722-842) + output, lane c = candidate c (see Assembler::selectAndEmit for the general form)
  // ------------------------------------------------------------------------------------------------
  WV_DEV void selectAndEmit(const unsigned locus)
  {
    const unsigned lane = unsigned(wv::lane());
    uint64_t       sup0 = 0, sup1 = 0, rej0 = 0, rej1 = 0;
    unsigned       nLeft = 0, nRight = 0, myLen = 0;
    int            consB = 0, consE = 0;
    if (lane < nCand) {
      const uint64_t* lb = A.lane_bits + size_t(candSlotV) * 2 * WQ_MAX;
      sup0               = lb[0];
      sup1               = lb[1];
      rej0               = lb[WQ_MAX];
      rej1               = lb[WQ_MAX + 1];
      const int32_t* m   = A.lane_meta + candSlotV * 8;
      nLeft              = unsigned(m[0]);
      nRight             = unsigned(m[1]);
      consB              = m[2];
      consE              = m[3];
      myLen              = nLeft + k + nRight;
    }
    uint64_t used0 = 0, used1 = 0;  // wave-uniform
    bool     aliveL     = lane < nCand;
    unsigned finalCount = 0;
    uint64_t chosen     = 0;  // chosen candidates in order, 6 bits each (maxAssemblyCount <= 10 fits a qword; more: second word)
    uint64_t chosenHi   = 0;
    while (finalCount < P.opt.maxAssemblyCount) {
      if (!wv::any(aliveL)) break;
      const unsigned usedNormal = unsigned(wv::popc(used0)) + unsigned(wv::popc(used1));  // (no pseudo reads on this path)
      if (nNormal - usedNormal < P.opt.minUnusedReads) break;  // :750
      const unsigned nFresh = unsigned(wv::popc(sup0 & ~used0)) + unsigned(wv::popc(sup1 & ~used1));
      if (aliveL && nFresh < P.opt.minSupportReads) aliveL = false;  // :779-788
      uint64_t key = aliveL ? ((uint64_t(nFresh) << 40) | (uint64_t(myLen) << 8) | uint64_t(63u - lane)) : 0;
      for (int off = 1; off < 64; off <<= 1) {
        const uint64_t o = wv::shfl(key, wv::lane() ^ off);
        key              = (o > key) ? o : key;
      }
      if ((key >> 40) == 0) break;  // :807
      const int selected = wv::first(int(63u - unsigned(key & 63u)));
      if (finalCount < 10)
        chosen |= uint64_t(selected) << (6 * finalCount);
      else
        chosenHi |= uint64_t(selected) << (6 * (finalCount - 10));
      if (int(lane) == selected) aliveL = false;
      used0 |= wv::readlane(sup0, selected);
      used1 |= wv::readlane(sup1, selected);
      finalCount++;
    }
    auto chosenAt = [&](const unsigned f) { return unsigned(((f < 10) ? (chosen >> (6 * f)) : (chosenHi >> (6 * (f - 10)))) & 63u); };

    AsmLocusOut out;
    out.status            = ASM_OK;
    out.n_contigs         = finalCount;
    out.n_words           = W;
    out.n_pseudo          = 0;
    out.final_word_length = k;
    out.n_iterations      = 1;
    out.cyclic_iterations = 0;
    out.reserved          = 0;
    uint64_t seqBytes = 0;
    for (unsigned f = 0; f < finalCount; ++f) seqBytes += wv::readlane(myLen, int(chosenAt(f)));
    const uint64_t     bitsWords = uint64_t(finalCount) * 2 * W;
    unsigned long long seqBase = 0, bitsBase = 0;
    if (lane == 0) {
      seqBase  = wv::atomic_add(P.seq_used, (unsigned long long)seqBytes);
      bitsBase = wv::atomic_add(P.bits_used, (unsigned long long)bitsWords);
    }
    seqBase  = wv::readlane(uint64_t(seqBase), 0);
    bitsBase = wv::readlane(uint64_t(bitsBase), 0);
    if (seqBase + seqBytes > P.seq_cap || bitsBase + bitsWords > P.bits_cap) {
      out.status         = ASM_E_OUT_CAPACITY;
      out.n_contigs      = 0;
      out.pseudo_off     = 0;
      out.pseudo_len_off = 0;
      if (lane == 0) P.loci[locus] = out;
      return;
    }
    uint64_t       so = seqBase, bo = bitsBase;
    const unsigned seqWords = P.max_contig_len / 16 + 2;
    for (unsigned f = 0; f < finalCount; ++f) {
      const int      c   = int(chosenAt(f));
      const unsigned sl  = wv::readlane(candSlotV, c);
      const unsigned nL  = wv::readlane(nLeft, c), nR = wv::readlane(nRight, c), len = nL + k + nR;
      const FRec     sr  = nodes[slotNode[sl]];
      const unsigned seedPb = recPb(sr.w0, sr.w1);
      const uint32_t* rightBuf = reinterpret_cast<const uint32_t*>(A.lane_seq) + size_t(sl) * 2 * seqWords;
      const uint32_t* leftBuf  = rightBuf + seqWords;
      for (unsigned i = lane; i < len; i += 64) {  // reverse(left) + seed + right
        unsigned code;
        if (i < nL) {
          const unsigned j = nL - 1 - i;
          code             = (leftBuf[j >> 4] >> (2 * (j & 15))) & 3;
        } else if (i < nL + k) {
          code = baseAt(seedPb + (i - nL));
        } else {
          const unsigned j = i - nL - k;
          code             = (rightBuf[j >> 4] >> (2 * (j & 15))) & 3;
        }
        P.seq_arena[so + i] = uint8_t("ACGT"[code]);
      }
      const uint64_t s0 = wv::readlane(sup0, c), s1 = wv::readlane(sup1, c), r0 = wv::readlane(rej0, c), r1 = wv::readlane(rej1, c);
      if (lane < 2 * W) {
        const unsigned half = lane / W, w = lane % W;
        P.bits_arena[bo + lane] = half ? (w ? r1 : r0) : (w ? s1 : s0);
      }
      const int cb = wv::readlane(consB, c), ce = wv::readlane(consE, c);
      if (lane == 0) {
        AsmContigOut o;
        o.seq_off    = so;
        o.bits_off   = bo;
        o.seq_len    = len;
        o.cons_begin = cb;
        o.cons_end   = int(len) - ce;  // :498
        o.reserved   = 0;
        P.contigs[size_t(locus) * P.opt.maxAssemblyCount + f] = o;
      }
      so += len;
      bo += 2 * W;
    }
    out.pseudo_off     = so;
    out.pseudo_len_off = bo;
    if (lane == 0) P.loci[locus] = out;
  }

  template <int KW>
  WV_DEV int runK(const unsigned locus)
  {
    if (!buildGraph<KW>()) return FA_PUNT;
    // from here on the first wave works alone (the walks are one dependent chain per lane); the others wait at the barrier
    int rc = FA_PUNT;
    if (tw == 0) {
      rc = afterGraph<KW>(locus);
      command(FA_CMD_EXIT);
    } else {
      workerLoop<KW>();
    }
    return rc;
  }

  /// first wave: the team runs operation `cmd` next (the caller goes on to run it too)
  WV_DEV void command(const unsigned cmd, const unsigned a0 = 0, const unsigned a1 = 0, const unsigned a2 = 0)
  {
    if (tn == 1) return;
    if (wv::lane() == 0) {
      hdr[FA_H_CMD]  = cmd;
      hdr[FA_H_ARG0] = a0;
      hdr[FA_H_ARG1] = a1;
      hdr[FA_H_ARG2] = a2;
    }
    teamSync();
  }
  /// the other waves: run what the first wave asks for until it is done with the locus
  template <int KW>
  WV_DEV void workerLoop()
  {
    while (true) {
      teamSync();
      const unsigned cmd = wv::first(wv::atomic_load(&hdr[FA_H_CMD]));
      if (cmd == FA_CMD_EXIT) break;
      const unsigned a0 = wv::first(wv::atomic_load(&hdr[FA_H_ARG0])), a1 = wv::first(wv::atomic_load(&hdr[FA_H_ARG1])),
                     a2 = wv::first(wv::atomic_load(&hdr[FA_H_ARG2]));
      if (cmd == FA_CMD_SEED) {
        (void)selectSeed<KW>();
      } else if (cmd == FA_CMD_TENT) {
        (void)selectTentative<KW>(a0 & 0xffffu, a0 >> 16, lds + a1, a2);
      }
    }
  }
  template <int KW>
  WV_DEV unsigned teamSeed()
  {
    command(FA_CMD_SEED);
    return selectSeed<KW>();
  }
  template <int KW>
  WV_DEV unsigned teamTentative(const unsigned T, const unsigned maxCount, char* area, const unsigned areaBytes)
  {
    command(FA_CMD_TENT, T | (maxCount << 16), unsigned(area - lds), areaBytes);
    return selectTentative<KW>(T, maxCount, area, areaBytes);
  }

  template <int KW>
  WV_DEV int afterGraph(const unsigned locus)
  {
    if (graphHasCycle() != 0) return FA_PUNT;  // cyclic (exact repeat search) or no room: general path
    tick(3);
    if (contigRounds<KW>() != 0) return FA_PUNT;
    selectAndEmit(locus);
    tick(7);
    FA_STAT(0, 1);
    FA_STAT(3, nCand);
    return FA_DONE;
  }

  /// the whole fast path for one locus.  FA_DONE: results emitted.  FA_PUNT: nothing emitted, the general path takes it.
  WV_DEV int run(const unsigned locus)
  {
    const unsigned minWL = P.locus_min_wl ? P.locus_min_wl[locus] : P.opt.minWordLength;
    const unsigned maxWL = P.locus_max_wl ? P.locus_max_wl[locus] : P.opt.maxWordLength;
    if (minWL == 0 || maxWL > 16u * ASM_MAX_KW || minWL > maxWL || 2 * P.opt.maxAssemblyCount > ASM_MAX_CAND) return FA_PUNT;
    if (P.opt.minCoverage > 255 || P.opt.maxAssemblyCount > 20) return FA_PUNT;
    k     = minWL;
    A.k   = k;  // the key helpers of the general path read it
    tMark = wv::clock();
    if (!pack(locus)) return FA_PUNT;
    tick(0);
    if (tw == 0) {
      unused_bits[wv::lane()] = 0;
      repeat_bits[wv::lane()] = 0;
    }
    wv::sync();  // (buildGraph's first barrier comes before anything reads these)
    const unsigned kw = (k + 15) >> 4;
    if (kw <= 2) return runK<2>(locus);
    if (kw <= 4) return runK<4>(locus);
    return runK<8>(locus);
  }
};

/// persistent workgroups of FA_TEAM cooperating wavefronts (one per SIMD of the CU), FA_BUDGET bytes of dynamic LDS each: one
/// locus at a time per workgroup; params as assemble_kernel.  (The block size chooses the team: 64 threads run the same code
/// with one wave.)  Loci this path does not cover are appended to P.punt_ids (P.punt_count counts them); assemble_kernel,
/// launched behind this kernel with n_loci_dev = punt_count and locus_ids = punt_ids, runs them.
// Register budget.  The walk loop (walkSlots inlined into contigRounds) needs ~200 VGPRs; a budget of 168 (three waves per
// SIMD, i.e. three workgroups of four waves per CU) makes the compiler spill 31 of them around that loop, and the spilled
// build produced wrong contig sets on the hardware for loci with more than one walk round (round 3, digests of 73 of 10 000
// loci; the unspilled build is exact) -- so the budget stays at two waves per SIMD and the launch uses teams of two.
#ifndef MANTA_FAST_WAVES_PER_SIMD
#define MANTA_FAST_WAVES_PER_SIMD 2
#endif
WV_KERNEL_WG(FA_TEAM) WV_WAVES_PER_SIMD(MANTA_FAST_WAVES_PER_SIMD) void assemble_fast_kernel(const AsmParams P)
{
  uint8_t*       wsBase = P.ws + uint64_t(wv::block_single()) * P.ws_stride;
  char*          lds    = wv::lds_single();
  uint32_t*      hdr    = reinterpret_cast<uint32_t*>(lds + FA_OFF_HDR);
  const unsigned tw     = unsigned(wv::wave_in_wg());
  while (true) {
    if (tw == 0 && wv::lane() == 0) hdr[FA_H_SLOT] = wv::atomic_add(P.counter, 1u);
    wv::sync();
    wv::wg_barrier();
    const unsigned slot = wv::first(wv::atomic_load(&hdr[FA_H_SLOT]));
    if (slot >= P.n_loci) break;
    const unsigned locus = P.locus_ids ? P.locus_ids[slot] : slot;
    Assembler      a(P, wsBase);
    int            rc = FA_PUNT;
    const bool arrived = !P.upload_chunks_done || asmWaitUploaded(P, locus);
    if (arrived) {
      FastAsm f(a, lds);
      rc = f.run(locus);
    }
    wv::sync();
    if (tw == 0 && rc != FA_DONE && wv::lane() == 0) P.punt_ids[wv::atomic_add(P.punt_count, 1u)] = locus;
    wv::sync();
    wv::wg_barrier();  // (the slot word is rewritten next)
  }
}

}  // namespace manta_dev
