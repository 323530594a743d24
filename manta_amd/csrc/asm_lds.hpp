// The iterative assembler's common case as an LDS pipeline of TWO kernels (assembly/IterativeAssembler.cpp:844-931; first word
// length of a locus whose k-mer graph is acyclic and small enough):
//
//   graph_kernel   one workgroup of LG_WAVES wavefronts per locus, LG_BUDGET bytes of LDS (two workgroups per CU).  Everything
//                  that is parallel over k-mer instances or over words: pack -> table pass (getKmerCounts :506-550; the slot of a
//                  word IS its identity: claim + one ds_or into the slot's read set) -> counts -> SORT of the words into the
//                  reference's seed order (:686-696: count descending, k-mer ascending) -> 8-byte node records with successor /
//                  predecessor links in that numbering -> proof of acyclicity from the reads' offsets -> the compact graph (8 bytes
//                  per word + 16 per word with more than one read) goes to a per-locus slab in global memory.
//   contig_kernel  (asm_contig.hpp) one single-wave workgroup per locus, only the compact graph in LDS (12-19 KB for a config-2
//                  locus: eight per CU): cycle test where there is no proof, the contig loop (:685-713) on speculative lane-per-seed
//                  walks (:149-501), selectContigs (:722-842), output.
//
// Why two kernels.  The fused predecessor of this file (assemble_fast_kernel, rounds 2-3) held a locus' 52 KB of LDS from the
// first byte to the last contig: three loci per CU, and for two thirds of a locus' time only one wave of the workgroup had work
// (the walks are one dependent chain per lane).  The parallel half wants many waves and room for simple dense structures; the
// serial half wants as many loci in flight as possible and needs neither the pile nor the hash table.  Splitting at the graph
// gives each half its own occupancy, register budget and LDS map; the hand-over costs ~16 KB per locus, written and read once,
// coalesced.
//
// What the sort buys: with node ids in seed order, "the next unused seed" is the lowest set bit of a bitmap, the two lowest
// count tiers are an id range, the words with more than one read are the ids below nFat (their bitsets need no reference), and
// the contig kernel never looks at a key -- no pile, no table, no key compares in LDS.
//
// Anything this path does not cover -- a cycle, a repeat hit that asks for the next word length, > LG_MAX_READS reads, a graph that
// does not fit, bytes outside {A,C,G,T,N} -- goes onto the punt list; assemble_kernel, launched behind, picks it up from device
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
static const unsigned LG_SIB_CAP   = 32;     // words without a predecessor that have siblings (side table)
static const unsigned LG_OVF_CAP   = 32;     // words with more than two successors / predecessors (side tables)
static const unsigned LG_CLASSES   = 4;      // LDS size classes of contig_kernel (one launch each)

// graph_kernel LDS map (bytes)
static const unsigned LG_OFF_HDR   = 0;                          // u32[64] header words, u16[128] byte offset of a read in the staged pile
static const unsigned LG_OFF_RD    = 512;                        // u32[128] read descriptors {code dword offset : 11, length : 16, has N : 1}
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
static_assert(LG_OFF_DYN + 12000 <= LG_BUDGET, "graph_kernel LDS map");
static_assert(8 * (LG_SIB_CAP + 2 * LG_OVF_CAP) + 528 <= 1024 * LG_WAVES, "side tables inside the histogram region");
// (dead after the sort: the per-wave histograms hold the side tables and the chain labels of the speculation list; sortB the
// words' potentials; the digit bases hold the reads' anchors during the table pass)
static const unsigned LG_OFF_SIB   = LG_OFF_WHIST;                    // u16[LG_SIB_CAP][4]
static const unsigned LG_OFF_SOVF  = LG_OFF_SIB + 8 * LG_SIB_CAP;     // u16[LG_OVF_CAP][4]
static const unsigned LG_OFF_POVF  = LG_OFF_SOVF + 8 * LG_OVF_CAP;    // u16[LG_OVF_CAP][4]
static const unsigned LG_OFF_CHAIN = LG_OFF_POVF + 8 * LG_OVF_CAP;    // u16[128] label, u16[128] distance, u32[4] duplicate bits
static const unsigned LG_FILTER_BITS = 16384;
static const unsigned LG_OFF_FILTER = LG_OFF_WHIST + 4096;            // u32[512] presence filter over the hash tags
static const unsigned LG_OFF_PHI   = LG_OFF_SORTB;                    // i16[2048]
static const unsigned LG_OFF_ANCH  = LG_OFF_DBASE;                    // u32[128] anchors, then i32[128] offsets
static const unsigned LG_NO_ANCHOR = 0xffffffffu;

// header words (graph_kernel LDS)
enum {
  LG_H_SLOT = 0,   ///< the queue slot of the workgroup's current locus
  LG_H_FLAG = 1,   ///< pack: a byte outside the alphabet was seen; table: full
  LG_H_N    = 2,   ///< sort: number of words
  LG_H_NSIB = 3,
  LG_H_NSOVF = 16, LG_H_NPOVF = 17,
  LG_H_CYC  = 18,  ///< an edge against the potential: no proof of acyclicity
  LG_H_PAR  = 32,  ///< [32] u8[128]: parent read of a read's anchor (offset resolution)
  LG_H_OFF_LO = 4, ///< slab offset of this locus in the arena (bytes)
  LG_H_OFF_HI = 5,
  LG_H_PUNT = 6,
  LG_H_TOT  = 8    ///< [4] scan: totals of the four digit quarters
};

/// A word of the compact graph: 8 bytes.
///   [0,11) [11,22)  successors 0, 1 (id + 1, 0 = none; A,C,G,T order)     [22,33) [33,44)  predecessors 0, 1 (same)
///   [44,48) count, saturated at 15   [48,55) the word's only read (count 1; words with more reads are the ids below nFat and
///   own bitset `id` of the pool)   [55,57) first base   [57,59) last base   59 self loop
///   60 / 61: a third / fourth successor / predecessor in the overflow tables (three-way branches: a handful per locus)
typedef uint64_t FRec8;
struct alignas(16) FSet {
  uint64_t w[2];
};
struct alignas(16) FBucket {
  uint32_t s[4];
};

/// what graph_kernel leaves in a locus' slab: this header, then (lgSlab) FRec8[nNodes], FSet[nFat], u16[64] speculation list,
/// u16[LG_SIB_CAP][4] sibling table, u16[LG_OVF_CAP][4] x 2 overflow tables, u16[nNodes] first occurrences, u32[codeWords] 2-bit
/// pile (the last two for the seeds' text)
struct alignas(16) LgHdr {
  uint32_t nNodes, nFat, k, nNormal;
  uint32_t nEligible;     ///< ids below it are seeds (:678-682)
  uint32_t nSpec;         ///< entries of the round-0 walk list (entry 0 = the first seed)
  uint32_t nSib, codeWords;
  uint32_t W, need;
  uint32_t nSovf, nPovf;
  uint32_t acyclic;       ///< graph_kernel has a proof that the graph has no cycle (potential from the reads' offsets): no peel
  uint32_t nPseudo;       ///< big class, later word lengths: reads nNormal .. nNormal + nPseudo - 1 are the previous length's contigs
  uint32_t cyclic;        ///< big class: repeat_big_kernel found a cycle and left the core / repeat-word bitmaps in the slab
  uint32_t nCore;         ///< ... words left by the two-sided peel (only these can be met twice by a walk)
};
struct LgSlab {
  uint32_t recs, pool, spec, sib, sovf, povf, pb, codes, total, rd1, lex, flags;
};
WV_HD LgSlab lgSlab(const unsigned nNodes, const unsigned nFat, const unsigned codeWords)
{
  LgSlab   L;
  uint32_t o = uint32_t(sizeof(LgHdr));
  L.rd1   = 0;
  L.lex   = 0;
  L.flags = 0;
  L.recs  = o;
  o += (8u * nNodes + 15u) & ~15u;
  L.pool  = o;
  o += 16u * nFat;
  L.spec  = o;
  o += 128u;
  L.sib   = o;
  o += 8u * LG_SIB_CAP;
  L.sovf  = o;
  o += 8u * LG_OVF_CAP;
  L.povf  = o;
  o += 8u * LG_OVF_CAP;
  L.pb    = o;
  o += (2u * nNodes + 15u) & ~15u;
  L.codes = o;
  o += (4u * codeWords + 15u) & ~15u;
  L.total = o;
  return L;
}
WV_HD uint64_t lgSlabBytes(const unsigned nNodes, const unsigned nFat, const unsigned codeWords)
{
  return lgSlab(nNodes, nFat, codeWords).total;
}

// contig_kernel LDS map
static const unsigned CK_OFF_HDR    = 0;     // LgHdr
static const unsigned CK_OFF_UNUSED = 64;    // u32[64] "unusedWords" bitmap over node ids
static const unsigned CK_OFF_TENT   = 320;   // u16[128] seed list of the round
static const unsigned CK_OFF_SLOTND = 576;   // u16[64] word walked by cache slot s
static const unsigned CK_OFF_TBL    = 704;   // u8[64]
static const unsigned CK_OFF_SIB    = 768;   // u16[LG_SIB_CAP][4]
static const unsigned CK_OFF_SOVF   = CK_OFF_SIB + 8 * LG_SIB_CAP;
static const unsigned CK_OFF_POVF   = CK_OFF_SOVF + 8 * LG_OVF_CAP;
static const unsigned CK_OFF_RECS   = CK_OFF_POVF + 8 * LG_OVF_CAP;
/// LDS the contig kernel needs for a graph: records + the larger of {bitset pool, cycle-test state (not for a graph that comes
/// with a proof of acyclicity)}
WV_HD unsigned ckNeed(const unsigned nNodes, const unsigned nFat, const bool acyclic)
{
  const unsigned kahn = acyclic ? 0u : ((4 * ((nNodes + 3) / 4) + 2 * nNodes + 32 + 15) & ~15u);
  const unsigned pool = 16 * (nFat ? nFat : 1u);
  return CK_OFF_RECS + ((8 * nNodes + 15) & ~15u) + ((pool > kahn) ? pool : kahn);
}

// ---- the pipeline's second size class ("big": asm_lds_big.hpp): piles of up to 256 reads, several thousand words -- the config-4/5
// shape (200 reads x 250 bases).  One graph workgroup of LGL_WAVES wavefronts owns a CU's whole LDS; what the two classes share is the
// shape of the compact graph, and contig_kernel's code is one template over these traits.
static const unsigned LGL_SLOTS     = 8192;
static const unsigned LGL_BUCKETS   = LGL_SLOTS / 4;
static const unsigned LGL_MAX_NODES = 7168;   // 0.875 x slots; ids are stored +1 in 13-bit link fields
static const unsigned LGL_MAX_READS = 256;    // read sets of four qwords
static const unsigned LGL_MAX_PILE  = 3598;   // code dwords (+ 2 of padding) of the locus' own reads
static const unsigned LGL_MAX_PILE_ALL = 4090; // ... with the pseudo reads of a later word length behind them: a packed base index must fit 16 bits
static const unsigned LGL_POOL_CAP  = 1280;   // read sets handed out during the table pass (words with more than one read) that live in LDS
static const unsigned LGL_POOL_OVF  = 704;    // ... and further ones in the workgroup's device-memory workspace (later word lengths: the pseudo reads share most of their words)
// graph_big_kernel's device-memory workspace per workgroup: the overflow read sets, then the words' lexicographic ranks by table slot and by id
// (they fall out of the seed-order sort's byte passes; the repeat search of a cyclic graph wants them: LdsGraphL::sortWords / lexOrder)
static const unsigned LGL_GWS_SETS  = 32 * LGL_POOL_OVF;
static const unsigned LGL_GWS_BYTES = LGL_GWS_SETS + 2 * 8192 + 2 * 8192;
static const unsigned LGL_WAVES     = 16;
static const unsigned LGL_BUDGET    = 163840;
static const unsigned LGL_OVF_CAP   = 128;
static const unsigned LGL_CLASSES   = 2;      // LDS size classes of contig_big_kernel

/// A word of the big class' compact graph, 8 bytes as well:
///   [0,13) [13,26) successors 0, 1 (id + 1, 0 = none; A,C,G,T order)   [26,39) [39,52) predecessors 0, 1
///   [52,56) count, saturated at 15   [56,58) first base   [58,60) last base   60 self loop
///   61 / 62: more than two successors / predecessors -- field 1 then holds the INDEX of the word's overflow entry {second, third,
///   fourth} instead of the second link (no search).  The only read of a single-read word does not fit: u8 array `rd1` by id.
struct LgS {
  static const bool     BIG       = false;
  static const unsigned ID_BITS   = 11;
  static const unsigned SETW      = 2;   ///< qwords of a read set
  static const unsigned MAX_NODES = LG_MAX_NODES;
  static const unsigned OVF_CAP   = LG_OVF_CAP;
  static const unsigned UNUSED_DW = 64;  ///< dwords of the "unusedWords" bitmap
  static const unsigned CNT_SH = 44, FIRST_SH = 55, LAST_SH = 57, SELF_SH = 59, SOVF_SH = 60, POVF_SH = 61;
};
struct LgL {
  static const bool     BIG       = true;
  static const unsigned ID_BITS   = 13;
  static const unsigned SETW      = 4;
  static const unsigned MAX_NODES = LGL_MAX_NODES;
  static const unsigned OVF_CAP   = LGL_OVF_CAP;
  static const unsigned UNUSED_DW = 256;
  static const unsigned CNT_SH = 52, FIRST_SH = 56, LAST_SH = 58, SELF_SH = 60, SOVF_SH = 61, POVF_SH = 62;
};
template <class C>
struct alignas(16) FSetT {
  uint64_t w[C::SETW];
};
/// record fields by class
template <class C>
struct LgRec {
  static const unsigned IDM  = (1u << C::ID_BITS) - 1u;
  static const uint64_t M2   = (uint64_t(1) << (2 * C::ID_BITS)) - 1;
  WV_DEV static unsigned succ(const FRec8 w, const unsigned i) { return unsigned(w >> (C::ID_BITS * i)) & IDM; }
  WV_DEV static unsigned pred(const FRec8 w, const unsigned i) { return unsigned(w >> (C::ID_BITS * (2 + i))) & IDM; }
  WV_DEV static unsigned cnt(const FRec8 w) { return unsigned(w >> C::CNT_SH) & 15u; }
  WV_DEV static unsigned firstBase(const FRec8 w) { return unsigned(w >> C::FIRST_SH) & 3u; }
  WV_DEV static unsigned lastBase(const FRec8 w) { return unsigned(w >> C::LAST_SH) & 3u; }
  WV_DEV static bool     selfLoop(const FRec8 w) { return (w >> C::SELF_SH) & 1u; }
  WV_DEV static bool     sOvf(const FRec8 w) { return (w >> C::SOVF_SH) & 1u; }
  WV_DEV static bool     pOvf(const FRec8 w) { return (w >> C::POVF_SH) & 1u; }
  /// the (up to four) neighbours of word `nd` as 4 x ID_BITS (id + 1)
  WV_DEV static uint64_t links(const FRec8 w, const unsigned nd, const bool succ, const uint16_t* ovf, const unsigned nOvf)
  {
    uint64_t l = (succ ? w : (w >> (2 * C::ID_BITS))) & M2;
    if (succ ? sOvf(w) : pOvf(w)) {
      if (C::BIG) {
        const unsigned e = unsigned(l >> C::ID_BITS) & IDM;  // (the overflow entry's index sits where the second link would)
        l = (l & IDM) | (uint64_t(ovf[4 * e]) << C::ID_BITS) | (uint64_t(ovf[4 * e + 1]) << (2 * C::ID_BITS)) | (uint64_t(ovf[4 * e + 2]) << (3 * C::ID_BITS));
      } else {
        for (unsigned e = 0; e < nOvf; ++e)
          if (unsigned(ovf[4 * e]) == nd) l |= (uint64_t(ovf[4 * e + 1]) << 22) | (uint64_t(ovf[4 * e + 2]) << 33);
      }
    }
    return l;
  }
  WV_DEV static unsigned linkField(const uint64_t l, const unsigned c) { return unsigned(l >> (C::ID_BITS * c)) & IDM; }  // id + 1
  WV_DEV static unsigned linkId(const uint64_t l, const unsigned c)
  {
    const unsigned f = linkField(l, c);
    return f ? f - 1 : ASM_NONE;
  }
};

/// the big class' slab: LgHdr, FRec8[nNodes], FSetT<LgL>[nFat], u16[64] speculation list, u16[LG_SIB_CAP][4] sibling table,
/// u16[LGL_OVF_CAP][4] x 2 overflow tables, u8[nNodes] the only read of a single-read word, u16[nNodes] first occurrences, the pile,
/// u16[nNodes] lexicographic ranks, the core / repeat-word bitmaps
WV_HD LgSlab lgSlabL(const unsigned nNodes, const unsigned nFat, const unsigned codeWords)
{
  LgSlab   L;
  uint32_t o = uint32_t(sizeof(LgHdr));
  L.recs  = o;
  o += (8u * nNodes + 15u) & ~15u;
  L.pool  = o;
  o += 32u * nFat;
  L.spec  = o;
  o += 128u;
  L.sib   = o;
  o += 8u * LG_SIB_CAP;
  L.sovf  = o;
  o += 8u * LGL_OVF_CAP;
  L.povf  = o;
  o += 8u * LGL_OVF_CAP;
  L.rd1   = o;
  o += (nNodes + 15u) & ~15u;
  L.pb    = o;
  o += (2u * nNodes + 15u) & ~15u;
  L.codes = o;
  o += (4u * codeWords + 15u) & ~15u;
  L.lex   = o;  // u16[nNodes]: the words' lexicographic ranks (written for a graph without a proof of acyclicity only)
  o += (2u * nNodes + 15u) & ~15u;
  L.flags = o;  // u32[LgL::UNUSED_DW] core words, u32[LgL::UNUSED_DW] repeat words (repeat_big_kernel)
  o += 8u * 256u;
  L.total = o;
  return L;
}
template <class C>
WV_HD LgSlab lgSlabOf(const unsigned nNodes, const unsigned nFat, const unsigned codeWords)
{
  return C::BIG ? lgSlabL(nNodes, nFat, codeWords) : lgSlab(nNodes, nFat, codeWords);
}

/// contig_kernel's LDS map by class (the small class' offsets are the CK_OFF_* above)
#ifndef MANTA_STRETCH_WIN
#define MANTA_STRETCH_WIN 4
#endif
static const unsigned LGL_STRETCH_WIN = MANTA_STRETCH_WIN;  ///< contig_big_kernel's walk-round lists: unused words looked at, per lane
template <class C>
struct CkMap {
  static const unsigned UNUSED = 64;
  static const unsigned TENT   = UNUSED + 4 * C::UNUSED_DW;
  static const unsigned SLOTND = TENT + (C::BIG ? 2 * 64 * LGL_STRETCH_WIN : 256);  // (big class: 64 x LGL_STRETCH_WIN words looked at per list, stretchSeedList)
  static const unsigned TBL    = SLOTND + 128;
  static const unsigned SIB    = TBL + 64;
  static const unsigned SOVF   = SIB + 8 * LG_SIB_CAP;
  static const unsigned POVF   = SOVF + 8 * C::OVF_CAP;
  static const unsigned RECS   = POVF + 8 * C::OVF_CAP;
};
static_assert(CkMap<LgS>::TENT == CK_OFF_TENT && CkMap<LgS>::SLOTND == CK_OFF_SLOTND && CkMap<LgS>::TBL == CK_OFF_TBL && CkMap<LgS>::SIB == CK_OFF_SIB &&
              CkMap<LgS>::RECS == CK_OFF_RECS, "the small class keeps its map");
/// ckNeed by class (big: + the u8 array of single-read words' reads behind the records)
template <class C>
WV_HD unsigned ckNeedOf(const unsigned nNodes, const unsigned nFat, const bool acyclic)
{
  const unsigned kahn = acyclic ? 0u : ((4 * ((nNodes + 3) / 4) + 2 * nNodes + 32 + 15) & ~15u);
  const unsigned pool = 8 * C::SETW * (nFat ? nFat : 1u);
  return CkMap<C>::RECS + ((8 * nNodes + 15) & ~15u) + (C::BIG ? ((nNodes + 15) & ~15u) : 0u) + ((pool > kahn) ? pool : kahn);
}

/// ... of a CYCLIC graph (big class): + core bitmap, its prefix counts, repeat-word bitmap, one visited bitmap over the core per walk
static const unsigned LGL_CORE_CAP = 4096;  ///< core words a cyclic graph may have on the pipeline (more: general path)
template <class C>
WV_HD unsigned ckCyclicExtra(const unsigned nCore)
{
  return 4u * C::UNUSED_DW + 2u * C::UNUSED_DW + 4u * C::UNUSED_DW + 64u * 4u * ((nCore + 31u) / 32u);
}
template <class C>
WV_HD unsigned ckNeedCyclic(const unsigned nNodes, const unsigned nFat, const unsigned nCore)
{
  const unsigned pool = 8 * C::SETW * (nFat ? nFat : 1u);
  return CkMap<C>::RECS + ((8 * nNodes + 15) & ~15u) + (C::BIG ? ((nNodes + 15) & ~15u) : 0u) + pool + ckCyclicExtra<C>(nCore);
}

/// A big-class locus between two word lengths (IterativeAssembler.cpp:856-910): what contig_big_kernel leaves for graph_big_kernel's
/// next round.  The pseudo reads lie in the pseudo arena as 2-bit codes in the pile's layout ((len + 15) / 16 + 1 dwords each).
static const unsigned LGL_MAX_PSEUDO = 40;  // 2 * maxAssemblyCount (<= 20 on the pipeline)
static const unsigned LGL_MAX_ROUNDS = 24;  // word lengths one launch sequence covers (more: general path)
struct alignas(16) LgIter {
  uint32_t k;             ///< the next word length
  uint32_t nIter;         ///< word lengths done
  uint32_t cyclicIters;   ///< ... of which had a cyclic graph
  uint32_t nPseudo;
  uint64_t off;           ///< dword offset of the pseudo reads' codes in the arena
  uint32_t codeWords;     ///< their dwords
  uint32_t slabCap;       ///< bytes of the locus' slab (graph_big_kernel: a later round builds in place when its graph fits)
  uint16_t len[LGL_MAX_PSEUDO];
};

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
  uint32_t            flags;       ///< LG_FLAG_*
  uint32_t*           stats;       ///< [0] loci whose graph came with a proof of acyclicity, [1] reads re-anchored by readOffsets' second pass
  uint8_t*            cws;         ///< contig_kernel workspaces
  uint64_t            cws_stride;
  // ---- the big class' word-length rounds (asm_lds_big.hpp / asm_repeat_big.hpp); iter == nullptr: one round, repeat hits are handed back
  uint32_t            round;       ///< this launch's round (0: the first word length)
  uint32_t            last_round;  ///< rounds launched - 1: a locus that needs one more is handed back
  LgIter*             iter;        ///< [n_loci]
  uint32_t*           parena;      ///< pseudo reads between rounds (dwords)
  uint64_t            parena_cap;
  unsigned long long* parena_used;
  uint32_t*           next_ids;    ///< contig_big_kernel: the loci of the next round
  uint32_t*           next_count;
  uint32_t*           cyc_ids;     ///< graph_big_kernel: loci whose graph came without a proof of acyclicity (repeat_big_kernel's list)
  uint32_t*           cyc_count;
  uint8_t*            rws;         ///< repeat_big_kernel workspaces (one per wave)
  uint64_t            rws_stride;
  uint8_t*            gws;         ///< graph_big_kernel workspaces (one per workgroup: LGL_GWS_BYTES -- the overflow read sets, the lexicographic ranks)
  unsigned long long* rprof;       ///< [8] repeat_big_kernel: shader clocks by phase, summed over its loci (debug line; nullptr: not kept)
};

static const uint32_t LG_FLAG_NO_PROOF = 1u;  ///< tests / A-B runs: never skip contig_kernel's cycle test
static const uint32_t LG_FLAG_NO_RESCUE = 2u;  ///< A-B runs: reads without an anchor stay roots of their own (readOffsets)

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
WV_DEV unsigned lg8Succ(const FRec8 w, const unsigned i) { return unsigned(w >> (11 * i)) & 0x7ffu; }         // id + 1
WV_DEV unsigned lg8Pred(const FRec8 w, const unsigned i) { return unsigned(w >> (22 + 11 * i)) & 0x7ffu; }    // id + 1
WV_DEV unsigned lg8Cnt(const FRec8 w) { return unsigned(w >> 44) & 15u; }
WV_DEV unsigned lg8Read(const FRec8 w) { return unsigned(w >> 48) & 0x7fu; }
WV_DEV unsigned lg8FirstBase(const FRec8 w) { return unsigned(w >> 55) & 3u; }
WV_DEV unsigned lg8LastBase(const FRec8 w) { return unsigned(w >> 57) & 3u; }
WV_DEV bool     lg8SelfLoop(const FRec8 w) { return (w >> 59) & 1u; }
WV_DEV bool     lg8SOvf(const FRec8 w) { return (w >> 60) & 1u; }
WV_DEV bool     lg8POvf(const FRec8 w) { return (w >> 61) & 1u; }
static const uint64_t LG_M22 = (uint64_t(1) << 22) - 1;
static const uint64_t LG_M44 = (uint64_t(1) << 44) - 1;
/// the (up to four) neighbours of word `nd` as 4 x 11 bits: the two of the record, the rest from the overflow table `ovf`
/// (u16 {word, third, fourth, -}, `nOvf` entries; searched only for the flagged words)
WV_DEV uint64_t lg8Links(const FRec8 w, const unsigned nd, const bool succ, const uint16_t* ovf, const unsigned nOvf)
{
  uint64_t l = (succ ? w : (w >> 22)) & LG_M22;
  if (succ ? lg8SOvf(w) : lg8POvf(w)) {
    for (unsigned e = 0; e < nOvf; ++e)
      if (unsigned(ovf[4 * e]) == nd) l |= (uint64_t(ovf[4 * e + 1]) << 22) | (uint64_t(ovf[4 * e + 2]) << 33);
  }
  return l;
}
WV_DEV unsigned lgLinkId(const uint64_t w, const unsigned c)
{
  const unsigned f = unsigned(w >> (11 * c)) & 0x7ffu;
  return f ? f - 1 : ASM_NONE;
}

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
  FRec8*           nodes;   // records (the sets' bytes, after the bitsets have left for the slab)
  uint64_t*        pred4;   // predecessors by symbol while the links are scattered (4 x 11 bits)
  int16_t*         phi;     // potential of a word (id): offset of its first read + position in it
  int16_t*         roff;    // offset of a read (overlays rdm after the table pass)
  uint32_t*        filter;  // presence filter (maybePresent)
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
    nodes  = reinterpret_cast<FRec8*>(lds + LG_OFF_SETS);
    pred4  = reinterpret_cast<uint64_t*>(lds + LG_OFF_SETS) + LG_SLOTS;
    phi    = reinterpret_cast<int16_t*>(lds + LG_OFF_PHI);
    roff   = reinterpret_cast<int16_t*>(lds + LG_OFF_RDM);
    filter = reinterpret_cast<uint32_t*>(lds + LG_OFF_FILTER);
    sortA  = reinterpret_cast<uint16_t*>(lds + LG_OFF_SORTA);
    sortB  = reinterpret_cast<uint16_t*>(lds + LG_OFF_SORTB);
    keyArr = reinterpret_cast<uint32_t*>(lds + LG_OFF_KEYS);
    slotId = reinterpret_cast<uint16_t*>(lds + LG_OFF_KEYS);
    cntArr = reinterpret_cast<uint8_t*>(lds + LG_OFF_CNT);
    codes  = reinterpret_cast<uint32_t*>(lds + LG_OFF_DYN);
    nmask  = codes;
  }

  /// per-phase shader clocks of the workgroup's first wave (-DMANTA_ASM_PROFILE).  -DMANTA_LG_PROFILE_GRAPH: the eight counters
  /// split graph_kernel alone (`fine`: 0 pack, 1 table, 2 successor links, 3 counts + radix passes, 4 ties + ids, 5 slab + bitsets +
  /// record init, 6 predecessors + siblings, 7 speculation list + slab write); contig_kernel counts nothing then
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

  /// Presence filter over the words' hash tags (16 K bits, built after the table pass): three of a word's four possible successors
  /// usually do not exist, and 93 % of those are answered by one bit test instead of a bucket probe.
  WV_DEV bool maybePresent(const uint32_t h) const
  {
    const unsigned t = (h >> 15) & (LG_FILTER_BITS - 1);
    return (filter[t >> 5] >> (t & 31)) & 1u;
  }
  /// lookupSlot behind the filter
  template <int KW>
  WV_DEV unsigned lookupFiltered(const Key<KW>& key) const
  {
    if (!maybePresent(keyHash(key))) return ASM_NONE;
    return lookupSlot<KW>(key);
  }
  WV_DEV void buildFilter()
  {
    for (unsigned i = tid(); i < LG_FILTER_BITS / 32; i += nThreads()) filter[i] = 0;
    teamSync();
    for (unsigned s = tid(); s < LG_SLOTS; s += nThreads()) {
      const uint32_t v = slots[s];
      if (v != LG_EMPTY) {
        const unsigned t = (v >> 15) & (LG_FILTER_BITS - 1);
        wv::atomic_or(&filter[t >> 5], 1u << (t & 31));
      }
    }
    teamSync();
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
    unsigned       cw = 0, mw = 0, nb = 0;  // code dwords, N-bitmap dwords, bases so far
    uint16_t*      rstart  = reinterpret_cast<uint16_t*>(hdr + 64);
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
      if (tw == 0 && r < nNormal && cwo <= 0x7ffu) {
        rd[r]     = cwo | ((len & 0xffffu) << 11);
        rdm[r]    = uint16_t(mwo);
        rstart[r] = uint16_t(nb + sb - len);
      }
      cw += wv::readlane(sc, 63);
      mw += wv::readlane(sm, 63);
      nb += wv::readlane(sb, 63);
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
    // The locus' bases are one contiguous run of the input arena: every thread fetches 16-byte pieces of it into LDS (the sets'
    // bytes, unused until the table pass) -- all loads in flight at once, one memory latency for the whole pile -- and the
    // conversion below reads its bytes from there.
    char* const stage = lds + LG_OFF_SETS;
    unsigned    lead  = 0;
    {
      const uintptr_t g0 = reinterpret_cast<uintptr_t>(P.bases + P.read_off[rBegin] + shift);
      lead               = unsigned(g0 & 15);
      const u32x4*   gsrc    = reinterpret_cast<const u32x4*>(g0 - lead);
      const unsigned nChunks = (lead + nb + 15) / 16 + 1;  // (+1: the byte funnel reads up to four bytes past a read's last dword; the arena is padded)
      for (unsigned c = tid(); c < nChunks; c += nThreads()) reinterpret_cast<u32x4*>(stage)[c] = gsrc[c];
    }
    teamSync();
    // 8 lanes per read, 8 reads per pass: lane (g, i) converts code dwords i, i+8, ... of read (base + g)
    for (unsigned base = 8 * tw; base < nNormal; base += 8 * tn) {
      const unsigned r = base + (lane >> 3);
      if (r >= nNormal) continue;
      const char*    src = stage + lead + rstart[r];
      const unsigned d = rd[r], cwo = d & 0x7ffu, len = (d >> 11) & 0xffffu, mwo = rdm[r];
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
    if (tid() < LG_MAX_READS) reinterpret_cast<uint32_t*>(lds + LG_OFF_ANCH)[tid()] = LG_NO_ANCHOR;
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
        bool                      haveAnchor = false;  // (see readOffsets)
        for (unsigned j0 = 0; j0 + k <= len; j0 += 64) {
          const unsigned j  = j0 + lane;
          const unsigned pb = cwo * 16 + j;
          bool           todo = (j + k <= len) && !(rdHasN && windowHasN(mwo, j));  // :531
          Key<KW>        key;
          uint32_t       mine = 0;
          unsigned       tag = 0, b = 0, slot = 0, skip = 0, foundPb = 0x8000u;
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
          // One probe round: read the bucket; the first slot (past `skip`) that is empty or carries the tag decides -- an empty
          // slot is claimed with a compare-and-swap, a tagged one has its word fetched and compared, both in the same LDS round
          // trip.  A lane that loses the swap simply goes round again: the bucket then shows who took the slot.
          while (wv::any(todo)) {
            const FBucket bk = *reinterpret_cast<const FBucket*>(slots + 4 * b);
            unsigned      at = 4;
            uint32_t      sv = LG_EMPTY;
            for (int i = 3; i >= 0; --i) {
              const bool hit = unsigned(i) >= skip && (bk.s[i] == LG_EMPTY || (bk.s[i] >> 15) == tag);
              at             = hit ? unsigned(i) : at;
              sv             = hit ? bk.s[i] : sv;
            }
            const bool tryClaim = todo && at < 4 && sv == LG_EMPTY;
            const bool tryMatch = todo && at < 4 && sv != LG_EMPTY;
            uint32_t   casOld   = 0;
            if (tryClaim) casOld = wv::atomic_cas(&slots[4 * b + at], LG_EMPTY, mine);
            const Key<KW> got  = keyAt<KW>(tryMatch ? (sv & 0x7fffu) : 0u);
            const bool    same = keyEq(got, key);
            if (tryClaim) {
              if (casOld == LG_EMPTY) {
                slot = 4 * b + at;
                todo = false;
              }
            } else if (tryMatch && same) {
              slot    = 4 * b + at;
              foundPb = sv & 0x7fffu;
              todo    = false;
            } else if (todo) {
              // another word under the tag (17 bits: rare): next slot of the bucket; a full bucket without the word: next bucket
              skip = tryMatch ? at + 1 : 4u;
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
          if ((j + k <= len) && !fail && !(rdHasN && windowHasN(mwo, j))) wv::atomic_or(setWord + 2 * slot, setBit);
          if (!haveAnchor) {
            // the read's anchor: its first word that another read had brought before (position here, first occurrence there)
            // (only words of EARLIER reads -- their codes lie below this read's -- so that the anchors form a forest: reads are
            // inserted concurrently, and two of them anchoring at each other would leave both without an offset)
            const bool     cand = foundPb < cwo * 16;
            const uint64_t mc   = wv::ballot(cand);
            if (mc) {
              const int      l   = wv::ctz(mc);
              const unsigned apb = wv::readlane(foundPb, l);
              if (lane == 0) reinterpret_cast<uint32_t*>(lds + LG_OFF_ANCH)[r] = apb | ((j0 + unsigned(l)) << 15);
              haveAnchor = true;
            }
          }
        }
      }
    }
    if (wv::any(fail) && lane == 0) wv::atomic_or(&hdr[LG_H_FLAG], 2u);
    teamSync();
    return wv::atomic_load(&hdr[LG_H_FLAG]) == 0;
  }

  // ------------------------------------------------------------------------------------------------
  // the words in seed order (:686-696: count descending, k-mer ascending).  Stable LSD radix sort of the occupied slots over
  // {first 8 bases (two passes), 255 - count}; what still ties (same count, same first 8 bases: a word or two) is ranked by full
  // key compares inside its run.  Result: sortA[id] = slot, slotId[slot] = id, nFat / nEligible / lowTier from the count digits.
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
    const unsigned chunk = (((n + tn - 1) / tn) + 63) & ~63u;  // elements of one wave, in order
    const unsigned c0 = chunk * tw, c1 = (c0 + chunk < n) ? (c0 + chunk) : n;
    uint16_t *     src = sortA, *dst = sortB;
    uint32_t*      myHist = whist + 256 * tw;
    // three passes: bits 16..23 and 24..31 of the first 16 bases, then the count
    for (int pass = 0; pass < 3; ++pass) {
      for (unsigned i = tid(); i < 256 * tn; i += nThreads()) whist[i] = 0;
      teamSync();
      auto digitOf = [&](const unsigned slot) -> unsigned {
        return (pass < 2) ? ((keyArr[slot] >> (16 + 8 * pass)) & 255u) : (255u - cntOf(slot));
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
      if (pass == 2 && tid() == 0) {
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
    // three passes: the sorted list sits in sortB (= src); runs of equal {count, first 8 bases} -> exact order into sortA
    for (unsigned i = tid(); i < n; i += nThreads()) {
      const unsigned slot = src[i];
      const unsigned c = cntOf(slot), p = keyArr[slot] >> 16;
      unsigned       lo = i, hi = i;
      while (lo > 0) {
        const unsigned o = src[lo - 1];
        if (cntOf(o) != c || (keyArr[o] >> 16) != p) break;
        lo--;
      }
      while (hi + 1 < n) {
        const unsigned o = src[hi + 1];
        if (cntOf(o) != c || (keyArr[o] >> 16) != p) break;
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
  // A proof that the graph has no cycle, for free in the common case.  Reads that come from one haplotype and differ by
  // substitutions only agree on a coordinate: read r sits at offset off[r] such that a word at position j of r has the
  // coordinate off[r] + j in every read that holds it.  The table pass notes one ANCHOR per read (its first word that another
  // read had brought before: position here, position there), the offsets follow by pointer doubling over the anchor forest, and
  // a word's potential phi = coordinate of its first occurrence.  If phi rises along EVERY edge the graph is acyclic -- that is
  // checked edge by edge in the links pass, so a wrong or inconsistent offset (indel haplotypes, an anchor cycle between reads
  // inserted at the same time) costs nothing but the proof: contig_kernel then runs its peel.
  // ------------------------------------------------------------------------------------------------
  /// read that owns packed base index pb (the reads' code offsets ascend)
  WV_DEV unsigned readOfPb(const unsigned pb) const
  {
    const unsigned cwd = pb >> 4;
    unsigned       lo = 0, hi = nNormal;  // rd[lo].cwo <= cwd < rd[hi].cwo
    while (hi - lo > 1) {
      const unsigned mid = (lo + hi) >> 1;
      if ((rd[mid] & 0x7ffu) <= cwd) lo = mid; else hi = mid;
    }
    return lo;
  }
  template <int KW>
  WV_DEV void readOffsets()
  {
    if (tw == 0) {
      const uint32_t* anch = reinterpret_cast<const uint32_t*>(lds + LG_OFF_ANCH);
      int32_t*        off  = reinterpret_cast<int32_t*>(lds + LG_OFF_ANCH) + LG_MAX_READS;
      uint8_t*        par  = reinterpret_cast<uint8_t*>(hdr + LG_H_PAR);
      int32_t         myOff[2];
      unsigned        myPar[2], myRoot[2];
      for (unsigned h = 0; h < 2; ++h) {
        const unsigned r = lane + 64 * h;
        myOff[h] = 0;
        myPar[h] = r;
        myRoot[h] = r;
        if (r < nNormal) {
          const uint32_t a = anch[r];
          if (a != LG_NO_ANCHOR) {
            const unsigned apb = a & 0x7fffu, j = a >> 15;
            const unsigned r0  = readOfPb(apb);
            myPar[h] = r0;
            myOff[h] = int32_t(apb - 16u * (rd[r0] & 0x7ffu)) - int32_t(j);
          }
        }
      }
      wv::sync();  // (every anchor has been read: the offsets take the second half of the same array... and rdm's bytes below)
      for (unsigned h = 0; h < 2; ++h) {
        off[lane + 64 * h] = myOff[h];
        par[lane + 64 * h] = uint8_t(myPar[h]);
      }
      wv::sync();
      for (int round = 0; round < 7; ++round) {
        int32_t  po[2];
        unsigned pp[2];
        for (unsigned h = 0; h < 2; ++h) {
          po[h] = off[myPar[h]];
          pp[h] = par[myPar[h]];
        }
        wv::sync();
        for (unsigned h = 0; h < 2; ++h) {
          const unsigned r = lane + 64 * h;
          if (myPar[h] != r) {  // (a root keeps its offset)
            myOff[h] += po[h];
            if (pp[h] == myPar[h]) {  // the parent is a root: done after this addition
              off[r] = myOff[h];
              par[r] = uint8_t(r);
              myRoot[h] = myPar[h];
              myPar[h] = r;
            } else {
              off[r]   = myOff[h];
              par[r]   = uint8_t(pp[h]);
              myPar[h] = pp[h];
            }
          }
        }
        wv::sync();
      }
#ifdef MANTA_WAVE_EMU
      if (std::getenv("MANTA_EMU_PROOF_TRACE"))
        for (unsigned h = 0; h < 2; ++h)
          if (lane + 64 * h < nNormal) std::fprintf(stderr, "  read %u: anchor %08x off %d par %u\n", lane + 64 * h, anch[lane + 64 * h], myOff[h], myPar[h]);
#endif
      // (myRoot is the read a read's offset was completed through: the root for the root's children, a finished inner read for the reads
      // below it.  The roots proper by pointer jumping -- the trees below are told apart by them)
      {
        uint8_t* rootTmp = reinterpret_cast<uint8_t*>(lds + LG_OFF_WHIST);  // (the histograms are not in use before the sort)
        for (unsigned h = 0; h < 2; ++h) rootTmp[lane + 64 * h] = uint8_t(myRoot[h]);
        wv::sync();
        for (int round = 0; round < 7; ++round) {
          unsigned up[2];
          for (unsigned h = 0; h < 2; ++h) up[h] = rootTmp[rootTmp[lane + 64 * h]];
          wv::sync();
          for (unsigned h = 0; h < 2; ++h) rootTmp[lane + 64 * h] = uint8_t(up[h]);
          wv::sync();
        }
        for (unsigned h = 0; h < 2; ++h) myRoot[h] = rootTmp[lane + 64 * h];
      }
      // ---- second chance.  A read whose words were all new when it went in -- the reads of a step are inserted by eight waves at
      // once, so which of two overlapping reads "was first" is a race -- has no anchor and is the root of a tree of its own, at
      // offset 0: one such tree beside the main one and the proof is lost.  Now that every word is in the table, such a root
      // looks its words up again and ties itself (and its tree) to the first word that a read of the MAIN tree brought.  Any
      // function phi proves acyclicity if it rises by one along every edge, so nothing here can make the proof unsound.
      if (!(G.flags & LG_FLAG_NO_RESCUE)) {
        uint8_t* rootOf = reinterpret_cast<uint8_t*>(lds + LG_OFF_WHIST);  // (the histograms are not in use before the sort)
        for (unsigned h = 0; h < 2; ++h) rootOf[lane + 64 * h] = uint8_t(myRoot[h]);
        wv::sync();
        // the main tree: the root with the most reads
        unsigned mainRoot = 0, mainSize = 0;
        for (unsigned h = 0; h < 2; ++h) {
          uint64_t roots = wv::ballot(lane + 64 * h < nNormal && myRoot[h] == lane + 64 * h);
          while (roots) {
            const unsigned c  = unsigned(wv::ctz(roots)) + 64 * h;
            roots &= roots - 1;
            const unsigned sz = unsigned(wv::popc(wv::ballot(lane < nNormal && myRoot[0] == c))) +
                                unsigned(wv::popc(wv::ballot(lane + 64 < nNormal && myRoot[1] == c)));
            if (sz > mainSize) {
              mainSize = sz;
              mainRoot = c;
            }
          }
        }
        for (unsigned round = 0; round < 3; ++round) {
          bool any = false;
          for (unsigned h = 0; h < 2; ++h) {
            uint64_t roots = wv::ballot(lane + 64 * h < nNormal && myRoot[h] == lane + 64 * h && lane + 64 * h != mainRoot);
            while (roots) {
              const unsigned s = unsigned(wv::ctz(roots)) + 64 * h;
              roots &= roots - 1;
              // A word that both trees hold ties them: it was brought by a read of one tree (its first occurrence) and sits in a read of
              // the other.  Which side brought the shared words is the same race -- a tree whose reads were always first holds no
              // word of the other's -- so both directions are tried: reads of s against words of the main tree, then reads of the main
              // tree against words of s; every second position of up to sixteen reads each (the words one tree brought come in runs;
              // reads at the far end of a tree may not reach the other tree at all, the ones in the middle do).
              bool    tied  = false;
              int32_t delta = 0;
              for (unsigned dir = 0; dir < 2 && !tied; ++dir) {
                const unsigned fromRoot = dir ? mainRoot : s, toRoot = dir ? s : mainRoot;
                uint64_t mem0 = wv::ballot(lane < nNormal && myRoot[0] == fromRoot), mem1 = wv::ballot(lane + 64 < nNormal && myRoot[1] == fromRoot);
                for (unsigned tries = 0; tries < 16 && !tied && (mem0 | mem1); ++tries) {
                  unsigned x;
                  if (mem0) {
                    x = unsigned(wv::ctz(mem0));
                    mem0 &= mem0 - 1;
                  } else {
                    x = 64 + unsigned(wv::ctz(mem1));
                    mem1 &= mem1 - 1;
                  }
                  const unsigned d = rd[x], cwo = d & 0x7ffu, len = (d >> 11) & 0xffffu, mwo = rdm[x];
                  const bool     rdHasN = (d >> 27) & 1u;
                  const int32_t  offX   = off[x];
                  for (unsigned j0 = 0; j0 + k <= len && !tied; j0 += 128) {
                    const unsigned j     = j0 + 2 * lane;
                    const bool     valid = (j + k <= len) && !(rdHasN && windowHasN(mwo, j));
                    unsigned       slot  = ASM_NONE;
                    if (valid) slot = lookupSlot<KW>(keyAt<KW>(cwo * 16 + j));
                    bool    ok   = false;
                    int32_t mine = 0;
                    if (slot != ASM_NONE) {
                      const unsigned fpb = slots[slot] & 0x7fffu;
                      const unsigned o   = readOfPb(fpb);
                      if (rootOf[o] == toRoot) {
                        ok = true;
                        const int32_t there = off[o] + int32_t(fpb - 16u * (rd[o] & 0x7ffu)), here = offX + int32_t(j);
                        mine = dir ? (here - there) : (there - here);  // the word's coordinate in the main tree minus its coordinate in s
                      }
                    }
                    const uint64_t m = wv::ballot(ok);
                    if (m) {
                      tied  = true;
                      delta = wv::readlane(mine, int(wv::ctz(m)));
                    }
                  }
                }
              }
#ifdef MANTA_WAVE_EMU
              if (std::getenv("MANTA_EMU_PROOF_TRACE") && lane == 0)
                std::fprintf(stderr, "  rescue: tree of read %u (main tree: read %u with %u reads): %s, delta %d\n", s, mainRoot, mainSize, tied ? "tied" : "no link", int(delta));
#endif
              if (!tied) continue;
              any = true;
              for (unsigned g = 0; g < 2; ++g)  // the whole tree of s moves
                if (myRoot[g] == s) {
                  myOff[g] += delta;
                  myRoot[g] = mainRoot;
                  off[lane + 64 * g]    = myOff[g];
                  rootOf[lane + 64 * g] = uint8_t(mainRoot);
                }
              if (lane == 0 && G.stats) wv::atomic_add(&G.stats[1], 1u);
              wv::sync();
            }
          }
          if (!any) break;
        }
      }
      // (rdm is dead after the table pass: its bytes take the offsets; one that does not fit 16 bits only loses the proof)
      for (unsigned h = 0; h < 2; ++h) {
        const int32_t v = myOff[h];
        roff[lane + 64 * h] = int16_t((v > 30000) ? 30000 : ((v < -30000) ? -30000 : v));
      }
    }
    teamSync();
  }

  // ------------------------------------------------------------------------------------------------
  // node records in the new numbering (FRec8), links (8 table lookups per word), side tables
  // ------------------------------------------------------------------------------------------------
  template <int KW>
  WV_DEV bool buildRecords(uint8_t* slab, const LgSlab& SL)
  {
    // bitsets of the words with more than one read: ids below nFat, straight into the slab; first occurrences likewise
    FSet*     gPool = reinterpret_cast<FSet*>(slab + SL.pool);
    uint16_t* gPb   = reinterpret_cast<uint16_t*>(slab + SL.pb);
    for (unsigned i = tid(); i < nFat; i += nThreads()) gPool[i] = sets[sortA[i]];
    buildFilter();  // (its barriers also end the sets' life: the records take their place)
    for (unsigned i = tid(); i < nNodes; i += nThreads()) {
      const unsigned slot = sortA[i], pb = slots[slot] & 0x7fffu, enc = cntArr[slot];
      const unsigned cnt = (enc & 0x80u) ? 1u : enc;
      const unsigned sup = (enc & 0x80u) ? (enc & 0x7fu) : 0u;
      const Key<KW>  key = keyAt<KW>(pb);
      unsigned       last = 0;
      for (int w = 0; w < KW; ++w)
        if (unsigned(w) == ((k - 1) >> 4)) last = (key.w[w] >> (30 - 2 * ((k - 1) & 15))) & 3u;
      nodes[i] = (uint64_t(cnt > 15 ? 15 : cnt) << 44) | (uint64_t(sup) << 48) | (uint64_t(key.w[0] >> 30) << 55) | (uint64_t(last) << 57);
      pred4[i] = 0;
      gPb[i]   = uint16_t(pb);
      const unsigned r = readOfPb(pb);
      phi[i]           = int16_t(int(roff[r]) + int(pb - 16u * (rd[r] & 0x7ffu)));
    }
    if (tid() == 0) {
      hdr[LG_H_NSIB]  = 0;
      hdr[LG_H_NSOVF] = 0;
      hdr[LG_H_NPOVF] = 0;
      hdr[LG_H_CYC]   = 0;
    }
    teamSync();
    tick(2, 5);
    // successor lookups, predecessor scatter (by symbol position; packed below), the potential along every edge
    uint16_t* sovf = reinterpret_cast<uint16_t*>(lds + LG_OFF_SOVF);
    uint16_t* povf = reinterpret_cast<uint16_t*>(lds + LG_OFF_POVF);
    bool      against = false;
    for (unsigned nb = 64 * tw; nb < nNodes; nb += 64 * tn) {
      const unsigned nd = nb + lane;
      if (nd < nNodes) {
        const unsigned pb   = slots[sortA[nd]] & 0x7fffu;
        const Key<KW>  key  = keyAt<KW>(pb);
        const unsigned firstBase = key.w[0] >> 30;
        const int      myPhi = phi[nd];
        unsigned       found[4];
        unsigned       m = 0;
        bool           selfLoop = false;
        unsigned       sslot[4];
        for (unsigned c = 0; c < 4; ++c) sslot[c] = lookupFiltered<KW>(keyShiftAppend<KW>(key, c));
        for (unsigned c = 0; c < 4; ++c) {
          const unsigned ss = sslot[c];
          if (ss == ASM_NONE) continue;
          const unsigned s = slotId[ss];
          found[m++]       = s + 1;
          if (s == nd) {
            selfLoop = true;
          } else if (int(phi[s]) <= myPhi) {
            against = true;
#ifdef MANTA_WAVE_EMU
            if (std::getenv("MANTA_EMU_PROOF_TRACE")) {
              const unsigned pbs = slots[sortA[s]] & 0x7fffu;
              std::fprintf(stderr, "  against: word %u (read %u pos %u phi %d) -> word %u (read %u pos %u phi %d)\n", nd, readOfPb(pb), pb - 16 * (rd[readOfPb(pb)] & 0x7ffu), myPhi,
                           s, readOfPb(pbs), pbs - 16 * (rd[readOfPb(pbs)] & 0x7ffu), int(phi[s]));
            }
#endif
          }
          wv::atomic_or(reinterpret_cast<unsigned long long*>(&pred4[s]), (unsigned long long)(uint64_t(nd + 1) << (11 * firstBase)));
        }
        uint64_t w = nodes[nd];
        if (m > 0) w |= uint64_t(found[0]);
        if (m > 1) w |= uint64_t(found[1]) << 11;
        if (selfLoop) w |= uint64_t(1) << 59;
        if (m > 2) {
          w |= uint64_t(1) << 60;
          const unsigned at = wv::atomic_add(&hdr[LG_H_NSOVF], 1u);
          if (at < LG_OVF_CAP) {
            sovf[4 * at + 0] = uint16_t(nd);
            sovf[4 * at + 1] = uint16_t(found[2]);
            sovf[4 * at + 2] = uint16_t(m > 3 ? found[3] : 0u);
            sovf[4 * at + 3] = 0;
          }
        }
        nodes[nd] = w;
      }
    }
    if (wv::any(against) && lane == 0) wv::atomic_or(&hdr[LG_H_CYC], 1u);
    teamSync();
    tick(2, 2);
    // predecessors: the first two into the record, further ones into the overflow table; words without a predecessor: their
    // siblings (the words that differ in the last base only, :185-210) cannot be found through a predecessor's successor
    // list -> side table
    uint16_t* sib = reinterpret_cast<uint16_t*>(lds + LG_OFF_SIB);
    for (unsigned nd = tid(); nd < nNodes; nd += nThreads()) {
      const uint64_t p4 = pred4[nd];
      unsigned       pf[4];
      unsigned       m = 0;
      for (unsigned c = 0; c < 4; ++c) {
        const unsigned f = unsigned(p4 >> (11 * c)) & 0x7ffu;
        if (f) pf[m++] = f;
      }
      uint64_t w = nodes[nd];
      if (m > 0) w |= uint64_t(pf[0]) << 22;
      if (m > 1) w |= uint64_t(pf[1]) << 33;
      if (m > 2) {
        w |= uint64_t(1) << 61;
        const unsigned at = wv::atomic_add(&hdr[LG_H_NPOVF], 1u);
        if (at < LG_OVF_CAP) {
          povf[4 * at + 0] = uint16_t(nd);
          povf[4 * at + 1] = uint16_t(pf[2]);
          povf[4 * at + 2] = uint16_t(m > 3 ? pf[3] : 0u);
          povf[4 * at + 3] = 0;
        }
      }
      nodes[nd] = w;
      if (m == 0) {
        const unsigned pb  = slots[sortA[nd]] & 0x7fffu;
        const Key<KW>  key = keyAt<KW>(pb);
        const unsigned lastBase = lg8LastBase(w);
        unsigned       found[3] = {LG_NO_SLOT, LG_NO_SLOT, LG_NO_SLOT};
        unsigned       nf = 0;
        for (unsigned c = 0; c < 4; ++c) {
          if (c == lastBase) continue;
          Key<KW> s2 = key;
          keySetBase(s2, k - 1, c);
          const unsigned ss = lookupFiltered<KW>(s2);
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
    if (wv::atomic_load(&hdr[LG_H_NSIB]) > LG_SIB_CAP || wv::atomic_load(&hdr[LG_H_NSOVF]) > LG_OVF_CAP ||
        wv::atomic_load(&hdr[LG_H_NPOVF]) > LG_OVF_CAP)
      return false;
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
    uint32_t* dupW  = reinterpret_cast<uint32_t*>(dist + 128);  // [4] duplicate bits of the 128 entries
    const unsigned e0 = lowTier, e1 = (nEligible < lowTier + 128) ? nEligible : (lowTier + 128);
    const unsigned nE = (e1 > e0) ? (e1 - e0) : 0u;
    // a word's only successor / number of predecessors, self loops aside (a word with an overflow entry has three or more)
    auto outOnly = [&](const FRec8 w, const unsigned nd, unsigned& od) -> unsigned {
      unsigned only = ASM_NONE;
      od            = lg8SOvf(w) ? 3u : 0u;
      for (unsigned c = 0; c < 2; ++c) {
        const unsigned f = lg8Succ(w, c);
        if (f && f - 1 != nd) {
          od++;
          only = f - 1;
        }
      }
      return only;
    };
    auto inDeg = [&](const FRec8 w, const unsigned nd) -> unsigned {
      unsigned id = lg8POvf(w) ? 3u : 0u;
      for (unsigned c = 0; c < 2; ++c) {
        const unsigned f = lg8Pred(w, c);
        if (f && f - 1 != nd) id++;
      }
      return id;
    };
    if (tid() < 4) dupW[tid()] = 0;
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
    // entry e is dropped if an earlier entry of its stretch lies upstream (more steps to the stretch's end); the earlier
    // entries are dealt out to the waves
    if (nE > 0) {
      unsigned eL[2], eD[2];
      bool     dup[2] = {false, false};
      for (unsigned h = 0; h < 2; ++h) {
        const unsigned i = lane + 64 * h;
        eL[h]            = (i < nE) ? unsigned(label[i]) : ASM_NONE;
        eD[h]            = (i < nE) ? unsigned(dist[i]) : 0u;
      }
      for (unsigned j = tw; j < nE; j += tn) {
        const unsigned lj = label[j], dj = dist[j];
        for (unsigned h = 0; h < 2; ++h)
          if (j < lane + 64 * h && lj == eL[h] && dj > eD[h]) dup[h] = true;
      }
      for (unsigned h = 0; h < 2; ++h) {
        const uint64_t m = wv::ballot(dup[h]);
        if (lane == 0 && m) {
          wv::atomic_or(&dupW[2 * h], uint32_t(m));
          wv::atomic_or(&dupW[2 * h + 1], uint32_t(m >> 32));
        }
      }
    }
    teamSync();
    unsigned n0 = 0;
    if (tw == 0) {
      if (nEligible > 0) {
        n0 = 1;
        for (unsigned h = 0; h < 2; ++h) {
          const unsigned i    = lane + 64 * h;
          const bool     dup  = (dupW[2 * h + (lane >> 5)] >> (lane & 31)) & 1u;
          const bool     keep = (i < nE) && !dup && (e0 + i) != 0u;
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
    readOffsets<KW>();
    tick(1, 1);
    if (!sortWords<KW>()) return false;
    // slab for this locus
    const LgSlab   SL    = lgSlab(nNodes, nFat, codeWords);
    const uint64_t bytes = SL.total;
    if (tid() == 0) {
      const unsigned long long off = wv::atomic_add(G.arena_used, (unsigned long long)bytes);
      hdr[LG_H_OFF_LO] = uint32_t(off);
      hdr[LG_H_OFF_HI] = uint32_t(off >> 32);
    }
    teamSync();
    const uint64_t off = (uint64_t(wv::atomic_load(&hdr[LG_H_OFF_HI])) << 32) | wv::atomic_load(&hdr[LG_H_OFF_LO]);
    if (off + bytes > G.arena_cap) return false;
    uint8_t* slab = G.arena + off;
    if (!buildRecords<KW>(slab, SL)) return false;
    const bool     acyclic = wv::atomic_load(&hdr[LG_H_CYC]) == 0 && !(G.flags & LG_FLAG_NO_PROOF);
    if (acyclic && tid() == 0 && G.stats) wv::atomic_add(&G.stats[0], 1u);
    const unsigned need    = ckNeed(nNodes, nFat, acyclic);
    unsigned       cls     = LG_CLASSES;
    for (unsigned c = LG_CLASSES; c-- > 0;)
      if (G.class_bytes[c] && need <= G.class_bytes[c]) cls = c;
    if (cls == LG_CLASSES) return false;
    uint16_t*      gSpec = reinterpret_cast<uint16_t*>(slab + SL.spec);
    const unsigned nSpec = speculationList(gSpec);
    // the rest of the slab
    FRec8* gRec = reinterpret_cast<FRec8*>(slab + SL.recs);
    for (unsigned i = tid(); i < nNodes; i += nThreads()) gRec[i] = nodes[i];
    const unsigned nSib = wv::atomic_load(&hdr[LG_H_NSIB]), nSovf = wv::atomic_load(&hdr[LG_H_NSOVF]), nPovf = wv::atomic_load(&hdr[LG_H_NPOVF]);
    {
      // the three side tables lie back to back in LDS and in the slab (fixed capacities)
      const uint16_t* tsrc = reinterpret_cast<const uint16_t*>(lds + LG_OFF_SIB);
      uint16_t*       tdst = reinterpret_cast<uint16_t*>(slab + SL.sib);
      for (unsigned i = tid(); i < 4 * (LG_SIB_CAP + 2 * LG_OVF_CAP); i += nThreads()) tdst[i] = tsrc[i];
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
      h.nPseudo = h.cyclic = h.nCore = 0;
      *reinterpret_cast<LgHdr*>(slab) = h;
      G.slab_off[locus]               = off;
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
    k     = minWL;
    tMark = wv::clock();
    if (!pack(locus)) return false;
    tick(0, 0);
    const unsigned kw = (k + 15) >> 4;
    if (kw > unsigned(MAXKW)) return false;
    if (kw <= 2) return runK<2>(locus);
    if (MAXKW >= 4 && kw <= 4) return runK<(MAXKW >= 4 ? 4 : 2)>(locus);
    return runK<MAXKW>(locus);
  }
};

/// persistent workgroups of LG_WAVES wavefronts, LG_BUDGET bytes of dynamic LDS each; params as assemble_kernel plus the
/// pipeline's own.  Loci this path does not cover are appended to P.punt_ids (P.punt_count counts them).
/// One instantiation per key width (word lengths up to 32 / 64 / 128): the register allocation of a kernel is that of its widest
/// path, and the 32-dword keys of the longest words would cost the ordinary word lengths their spill-free build.
#ifndef MANTA_LG_WAVES_PER_SIMD
#define MANTA_LG_WAVES_PER_SIMD 4
#endif
template <int MAXKW>
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
      ok = g.template run<MAXKW>(locus);
    }
    wv::sync();
    if (tw == 0 && !ok && wv::lane() == 0) P.punt_ids[wv::atomic_add(P.punt_count, 1u)] = locus;
    wv::sync();
    wv::wg_barrier();  // (the slot word is rewritten next)
  }
}

#if MANTA_TU != MANTA_TU_ALL
#if MANTA_TU == MANTA_TU_GRAPH
#define MANTA_X
#else
#define MANTA_X extern
#endif
MANTA_X template __global__ void graph_kernel<2>(const LgArgs);
MANTA_X template __global__ void graph_kernel<4>(const LgArgs);
MANTA_X template __global__ void graph_kernel<8>(const LgArgs);
#undef MANTA_X
#endif

}  // namespace manta_dev

#include "asm_contig.hpp"
