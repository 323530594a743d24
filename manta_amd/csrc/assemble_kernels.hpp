// HIP kernel (gfx950) for Manta's iterative de Bruijn contig assembler
//   runIterativeAssembler            assembly/IterativeAssembler.cpp:844-931
//     buildContigs                   :644-720   (getKmerCounts :506-550, getRepeatKmers :627-642, walk :149-501)
//     selectContigs                  :722-842
// (paths relative to /root/reference/src/c++/lib).
//
// Mapping to the hardware: ONE 64-lane wavefront per candidate locus, pulled from an atomic work queue.
// The reference's three string-keyed hash maps and its std::set<unsigned> algebra become
//   * a 2-bit packed read pile (16 bases / dword, MSB first so dword order == lexicographic order) + N bitmap,
//   * an open-addressing table of distinct k-mers keyed by "packed base index of the first occurrence"
//     (lanes insert all k-mer instances of a read in parallel, one atomicCAS each),
//   * dense node arrays: read-support BITSETS (W = ceil((reads+pseudo)/64) qwords, lane w owns word w),
//     counts, and precomputed successor / predecessor links (8 table lookups per node, lanes over nodes), so
//     that the serial greedy walk touches no keys at all: every extension step is 2 link loads + <=7 bitset
//     loads + a handful of and/or/popcount lane ops.
// Everything between the packed input and the selected contigs stays on the device.
//
// Repeat k-mers: homopolymer words are found from self links.  Whether the k-mer graph has any other cycle is
// decided by a wave-parallel in-degree peel (Kahn); if it has none (the normal case for k >= 25) the
// reference's DFS cannot mark anything else.  For cyclic graphs the reference's result depends on libstdc++'s
// unordered_map iteration order (SURVEY.md hard part 1); that order is re-derived on the device
// (repeat_exact.hpp) and the same DFS is replayed by one lane.
#pragma once
#include "wave.hpp"

namespace manta_dev {

enum {
  ASM_OK               = 0,
  ASM_E_ALPHABET       = 1,  // reads hold bytes outside {A,C,G,T,N} AND the result may depend on the words that contain them
  ASM_E_TABLE_FULL     = 2,  // more k-mer instances / nodes than the workspace was sized for
  ASM_E_CONTIG_TOO_LONG = 3,
  ASM_E_OUT_CAPACITY   = 4,  // output arenas exhausted
  ASM_E_WORD_TOO_LONG  = 5,  // word length > 128
  ASM_E_TOO_MANY_READS = 6,
  ASM_E_INTERNAL       = 7
};

static const unsigned ASM_FLAG_SERIAL_WALK = 1u;  // debug/A-B knob: build contigs one at a time even for small read sets
static const unsigned ASM_NONE     = 0xffffffffu;
static const int      ASM_MAX_KW   = 8;    // k <= 128
static const unsigned ASM_MAX_W    = 16;   // <= 1024 reads incl. pseudo reads
static const unsigned ASM_MAX_CAND = 64;   // 2*maxAssemblyCount must not exceed this
static const unsigned TENT_CAP     = 256;  // tentative-seed list capacity of one speculative round
#ifndef MANTA_ASM_LDS
#define MANTA_ASM_LDS 10240
#endif
static const unsigned ASM_LDS_BYTES = MANTA_ASM_LDS;
// first table size of a locus and word length: instances >> this, rounded up to a power of two (doubles when more than 70 % full).
// Measured per 10 000 config-2 loci: >> 0: 11.17 ms, >> 1: 11.10, >> 2: 11.40 (collisions)
#ifndef MANTA_ASM_TABLE_SHIFT
#define MANTA_ASM_TABLE_SHIFT 1
#endif
#ifndef MANTA_ASM_STRETCH
#define MANTA_ASM_STRETCH 96
#endif
static const unsigned ASM_STRETCH_MAX = MANTA_ASM_STRETCH;  // frontier size up to which the cycle check tries stretch peels  // dynamic LDS per wavefront (visited bitmaps of a speculative round)
static const unsigned WQ_MAX       = 4;    // lane-private walks hold read sets of up to 4 qwords (<= 256 reads) in registers

struct AsmOptsDev {
  uint32_t minWordLength, maxWordLength, wordStepSize, minCoverage, minConservativeCoverage, minUnusedReads,
      minSupportReads, maxAssemblyCount;
};

struct AsmContigOut {
  uint64_t seq_off;   ///< into seq_arena
  uint64_t bits_off;  ///< into bits_arena: W words support, then W words reject
  uint32_t seq_len;
  int32_t  cons_begin, cons_end;
  uint32_t reserved;
};

struct AsmLocusOut {
  int32_t  status;
  uint32_t n_contigs;
  uint32_t n_words;        ///< W: qwords per bitset
  uint32_t n_pseudo;       ///< pseudo reads left appended to `reads` on return
  uint64_t pseudo_off;     ///< into seq_arena: n_pseudo sequences back to back
  uint64_t pseudo_len_off; ///< into bits_arena: n_pseudo lengths (one qword each)
  uint32_t final_word_length;
  uint32_t n_iterations;
  uint32_t cyclic_iterations;  ///< iterations that needed the exact (order-emulating) repeat search
  uint32_t reserved;
};

struct AsmParams {
  const uint8_t*      bases;
  const uint64_t*     read_off;          ///< n_reads_total + 1
  const uint32_t*     locus_read_begin;  ///< n_loci + 1
  uint32_t            n_loci;
  AsmOptsDev          opt;
  uint32_t*           counter;
  uint8_t*            ws;
  uint64_t            ws_stride;
  // workspace capacities (identical for every workgroup)
  uint32_t cap_slots;       ///< power of two
  uint32_t cap_nodes;
  uint32_t cap_words;       ///< packed code dwords (16 bases each) incl. pseudo reads and padding
  uint32_t cap_reads;       ///< normal + pseudo
  uint32_t max_contig_len;
  uint32_t w_max;           ///< bitset qwords the workspace is sized for
  // outputs
  AsmLocusOut*        loci;
  AsmContigOut*       contigs;  ///< n_loci * opt.maxAssemblyCount
  uint8_t*            seq_arena;
  uint64_t            seq_cap;
  unsigned long long* seq_used;
  uint64_t*           bits_arena;
  uint64_t            bits_cap;
  unsigned long long* bits_used;
  // libstdc++ bucket growth schedule for the exact repeat search (host records it from the live library)
  unsigned long long* phase_cycles;  ///< optional [8]: pack, table, links, cycle-check, exact, seed, walk, select (summed over loci)
  const uint32_t* growth_size;     ///< map.size() right before the insertion that rehashes
  const uint32_t* growth_buckets;  ///< bucket count after it
  uint32_t        n_growth;
  uint32_t        flags;  ///< ASM_FLAG_*
  // optional per-locus word lengths (mixed-k batches); nullptr = opt.minWordLength / maxWordLength
  const uint32_t* locus_min_wl;
  const uint32_t* locus_max_wl;
  // optional locus list of this launch (the kernel then works on loci[locus_ids[i]], i < n_loci); nullptr = identity
  const uint32_t* locus_ids;
  // optional packed read piles (manta_packed_piles_t): when pl_codes is set, bases / read_off are unused
  // and stage 0 is a copy -- the piles arrive in the layout this kernel packs into (2-bit codes MSB first + N bitmap)
  const uint32_t* pl_codes;
  const uint32_t* pl_nmask;
  const uint32_t* pl_read_len;
  const uint64_t* pl_code_off;  ///< per read, dwords
  const uint64_t* pl_mask_off;
  // optional streamed upload (whole-batch calls): the kernel is launched while the read bases are still arriving.  The
  // upload goes chunk by chunk (chunk_loci loci each, every chunk at a 256-byte aligned device offset so that no cache
  // line is shared between chunks); *upload_chunks_done (fine-grained device memory, bumped by a 4-byte copy queued behind
  // every chunk on the copy stream) counts the chunks that have landed.  chunk_shift[c] = device offset - caller offset of chunk c.
  const uint32_t* upload_chunks_done;
  const uint32_t* chunk_shift;
  uint32_t        chunk_loci;
  uint32_t        reserved2;
  // streamed upload of packed piles: per chunk three shifts {read index, code dwords, mask dwords} (device position minus the
  // caller's position, modulo 2^64): every chunk's slices of the five pile arrays start on their own cache lines
  const uint64_t* pl_chunk_shift;
  // small_assemble_kernel only (small_asm.hpp): SmallAssemblerOptions::minSeedReads / maxAssemblyIterations
  uint32_t        small_min_seed_reads;
  uint32_t        small_max_iterations;
  // The LDS pipeline (graph_kernel / contig_kernel, asm_lds.hpp) hands the loci it does not cover to the general kernel through
  // device memory: it appends their ids to punt_ids and counts them in *punt_count; the general kernel launched behind it takes
  // locus_ids = punt_ids and reads its number of loci from *n_loci_dev (nullptr: n_loci) -- no host round trip in between
  uint32_t*       punt_ids;
  uint32_t*       punt_count;
  const uint32_t* n_loci_dev;
  /// dynamic LDS per wave of assemble_kernel (ASM_LDS_BYTES; a launch may be given less): peel state / visited bitmaps that do
  /// not fit go to the slab
  uint32_t        lds_bytes;
  /// assemble_kernel takes no further locus once the shared work counter has reached this value (0: no limit; unused since the
  /// LDS pipeline and this kernel no longer share a queue)
  uint32_t        stop_before;
};



// --------------------------------------------------------------------------------------------------
// per-workgroup workspace carve (host and device agree through asmWorkspaceLayout)
// --------------------------------------------------------------------------------------------------
struct AsmWsLayout {
  uint64_t codes, nmask, rd_cw, rd_mw, rd_len, rd_hasn, slots, node_key, node_cnt, node_flag, node_aux, rec, links,
      frontier, cand_seq, cand_bits, cand_meta, walk_left, walk_right, pseudo_seq, pseudo_len, exact, node_k32, tent, lane_seq,
      lane_bits, lane_meta, lane_vis, unused, total;
};

WV_HD uint64_t asmAlign16(uint64_t v)
{
  return (v + 15) & ~uint64_t(15);
}

/// bytes of one node record, the unit the contig walks fetch:
///   [ 0,16) successors  : four 21-bit node ids + the word's count (11 bits), see packLinks
///   [16,32) predecessors: same packing
///   [32, ..) read-support bitset, W qwords
/// -> exactly one 64-byte line for W <= 4 (up to 256 reads incl. pseudo reads)
WV_HD uint32_t asmRecStride(const uint32_t W)
{
  const uint32_t b = 32 + 8 * W;
  return (b < 64) ? 64u : ((b + 15u) & ~15u);
}

static const uint32_t LINK_NONE21 = 0x1fffffu;  // packed form of ASM_NONE; node ids must stay below it
static const uint32_t LINK_CNT_MAX = 0x7ffu;    // counts are stored saturated; 0x7ff means "read node_cnt"

WV_HD uint64_t asmPut(uint64_t& cursor, const uint64_t bytes)
{
  const uint64_t at = cursor;
  cursor            = asmAlign16(cursor + bytes);
  return at;
}

WV_HD AsmWsLayout asmWorkspaceLayout(
    const uint32_t cap_slots, const uint32_t cap_nodes, const uint32_t cap_words, const uint32_t cap_reads,
    const uint32_t max_contig_len, const uint32_t w_max, const uint32_t maxAssemblyCount)
{
  AsmWsLayout    L;
  uint64_t       o     = 0;
  const uint64_t nCand = 2ull * maxAssemblyCount;
  L.codes      = asmPut(o, 4ull * (cap_words + 2));
  L.nmask      = asmPut(o, 4ull * (cap_words / 2 + cap_reads + 8));
  L.rd_cw      = asmPut(o, 4ull * (cap_reads + 1));
  L.rd_mw      = asmPut(o, 4ull * (cap_reads + 1));
  L.rd_len     = asmPut(o, 4ull * (cap_reads + 1));
  L.rd_hasn    = asmPut(o, 4ull * (cap_reads + 1));
  L.slots      = asmPut(o, 8ull * cap_slots);  // {packed base index of the word's first occurrence, node id} per slot
  L.node_key   = asmPut(o, 4ull * cap_nodes);
  L.node_cnt   = asmPut(o, 4ull * cap_nodes);
  L.node_flag  = asmPut(o, 4ull * cap_nodes);
  L.node_aux   = asmPut(o, 4ull * cap_nodes);
  L.rec        = asmPut(o, uint64_t(cap_nodes) * asmRecStride(w_max));
  L.links      = asmPut(o, 32ull * cap_nodes);
  L.frontier   = asmPut(o, 8ull * cap_nodes + 256);
  L.cand_seq   = asmPut(o, nCand * max_contig_len);
  L.cand_bits  = asmPut(o, nCand * 2 * 8ull * w_max);
  L.cand_meta  = asmPut(o, nCand * 16);
  L.walk_left  = asmPut(o, max_contig_len);
  L.walk_right = asmPut(o, max_contig_len);
  L.pseudo_seq = asmPut(o, nCand * max_contig_len);
  L.pseudo_len = asmPut(o, nCand * 4);
  L.exact      = asmPut(o, 4ull * (14ull * cap_nodes + 256));
  L.node_k32   = asmPut(o, 4ull * cap_nodes);
  L.tent       = asmPut(o, 4ull * 2 * TENT_CAP);
  L.lane_seq   = asmPut(o, 64ull * 2 * 4 * (max_contig_len / 16 + 2));
  L.lane_bits  = asmPut(o, 64ull * 2 * WQ_MAX * 8);
  L.lane_meta  = asmPut(o, 64ull * 8 * 4);
  L.lane_vis   = asmPut(o, 64ull * 4 * ((cap_nodes + 31) / 32));
  L.unused     = asmPut(o, 4ull * ((cap_nodes + 31) / 32 + 2));  // "unusedWords" (:678-682) as a bitmap over node ids
  L.total      = asmAlign16(o);
  return L;
}

// node_flag bits
static const unsigned NF_REPEAT = 2u;   // member of repeatWords
static const unsigned NF_JUNK = 1u;     // byte-generic form: the word holds a byte outside the alphabet.  Such a word cannot lie on a
                                        // cycle (its odd byte only moves towards the front and out as symbols are appended), and its
                                        // links are not symmetric (the walks prepend / append A,C,G,T only): the cycle test leaves it out
// bits 8.. : serial of the last contig that used the word ("wordsInContig", :182)

WV_DEV unsigned baseCode(const uint8_t c)
{
  return (c == 'A') ? 0u : (c == 'C') ? 1u : (c == 'G') ? 2u : (c == 'T') ? 3u : (c == 'N') ? 4u : 5u;
}

WV_DEV uint32_t hashMix(uint32_t h, const uint32_t v)
{
  h ^= v;
  h *= 0x9E3779B1u;
  h ^= h >> 15;
  return h;
}

/// 16-byte vector for one-instruction loads of a link quad / count block / support pair
struct alignas(16) u32x4 {
  uint32_t x, y, z, w;
};

template <int KW>
struct Key {
  uint32_t w[KW];
};

/// SB = bits per symbol of the packed pile.  2: the four bases as codes 0..3, 16 per dword (+ the N bitmap) -- the production
/// form.  8: the reads' bytes as they are, 4 per dword -- the byte-generic form for piles that hold bytes outside {A,C,G,T,N}
/// where masking them is not provably exact (the reference treats any byte as a symbol; only 'N' words are skipped and only
/// A,C,G,T extend a contig); keys of up to 32 dwords there (128 symbols, as in the 2-bit form).  Everything below a key is the same code.
template <int SB>
struct AssemblerT {
  static const unsigned SPD      = 32u / SB;          ///< symbols per code dword
  static const unsigned SYM_MASK = (1u << SB) - 1u;
  static const int      GEN_KW   = (SB == 8) ? 32 : ASM_MAX_KW;  ///< key dwords of the longest word (128 symbols in either form)
  static const unsigned MAX_K    = SPD * unsigned(GEN_KW);       ///< longest word
  /// code of alphabet symbol c (0..3 = A,C,G,T); alphabet index of a code (4: not in the alphabet); its character
  WV_DEV static unsigned symOfIndex(const unsigned c) { return (SB == 2) ? c : unsigned(uint8_t("ACGT"[c & 3u])); }
  WV_DEV static unsigned indexOfSym(const unsigned s)
  {
    if (SB == 2) return s;
    return (s == 'A') ? 0u : (s == 'C') ? 1u : (s == 'G') ? 2u : (s == 'T') ? 3u : 4u;
  }
  WV_DEV static uint8_t charOfSym(const unsigned s) { return (SB == 2) ? uint8_t("ACGT"[s & 3u]) : uint8_t(s); }
  const AsmParams& P;
  uint8_t*         ws;
  AsmWsLayout      L;
  // workspace views
  uint32_t *codes, *nmask, *rd_cw, *rd_mw, *rd_len, *rd_hasn, *slots, *node_key, *node_cnt, *node_flag, *node_aux;
  uint8_t*  rec;        // node records (see asmRecStride)
  unsigned  recStride;
  uint32_t* links;      // succ[4] | pred[4] per node, plain 32-bit ids (cycle test, exact repeat search, wide-set walk)
  uint32_t* frontier;
  uint8_t * cand_seq, *walk_left, *walk_right, *pseudo_seq;
  uint64_t* cand_bits;
  int32_t*  cand_meta;  // per candidate: len, consBegin, consEnd, pad
  uint32_t* pseudo_len;
  uint32_t* exact_ws;
  uint32_t *node_k32, *tent_raw, *tent_sorted, *lane_vis, *unused_bits;
  uint64_t *small_active, *small_repeat;  // SmallAssembler read sets (alias lane_bits: the lane walks are not used there)
  uint8_t*  lane_seq;
  uint64_t* lane_bits;
  int32_t*  lane_meta;
  // per-locus state (wave-uniform)
  unsigned nNormal, nReads, W, k, nNodes, nCodeWordsNormal, nMaskWordsNormal, nCand;
  unsigned codeWordsUsed, maskWordsUsed, slotMask;
  const uint32_t* presentMap;  // see lookup4 (valid from the table pass to the end of the links pass)
  int      status;
  unsigned cyclicIters;
  uint64_t tPhase[8];
  uint64_t tMark;

  WV_DEV AssemblerT(const AsmParams& p, uint8_t* wsBase) : P(p), ws(wsBase)
  {
    L = asmWorkspaceLayout(p.cap_slots, p.cap_nodes, p.cap_words, p.cap_reads, p.max_contig_len, p.w_max, p.opt.maxAssemblyCount);
    codes      = reinterpret_cast<uint32_t*>(ws + L.codes);
    nmask      = reinterpret_cast<uint32_t*>(ws + L.nmask);
    rd_cw      = reinterpret_cast<uint32_t*>(ws + L.rd_cw);
    rd_mw      = reinterpret_cast<uint32_t*>(ws + L.rd_mw);
    rd_len     = reinterpret_cast<uint32_t*>(ws + L.rd_len);
    rd_hasn    = reinterpret_cast<uint32_t*>(ws + L.rd_hasn);
    slots      = reinterpret_cast<uint32_t*>(ws + L.slots);
    node_key   = reinterpret_cast<uint32_t*>(ws + L.node_key);
    node_cnt   = reinterpret_cast<uint32_t*>(ws + L.node_cnt);
    node_flag  = reinterpret_cast<uint32_t*>(ws + L.node_flag);
    node_aux   = reinterpret_cast<uint32_t*>(ws + L.node_aux);
    rec        = ws + L.rec;
    recStride  = 64;
    links      = reinterpret_cast<uint32_t*>(ws + L.links);
    frontier   = reinterpret_cast<uint32_t*>(ws + L.frontier);
    cand_seq   = ws + L.cand_seq;
    cand_bits  = reinterpret_cast<uint64_t*>(ws + L.cand_bits);
    cand_meta  = reinterpret_cast<int32_t*>(ws + L.cand_meta);
    walk_left  = ws + L.walk_left;
    walk_right = ws + L.walk_right;
    pseudo_seq = ws + L.pseudo_seq;
    pseudo_len = reinterpret_cast<uint32_t*>(ws + L.pseudo_len);
    exact_ws   = reinterpret_cast<uint32_t*>(ws + L.exact);
    node_k32   = reinterpret_cast<uint32_t*>(ws + L.node_k32);
    tent_raw   = reinterpret_cast<uint32_t*>(ws + L.tent);
    tent_sorted = tent_raw + TENT_CAP;
    lane_seq   = ws + L.lane_seq;
    lane_bits  = reinterpret_cast<uint64_t*>(ws + L.lane_bits);
    small_active = lane_bits;
    small_repeat = lane_bits + ASM_MAX_W;
    lane_meta  = reinterpret_cast<int32_t*>(ws + L.lane_meta);
    lane_vis   = reinterpret_cast<uint32_t*>(ws + L.lane_vis);
    unused_bits = reinterpret_cast<uint32_t*>(ws + L.unused);
  }

  /// optional per-phase shader-clock profile (compiled in with -DMANTA_ASM_PROFILE; costs registers)
  WV_DEV void tick(int phase)
  {
#if defined(MANTA_ASM_PROFILE) && !defined(MANTA_LG_PROFILE_GRAPH)  // (-DMANTA_LG_PROFILE_GRAPH: the eight counters are the graph kernels' alone)
#ifdef MANTA_ASM_PROFILE_EXACT  // developer build: the eight counters split the exact repeat search (tickExact), everything else is counter 7
    phase = 7;
#endif
    const uint64_t now = wv::clock();
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i == phase) tPhase[i] += now - tMark;
    tMark = now;
#else
    (void)phase;
#endif
  }
  /// sub-phases of exactRepeatSearch (0 insertion sequence + hashes, 1 rank inside the groups, 2 / 3 the two unordered_map orders,
  /// 5 DFS set-up, 6 DFS) in a -DMANTA_ASM_PROFILE -DMANTA_ASM_PROFILE_EXACT build
  WV_DEV void tickExact(const int phase)
  {
#if defined(MANTA_ASM_PROFILE) && defined(MANTA_ASM_PROFILE_EXACT)
    const uint64_t now = wv::clock();
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i == phase) tPhase[i] += now - tMark;
    tMark = now;
#else
    (void)phase;
#endif
  }

  WV_DEV uint32_t* recSucc(const unsigned n) const { return links + size_t(n) * 8; }
  WV_DEV uint32_t* recPred(const unsigned n) const { return links + size_t(n) * 8 + 4; }
  WV_DEV uint64_t* recPacked(const unsigned n, const unsigned dirOff16) const  // dirOff16: 0 = successors, 16 = predecessors
  {
    return reinterpret_cast<uint64_t*>(rec + size_t(n) * recStride + dirOff16);
  }
  WV_DEV uint64_t* recSup(const unsigned n) const { return reinterpret_cast<uint64_t*>(rec + size_t(n) * recStride + 32); }
  /// four node ids (ASM_NONE -> LINK_NONE21) + saturated count in 128 bits
  WV_DEV static void packLinks(const unsigned id[4], const unsigned cnt, uint64_t& lo, uint64_t& hi)
  {
    uint64_t v[4];
    for (int i = 0; i < 4; ++i) v[i] = (id[i] == ASM_NONE) ? uint64_t(LINK_NONE21) : uint64_t(id[i]);
    lo = v[0] | (v[1] << 21) | (v[2] << 42);
    hi = v[3] | (uint64_t(cnt < LINK_CNT_MAX ? cnt : LINK_CNT_MAX) << 21);
  }
  WV_DEV static void unpackLinks(const uint64_t lo, const uint64_t hi, unsigned id[4], unsigned& cnt)
  {
    id[0] = unsigned(lo) & LINK_NONE21;
    id[1] = unsigned(lo >> 21) & LINK_NONE21;
    id[2] = unsigned(lo >> 42) & LINK_NONE21;
    id[3] = unsigned(hi) & LINK_NONE21;
    for (int i = 0; i < 4; ++i)
      if (id[i] == LINK_NONE21) id[i] = ASM_NONE;
    cnt = unsigned(hi >> 21) & LINK_CNT_MAX;
  }

  // ------------------------------------------------------------------------------------------------
  // wave helpers (all lanes must call)
  // ------------------------------------------------------------------------------------------------
  WV_DEV static unsigned waveSum(unsigned v)
  {
    for (int off = 1; off < 64; off <<= 1) v += wv::shfl(v, wv::lane() ^ off);
    return v;
  }
  WV_DEV static unsigned waveMax(unsigned v)
  {
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned o = wv::shfl(v, wv::lane() ^ off);
      v                = (o > v) ? o : v;
    }
    return v;
  }
  WV_DEV static uint64_t waveSum64(uint64_t v)
  {
    for (int off = 1; off < 64; off <<= 1) v += wv::shfl(v, wv::lane() ^ off);
    return v;
  }
  WV_DEV static bool waveAny64(const uint64_t v) { return wv::any(v != 0); }

  // ------------------------------------------------------------------------------------------------
  // packed reads
  // ------------------------------------------------------------------------------------------------
  /// the symbol at packed base index pb
  WV_DEV unsigned symAt(const unsigned pb) const { return (codes[pb / SPD] >> (32 - SB - SB * (pb % SPD))) & SYM_MASK; }
  /// one dword of symbols (16 bases / 4 bytes) starting at packed base index pb, MSB first
  WV_DEV uint32_t codes16(const unsigned pb) const
  {
    const unsigned wi = pb / SPD, sh = (pb % SPD) * SB;
    const uint32_t a = codes[wi];
    if (sh == 0) return a;
    return (a << sh) | (codes[wi + 1] >> (32 - sh));
  }

  template <int KW>
  WV_DEV Key<KW> keyAt(const unsigned pb) const
  {
    Key<KW>        key;
    const unsigned kw = (k + SPD - 1) / SPD;
    const unsigned wi = pb / SPD, sh = (pb % SPD) * SB;
    uint32_t       raw[KW + 1];
    if (KW <= 4) {
      // one wide (dword-aligned) load of KW+1 code dwords; the slab is padded, over-reading is harmless
      struct __attribute__((packed, aligned(4))) Raw {
        uint32_t v[KW + 1];
      };
      const Raw r = *reinterpret_cast<const Raw*>(codes + wi);
      for (int i = 0; i <= KW; ++i) raw[i] = r.v[i];
    } else {
      for (int i = 0; i <= KW; ++i) raw[i] = (unsigned(i) <= kw) ? codes[wi + i] : 0u;
    }
    for (int i = 0; i < KW; ++i) {
      uint32_t v = 0;
      if (unsigned(i) < kw) {
        v = uint32_t((((uint64_t(raw[i]) << 32) | raw[i + 1]) << sh) >> 32);  // funnel shift (v_alignbit_b32)
        const unsigned have = k - SPD * unsigned(i);  // symbols that belong to the word in this dword
        if (have < SPD) v &= ~((1u << (32 - SB * have)) - 1u);
      }
      key.w[i] = v;
    }
    return key;
  }

  /// hashes only the dwords the k-mer occupies, so the value does not depend on the KW instantiation
  template <int KW>
  WV_DEV uint32_t keyHash(const Key<KW>& key) const
  {
    const unsigned kw = (k + SPD - 1) / SPD;
    uint32_t       h  = 0x811C9DC5u;
    for (int i = 0; i < KW; ++i)
      if (unsigned(i) < kw) h = hashMix(h, key.w[i]);
    h ^= h >> 13;
    return h;
  }
  template <int KW>
  WV_DEV static bool keyEq(const Key<KW>& a, const Key<KW>& b)
  {
    bool eq = true;
    for (int i = 0; i < KW; ++i) eq = eq && (a.w[i] == b.w[i]);
    return eq;
  }
  /// lexicographic a < b (dword order == base order because bases are packed MSB first)
  template <int KW>
  WV_DEV static bool keyLess(const Key<KW>& a, const Key<KW>& b)
  {
    for (int i = 0; i < KW; ++i) {
      if (a.w[i] != b.w[i]) return a.w[i] < b.w[i];
    }
    return false;
  }
  template <int KW>
  WV_DEV void keySetBase(Key<KW>& key, const unsigned i, const unsigned c) const  // c: alphabet index 0..3
  {
    const unsigned sh = 32 - SB - SB * (i % SPD);
    for (int w = 0; w < KW; ++w)
      if (unsigned(w) == (i / SPD)) key.w[w] = (key.w[w] & ~(SYM_MASK << sh)) | (symOfIndex(c) << sh);
  }
  /// word[1..k-1] + c
  template <int KW>
  WV_DEV Key<KW> keyShiftAppend(const Key<KW>& key, const unsigned c) const
  {
    Key<KW> r;
    for (int w = 0; w < KW; ++w) r.w[w] = (key.w[w] << SB) | ((w + 1 < KW) ? (key.w[w + 1] >> (32 - SB)) : 0u);
    keySetBase(r, k - 1, c);
    return r;
  }
  /// c + word[0..k-2]
  template <int KW>
  WV_DEV Key<KW> keyShiftPrepend(const Key<KW>& key, const unsigned c) const
  {
    Key<KW> r;
    for (int w = 0; w < KW; ++w) r.w[w] = (key.w[w] >> SB) | ((w > 0) ? (key.w[w - 1] << (32 - SB)) : 0u);
    // drop the base that moved to position k
    const unsigned kw = (k + SPD - 1) / SPD;
    for (int w = 0; w < KW; ++w) {
      if (unsigned(w) >= kw) {
        r.w[w] = 0;
      } else if (unsigned(w) == kw - 1) {
        const unsigned have = k - SPD * unsigned(w);
        if (have < SPD) r.w[w] &= ~((1u << (32 - SB * have)) - 1u);
      }
    }
    keySetBase(r, 0, c);
    return r;
  }

  /// does [j, j+k) of read r contain an 'N' ?
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

  /// one table slot = 8 bytes {pb, node id}: a lookup touches one line
  WV_DEV uint64_t slotPair(const unsigned sidx) const { return *reinterpret_cast<const uint64_t*>(&slots[2 * size_t(sidx)]); }

  /// node id of the k-mer `key`, or ASM_NONE
  template <int KW>
  WV_DEV unsigned lookup(const Key<KW>& key) const
  {
    const unsigned mask = slotMask;
    unsigned       s    = keyHash(key) & mask;
    for (unsigned probe = 0; probe <= mask; ++probe) {
      const uint64_t pr  = slotPair(s);
      const uint32_t cur = uint32_t(pr);
      if (cur == ASM_NONE) return ASM_NONE;
      if (keyEq(keyAt<KW>(cur), key)) return uint32_t(pr >> 32);
      s = (s + 1) & mask;
    }
    return ASM_NONE;
  }

  /// four lookups with their memory round trips overlapped: all slot loads first, then all key fetches; only a
  /// collision (first probed slot holds another word) falls back to the serial probe loop
  /// `present` (links pass only): one bit per table index, set if some word's probe sequence STARTS there (buildGraph) -- three of a
  /// word's four possible successors usually do not exist, and two thirds of those are answered by this bitmap (a few hundred bytes,
  /// cache resident) instead of a slot sector each
  template <int KW>
  WV_DEV void lookup4(const Key<KW> (&keys)[4], unsigned (&out)[4], const uint32_t* present = nullptr) const
  {
    const unsigned mask = slotMask;
    uint64_t       pr[4];
    unsigned       h0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) h0[i] = keyHash(keys[i]) & mask;
    if (present) {
      uint32_t pw[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) pw[i] = present[h0[i] >> 5];
#pragma unroll
      for (int i = 0; i < 4; ++i) pr[i] = ((pw[i] >> (h0[i] & 31)) & 1u) ? slotPair(h0[i]) : ~uint64_t(0);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) pr[i] = slotPair(h0[i]);
    }
    Key<KW> got[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (uint32_t(pr[i]) != ASM_NONE) {
        got[i] = keyAt<KW>(uint32_t(pr[i]));
      } else {
        for (int w = 0; w < KW; ++w) got[i].w[w] = 0;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (uint32_t(pr[i]) == ASM_NONE)
        out[i] = ASM_NONE;
      else if (keyEq(got[i], keys[i]))
        out[i] = uint32_t(pr[i] >> 32);
      else
        out[i] = lookup<KW>(keys[i]);
    }
  }

  // ------------------------------------------------------------------------------------------------
  // stage 0: pack this locus' reads (bytes -> 2 bit + N bitmap).  Coalesced: lane i converts 16 bases.
  // ------------------------------------------------------------------------------------------------
  WV_DEV void packNormalReads(const unsigned locus)
  {
    const unsigned rBegin = P.locus_read_begin[locus], rEnd = P.locus_read_begin[locus + 1];
    nNormal               = rEnd - rBegin;
    status                = ASM_OK;

    if (nNormal + 2 * P.opt.maxAssemblyCount > P.cap_reads || nNormal + 2 * P.opt.maxAssemblyCount > 64 * P.w_max) {
      status = ASM_E_TOO_MANY_READS;
      return;
    }
    // word offsets: serial prefix by lane 0 would be slow for 1000 reads -> chunked scan
    unsigned cw = 0, mw = 0;
    for (unsigned base = 0; base < nNormal; base += 64) {
      const unsigned r   = base + unsigned(wv::lane());
      unsigned       len = 0;
      if (r < nNormal) len = P.pl_codes ? P.pl_read_len[rBegin + r + plShift(locus, 0)] : unsigned(P.read_off[rBegin + r + 1] - P.read_off[rBegin + r]);
      const unsigned myC = (r < nNormal) ? (len + SPD - 1) / SPD + 1 : 0u;  // +1 padding dword so key fetches may read one past
      const unsigned myM = (r < nNormal) ? (len + 31) / 32 + 1 : 0u;
      // inclusive scan over lanes
      unsigned sc = myC, sm = myM;
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned oc = wv::shfl(sc, wv::lane() - off), om = wv::shfl(sm, wv::lane() - off);
        if (wv::lane() >= off) {
          sc += oc;
          sm += om;
        }
      }
      if (r < nNormal) {
        rd_cw[r]               = cw + sc - myC;
        rd_len[r]              = len;
        rd_hasn[r]             = 0;
        rd_mw[r]               = mw + sm - myM;
      }
      cw += wv::readlane(sc, 63);
      mw += wv::readlane(sm, 63);
    }
    if (cw + 2 > P.cap_words || mw + 2 > maskWordCap()) {
      status = ASM_E_TABLE_FULL;
      return;
    }
    nCodeWordsNormal = cw;
    nMaskWordsNormal = mw;
    wv::sync();
    if (SB == 8) {
      // byte-generic pile: the reads' bytes as they are, 4 per dword, first byte in the top bits (dword order == string
      // order); 'N' positions go into the bitmap as usual (their words are skipped, :531), every other byte is a symbol
      if (P.pl_codes) {
        status = ASM_E_INTERNAL;  // (packed piles cannot hold such bytes; the host never sends them here)
        return;
      }
      const unsigned lane8 = unsigned(wv::lane());
      const uint32_t shift8 = P.chunk_shift ? P.chunk_shift[locus / P.chunk_loci] : 0u;
      for (unsigned r = 0; r < nNormal; ++r) {
        const uint8_t* src = P.bases + P.read_off[rBegin + r] + shift8;
        const unsigned len = rd_len[r], cwo = rd_cw[r], mwo = rd_mw[r];
        const unsigned nCw = (len + SPD - 1) / SPD + 1;
        bool           sawN = false;
        for (unsigned wi = lane8; wi < nCw; wi += 64) {
          uint32_t code = 0;
          for (unsigned b = 0; b < 4; ++b) {
            const unsigned i = wi * 4 + b;
            if (i >= len) continue;
            const unsigned c = src[i];
            code |= c << (24 - 8 * b);
            if (c == 'N') {
              wv::atomic_or(&nmask[mwo + (i >> 5)], 1u << (i & 31));
              sawN = true;
            }
          }
          codes[cwo + wi] = code;
        }
        if (wv::any(sawN) && lane8 == 0) rd_hasn[r] = 1u;
      }
      return;
    }
    if (P.pl_codes) {  // packed piles: copy (8 lanes per read, as below), add the pad dwords, note which reads hold an 'N'
      const unsigned lane = unsigned(wv::lane());
      const uint64_t plShiftR = plShift(locus, 0), plShiftC = plShift(locus, 1), plShiftM = plShift(locus, 2);
      for (unsigned base = 0; base < nNormal; base += 8) {
        const unsigned r = base + (lane >> 3);
        if (r >= nNormal) continue;
        const unsigned  len = rd_len[r], cwo = rd_cw[r], mwo = rd_mw[r];
        const unsigned  nCw = (len + 15) / 16, nMw = (len + 31) / 32;
        const uint32_t* sc  = P.pl_codes + (P.pl_code_off[rBegin + r + plShiftR] + plShiftC);
        const uint32_t* sm  = P.pl_nmask + (P.pl_mask_off[rBegin + r + plShiftR] + plShiftM);
        for (unsigned wi = (lane & 7); wi <= nCw; wi += 8) codes[cwo + wi] = (wi < nCw) ? sc[wi] : 0u;
        bool sawN = false;
        for (unsigned wi = (lane & 7); wi <= nMw; wi += 8) {
          const uint32_t m = (wi < nMw) ? sm[wi] : 0u;
          nmask[mwo + wi]  = m;
          sawN             = sawN || (m != 0);
        }
        if (sawN) wv::atomic_or(&rd_hasn[r], 1u);
      }
      return;
    }
    bool badAlphabet = false;
    // 8 lanes per read, 8 reads per pass: lane (g, i) converts code dwords i, i+8, ... of read (base + g)
    const unsigned lane = unsigned(wv::lane());
    for (unsigned base = 0; base < nNormal; base += 8) {
      const unsigned r = base + (lane >> 3);
      if (r >= nNormal) continue;
      const uint8_t* src = P.bases + P.read_off[rBegin + r] + (P.chunk_shift ? P.chunk_shift[locus / P.chunk_loci] : 0u);
      const unsigned len = rd_len[r], cwo = rd_cw[r], mwo = rd_mw[r];
      const unsigned nCw = (len + 15) / 16 + 1;
      for (unsigned wi = (lane & 7); wi < nCw; wi += 8) {
        uint32_t code = 0, nbits = 0;
        if (wi * 16 < len) {
          // 16 bases = five aligned dword loads + a byte funnel instead of sixteen byte loads (the input arena is padded)
          const uintptr_t addr  = reinterpret_cast<uintptr_t>(src + wi * 16);
          const uint32_t* ap    = reinterpret_cast<const uint32_t*>(addr & ~uintptr_t(3));
          const unsigned  shift = unsigned(addr & 3) * 8;
          uint32_t        d[5];
          for (int q = 0; q < 5; ++q) d[q] = ap[q];
          for (unsigned q = 0; q < 4; ++q) {
            const uint32_t four = shift ? ((d[q] >> shift) | (d[q + 1] << (32 - shift))) : d[q];
            for (unsigned b4 = 0; b4 < 4; ++b4) {
              const unsigned b = q * 4 + b4;
              const unsigned i = wi * 16 + b;
              unsigned       c = 0;
              if (i < len) {
                c = baseCode(uint8_t(four >> (8 * b4)));
                if (c == 5) badAlphabet = true;  // handled below: masked like 'N' where that is provably exact
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
        if (nbits) {  // the N bitmap was zeroed up front; two code dwords share one mask dword
          wv::atomic_or(&nmask[mwo + (wi >> 1)], (wi & 1) ? (nbits << 16) : nbits);
          wv::atomic_or(&rd_hasn[r], 1u);
        }
      }
      if (badAlphabet) wv::atomic_or(&rd_hasn[r], 2u);  // (lane-private flag: this lane only ever works on read r in this pass)
      badAlphabet = false;
    }
    // Bytes outside {A,C,G,T,N} ("junk"; the reference is byte-generic).  A word that contains one is an ordinary word for
    // the reference, but it is never a successor or predecessor of anything (extensions append alphabet symbols only,
    // IterativeAssembler.cpp:241-247), so it can matter in exactly two ways: as a seed -- needs wordCount >= minCoverage
    // (:678-682) -- and through the visiting order of the repeat search, which only changes results when the graph has a
    // cycle (:555-642).  A junk word occurs at most once per read that holds junk, so with fewer such reads than
    // minCoverage no junk word can be a seed: the junk positions are then masked like 'N' (their words are dropped, every
    // other word of the read counts as usual) and the locus is exact as long as its graphs stay acyclic (contigsFromGraph
    // reports ASM_E_ALPHABET otherwise).  Anything else is reported, never guessed.
    // (resolveJunk(), called once the flags are visible, applies the rule)
  }

  /// number of reads that hold a byte outside {A,C,G,T,N} (flag bit 1 of rd_hasn; kept in the slab, not in registers: it is
  /// needed once per locus and again only for cyclic graphs).  Call after packNormalReads + sync + fence.
  WV_DEV_COLD unsigned countJunkReads()
  {
    unsigned junkReads = 0;
    if (!P.pl_codes && status == ASM_OK)
      for (unsigned r = unsigned(wv::lane()); r < nNormal; r += 64) junkReads += (rd_hasn[r] >> 1) & 1u;
    return waveSum(junkReads);
  }

  WV_DEV unsigned maskWordCap() const { return P.cap_words / 2 + P.cap_reads + 2; }

  /// streamed packed piles: where this locus' chunk landed (i = 0 read index, 1 code dwords, 2 mask dwords); 0 otherwise
  WV_DEV uint64_t plShift(const unsigned locus, const unsigned i) const
  {
    return P.pl_chunk_shift ? P.pl_chunk_shift[3 * size_t(locus / P.chunk_loci) + i] : uint64_t(0);
  }

  // ------------------------------------------------------------------------------------------------
  // k-mer graph for the current word length
  // ------------------------------------------------------------------------------------------------
  /// SMALL = the SmallAssembler's k-mer pass (assembly/SmallAssembler.cpp:396-455): only the reads of small_active[]
  /// take part, and a read that holds a word twice is noted in small_repeat[] (both: bitsets over read indices in the slab)
  template <int KW, bool SMALL = false>
  WV_DEV void buildGraph()
  {
    const unsigned lane = unsigned(wv::lane());
    // table sized for this locus and word length: 2x the k-mer instances, rounded up to a power of two
    unsigned inst = 0;
    for (unsigned r = lane; r < nReads; r += 64) {
      const unsigned len = rd_len[r];
      inst += (len >= k) ? (len - k + 1) : 0u;
    }
    inst = waveSum(inst);
    // The table is sized for the DISTINCT words, which is usually a small fraction of the instances (reads overlap):
    // start at half the instance count and double on overflow (load factor > 0.7), which re-runs the pass.
    unsigned tableSlots = 64;
    while (tableSlots < (inst >> MANTA_ASM_TABLE_SHIFT) && tableSlots < P.cap_slots) tableSlots <<= 1;
    bool full = false;
    while (true) {
      slotMask = tableSlots - 1;
      for (unsigned s = lane; s < tableSlots; s += 64) *reinterpret_cast<uint64_t*>(&slots[2 * size_t(s)]) = ~uint64_t(0);  // {no word, no id}
      // which table indices start a probe sequence (lookup4 of the links pass); lives in the frontier region, idle until the cycle test
      uint32_t* const present = (uint64_t(tableSlots) / 8 + 4 <= 8ull * P.cap_nodes + 256) ? frontier : nullptr;  // (always, for any sane capacity pair)
      presentMap              = present;
      for (unsigned s = lane; present && s < (tableSlots + 31) / 32; s += 64) present[s] = 0;
      wv::sync();

      // one fused pass over the k-mer instances (assembly/IterativeAssembler.cpp:516-548):
      //   claim-or-find the word's slot with one atomicCAS; the claimers of this step get dense node ids from a
      //   ballot prefix and initialise their node; after a wave-local fence every instance ORs its read into the
      //   word's support set.  Per-read de-dup is implicit (sets).
      const unsigned mask     = slotMask;
      const unsigned maxNodes = (tableSlots < P.cap_slots) ? unsigned((uint64_t(tableSlots) * 7) / 10) : tableSlots;
      // Read sets are gathered in a DENSE array (W qwords per word, no links in between) and copied into the 64-byte
      // records by the pass below: the words of a read have mostly consecutive ids, so the 64 atomics of a step fall into
      // 64 * 8 W / 64 sectors (16 for W = 2) instead of one record sector each.  The array borrows the exact-search
      // workspace (14 dwords per node, idle until the graph is complete); wider read sets keep the records.
      const bool      denseSup = (2 * W <= 14);
      uint64_t* const supBuild = reinterpret_cast<uint64_t*>(exact_ws + 16);
      bool           overflow = false;
      full                    = false;  // a lane can run out of slots while the table overflows; only the final pass counts
      nNodes                  = 0;
      if (SMALL) {
        if (lane < ASM_MAX_W) small_repeat[lane] = 0;
        wv::sync();
      }
      for (unsigned rBase = 0; rBase < nReads && !overflow; rBase += 64) {
        // read descriptors of up to 64 reads live in lane registers; v_readlane hands them out per read
        const unsigned rMine = rBase + lane;
        const unsigned lenV  = (rMine < nReads) ? rd_len[rMine] : 0u;
        const unsigned cwoV  = (rMine < nReads) ? rd_cw[rMine] : 0u;
        const unsigned mwoV  = (rMine < nReads) ? rd_mw[rMine] : 0u;
        const unsigned hasnV = (rMine < nReads) ? rd_hasn[rMine] : 0u;
        const unsigned rEnd  = (nReads - rBase < 64) ? (nReads - rBase) : 64u;
        for (unsigned ri = 0; ri < rEnd && !overflow; ++ri) {
          const unsigned r   = rBase + ri;
          const unsigned len = wv::readlane(lenV, int(ri));
          if (len < k) continue;  // :522
          if (SMALL && !((small_active[r >> 6] >> (r & 63)) & 1u)) continue;  // SmallAssembler.cpp:413 (isUsed)
          bool twice = false;
          const unsigned cwo = wv::readlane(cwoV, int(ri)), mwo = wv::readlane(mwoV, int(ri));
          const bool     rdHasN = wv::readlane(hasnV, int(ri)) != 0;  // most reads have no 'N': skip the bitmap test
          const uint64_t bit = uint64_t(1) << (r & 63);
          for (unsigned j0 = 0; j0 + k <= len && !overflow; j0 += 64) {
            const unsigned j    = j0 + lane;
            const unsigned pb   = cwo * SPD + j;
            unsigned       slot = ASM_NONE;
            bool           won  = false;
            unsigned       foundId = ASM_NONE;
            unsigned       h0      = 0;  // first table index of the word's probe sequence
            const bool     valid = (j + k <= len) && !(rdHasN && windowHasN(mwo, j));  // :531
            Key<KW>        key;
            if (valid) key = keyAt<KW>(pb);
            // (guessing the ids of consecutive positions instead of probing the table was measured: no gain)
            if (valid) {
              unsigned      s   = keyHash(key) & mask;
              h0                = s;
              for (unsigned probe = 0; probe <= mask; ++probe) {
                // one 8-byte load brings the word AND (for words claimed in an earlier step) its node id
                const uint64_t pr  = wv::atomic_load(reinterpret_cast<const unsigned long long*>(&slots[2 * size_t(s)]));
                uint32_t       cur = uint32_t(pr);
                foundId            = uint32_t(pr >> 32);
                if (cur == ASM_NONE) {
                  cur     = wv::atomic_cas(&slots[2 * size_t(s)], ASM_NONE, pb);
                  foundId = ASM_NONE;  // claimed just now (by this lane or a neighbour): the id is not written yet
                  if (cur == ASM_NONE) {
                    slot = s;
                    won  = true;
                    break;
                  }
                }
                if (keyEq(keyAt<KW>(cur), key)) {
                  slot = s;
                  break;
                }
                s = (s + 1) & mask;
              }
              if (slot == ASM_NONE) full = true;
            }
            const uint64_t m  = wv::ballot(won);
            const unsigned id = nNodes + unsigned(wv::popc(m & ((uint64_t(1) << lane) - 1)));
            nNodes += unsigned(wv::popc(m));
            if (nNodes > P.cap_nodes) {
              status = ASM_E_TABLE_FULL;
              return;
            }
            if (nNodes > maxNodes) {  // uniform: the table is too small for this locus, start over with twice the slots
              overflow = true;
              break;
            }
            unsigned myId = won ? id : foundId;
            if (m) {  // some lane created a word in this step: publish the new nodes before anybody ORs into them
              if (won) {
                slots[2 * size_t(slot) + 1] = id;
                if (present) wv::atomic_or(&present[h0 >> 5], 1u << (h0 & 31));
                node_key[id]                = pb;
                for (unsigned w = 0; w < W; ++w) (denseSup ? supBuild + size_t(id) * W : recSup(id))[w] = 0;
                for (unsigned c = 0; c < 4; ++c) recPred(id)[c] = ASM_NONE;  // filled by the successors' scatter
              }
              wv::sync();
              if (slot != ASM_NONE && myId == ASM_NONE) myId = wv::atomic_load(&slots[2 * size_t(slot) + 1]);
            }
            if (slot != ASM_NONE) {
              uint64_t* const supAt = denseSup ? supBuild + size_t(myId) * W : recSup(myId);
              if (SMALL) {
                const unsigned long long before = wv::atomic_or(reinterpret_cast<unsigned long long*>(&supAt[r >> 6]), bit);
                if (before & bit) twice = true;  // the word is already in this read's word set (SmallAssembler.cpp:432)
              } else {
                // No atomic needed: the slab belongs to this wave, every lane of the step adds the SAME bit (one read per step),
                // and two lanes that meet in one word (the read holds it twice) store the same value.
                supAt[r >> 6] |= bit;
              }
            }
          }
          if (SMALL && wv::any(twice) && lane == 0) small_repeat[r >> 6] |= bit;
        }
      }
      wv::sync();
      wv::fence_acquire();  // slots / supports were filled by L2 atomics: drop stale L1 lines before plain re-reads
      if (!overflow) break;
      tableSlots <<= 1;
    }
    if (wv::any(full)) {
      status = ASM_E_TABLE_FULL;
      return;
    }
    tick(1);

    // counts (:541-545: a pseudo read weighs minCoverage), seed eligibility (:679-682), links
    for (unsigned nd = lane; nd < nNodes; nd += 64) {
      unsigned cnt = 0;
      for (unsigned w = 0; w < W; ++w) {
        uint64_t s;
        if (2 * W <= 14) {  // (denseSup of the table pass; the record is written once, whole, by the next pass)
          s = reinterpret_cast<const uint64_t*>(exact_ws + 16)[size_t(nd) * W + w];
        } else {
          s = recSup(nd)[w];
        }
        cnt += unsigned(wv::popc(s & normalMask(w))) + P.opt.minCoverage * unsigned(wv::popc(s & ~normalMask(w)));
      }
      node_cnt[nd]  = cnt;
      const Key<KW> key = keyAt<KW>(node_key[nd]);
      node_k32[nd]  = key.w[0];
      bool          selfLoop = false;
      unsigned      sIds[4];
      {
        Key<KW> ks[4];
        for (unsigned c = 0; c < 4; ++c) ks[c] = keyShiftAppend<KW>(key, c);
        lookup4<KW>(ks, sIds, presentMap);
      }
      // Predecessor links are not looked up: nd is the predecessor of each of its successors through nd's own first
      // base, so the successor lookups scatter them (every (word, symbol) slot has exactly one writer).
      // (a word whose first symbol is not in the alphabet is nobody's predecessor candidate: extensions prepend A,C,G,T only)
      const unsigned firstBase = indexOfSym(key.w[0] >> (32 - SB));
      for (unsigned c = 0; c < 4; ++c) {
        const unsigned s = sIds[c];
        recSucc(nd)[c]   = s;
        if (s != ASM_NONE && firstBase < 4) recPred(s)[firstBase] = nd;
        if (s == nd) selfLoop = true;  // homopolymer (:574-577)
      }
      bool hasJunk = false;
      if (SB == 8)
        for (unsigned i = 0; i < k; ++i) hasJunk = hasJunk || indexOfSym(symAt(node_key[nd] + i)) >= 4;
      if (SB == 8 && indexOfSym(symAt(node_key[nd] + k - 1)) >= 4) {
        // a word that ENDS in a byte outside the alphabet is nobody's successor (extensions append A,C,G,T only), so no scatter
        // reaches its predecessor slots: look its predecessors up (c + word[0..k-2] drops that byte and may well exist) -- the
        // walk to the left takes them (:241-251), the successor graph of the cycle test / repeat search does not have them
        for (unsigned c = 0; c < 4; ++c) recPred(nd)[c] = lookup<KW>(keyShiftPrepend<KW>(key, c));
      }
      if (2 * W > 14) packLinks(sIds, cnt, recPacked(nd, 0)[0], recPacked(nd, 0)[1]);
      node_flag[nd] = (selfLoop ? NF_REPEAT : 0u) | (hasJunk ? NF_JUNK : 0u);
    }
    wv::sync();
    wv::fence_acquire();
    for (unsigned nd = lane; nd < nNodes; nd += 64) {
      unsigned pIds[4], indeg = 0;
      for (unsigned c = 0; c < 4; ++c) {
        const unsigned p = recPred(nd)[c];
        pIds[c]          = p;
        if (p != ASM_NONE && p != nd && !(SB == 8 && ((node_flag[nd] | node_flag[p]) & NF_JUNK))) indeg++;
      }
      packLinks(pIds, node_cnt[nd], recPacked(nd, 16)[0], recPacked(nd, 16)[1]);
      if (2 * W <= 14) {
        // the rest of the 64-byte record in the same go (one full-sector write per word instead of three partial ones at three
        // different times): successors from the links array, read set from the dense array of the table pass
        unsigned sIds[4];
        for (unsigned c = 0; c < 4; ++c) sIds[c] = recSucc(nd)[c];
        packLinks(sIds, node_cnt[nd], recPacked(nd, 0)[0], recPacked(nd, 0)[1]);
        for (unsigned w = 0; w < W; ++w) recSup(nd)[w] = reinterpret_cast<const uint64_t*>(exact_ws + 16)[size_t(nd) * W + w];
      }
      node_aux[nd] = indeg;
    }
    wv::sync();
    // seed eligibility (:679-682) as a bitmap over node ids: one ballot per 64 nodes
    for (unsigned base = 0; base < nNodes; base += 64) {
      const unsigned nd = base + lane;
      const uint64_t m  = wv::ballot(nd < nNodes && node_cnt[nd] >= P.opt.minCoverage);
      if (lane < 2) unused_bits[(base >> 5) + lane] = uint32_t(m >> (32 * lane));
    }
    wv::sync();
    tick(2);
  }

  WV_DEV bool isUnused(const unsigned nd) const { return (unused_bits[nd >> 5] >> (nd & 31)) & 1u; }

  WV_DEV uint64_t normalMask(const unsigned w) const
  {
    const unsigned lo = w * 64;
    if (nNormal >= lo + 64) return ~uint64_t(0);
    if (nNormal <= lo) return 0;
    return (uint64_t(1) << (nNormal - lo)) - 1;
  }

  /// true if the k-mer graph (self loops ignored) has a directed cycle.  Wave-parallel peel of sources (in-degree 0)
  /// AND sinks (out-degree 0): what survives has in- and out-degree >= 1 inside the survivor set, i.e. contains a
  /// cycle.  Peeling from both ends halves the number of rounds on the (mostly linear) graphs.
  /// Per-node state = one 16-bit field {in-degree:4, out-degree:4, peeled:1}, two nodes per dword, updated with
  /// atomics; it lives in LDS when the graph is small enough (<= ASM_LDS_BYTES/2 nodes), else in global scratch.
  /// byte-generic form only: is the edge between a and b outside the cycle test (one of them holds a non-alphabet byte)?
  WV_DEV bool isJunk(const unsigned a, const unsigned b) const
  {
    if (SB != 8 || b == ASM_NONE) return false;
    return ((node_flag[a] | node_flag[b]) & NF_JUNK) != 0;
  }

  WV_DEV bool graphHasCycle()
  {
    const unsigned lane   = unsigned(wv::lane());
    uint32_t*      cur    = frontier;
    uint32_t*      nxt    = frontier + P.cap_nodes;
    uint32_t*      cnt    = exact_ws;  // frontier sizes [0],[1]
    const bool     inLds  = (nNodes * 2 <= P.lds_bytes);
    uint32_t*      st     = inLds ? reinterpret_cast<uint32_t*>(wv::lds(P.lds_bytes)) : (exact_ws + 64);
    const unsigned nWords = (nNodes + 1) / 2;
    if (lane == 0) {
      cnt[0] = 0;
      cnt[1] = 0;
    }
    wv::sync();  // (the other lanes' atomics on cnt[0] below must find the zero)
    for (unsigned w = lane; w < nWords; w += 64) {
      uint32_t v = 0;
      for (unsigned h = 0; h < 2; ++h) {
        const unsigned nd = w * 2 + h;
        if (nd >= nNodes) continue;
        unsigned od = 0, only = ASM_NONE;
        for (unsigned c = 0; c < 4; ++c) {
          const unsigned s = recSucc(nd)[c];
          if (s != ASM_NONE && s != nd && !isJunk(nd, s)) {
            od++;
            only = s;
          }
        }
        const unsigned id  = node_aux[nd];
        const bool     src = (id == 0 || od == 0);
        // "simple edge" nd -> nd+1: the only way out of nd and the only way into nd+1 (node ids follow read order, so
        // the unbranched stretches of the graph are mostly runs of consecutive ids)
        const bool simple = (od == 1 && only == nd + 1 && nd + 1 < nNodes && node_aux[nd + 1] == 1);
        v |= (id | (od << 4) | (src ? 0x100u : 0u) | (simple ? 0x200u : 0u)) << (16 * h);
        if (src) cur[wv::atomic_add(&cnt[0], 1u)] = nd;
      }
      st[w] = v;
    }
    wv::sync();
    unsigned removed = 0;
    unsigned which   = 0;
    while (true) {
      const unsigned nCur = wv::first(wv::atomic_load(&cnt[which]));
      if (nCur == 0) break;
      removed += nCur;
      // (measured: letting a lane chase the chain it is peeling is slower than these wave-wide rounds)
      for (unsigned i = lane; i < nCur; i += 64) {
        const unsigned nd = cur[i];
        for (unsigned c = 0; c < 4; ++c) {
          const unsigned s = isJunk(nd, recSucc(nd)[c]) ? ASM_NONE : recSucc(nd)[c];
          if (s != ASM_NONE && s != nd) {
            const unsigned sh  = 16 * (s & 1);
            const unsigned old = wv::atomic_sub(&st[s >> 1], 1u << sh) >> sh;
            if ((old & 0xfu) == 1u && !(wv::atomic_or(&st[s >> 1], 0x100u << sh) & (0x100u << sh)))
              nxt[wv::atomic_add(&cnt[which ^ 1], 1u)] = s;
          }
          const unsigned p = isJunk(nd, recPred(nd)[c]) ? ASM_NONE : recPred(nd)[c];
          if (p != ASM_NONE && p != nd) {
            const unsigned sh  = 16 * (p & 1);
            const unsigned old = wv::atomic_sub(&st[p >> 1], 0x10u << sh) >> sh;
            if ((old & 0xf0u) == 0x10u && !(wv::atomic_or(&st[p >> 1], 0x100u << sh) & (0x100u << sh)))
              nxt[wv::atomic_add(&cnt[which ^ 1], 1u)] = p;
          }
        }
      }
      wv::sync();
      // Stretch peel: a new source (sink) at the head (tail) of a run of simple edges drags the whole run with it, one
      // node per round.  When the next frontier is small -- the steady state: the two ends of the main path -- take such
      // runs 64 nodes at a time: everything strictly inside the run has no other neighbour, so it leaves silently; the
      // node at the far end is marked and queued, and the regular round updates ITS neighbours.
      {
        const unsigned nNext = wv::first(wv::atomic_load(&cnt[which ^ 1]));
        if (nNext > 0 && nNext <= ASM_STRETCH_MAX) {
          auto stateOf = [&](const unsigned n) { return (wv::atomic_load(&st[n >> 1]) >> (16 * (n & 1))) & 0xffffu; };
          for (unsigned qi = 0; qi < nNext; ++qi) {
            const unsigned f  = wv::first(nxt[qi]);
            const unsigned sf = wv::first(stateOf(f));
            for (int dir = 0; dir < 2; ++dir) {
              if (dir == 0 ? ((sf & 0xfu) != 0) : ((sf & 0xf0u) != 0)) continue;  // forward from a source, backward from a sink
              unsigned c = f, total = 0;
              while (true) {
                // lane l looks at the edge between c+-l and c+-(l+1) and at the far node of that edge
                bool ok = false;
                if (dir == 0) {
                  const unsigned a = c + lane;
                  if (a + 1 < nNodes) ok = (stateOf(a) & 0x200u) && !(stateOf(a + 1) & 0x100u);
                } else {
                  if (c >= lane + 1) {
                    const unsigned b = c - lane - 1;
                    ok               = (stateOf(b) & 0x200u) && !(stateOf(b) & 0x100u);
                  }
                }
                const uint64_t good = wv::ballot(ok);
                const unsigned take = (~good == 0) ? 64u : unsigned(wv::ctz(~good));
                if (take == 0) break;
                if (lane < take) {
                  const unsigned n = (dir == 0) ? (c + lane + 1) : (c - lane - 1);
                  wv::atomic_or(&st[n >> 1], 0x100u << (16 * (n & 1)));
                }
                total += take;
                c = (dir == 0) ? (c + take) : (c - take);
                wv::sync();
                if (take < 64) break;
              }
              if (total > 0) {
                if (lane == 0) nxt[wv::atomic_add(&cnt[which ^ 1], 1u)] = c;  // far end: counted and expanded next round
                removed += total - 1;
                wv::sync();
              }
            }
          }
        }
      }
      if (lane == 0) cnt[which] = 0;
      wv::sync();
      which ^= 1;
      uint32_t* t = cur;
      cur         = nxt;
      nxt         = t;
    }
    return removed != nNodes;
  }

  // ------------------------------------------------------------------------------------------------
  // seed selection (:686-696): highest count among unused words, ties -> lexicographically smallest
  // ------------------------------------------------------------------------------------------------
  WV_DEV_COLD unsigned selectSeed()
  {
    typedef Key<GEN_KW> GKey;
    const unsigned lane = unsigned(wv::lane());
    unsigned       best = 0;
    for (unsigned nd = lane; nd < nNodes; nd += 64) {
      if (isUnused(nd)) {
        const unsigned c = node_cnt[nd];
        best             = (c > best) ? c : best;
      }
    }
    best = waveMax(best);
    if (best == 0) return ASM_NONE;
    // smallest 16-base prefix among the best-count words, then the full k-mer only among prefix ties
    unsigned minPre = 0xffffffffu;
    for (unsigned nd = lane; nd < nNodes; nd += 64)
      if (isUnused(nd) && node_cnt[nd] == best) minPre = (node_k32[nd] < minPre) ? node_k32[nd] : minPre;
    minPre = ~waveMax(~minPre);
    unsigned mine = ASM_NONE;
    GKey     mineKey;
    for (int i = 0; i < GEN_KW; ++i) mineKey.w[i] = 0xffffffffu;
    for (unsigned nd = lane; nd < nNodes; nd += 64) {
      if (isUnused(nd) && node_cnt[nd] == best && node_k32[nd] == minPre) {
        const GKey key = keyAt<GEN_KW>(node_key[nd]);
        if (mine == ASM_NONE || keyLess(key, mineKey)) {
          mine    = nd;
          mineKey = key;
        }
      }
    }
    for (int off = 1; off < 64; off <<= 1) {
      const int      src = wv::lane() ^ off;
      const unsigned on  = wv::shfl(mine, src);
      GKey           ok;
      for (int i = 0; i < GEN_KW; ++i) ok.w[i] = wv::shfl(mineKey.w[i], src);
      if (on != ASM_NONE && (mine == ASM_NONE || keyLess(ok, mineKey))) {
        mine    = on;
        mineKey = ok;
      }
    }
    return mine;
  }

  // ------------------------------------------------------------------------------------------------
  // walk (:149-501).  Lane w (< W) owns qword w of every read set; decisions are wave-uniform.
  // ------------------------------------------------------------------------------------------------
  WV_DEV uint64_t supWord(const unsigned node) const
  {
    const unsigned lane = unsigned(wv::lane());
    return (node != ASM_NONE && lane < W) ? recSup(node)[lane] : uint64_t(0);
  }

  /// four 16-bit counts packed in a qword, summed over the wave, returned wave-uniform
  WV_DEV static uint64_t packedCountSum(const uint64_t v)
  {
    uint64_t s = v;
    for (int off = 1; off < int(ASM_MAX_W); off <<= 1) s += wv::shfl(s, wv::lane() ^ off);
    return wv::readlane(s, 0);
  }

  WV_DEV_COLD bool walk(const unsigned seed, const unsigned serial, const unsigned candIdx)
  {
    static const int KW = GEN_KW;  // generic key width: this path is the wide-read-set fallback, not the hot one
    const unsigned lane = unsigned(wv::lane());
    uint64_t       S    = supWord(seed);  // contig.supportReads (:168)
    uint64_t       Rj   = 0;              // contig.rejectReads
    uint8_t*       outSeq  = cand_seq + size_t(candIdx) * P.max_contig_len;
    uint64_t*      outBits = cand_bits + size_t(candIdx) * 2 * W;
    int32_t*       meta    = cand_meta + candIdx * 4;

    // node_flag is read AND written inside the walk: only lane 0 touches it, decisions are broadcast
    unsigned seedFlag = 0;
    if (lane == 0) seedFlag = node_flag[seed];
    seedFlag                = wv::readlane(seedFlag, 0);
    const bool seedIsRepeat = (seedFlag & NF_REPEAT) != 0;
    if (lane == 0) {
      node_flag[seed] = (seedFlag & NF_REPEAT) | (serial << 8);  // wordsInContig
      unused_bits[seed >> 5] &= ~(1u << (seed & 31));          // unused.erase(seed)
    }

    // seed k-mer text
    const unsigned seedPb = node_key[seed];
    for (unsigned i = lane; i < k; i += 64) {
      const unsigned pb = seedPb + i;
      outSeq[i]         = charOfSym(symAt(pb));
    }
    if (seedIsRepeat) {  // :172-179
      if (lane < W) {
        outBits[lane]     = S;
        outBits[W + lane] = 0;
      }
      if (lane == 0) {
        meta[0] = int(k);
        meta[1] = 0;
        meta[2] = int(k);
      }
      wv::sync();
      return true;
    }

    // reads of the unselected siblings of the seed (same (k-1)-prefix) reject the contig (:185-210)
    {
      const Key<KW>  key      = keyAt<KW>(seedPb);
      const unsigned lastBase = indexOfSym(symAt(seedPb + k - 1));
      for (unsigned c = 0; c < 4; ++c) {
        if (c == lastBase) continue;
        Key<KW> sib = key;
        keySetBase(sib, k - 1, c);
        const unsigned n = lookup<KW>(sib);
        Rj |= supWord(n);
      }
    }

    bool     isRepeatFound = false;
    unsigned nLeft = 0, nRight = 0;
    bool     tooLong = false;
    int      consEnd = 0, consBegin = 0;
    for (unsigned mode = 0; mode < 2; ++mode) {
      const bool      isEnd = (mode == 0);
      const unsigned  fwdOff = isEnd ? 0u : 4u, bwdOff = isEnd ? 4u : 0u;  // succ[4] | pred[4] inside a node record
      unsigned        consOffset = 0;
      unsigned        cur        = seed;
      while (true) {
        unsigned cand[4];
        uint64_t cw[4];
        unsigned ccount[4];
        uint64_t pk = 0;
        for (unsigned c = 0; c < 4; ++c) {
          cand[c]   = recSucc(cur)[fwdOff + c];
          cw[c]     = supWord(cand[c]);
          ccount[c] = (cand[c] != ASM_NONE) ? node_cnt[cand[c]] : 0u;
          pk += uint64_t(wv::popc(S & cw[c])) << (16 * c);
        }
        pk = packedCountSum(pk);

        unsigned maxBaseCount = 0, maxCnt = 0, maxNode = ASM_NONE, maxSym = 0;
        uint64_t maxWR = 0, maxCW = 0, rm = 0, add = 0;
        for (unsigned c = 0; c < 4; ++c) {  // :241-336
          if (cand[c] == ASM_NONE) continue;
          const unsigned cnt = unsigned(pk >> (16 * c)) & 0xffffu;
          if (cnt == 0) continue;  // :280
          const uint64_t CW = S & cw[c];
          const uint64_t SH = maxCW & cw[c];
          if (cnt > maxCnt) {  // :283-316
            rm |= maxCW & ~SH;
            add |= maxWR & ~SH;
            maxWR        = cw[c];
            maxCnt       = cnt;
            maxCW        = CW;
            maxBaseCount = ccount[c];
            maxSym       = c;
            maxNode      = cand[c];
          } else {  // :317-335
            rm |= CW & ~SH;
            add |= cw[c] & ~SH;
          }
        }
        if (maxBaseCount < P.opt.minCoverage) break;  // :343
        unsigned maxFlag = 0;
        if (lane == 0) maxFlag = node_flag[maxNode];
        maxFlag = wv::readlane(maxFlag, 0);
        if ((maxFlag >> 8) == serial) {  // :352-358
          isRepeatFound = true;
          break;
        }
        if (isEnd) {  // :363
          if (k + nRight >= P.max_contig_len) {
            tooLong = true;
            break;
          }
          if (lane == 0) walk_right[nRight] = uint8_t("ACGT"[maxSym]);
          nRight++;
        } else {
          if (k + nRight + nLeft >= P.max_contig_len) {
            tooLong = true;
            break;
          }
          if (lane == 0) walk_left[nLeft] = uint8_t("ACGT"[maxSym]);
          nLeft++;
        }
        if ((consOffset != 0) || (maxBaseCount < P.opt.minConservativeCoverage)) consOffset += 1;  // :368-369

        // one step backwards at the branching point (:377-427); runs every step because the reference's
        // previousWordReads is always empty at the test (:237)
        for (unsigned c = 0; c < 4; ++c) {
          const unsigned n = recSucc(maxNode)[bwdOff + c];
          if (n == cur) continue;      // the word we came from (:381)
          if (n == maxNode) continue;  // :389
          if (n == ASM_NONE) continue;
          const uint64_t bw  = supWord(n);
          const uint64_t upd = bw & ~maxCW;  // :400-414
          add |= upd;
          rm |= upd;
        }
        Rj |= add;            // :440-442
        S |= maxWR & ~Rj;     // :458-464
        S &= ~rm;             // :471-473
        if (lane == 0) {  // :482-484
          node_flag[maxNode] = (maxFlag & NF_REPEAT) | (serial << 8);
          unused_bits[maxNode >> 5] &= ~(1u << (maxNode & 31));
        }
        cur = maxNode;
      }
      if (isEnd)
        consEnd = int(consOffset);  // :488-491
      else
        consBegin = int(consOffset);
      if (tooLong) break;
    }
    if (tooLong) {
      status = ASM_E_CONTIG_TOO_LONG;
      return false;
    }
    wv::sync();
    // assemble the text: reverse(left) + seed + right
    const unsigned len = nLeft + k + nRight;
    // reverse(left) + seed (re-derived from the packed reads) + right
    for (unsigned i = lane; i < len; i += 64) {
      uint8_t ch;
      if (i < nLeft) {
        ch = walk_left[nLeft - 1 - i];
      } else if (i < nLeft + k) {
        const unsigned pb = seedPb + (i - nLeft);
        ch                = charOfSym(symAt(pb));
      } else {
        ch = walk_right[i - nLeft - k];
      }
      outSeq[i] = ch;
    }
    if (lane < W) {
      outBits[lane]     = S;
      outBits[W + lane] = Rj;
    }
    if (lane == 0) {
      meta[0] = int(len);
      meta[1] = consBegin;
      meta[2] = int(len) - consEnd;  // :498
    }
    wv::sync();
    return isRepeatFound;
  }

  // defined in repeat_exact.hpp
  WV_DEV_COLD void exactRepeatSearch();
  // defined in walk_lanes.hpp
  WV_DEV unsigned selectTentative(const unsigned T);
  template <int WQ>
  WV_DEV void walkLanes(const unsigned nT);
  template <int WQ>
  WV_DEV bool contigRounds();

  /// buildContigs (:644-720).  Returns isAssemblySuccess.  Only the graph build is specialised on the key width;
  /// everything after it works on node ids / links and uses the generic key width for its few k-mer compares.
  template <int KW>
  WV_DEV bool buildContigs()
  {
    buildGraph<KW>();
    if (status != ASM_OK) return true;
#ifdef MANTA_ASM_STOP_AFTER_GRAPH  // timing experiments only (tools/ab_probe.py): pack + table + links, no contigs
    nCand = 0;
    return true;
#endif
    return contigsFromGraph();
  }

  WV_DEV bool contigsFromGraph()
  {
    const bool cyclic = graphHasCycle();
    tick(3);
#ifdef MANTA_WAVE_EMU
    if (std::getenv("MANTA_EMU_DUMP_REPEATS") && wv::lane() == 0) std::fprintf(stderr, "DEV SB=%d k=%u nodes=%u cyclic=%d\n", SB, k, nNodes, int(cyclic));
#endif
    if (SB == 2 && cyclic && countJunkReads() > 0) {  // the repeat search's visiting order would include the dropped junk words
      status = ASM_E_ALPHABET;
      return true;
    }
    if (cyclic) {
      cyclicIters++;
      exactRepeatSearch();
      tickExact(6);  // (developer build: the DFS; a no-op otherwise)
      tick(4);
      if (status != ASM_OK) return true;
#ifdef MANTA_WAVE_EMU
      if (std::getenv("MANTA_EMU_DUMP_REPEATS") && wv::lane() == 0) {
        std::fprintf(stderr, "DEV repeats k=%u:", k);
        for (unsigned nd = 0; nd < nNodes; ++nd)
          if (node_flag[nd] & NF_REPEAT) {
            std::fprintf(stderr, " ");
            for (unsigned i = 0; i < k; ++i) std::fputc(charOfSym(symAt(node_key[nd] + i)), stderr);
          }
        std::fprintf(stderr, "\n");
      }
#endif
    }
    if (!(P.flags & ASM_FLAG_SERIAL_WALK)) {
      if (W <= 2) return contigRounds<2>();
      if (W <= 4) return contigRounds<4>();
    }
    // more than 256 reads (or forced): one contig at a time, lanes spread over the qwords of the read sets
    nCand        = 0;
    bool success = true;
    while (nCand < 2 * P.opt.maxAssemblyCount) {  // :685
      const unsigned seed = selectSeed();
      tick(5);
      if (seed == ASM_NONE) break;
      const bool rep = walk(seed, nCand + 1, nCand);
      tick(6);
      if (status != ASM_OK) return true;
      if (rep) success = false;
      nCand++;
    }
    return success;
  }

  WV_DEV bool buildContigsForK()
  {
    const unsigned kw = (k + SPD - 1) / SPD;
    if (kw <= 2) return buildContigs<2>();
    if (kw <= 4) return buildContigs<4>();
    if (SB != 8 || kw <= 8) return buildContigs<8>();
    if (kw <= 16) return buildContigs<(SB == 8) ? 16 : 8>();  // (byte form only: 4 symbols per dword)
    return buildContigs<(SB == 8) ? 32 : 8>();
  }

  // ------------------------------------------------------------------------------------------------
  // pseudo reads (:882-910)
  // ------------------------------------------------------------------------------------------------
  WV_DEV_COLD unsigned appendPseudoReads()
  {
    const unsigned lane = unsigned(wv::lane());
    unsigned       cw = nCodeWordsNormal, mw = nMaskWordsNormal;
    unsigned       nPseudo = 0;
    for (unsigned ci = 0; ci < nCand; ++ci) {
      const unsigned len = unsigned(cand_meta[ci * 4 + 0]);
      if (!(len > k + P.opt.wordStepSize)) continue;  // :898
      const unsigned nCw = (len + SPD - 1) / SPD + 1, nMw = (len + 31) / 32 + 1;
      if (cw + nCw + 2 > P.cap_words || mw + nMw + 2 > maskWordCap() || nNormal + nPseudo >= P.cap_reads) {
        status = ASM_E_TABLE_FULL;
        return 0;
      }
      const uint8_t* src = cand_seq + size_t(ci) * P.max_contig_len;
      const unsigned r   = nNormal + nPseudo;
      for (unsigned wi = lane; wi < nCw; wi += 64) {
        uint32_t code = 0;
        for (unsigned b = 0; b < SPD; ++b) {
          const unsigned i = wi * SPD + b;
          const unsigned c = (i < len) ? ((SB == 2) ? (baseCode(src[i]) & 3u) : unsigned(src[i])) : 0u;
          code |= c << (32 - SB - SB * b);
        }
        codes[cw + wi] = code;
      }
      for (unsigned wi = lane; wi < nMw; wi += 64) nmask[mw + wi] = 0;
      uint8_t* keep = pseudo_seq + size_t(nPseudo) * P.max_contig_len;
      for (unsigned i = lane; i < len; i += 64) keep[i] = src[i];
      if (lane == 0) {
        rd_cw[r]            = cw;
        rd_len[r]           = len;
        rd_mw[r]          = mw;
        rd_hasn[r]        = 0;
        pseudo_len[nPseudo] = len;
      }
      cw += nCw;
      mw += nMw;
      nPseudo++;
    }
    wv::sync();
    return nPseudo;
  }

  // ------------------------------------------------------------------------------------------------
  // selectContigs (:722-842) + output
  // ------------------------------------------------------------------------------------------------
  WV_DEV_COLD void selectAndEmit(const unsigned locus, const unsigned nPseudoFinal, const unsigned nIter)
  {
    const unsigned lane = unsigned(wv::lane());
    AsmLocusOut    out;
    out.status            = status;
    out.n_contigs         = 0;
    out.n_words           = W;
    out.n_pseudo          = 0;
    out.pseudo_off        = 0;
    out.pseudo_len_off    = 0;
    out.final_word_length = k;
    out.n_iterations      = nIter;
    out.cyclic_iterations = cyclicIters;
    // (introspection, manta_debug_repeat_words: size of the last word length's graph and the workspace slab it sits in)
    out.reserved          = (nNodes & 0x3ffffffu) | ((unsigned(wv::block()) & 63u) << 26);
    if (status != ASM_OK) {
      if (lane == 0) P.loci[locus] = out;  // (ASM_E_TABLE_FULL: the host runs the locus again on a worst-case workspace, api.cpp)
      return;
    }
    // index >= nNormal <=> pseudo read (see oracle/manta_oracle.cpp selectContigs on stale indices)
    const uint64_t pseudoMaskW = (lane < W) ? ~normalMask(lane) : 0;
    uint64_t       used        = 0;  // lane w owns word w
    uint64_t       alive       = (nCand >= 64) ? ~uint64_t(0) : ((uint64_t(1) << nCand) - 1);
    unsigned       finalCount  = 0;
    unsigned       chosenReg   = 0;  // lane f: the f-th chosen candidate (maxAssemblyCount <= 32 < 64)
    if (W <= WQ_MAX) {
      // Lane-per-candidate form of the same loop for read sets of up to WQ_MAX qwords (the usual case): lane ci keeps
      // candidate ci's support set and length in registers, the set of used reads is wave-uniform, and one selection round
      // is a handful of popcounts plus one wave arg-max instead of a dependent slab load + reduction per candidate.
      // Order of the reference's scan (:762-803): strict '>' on (fresh support, length) = first index wins a tie.
      uint64_t sup[WQ_MAX], usedS[WQ_MAX], pseudoS[WQ_MAX];
      unsigned myLen = 0;
      for (unsigned w = 0; w < WQ_MAX; ++w) {
        sup[w]     = (lane < nCand && w < W) ? cand_bits[size_t(lane) * 2 * W + w] : 0;
        usedS[w]   = 0;
        pseudoS[w] = (w < W) ? ~normalMask(w) : 0;
      }
      if (lane < nCand) myLen = unsigned(cand_meta[lane * 4 + 0]);
      bool aliveL = lane < nCand;
      while (finalCount < P.opt.maxAssemblyCount) {
        if (!wv::any(aliveL)) break;
        unsigned usedNormal = 0;
        for (unsigned w = 0; w < WQ_MAX; ++w) usedNormal += unsigned(wv::popc(usedS[w] & ~pseudoS[w]));
        if (nNormal - usedNormal < P.opt.minUnusedReads) break;  // :750
        unsigned nFresh = 0, nFreshNormal = 0;
        for (unsigned w = 0; w < WQ_MAX; ++w) {
          const uint64_t fresh = sup[w] & ~usedS[w];
          nFresh += unsigned(wv::popc(fresh));
          nFreshNormal += unsigned(wv::popc(fresh & ~pseudoS[w]));
        }
        if (aliveL && nFreshNormal < P.opt.minSupportReads) aliveL = false;  // :779-788
        uint64_t key = aliveL ? ((uint64_t(nFresh) << 40) | (uint64_t(myLen) << 8) | uint64_t(63u - lane)) : 0;
        for (int off = 1; off < 64; off <<= 1) {
          const uint64_t o = wv::shfl(key, wv::lane() ^ off);
          key              = (o > key) ? o : key;
        }
        if ((key >> 40) == 0) break;  // :807
        const int selected = wv::first(int(63u - unsigned(key & 63u)));
        if (lane == finalCount) chosenReg = unsigned(selected);
        if (int(lane) == selected) aliveL = false;
        for (unsigned w = 0; w < WQ_MAX; ++w) usedS[w] |= wv::shfl(sup[w], selected);
        finalCount++;
      }
      alive = 0;  // (the general loop below is skipped)
    }
    while (alive != 0 && finalCount < P.opt.maxAssemblyCount) {
      const unsigned usedAll      = waveSum(unsigned(wv::popc(used)));
      const unsigned usedPseudo   = waveSum(unsigned(wv::popc(used & pseudoMaskW)));
      const unsigned unusedNormal = nNormal - (usedAll - usedPseudo);
      if (unusedNormal < P.opt.minUnusedReads) break;  // :750 (a `return`; nothing follows the loop anyway)
      int      selected   = -1;
      unsigned maxSupport = 0, maxLength = 0;
      for (unsigned ci = 0; ci < nCand; ++ci) {
        if (!((alive >> ci) & 1)) continue;
        const uint64_t sup    = (lane < W) ? cand_bits[size_t(ci) * 2 * W + lane] : 0;
        const uint64_t fresh  = sup & ~used;
        const uint64_t pk     = packedCountSum(uint64_t(wv::popc(fresh)) | (uint64_t(wv::popc(fresh & ~pseudoMaskW)) << 16));
        const unsigned nFresh = unsigned(pk) & 0xffffu, nFreshNormal = unsigned(pk >> 16) & 0xffffu;
        if (nFreshNormal < P.opt.minSupportReads) {  // :779-788
          alive &= ~(uint64_t(1) << ci);
          continue;
        }
        const unsigned len = unsigned(cand_meta[ci * 4 + 0]);
        if ((nFresh > maxSupport) || ((nFresh == maxSupport) && (len > maxLength))) {  // :794-801
          selected   = int(ci);
          maxSupport = nFresh;
          maxLength  = len;
        }
      }
      if (maxSupport == 0) break;  // :807
      if (lane == finalCount) chosenReg = unsigned(selected);
      alive &= ~(uint64_t(1) << unsigned(selected));
      if (lane < W) used |= cand_bits[size_t(selected) * 2 * W + lane];
      finalCount++;
    }

    // ---- reserve output space ----
    // lane f holds the f-th chosen contig's numbers, lane p the p-th pseudo read's length: the loops below run on register
    // exchanges instead of a chain of dependent slab loads per contig
    const bool     isMine = lane < finalCount;
    const unsigned myLen  = isMine ? unsigned(cand_meta[chosenReg * 4 + 0]) : 0u;
    const int      myCb   = isMine ? cand_meta[chosenReg * 4 + 1] : 0;
    const int      myCe   = isMine ? cand_meta[chosenReg * 4 + 2] : 0;
    unsigned       lenScan = myLen;
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned o = wv::shfl(lenScan, int(lane) - off);
      if (int(lane) >= off) lenScan += o;
    }
    const uint64_t seqBytes = wv::shfl(lenScan, 63);
    uint64_t       pseudoBytes = 0;
    for (unsigned p0 = 0; p0 < nPseudoFinal; p0 += 64) pseudoBytes += waveSum((p0 + lane < nPseudoFinal) ? pseudo_len[p0 + lane] : 0u);
    const uint64_t bitsWords = uint64_t(finalCount) * 2 * W + nPseudoFinal;
    unsigned long long seqBase = 0, bitsBase = 0;
    if (lane == 0) {
      seqBase  = wv::atomic_add(P.seq_used, (unsigned long long)(seqBytes + pseudoBytes));
      bitsBase = wv::atomic_add(P.bits_used, (unsigned long long)(bitsWords));
    }
    seqBase  = wv::readlane(uint64_t(seqBase), 0);
    bitsBase = wv::readlane(uint64_t(bitsBase), 0);
    if (seqBase + seqBytes + pseudoBytes > P.seq_cap || bitsBase + bitsWords > P.bits_cap) {
      out.status = ASM_E_OUT_CAPACITY;
      if (lane == 0) P.loci[locus] = out;
      return;
    }
    if (isMine) {  // every contig's record at once
      AsmContigOut c;
      c.seq_off    = seqBase + (lenScan - myLen);
      c.bits_off   = bitsBase + uint64_t(lane) * 2 * W;
      c.seq_len    = myLen;
      c.cons_begin = myCb;
      c.cons_end   = myCe;
      c.reserved   = 0;
      P.contigs[size_t(locus) * P.opt.maxAssemblyCount + lane] = c;
    }
    uint64_t so = seqBase, bo = bitsBase;
    for (unsigned f = 0; f < finalCount; ++f) {
      const unsigned ci  = wv::readlane(chosenReg, int(f));
      const unsigned len = wv::readlane(myLen, int(f));
      const uint8_t* src = cand_seq + size_t(ci) * P.max_contig_len;
      for (unsigned i = lane; i < len; i += 64) P.seq_arena[so + i] = src[i];
      if (lane < 2 * W) P.bits_arena[bo + lane] = cand_bits[size_t(ci) * 2 * W + lane];
      so += len;
      bo += 2 * W;
    }
    out.pseudo_off     = so;
    out.pseudo_len_off = bo;
    for (unsigned p = 0; p < nPseudoFinal; ++p) {
      const unsigned len = pseudo_len[p];
      const uint8_t* src = pseudo_seq + size_t(p) * P.max_contig_len;
      for (unsigned i = lane; i < len; i += 64) P.seq_arena[so + i] = src[i];
      if (lane == 0) P.bits_arena[bo + p] = len;
      so += len;
    }
    out.n_contigs = finalCount;
    out.n_pseudo  = nPseudoFinal;
    if (lane == 0) P.loci[locus] = out;
  }

  // ------------------------------------------------------------------------------------------------
  // runIterativeAssembler (:844-931)
  // ------------------------------------------------------------------------------------------------
  WV_DEV void run(const unsigned locus)
  {
    const unsigned lane = unsigned(wv::lane());
    for (int i = 0; i < 8; ++i) tPhase[i] = 0;
    tMark               = wv::clock();
    cyclicIters         = 0;
    nCand               = 0;
    const unsigned minWL = P.locus_min_wl ? P.locus_min_wl[locus] : P.opt.minWordLength;
    const unsigned maxWL = P.locus_max_wl ? P.locus_max_wl[locus] : P.opt.maxWordLength;
    k                   = minWL;
    // zero the N bitmap region (filled with atomic_or)
    for (unsigned i = lane; i < maskWordCap() + 2; i += 64) nmask[i] = 0;
    wv::sync();
    packNormalReads(locus);
    wv::sync();
    wv::fence_acquire();
    if (SB == 2 && countJunkReads() >= P.opt.minCoverage && status == ASM_OK) status = ASM_E_ALPHABET;  // see packNormalReads
    tick(0);
    W = (nNormal + 2 * P.opt.maxAssemblyCount + 63) / 64;
    if (W == 0) W = 1;
    recStride = asmRecStride(W);
    if (status == ASM_OK && (maxWL > MAX_K || minWL == 0)) status = ASM_E_WORD_TOO_LONG;
    if (status == ASM_OK && 2 * P.opt.maxAssemblyCount > ASM_MAX_CAND) status = ASM_E_INTERNAL;

    nReads            = nNormal;
    unsigned nPseudo  = 0;
    unsigned nIter    = 0;
    if (status == ASM_OK) {
      for (unsigned wl = minWL; wl <= maxWL; wl += P.opt.wordStepSize) {
        k = wl;
        nIter++;
        const bool ok = buildContigsForK();
        if (status != ASM_OK) break;
        if (ok) break;  // :872-877
        // drop the previous pseudo reads, add this iteration's contigs (:882-910)
        nPseudo = appendPseudoReads();
        if (status != ASM_OK) break;
        nReads = nNormal + nPseudo;
      }
    }
    selectAndEmit(locus, nPseudo, nIter);
    tick(7);
#ifdef MANTA_ASM_PROFILE
    if (P.phase_cycles && lane == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) wv::atomic_add(&P.phase_cycles[i], (unsigned long long)tPhase[i]);
    }
#endif
  }
};

typedef AssemblerT<2> Assembler;  ///< the production form: 2-bit codes

}  // namespace manta_dev

#include "repeat_exact.hpp"
#include "walk_lanes.hpp"

namespace manta_dev {

/// streamed upload: wait until the chunk holding `locus` has landed.  Polls a system-scope counter with back-off; gives
/// up after ~1 minute of shader clocks (a copy that never completes must not hang the device for good).  The host side
/// never runs two streamed assemblers on one device at once (api.cpp: streamedAsmMu), so a copy always finds a free workgroup slot.
WV_DEV bool asmWaitUploaded(const AsmParams& P, const unsigned locus)
{
  const unsigned need = locus / P.chunk_loci + 1;
  bool           ok   = true;
  if (wv::lane() == 0) {
    const uint64_t t0 = wv::clock();
    while (wv::atomic_load_system(P.upload_chunks_done) < need) {
      wv::sleep();
      if (wv::clock() - t0 > 150000000000ull) {
        ok = false;
        break;
      }
    }
  }
  ok = wv::first(int(ok)) != 0;
  wv::sync();
  return ok;
}

#ifndef MANTA_ASM_OCC
#define MANTA_ASM_OCC 4
#endif
#if !MANTA_TU_DEFINES(MANTA_TU_ASM)
WV_KERNEL_OCC(MANTA_ASM_OCC) void assemble_kernel(const AsmParams P);
#else
WV_KERNEL_OCC(MANTA_ASM_OCC) void assemble_kernel(const AsmParams P)
{
  uint8_t* wsBase = P.ws + uint64_t(wv::block()) * P.ws_stride;
  const unsigned nLoci = P.n_loci_dev ? wv::first(wv::atomic_load(P.n_loci_dev)) : P.n_loci;
  while (true) {
    unsigned slot = 0;
    if (wv::lane() == 0) {
      slot = (P.stop_before && wv::atomic_load(P.counter) >= P.stop_before) ? nLoci : wv::atomic_add(P.counter, 1u);
    }
    slot = wv::first(slot);
    if (slot >= nLoci) break;
    const unsigned locus = P.locus_ids ? P.locus_ids[slot] : slot;
    Assembler      a(P, wsBase);
    // (single reconvergence point per work item: both outcomes fall through to the sync below)
    const bool arrived = !P.upload_chunks_done || asmWaitUploaded(P, locus);
    if (arrived) {
      a.run(locus);
    } else if (wv::lane() == 0) {  // the chunk never arrived: report, do not hang
      AsmLocusOut out;
      out.status = ASM_E_INTERNAL;
      out.n_contigs = out.n_words = out.n_pseudo = 0;
      out.pseudo_off = out.pseudo_len_off = 0;
      out.final_word_length = out.n_iterations = out.cyclic_iterations = out.reserved = 0;
      P.loci[locus] = out;
    }
    wv::sync();
  }
}
#endif

#ifndef MANTA_DEV_NO_GENERIC  // (developer variants of the library leave the byte-generic kernel out: it is most of the compile time)
/// the byte-generic form (AssemblerT<8>) for the loci assemble_kernel reports ASM_E_ALPHABET for: same launch contract; the
/// workspace capacities count code dwords of 4 symbols.  Rare by construction (a byte outside {A,C,G,T,N} that cannot be
/// masked exactly), so this kernel is about being right, not fast.
#if !MANTA_TU_DEFINES(MANTA_TU_ASM_GENERIC)
WV_KERNEL_OCC(MANTA_ASM_OCC) void assemble_generic_kernel(const AsmParams P);
#else
WV_KERNEL_OCC(MANTA_ASM_OCC) void assemble_generic_kernel(const AsmParams P)
{
  uint8_t*       wsBase = P.ws + uint64_t(wv::block()) * P.ws_stride;
  const unsigned nLoci  = P.n_loci_dev ? wv::first(wv::atomic_load(P.n_loci_dev)) : P.n_loci;
  while (true) {
    unsigned slot = 0;
    if (wv::lane() == 0) slot = wv::atomic_add(P.counter, 1u);
    slot = wv::first(slot);
    if (slot >= nLoci) break;
    const unsigned locus = P.locus_ids ? P.locus_ids[slot] : slot;
    AssemblerT<8>  a(P, wsBase);
    const bool     arrived = !P.upload_chunks_done || asmWaitUploaded(P, locus);
    if (arrived) {
      a.run(locus);
    } else if (wv::lane() == 0) {
      AsmLocusOut out;
      out.status = ASM_E_INTERNAL;
      out.n_contigs = out.n_words = out.n_pseudo = 0;
      out.pseudo_off = out.pseudo_len_off = 0;
      out.final_word_length = out.n_iterations = out.cyclic_iterations = out.reserved = 0;
      P.loci[locus] = out;
    }
    wv::sync();
  }
}
#endif
#endif

}  // namespace manta_dev
