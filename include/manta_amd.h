/*
 * manta_amd.h -- C ABI of the MI355X-native assemble+align hot path of Manta's GenerateSVCandidates.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  Each
 * entry point names the reference interface it replaces (paths relative to /root/reference/src/c++/lib).
 * The C++ adapters in manta_amd/host/ re-create the reference's own signatures
 * (runIterativeAssembler, GlobalAligner<int>::align, ...) on top of these functions; INTEGRATION.md shows the
 * edits a Manta maintainer makes to call them.
 *
 * Conventions
 *   - every function returns MANTA_OK (0) or a negative MANTA_E_* code; manta_last_error() gives the message
 *     (the reference reports errors as C++ exceptions, common/Exceptions.hpp:54-85; the C++ adapter re-throws).
 *   - sequences are raw bytes exactly as the reference's std::string holds them (1 byte per base).
 *   - all calls are synchronous; a context is bound to one GPU and is NOT re-entrant (the reference's
 *     aligner/refiner objects are not either: alignment/GlobalJumpAligner.hpp:117-124) -- use one context per
 *     host worker thread (GenerateSVCandidates.cpp:232-250).
 *   - there is no CPU fallback: if no gfx950 device / HIP runtime is usable, manta_ctx_create fails.
 */
#ifndef MANTA_AMD_H
#define MANTA_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MANTA_OK 0
#define MANTA_E_INVALID_ARG (-1)  /* null pointer, bad enum, inconsistent sizes */
#define MANTA_E_NO_DEVICE (-2)    /* HIP runtime / device unavailable */
#define MANTA_E_HIP (-3)          /* a HIP call failed; see manta_last_error */
#define MANTA_E_EMPTY_SEQ (-4)    /* empty query/reference (GlobalJumpAlignerImpl.hpp:50-58) */
#define MANTA_E_UNSUPPORTED (-5)  /* input outside the supported envelope (see DESIGN.md limits) */
#define MANTA_E_CAPACITY (-6)     /* caller-provided output arena too small */
#define MANTA_E_DEVICE_FAULT (-7) /* kernel reported an internal overflow for some locus/task */
#define MANTA_E_SPLIT_QUERY_NOT_SHORTER (-8) /* splitReadAligner: querySize >= targetSize (SplitReadAlignment.cpp:237-247) */
#define MANTA_E_SPLIT_EMPTY_SCAN (-9)        /* splitReadAligner: scanEnd < scanStart (SplitReadAlignment.cpp:265-273) */
#define MANTA_E_NOT_TAKEN (-10)   /* whole-batch call on a shared block queue: another process of the node took this locus' block */

typedef struct manta_ctx manta_ctx_t;

/* device_id < 0 selects the current HIP device */
int         manta_ctx_create(int device_id, manta_ctx_t** ctx);
void        manta_ctx_destroy(manta_ctx_t* ctx);
const char* manta_last_error(const manta_ctx_t* ctx); /* ctx may be NULL: error of the last failed create */
/* "gfx950 / <CUs> CUs / <bytes> HBM" style description, for logs */
const char* manta_ctx_device_name(const manta_ctx_t* ctx);

/* ------------------------------------------------------------------------------------------------------
 * Aligners.  Replaces
 *   GlobalAligner<int>::align            alignment/GlobalAligner.hpp:36-46   (kind MANTA_ALIGNER_GLOBAL)
 *   GlobalLargeIndelAligner<int>::align  alignment/GlobalLargeIndelAligner.hpp:39-54 (MANTA_ALIGNER_LARGE_INDEL)
 *   GlobalJumpAligner<int>::align        alignment/GlobalJumpAligner.hpp:36-53       (MANTA_ALIGNER_JUMP)
 * called at SVCandidateAssemblyRefiner.cpp:933, 1668, 1706, 2032.
 * ---------------------------------------------------------------------------------------------------- */
enum { MANTA_ALIGNER_GLOBAL = 0, MANTA_ALIGNER_LARGE_INDEL = 1, MANTA_ALIGNER_JUMP = 2 };

/* alignment/AlignmentScores.hpp:23-57 */
typedef struct {
  int32_t match, mismatch, open, extend, off_edge;
  int32_t is_allow_edge_insertion; /* must be 0 for MANTA_ALIGNER_JUMP (GlobalJumpAligner.hpp:41-42) */
} manta_align_scores_t;

/* one query/reference(s) problem; offsets index the caller's sequence arena */
typedef struct {
  uint64_t query_off;
  uint64_t ref1_off;
  uint64_t ref2_off; /* MANTA_ALIGNER_JUMP only */
  uint32_t query_len;
  uint32_t ref1_len;
  uint32_t ref2_len;
  uint32_t reserved;
} manta_align_task_t;

/* CIGAR segments are BAM-packed: (length << 4) | op, op = 0 M,1 I,2 D,3 N,4 S,5 H,6 P,7 '=',8 'X'
 * (ALIGNPATH::align_t, blt_util/align_path.hpp:35-62).  Paths are always in '='/'X' form, as the reference's
 * aligners emit them (SingleRefAlignerSharedImpl.hpp:160-166). */
typedef struct {
  int32_t  status;    /* MANTA_OK or MANTA_E_* for this task */
  int32_t  score;     /* AlignmentResult::score / JumpAlignmentResult::score */
  int32_t  is_jumped; /* AlignmentResult::isJumped (single-reference aligners) */
  int32_t  begin_pos1; /* align.beginPos / align1.beginPos */
  int32_t  begin_pos2; /* align2.beginPos (jump) */
  uint32_t jump_insert_size; /* JumpAlignmentResult::jumpInsertSize */
  uint32_t jump_range;       /* JumpAlignmentResult::jumpRange */
  uint32_t cigar1_len, cigar2_len; /* number of segments */
  uint64_t cigar1_off, cigar2_off; /* into the caller's cigar arena */
} manta_align_result_t;

/* extra_score = largeIndelScore (LARGE_INDEL) or jumpScore (JUMP); ignored for GLOBAL.
 * cigar arena: 2*query_len+8 segments per task always suffice. */
int manta_align_batch(
    manta_ctx_t* ctx, int kind, const manta_align_scores_t* scores, int32_t extra_score, uint32_t n_tasks,
    const manta_align_task_t* tasks, const uint8_t* seq_arena, uint64_t seq_arena_bytes,
    manta_align_result_t* results, uint32_t* cigar_arena, uint64_t cigar_arena_cap, uint64_t* cigar_arena_used);

/* ------------------------------------------------------------------------------------------------------
 * Assembler.  Replaces
 *   void runIterativeAssembler(const IterativeAssemblerOptions&, AssemblyReadInput& reads,
 *                              AssemblyReadOutput& readInfo, Assembly& contigs)
 *                                                   assembly/IterativeAssembler.hpp:43-47
 * called at manta/SVCandidateAssembler.cpp:674,697, for a whole batch of candidate loci at once.
 * ---------------------------------------------------------------------------------------------------- */

/* options/IterativeAssemblerOptions.hpp:26-59.  The alphabet is the reference's default "ACGT"; minQval and
 * maxError are not read by the assembler itself (they act upstream, in read gathering). */
typedef struct {
  uint32_t min_word_length, max_word_length, word_step_size, min_contig_length;
  uint32_t min_coverage, min_conservative_coverage, min_unused_reads, min_support_reads, max_assembly_count;
} manta_asm_options_t;

/* assembly/AssembledContig.hpp:38-54 */
typedef struct {
  uint64_t seq_off;     /* contig.seq in the caller's sequence arena */
  uint64_t support_off; /* contig.supportReads as a bitset of n_words qwords in the caller's bitset arena */
  uint64_t reject_off;  /* contig.rejectReads, same form */
  uint32_t seq_len;
  uint32_t seed_read_count; /* always 0: never written by the reference (AssembledContig.hpp:45) */
  int32_t  conservative_begin, conservative_end; /* contig.conservativeRange */
} manta_asm_contig_t;

typedef struct {
  int32_t  status;       /* MANTA_OK or MANTA_E_* for this locus */
  uint32_t n_contigs;    /* contigs.size() */
  uint32_t first_contig; /* index of this locus' first record in the contigs array */
  uint32_t n_words;      /* qwords per read bitset: ceil((reads + 2*max_assembly_count)/64).  Bit r == read r;
                            r >= n_reads are pseudo reads (contigs fed back as reads, IterativeAssembler.cpp:897-910) */
  uint32_t n_pseudo;     /* pseudo reads the reference leaves appended to `reads` on return (:902) */
  uint32_t final_word_length;
  uint32_t n_iterations;
  uint32_t cyclic_iterations; /* word lengths whose k-mer graph was cyclic (exact repeat search taken) */
  uint64_t pseudo_seq_off;    /* n_pseudo sequences back to back in the sequence arena */
  uint64_t pseudo_len_off;    /* their lengths, one qword each, in the bitset arena */
} manta_asm_locus_result_t;

/* Input layout (what SVCandidateAssembler hands to runIterativeAssembler, flattened):
 *   bases              all reads of all loci, 1 byte per base {A,C,G,T,N}
 *   read_off[R+1]      byte offsets of the reads
 *   locus_read_begin[n_loci+1]  read-index range of every locus
 * Output: one manta_asm_locus_result_t per locus, at most max_assembly_count contig records per locus.
 * readInfo (AssemblyReadInfo.hpp:31-46) is implied: read r isUsed <=> some contig's support holds r, its contigIds
 * are those contigs in order (IterativeAssembler.cpp:826-834); the C++ adapter materialises it. */
int manta_assemble_batch(
    manta_ctx_t* ctx, const manta_asm_options_t* opt, uint32_t n_loci, const uint8_t* bases, const uint64_t* read_off,
    const uint32_t* locus_read_begin, manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs, uint64_t contigs_cap,
    uint8_t* seq_arena, uint64_t seq_arena_cap, uint64_t* seq_arena_used, uint64_t* bits_arena, uint64_t bits_arena_cap,
    uint64_t* bits_arena_used);

/* Introspection for tests: the repeat-word set of getRepeatKmers (assembly/IterativeAssembler.cpp:627-642) as the
 * device computed it for ONE read pile at word length opt->min_word_length -- the observable of the reference's
 * test_CircleDetector (assembly/test/IterativeAssemblerTest.cpp:30-61).  `out` receives the words, sorted, one per
 * line; *n_words their number.  Not a production entry point. */
int manta_debug_repeat_words(
    manta_ctx_t* ctx, const manta_asm_options_t* opt, uint32_t n_reads, const uint8_t* bases, const uint64_t* read_off, char* out,
    uint64_t out_cap, uint32_t* n_words);

/* ------------------------------------------------------------------------------------------------------
 * SmallAssembler.  Replaces
 *   void runSmallAssembler(const SmallAssemblerOptions&, const AssemblyReadInput& reads,
 *                          AssemblyReadOutput& assembledReadInfo, Assembly& contigs)
 *                                                   assembly/SmallAssembler.hpp:43-47 (SmallAssembler.cpp:622-685)
 * for a batch of read piles.  (The reference has no production caller for this assembler; its unit tests are
 * assembly/test/SmallAssemblerTest.cpp.)  Input layout and output records as manta_assemble_batch, with
 *   - n_words = ceil(reads / 64), no pseudo reads;
 *   - seed_read_count = contig.seedReadCount (SmallAssembler.cpp:577-580);
 *   - n_contigs = contigs.size() + 1: the LAST record of a locus (seq_len 0, seed_read_count 0xffffffff) carries the reads the
 *     reference marks isUsed && isFiltered (repeat reads at the last word length, :496-503) in its support bitset.
 *   readInfo: read r isUsed <=> it is in that set or in some contig's support; contigIds = {the one contig that holds it}.
 * ---------------------------------------------------------------------------------------------------- */

/* options/SmallAssemblerOptions.hpp:24-56; alphabet "ACGT"; minQval / maxError / minContigLength are not read by the assembler */
typedef struct {
  uint32_t min_word_length, max_word_length, word_step_size, min_contig_length;
  uint32_t min_coverage, min_conservative_coverage, min_seed_reads, max_assembly_iterations; /* <= 31 */
} manta_small_asm_options_t;

int manta_small_assemble_batch(
    manta_ctx_t* ctx, const manta_small_asm_options_t* opt, uint32_t n_loci, const uint8_t* bases, const uint64_t* read_off,
    const uint32_t* locus_read_begin, manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs, uint64_t contigs_cap,
    uint8_t* seq_arena, uint64_t seq_arena_cap, uint64_t* seq_arena_used, uint64_t* bits_arena, uint64_t bits_arena_cap,
    uint64_t* bits_arena_used);

/* ------------------------------------------------------------------------------------------------------
 * Fused "small SV" locus pipeline: the arithmetic core of
 *   SVCandidateAssemblyRefiner::getSmallSVAssembly   applications/GenerateSVCandidates/SVCandidateAssemblyRefiner.cpp:1860-2038
 * for a batch of candidate loci:  runIterativeAssembler (:1921-1926 via SVCandidateAssembler.cpp:661-675)
 *   -> per contig 10-mer reference trim (:1984-2011) -> GlobalLargeIndelAligner::align (:2032-2038)
 *   -> alignment.beginPos += adjustedLeadingCut (:2039).
 * Contigs stay on the device between the stages.  The staged form (upload / run / download) lets a caller keep
 * inputs resident in HBM and overlap host work; manta_smallsv_run is the timed region of bench.py.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct manta_smallsv manta_smallsv_t;

/* the *Cut values of SVCandidateAssemblyRefiner.cpp:1912-1915, per locus */
typedef struct {
  int32_t leading_cut, trailing_cut, max_leading_cut, max_trailing_cut;
} manta_ref_cuts_t;

typedef struct {
  int32_t              adjusted_leading_cut, adjusted_trailing_cut; /* :1994-2010 */
  manta_align_result_t align; /* begin_pos1 already includes adjusted_leading_cut (:2039) */
} manta_smallsv_alignment_t;

typedef struct {
  float    assemble_ms, schedule_ms, align_ms, total_ms; /* HIP-event times of the last run, per stage */
  uint32_t n_align_launches;
  uint32_t n_alignments;
  uint64_t ptr_matrix_bytes; /* back-pointer bytes the alignments of the last run wrote (private layout) */
  uint64_t dp_cells;         /* sum over alignments of query_len * ref_len */
} manta_smallsv_stats_t;

int  manta_smallsv_create(manta_ctx_t* ctx, const manta_asm_options_t* opt, const manta_align_scores_t* scores,
                          int32_t large_indel_score, manta_smallsv_t** out);
void manta_smallsv_destroy(manta_smallsv_t* b);
/* host -> HBM.  refs/ref_off[n_loci+1]: the fetched reference window of every locus; cuts[n_loci] */
int manta_smallsv_upload(manta_smallsv_t* b, uint32_t n_loci, const uint8_t* bases, const uint64_t* read_off,
                         const uint32_t* locus_read_begin, const uint8_t* refs, const uint64_t* ref_off,
                         const manta_ref_cuts_t* cuts);
/* all three stages on the device, synchronous; inputs must have been uploaded */
int manta_smallsv_run(manta_smallsv_t* b);
int manta_smallsv_stats(const manta_smallsv_t* b, manta_smallsv_stats_t* stats);
/* sizes that manta_smallsv_download will need after this run (upper bounds): contig records, sequence bytes, bitset
 * qwords, cigar u32 words -- so that the caller can size its arenas exactly */
int manta_smallsv_output_sizes(const manta_smallsv_t* b, uint64_t* contigs, uint64_t* seq_bytes, uint64_t* bits_words,
                               uint64_t* cigar_words);
/* HBM -> host.  alignments[i] belongs to contigs[i]. */
int manta_smallsv_download(manta_smallsv_t* b, manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs,
                           manta_smallsv_alignment_t* alignments, uint64_t contigs_cap, uint8_t* seq_arena,
                           uint64_t seq_arena_cap, uint64_t* seq_arena_used, uint64_t* bits_arena, uint64_t bits_arena_cap,
                           uint64_t* bits_arena_used, uint32_t* cigar_arena, uint64_t cigar_arena_cap,
                           uint64_t* cigar_arena_used);

/* ------------------------------------------------------------------------------------------------------
 * Fused "spanning" locus pipeline: the arithmetic core of
 *   SVCandidateAssemblyRefiner::getJumpAssembly      applications/GenerateSVCandidates/SVCandidateAssemblyRefiner.cpp:1745-1849
 * for a batch of breakend-pair candidates (DNA):  runIterativeAssembler (via assembleJumpContigs :1504-1511 ->
 * manta/SVCandidateAssembler.cpp:677-698) -> alignJumpContigs (:1525-1743): GlobalJumpAligner::align of every contig
 * against the CUT references (:1663-1670), the re-align rule on the uncut references (:1672-1713, including its
 * carry-over to the later contigs of the locus), and beginPos += leading cut (:1716-1717).
 * Inputs per locus are what alignJumpContigs holds after its orientation step (:1533-1550): the two reference strings
 * in alignment order (align1RefStr / align2RefStr, reverse-complemented where the breakend is reversed) and the four
 * cuts in the same order.  Same staging and download conventions as manta_smallsv_*.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct manta_spanning manta_spanning_t;

typedef struct {
  int32_t align1_leading_cut, align1_trailing_cut, align2_leading_cut, align2_trailing_cut; /* AlignData :1400-1411 */
} manta_jump_cuts_t;

typedef struct {
  int32_t              is_uncut; /* 1: the alignment is the re-alignment against the uncut references (:1699-1712) */
  int32_t              reserved;
  manta_align_result_t align;    /* begin_pos1/2 already include the leading cuts that applied (:1716-1717) */
} manta_spanning_alignment_t;

int  manta_spanning_create(manta_ctx_t* ctx, const manta_asm_options_t* opt, const manta_align_scores_t* scores,
                           int32_t jump_score, manta_spanning_t** out);
void manta_spanning_destroy(manta_spanning_t* b);
int  manta_spanning_upload(manta_spanning_t* b, uint32_t n_loci, const uint8_t* bases, const uint64_t* read_off,
                           const uint32_t* locus_read_begin, const uint8_t* refs1, const uint64_t* ref1_off,
                           const uint8_t* refs2, const uint64_t* ref2_off, const manta_jump_cuts_t* cuts);
int  manta_spanning_run(manta_spanning_t* b);
/* align_ms covers both alignment rounds and the re-align decision kernel */
int  manta_spanning_stats(const manta_spanning_t* b, manta_smallsv_stats_t* stats);
int  manta_spanning_output_sizes(const manta_spanning_t* b, uint64_t* contigs, uint64_t* seq_bytes, uint64_t* bits_words,
                                 uint64_t* cigar_words);
int  manta_spanning_download(manta_spanning_t* b, manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs,
                             manta_spanning_alignment_t* alignments, uint64_t contigs_cap, uint8_t* seq_arena,
                             uint64_t seq_arena_cap, uint64_t* seq_arena_used, uint64_t* bits_arena, uint64_t bits_arena_cap,
                             uint64_t* bits_arena_used, uint32_t* cigar_arena, uint64_t cigar_arena_cap,
                             uint64_t* cigar_arena_used);

/* ------------------------------------------------------------------------------------------------------
 * Split-read scoring.  Replaces
 *   void splitReadAligner(flankScoreSize, querySeq, qualConvert, queryQual, targetSeq, targetBpOffsetRange, alignment)
 *                               applications/GenerateSVCandidates/SplitReadAlignment.hpp:57-64 (.cpp:223-350)
 * called per read and per allele from SVScorerSplit.cpp (scoreSplitReads), for a whole batch of (read, target) pairs.
 * The two tables are the caller's own qscore_snp (blt_util/qscore_snp.hpp:33-55: qphred_to_ln_comp_error_prob and
 * qphred_to_ln_error_prob for q = 0..MAX_QSCORE), ln_one_third = std::log(1/3.f) and ln_random_base = -std::log(4.f) as the
 * reference computes them (:50, :76): passed in, so the sums are formed from the caller's exact values.
 * Float policy: best_ln_lhood is BIT-IDENTICAL to the reference's alignLnLhood (same operations in the same order: a float
 * accumulator, each term added in double and rounded back, N terms added in float); best_pos follows the reference's
 * strict-'>' scan.  The remaining SRAlignmentInfo fields (alignScore, isEvidence, evidence: three float ratio tests) are
 * scalar glue on the sizes and mismatch counts returned here; manta_amd/host/split_read.hpp forms them with the
 * reference's expressions.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct {
  uint64_t query_off, qual_off, target_off; /* into the caller's arena; qual = query_len basecall qualities */
  uint32_t query_len, target_len;
  int32_t  bp_begin, bp_end;  /* targetBpOffsetRange */
  uint32_t flank_score_size;
  uint32_t reserved;
} manta_split_task_t;

typedef struct {
  int32_t  status;        /* MANTA_OK, MANTA_E_SPLIT_* (the reference throws), MANTA_E_UNSUPPORTED (quality above the table) */
  uint32_t best_pos;      /* SRAlignmentInfo::alignPos */
  float    best_ln_lhood; /* SRAlignmentInfo::alignLnLhood */
  uint32_t left_size, hom_size, right_size; /* :309-337; left_size > query_len is the reference's "unexpected outcome" throw */
  uint32_t left_mismatches, hom_mismatches, right_mismatches; /* calculateAlignScore :95-121 */
} manta_split_result_t;

int manta_split_read_batch(
    manta_ctx_t* ctx, const double* ln_comp_error_prob, const double* ln_error_prob, uint32_t n_qscores, float ln_one_third,
    float ln_random_base, uint32_t n_tasks, const manta_split_task_t* tasks, const uint8_t* arena, uint64_t arena_bytes,
    manta_split_result_t* results);

/* ------------------------------------------------------------------------------------------------------
 * Mixed word lengths in one batch (minWordLength drawn per locus).  Overrides
 * IterativeAssemblerOptions::minWordLength / maxWordLength (options/IterativeAssemblerOptions.hpp:38-40) per locus for
 * the NEXT upload of the pipeline; pass NULL, NULL to go back to the option block.
 * ---------------------------------------------------------------------------------------------------- */
int manta_smallsv_set_word_lengths(manta_smallsv_t* b, uint32_t n_loci, const uint32_t* min_word_length,
                                   const uint32_t* max_word_length);
int manta_spanning_set_word_lengths(manta_spanning_t* b, uint32_t n_loci, const uint32_t* min_word_length,
                                    const uint32_t* max_word_length);

/* ------------------------------------------------------------------------------------------------------
 * Packed read piles.  The read-pile builder (manta_amd/host/read_pile.hpp) turns what
 * SVCandidateAssembler::getBreakendReads keeps of every BAM record -- insertAssemblyRead, manta/SVCandidateAssembler.cpp:
 * 102-136: the 4-bit BAM sequence (htsapi/bam_seq.hpp:41-59) with Q < minQval masked to N, reverse-complemented where the
 * breakend is reversed -- directly into the layout the kernels pack into, instead of the reference's std::string per read:
 *   codes  2 bits per base (A,C,G,T = 0..3; 0 under an N), 16 bases per dword, first base in the top bits
 *   nmask  1 bit per base (1 = 'N'), 32 bases per dword, first base in bit 0
 * 0.375 byte per base over PCIe instead of 1, and the device's stage 0 becomes a copy.  Unused bits of a read's last
 * dwords must be zero.  Offsets are in dwords; read r of the batch owns ceil(len/16) code and ceil(len/32) mask dwords.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct {
  const uint32_t* codes;
  const uint32_t* nmask;
  const uint32_t* read_len;         /* [R] bases per read */
  const uint64_t* read_code_off;    /* [R+1] */
  const uint64_t* read_mask_off;    /* [R+1] */
  const uint32_t* locus_read_begin; /* [n_loci+1] read-index range of every locus */
} manta_packed_piles_t;

int manta_smallsv_upload_piles(manta_smallsv_t* b, uint32_t n_loci, const manta_packed_piles_t* piles, const uint8_t* refs,
                               const uint64_t* ref_off, const manta_ref_cuts_t* cuts);
int manta_spanning_upload_piles(manta_spanning_t* b, uint32_t n_loci, const manta_packed_piles_t* piles, const uint8_t* refs1,
                                const uint64_t* ref1_off, const uint8_t* refs2, const uint64_t* ref2_off,
                                const manta_jump_cuts_t* cuts);

/* page-locked host memory: input/output buffers allocated here are copied by DMA without a staging pass */
int  manta_host_alloc(uint64_t bytes, void** out);
void manta_host_free(void* p);

/* ------------------------------------------------------------------------------------------------------
 * Whole-batch calls: what a GenerateSVCandidates worker pool does with its edges
 * (GenerateSVCandidates.cpp:232-266 hands EdgeRetriever ranges to --threads workers, each with a private refiner;
 * EdgeRetrieverBin.cpp:38-57 splits the edge list into contiguous bins), here for one GPU: the batch is cut into
 * contiguous blocks of block_loci loci, the blocks go into a queue ordered by decreasing cost (reads x bases), and
 * n_workers host threads -- each with a private pipeline on its own HIP stream -- pull blocks:
 * upload -> kernels -> download.  The copies of one block overlap the kernels of the others, and the VALU-bound
 * aligner of one block runs beside the assembler of the next.  Synchronous: on return every result is in the caller's
 * records and arenas (same layout as manta_*_download; offsets are relative to the arena starts; loci[i] belongs to
 * input locus i, contig records of different blocks are not in locus order -- use first_contig).
 * Per-item failures (MANTA_E_UNSUPPORTED / MANTA_E_DEVICE_FAULT in a locus or alignment status) do not stop the batch:
 * the call returns the code, every other item is valid.
 * ---------------------------------------------------------------------------------------------------- */
#define MANTA_BATCH_SERIAL_KERNELS 1u /* one block's kernels at a time; uploads / downloads of the others still overlap them */
#define MANTA_BATCH_NO_STREAMED_UPLOAD 2u /* wait for a block's read bases before its kernels start (default: the assembler
                                             is launched at once and consumes the bases chunk by chunk as they land) */
typedef struct {
  uint32_t block_loci; /* 0 = automatic: the whole batch as one block when it fits the device-side budget (measured optimum,
                          DESIGN.md 5), otherwise equal blocks of at most 65536 loci / 4 GiB of read bases */
  uint32_t n_workers;  /* 0 = 1: blocks one after the other (measured optimum: two blocks' kernels sharing the device lose more
                          than the hidden transfers win).  With several workers at most one block assembles and
                          one block aligns at any time (stage gates); the others upload / download / compact meanwhile */
  uint32_t flags;      /* MANTA_BATCH_* */
  uint32_t reserved;
  uint32_t* shared_queue; /* nullable.  A 32-bit counter (zeroed before the calls) in memory SHARED by several processes of one node
                             (e.g. POSIX shared memory): every process makes the same call on the same batch, each with its own GPU,
                             and blocks are handed out through this counter instead of the call's own -- one work queue across the
                             node's GPUs for the one-process-per-GPU deployment.  A process fills the records and arenas of the
                             blocks it took; the loci of the others carry MANTA_E_NOT_TAKEN and the caller merges (block b covers
                             loci [b * block_loci, ...): set block_loci explicitly so that every process cuts the same blocks).
                             The library does not validate the counter: it must be ZERO before the first process calls and every
                             process must pass the same batch and block_loci.  The caller verifies coverage when it merges -- every
                             locus taken by exactly one process (a counter that was not reset, or a process that failed after taking
                             blocks, shows up as loci nobody filled: status MANTA_E_NOT_TAKEN in every process' records) */
} manta_batch_plan_t;

/* The ABI's version and the size of the statistics record the library writes.  manta_batch_stats_t has grown between rounds (the
   routing counters), and the library writes the whole record: a caller -- or a ctypes / cgo mirror of the struct -- compiled against
   an older header would be written past its end.  Check both once after loading the library:
     manta_abi_version() == MANTA_ABI_VERSION  and  manta_batch_stats_size() == sizeof(manta_batch_stats_t)
   (manta_amd/_capi.py does).  The version changes whenever a struct of this header changes size or meaning. */
#define MANTA_ABI_VERSION 6
uint32_t manta_abi_version(void);
uint64_t manta_batch_stats_size(void);

typedef struct {
  double   wall_ms;                    /* call entry -> all results host-visible */
  double   h2d_ms, kernel_ms, d2h_ms;  /* host wall time per phase, summed over blocks (blocks overlap) */
  float    assemble_ms, schedule_ms, align_ms; /* HIP-event times, summed over blocks */
  uint32_t n_blocks, n_workers;
  uint64_t n_alignments, n_align_launches, dp_cells, ptr_matrix_bytes;
  uint64_t h2d_bytes, d2h_bytes;       /* PCIe traffic of the call */
  /* how the assembler stage routed the loci: the LDS pipeline's two size classes (<= 128 / <= 256 reads), the loci they handed
     back to the general kernel (cyclic graph, next word length, capacity), the loci outside both envelopes (general kernel) */
  uint64_t n_loci_lds_small, n_loci_lds_big, n_loci_handed_back, n_loci_general;
} manta_batch_stats_t;

int manta_smallsv_batch(
    manta_ctx_t* ctx, const manta_asm_options_t* opt, const manta_align_scores_t* scores, int32_t large_indel_score,
    uint32_t n_loci, const uint8_t* bases, const uint64_t* read_off, const uint32_t* locus_read_begin, const uint8_t* refs,
    const uint64_t* ref_off, const manta_ref_cuts_t* cuts, const uint32_t* locus_min_word_length /* nullable */,
    const uint32_t* locus_max_word_length /* nullable */, manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs,
    manta_smallsv_alignment_t* alignments, uint64_t contigs_cap, uint8_t* seq_arena, uint64_t seq_arena_cap,
    uint64_t* seq_arena_used, uint64_t* bits_arena, uint64_t bits_arena_cap, uint64_t* bits_arena_used, uint32_t* cigar_arena,
    uint64_t cigar_arena_cap, uint64_t* cigar_arena_used, const manta_batch_plan_t* plan /* nullable */,
    manta_batch_stats_t* stats /* nullable */);

/* the same with the read piles in packed form */
int manta_smallsv_batch_piles(
    manta_ctx_t* ctx, const manta_asm_options_t* opt, const manta_align_scores_t* scores, int32_t large_indel_score,
    uint32_t n_loci, const manta_packed_piles_t* piles, const uint8_t* refs, const uint64_t* ref_off, const manta_ref_cuts_t* cuts,
    const uint32_t* locus_min_word_length /* nullable */, const uint32_t* locus_max_word_length /* nullable */,
    manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs, manta_smallsv_alignment_t* alignments, uint64_t contigs_cap,
    uint8_t* seq_arena, uint64_t seq_arena_cap, uint64_t* seq_arena_used, uint64_t* bits_arena, uint64_t bits_arena_cap,
    uint64_t* bits_arena_used, uint32_t* cigar_arena, uint64_t cigar_arena_cap, uint64_t* cigar_arena_used,
    const manta_batch_plan_t* plan /* nullable */, manta_batch_stats_t* stats /* nullable */);

int manta_spanning_batch(
    manta_ctx_t* ctx, const manta_asm_options_t* opt, const manta_align_scores_t* scores, int32_t jump_score, uint32_t n_loci,
    const uint8_t* bases, const uint64_t* read_off, const uint32_t* locus_read_begin, const uint8_t* refs1,
    const uint64_t* ref1_off, const uint8_t* refs2, const uint64_t* ref2_off, const manta_jump_cuts_t* cuts,
    const uint32_t* locus_min_word_length /* nullable */, const uint32_t* locus_max_word_length /* nullable */,
    manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs, manta_spanning_alignment_t* alignments, uint64_t contigs_cap,
    uint8_t* seq_arena, uint64_t seq_arena_cap, uint64_t* seq_arena_used, uint64_t* bits_arena, uint64_t bits_arena_cap,
    uint64_t* bits_arena_used, uint32_t* cigar_arena, uint64_t cigar_arena_cap, uint64_t* cigar_arena_used,
    const manta_batch_plan_t* plan /* nullable */, manta_batch_stats_t* stats /* nullable */);

/* the same with the read piles in packed form (what manta_read_piles_batch emits for the breakends of a spanning candidate:
 * SVCandidateAssembler.cpp:677-698 hands assembleSVBreakends' pile to runIterativeAssembler) */
int manta_spanning_batch_piles(
    manta_ctx_t* ctx, const manta_asm_options_t* opt, const manta_align_scores_t* scores, int32_t jump_score, uint32_t n_loci,
    const manta_packed_piles_t* piles, const uint8_t* refs1, const uint64_t* ref1_off, const uint8_t* refs2, const uint64_t* ref2_off,
    const manta_jump_cuts_t* cuts, const uint32_t* locus_min_word_length /* nullable */, const uint32_t* locus_max_word_length /* nullable */,
    manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs, manta_spanning_alignment_t* alignments, uint64_t contigs_cap,
    uint8_t* seq_arena, uint64_t seq_arena_cap, uint64_t* seq_arena_used, uint64_t* bits_arena, uint64_t bits_arena_cap,
    uint64_t* bits_arena_used, uint32_t* cigar_arena, uint64_t cigar_arena_cap, uint64_t* cigar_arena_used,
    const manta_batch_plan_t* plan /* nullable */, manta_batch_stats_t* stats /* nullable */);

/* ------------------------------------------------------------------------------------------------------
 * One node, several GPUs, ONE work queue (the reference's partition primitives are the static bins of
 * EdgeRetrieverBin.cpp:38-57 and the --threads worker pool of GenerateSVCandidates.cpp:232-266).
 *
 * manta_node_t owns one context per device.  manta_node_*_batch is the whole-batch call over all of them: the batch is
 * cut into blocks (about 64 per call, at least ~1000 loci each, ordered by decreasing cost), and one host worker per
 * device (n_workers per device) pulls blocks from the single queue -- a device that draws expensive blocks simply takes
 * fewer.  Results land in the caller's records and arenas exactly as with the single-device call; no collective is
 * involved: one host process owns all the GPUs, so every block's results come back over that GPU's own PCIe link
 * (a device-to-device gather would only add a hop).  loci_per_device (nullable, n_devices entries) reports how the
 * queue spread the loci.  For one PROCESS per GPU use manta_batch_plan_t::shared_queue instead.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct manta_node manta_node_t;
int         manta_node_create(const int32_t* device_ids, uint32_t n_devices, manta_node_t** out);
void        manta_node_destroy(manta_node_t* node);
uint32_t    manta_node_device_count(const manta_node_t* node);
const char* manta_node_last_error(const manta_node_t* node);

int manta_node_smallsv_batch(
    manta_node_t* node, const manta_asm_options_t* opt, const manta_align_scores_t* scores, int32_t large_indel_score,
    uint32_t n_loci, const uint8_t* bases, const uint64_t* read_off, const uint32_t* locus_read_begin, const uint8_t* refs,
    const uint64_t* ref_off, const manta_ref_cuts_t* cuts, const uint32_t* locus_min_word_length /* nullable */,
    const uint32_t* locus_max_word_length /* nullable */, manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs,
    manta_smallsv_alignment_t* alignments, uint64_t contigs_cap, uint8_t* seq_arena, uint64_t seq_arena_cap,
    uint64_t* seq_arena_used, uint64_t* bits_arena, uint64_t bits_arena_cap, uint64_t* bits_arena_used, uint32_t* cigar_arena,
    uint64_t cigar_arena_cap, uint64_t* cigar_arena_used, const manta_batch_plan_t* plan /* nullable */,
    manta_batch_stats_t* stats /* nullable */, uint32_t* loci_per_device /* nullable */);

int manta_node_spanning_batch(
    manta_node_t* node, const manta_asm_options_t* opt, const manta_align_scores_t* scores, int32_t jump_score, uint32_t n_loci,
    const uint8_t* bases, const uint64_t* read_off, const uint32_t* locus_read_begin, const uint8_t* refs1,
    const uint64_t* ref1_off, const uint8_t* refs2, const uint64_t* ref2_off, const manta_jump_cuts_t* cuts,
    const uint32_t* locus_min_word_length /* nullable */, const uint32_t* locus_max_word_length /* nullable */,
    manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs, manta_spanning_alignment_t* alignments, uint64_t contigs_cap,
    uint8_t* seq_arena, uint64_t seq_arena_cap, uint64_t* seq_arena_used, uint64_t* bits_arena, uint64_t bits_arena_cap,
    uint64_t* bits_arena_used, uint32_t* cigar_arena, uint64_t cigar_arena_cap, uint64_t* cigar_arena_used,
    const manta_batch_plan_t* plan /* nullable */, manta_batch_stats_t* stats /* nullable */, uint32_t* loci_per_device /* nullable */);


/* ------------------------------------------------------------------------------------------------------
 * Read-pile construction (second half): the per-read tests of SVCandidateAssembler::getBreakendReads
 * (manta/SVCandidateAssembler.cpp:271-659) over DECODED BAM records, emitting the packed piles above.
 *
 * What stays with the caller: the BAM layer.  The caller runs the region queries the reference runs -- for every breakend of a
 * candidate and every alignment file, bamStream.resetRegion(tid, searchBegin, searchEnd) (:381; the search range is the
 * breakend interval widened to 400 bases, :285-303, see manta_read_search_range) -- and hands over the records in file order.
 * What this call does, per candidate, with the reference's order-dependent rules intact:
 *   isReadFilteredCore (ReadFilter.cpp:32-50), the normal-sample depth estimate and its two thresholds (:85-100, :404-425),
 *   isNonStrictSupplement (bam_record.hpp:139-144), the indel test (:473-483), the semi-aligned / soft-clip test
 *   (getSVBreakendCandidateSemiAligned, SVLocusScannerSemiAligned.cpp:25-330, incl. is_overlapping_pair / is_adapter_pair,
 *   bam_record_util.cpp:54-108), ShadowReadFinder::check (ShadowReadFinder.cpp:33-113), the remote-mate candidate tests
 *   (RemoteMateReadUtil.cpp:29-55; flags only -- fetching the mates is a BAM seek and stays with the caller), the read-key
 *   de-duplication and the 1000-read cap of insertAssemblyRead / the scan loop (:102-119, :387-393), then the string handling
 *   of insertAssemblyRead (:121-135: Q mask, reverse complement) straight into 2 bit + N bitmap.
 * Coordinates are htslib's: 0-based pos / mate_pos (bam1_core_t), i.e. the reference's bam_record::pos() - 1.
 * ---------------------------------------------------------------------------------------------------- */
#define MANTA_READ_TAG_SA 1u /* the record carries an SA tag (bam_record::isSASplit) */
#define MANTA_READ_TAG_MC 2u /* the record carries an MC tag; its cigar is given as mate_cigar (bam_record::hasMateCigar) */

typedef struct {
  int32_t  tid, pos;           /* bam1_core_t::tid, ::pos */
  int32_t  mate_tid, mate_pos; /* ::mtid, ::mpos */
  uint16_t flag;               /* ::flag (BAM_F*) */
  uint8_t  mapq;               /* ::qual */
  uint8_t  tags;               /* MANTA_READ_TAG_* */
  uint32_t read_len;           /* ::l_qseq */
  uint32_t n_cigar, cigar_off; /* BAM cigar words (length << 4 | op) in the cigar arena, as bam_get_cigar */
  uint32_t n_mate_cigar, mate_cigar_off; /* the MC tag in the same encoding, P and zero-length operations dropped
                                            (cigar_to_apath, blt_util/align_path.cpp:66-95) */
  uint32_t qname_len, qname_off; /* bytes in the name arena (no terminator needed) */
  uint64_t seq_off;            /* bam_get_seq: 4-bit codes, two bases per byte, (read_len + 1) / 2 bytes in the sequence arena */
  uint64_t qual_off;           /* bam_get_qual: read_len bytes in the quality arena */
} manta_bam_read_t;

/* one region query of getBreakendReads (:371-384): one breakend x one alignment file */
typedef struct {
  uint32_t read_begin, read_end; /* the query's records in file order (coordinate-sorted, as a BAM region query returns them: the depth
                                    estimate relies on it) */
  uint32_t bam_index;            /* alignment file: part of the read key (:110-111) */
  uint8_t  is_tumor;             /* _isAlignmentTumor[bamIndex] (:372): normal samples feed the depth estimate */
  uint8_t  is_locus_reversed;    /* getBreakendReads' isLocusReversed */
  uint8_t  first_of_breakend;    /* 1 on the first file of a breakend: the depth buffer starts over (:340) */
  uint8_t  reserved;
  int32_t  bp_begin, bp_end;     /* SVBreakend::interval.range */
  int32_t  bp_state;             /* SVBreakendState::index_t (manta/SVBreakend.hpp:146-153): 1 RIGHT_OPEN, 2 LEFT_OPEN, other: both */
  int32_t  ref_begin;            /* reference_contig_segment::get_offset() of the window the refiner fetched */
  uint32_t ref_len;
  uint64_t ref_off;              /* its text in the reference arena */
} manta_read_scan_t;

typedef struct {
  uint32_t scan_begin, scan_end; /* the candidate's queries in the reference's order: breakend 1 x files, breakend 2 x files */
  uint8_t  is_max_depth;         /* ChromDepthFilterUtil::isMaxDepthFilter() */
  uint8_t  search_remote;        /* isSearchRemoteInsertionReads: also flag remote-mate candidates */
  uint8_t  reserved[2];
  float    max_depth;            /* _dFilter.maxDepth(tid) as the reference's float (:333) */
  float    max_local_depth_remote; /* _dFilterLocalDepthForRemoteReadRetrieval.maxDepth(tid) (:334) */
} manta_read_locus_t;

typedef struct {
  uint32_t min_qval;                      /* IterativeAssemblerOptions::minQval (5) */
  uint32_t min_candidate_variant_size;    /* ReadScannerOptions (10): indels of at least half of it keep a read (:317) */
  uint32_t min_singleton_mapq_candidates; /* ReadScannerOptions (15): shadow anchors */
  uint32_t min_mapq;                      /* ReadScannerOptions (15): remote-mate candidates */
  uint32_t use_overlap_pair_evidence;     /* ReadScannerOptions (0) */
  uint32_t max_reads;                     /* 0: the reference's 1000 (:342) */
} manta_read_class_options_t;

/* per-record decision bits */
#define MANTA_READ_INDEL 1u        /* isIndelKeeper */
#define MANTA_READ_SEMI_ALIGNED 2u /* isSemiAlignedKeeper */
#define MANTA_READ_SHADOW 4u       /* isShadowKeeper */
#define MANTA_READ_IN_PILE 8u      /* inserted: pile_index is valid */
#define MANTA_READ_REVERSED 16u    /* inserted reverse-complemented */
#define MANTA_READ_REMOTE_MATE 32u /* remoteReads.emplace_back (:449-456): the caller may fetch the mate */
#define MANTA_READ_DEPTH_FILTERED 64u
#define MANTA_READ_DUPLICATE_KEY 128u /* kept by the tests, dropped by the read index (:112-119) */

typedef struct {
  int32_t  status;          /* MANTA_OK; MANTA_E_UNSUPPORTED: a kept read holds the BAM code '=' (not representable in two bits;
                               decisions are valid, the pile is not -- run the locus through the text interface) */
  uint32_t n_pile_reads;
  uint32_t retrieve_remote; /* isRetrieveRemoteReads (:575): no record tripped the local depth threshold */
  uint32_t reserved;
} manta_read_locus_result_t;

/* [searchBegin, searchEnd) of a breakend interval (:285-303) */
void manta_read_search_range(int32_t bp_begin, int32_t bp_end, int32_t* search_begin, int32_t* search_end);

/* Piles of n_loci candidates.  Outputs: decision / pile_index per record (pile_index: position inside the candidate's pile,
 * 0xffffffff when not inserted); the piles in manta_packed_piles_t layout in caller memory: codes_cap / mask_cap dwords,
 * reads_cap reads (MANTA_E_CAPACITY if exceeded; *_used report what is needed); pile_read[r] = record of pile read r.
 * read_code_off / read_mask_off take reads_cap + 1 entries (the last one = one past the last read) and are required whenever
 * n_loci > 0, also for a batch without a single record. */
int manta_read_piles_batch(
    manta_ctx_t* ctx, const manta_read_class_options_t* opt, uint32_t n_loci, const manta_read_locus_t* loci, uint32_t n_scans,
    const manta_read_scan_t* scans, uint32_t n_reads, const manta_bam_read_t* reads, const uint32_t* cigars, uint64_t n_cigar_words,
    const uint8_t* names, uint64_t names_bytes, const uint8_t* seqs, uint64_t seqs_bytes, const uint8_t* quals, uint64_t quals_bytes,
    const uint8_t* refs, uint64_t refs_bytes, uint8_t* decision, uint32_t* pile_index, manta_read_locus_result_t* results,
    uint32_t* codes, uint64_t codes_cap, uint64_t* codes_used, uint32_t* nmask, uint64_t mask_cap, uint64_t* mask_used,
    uint32_t* read_len, uint64_t* read_code_off, uint64_t* read_mask_off, uint32_t* pile_read, uint64_t reads_cap,
    uint64_t* reads_used, uint32_t* locus_read_begin);

#ifdef __cplusplus
}
#endif
#endif /* MANTA_AMD_H */
