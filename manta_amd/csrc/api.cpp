// C-ABI implementation (include/manta_amd.h).  Compiled by hipcc for gfx950 into manta_amd/libmanta_amd.so.
// Host code here only stages buffers, buckets work and launches the kernels; all arithmetic of the hot path
// runs in the HIP kernels of align_kernels.hpp / assemble_kernels.hpp.
// (This file: contexts, the single-call entry points and the two staged pipelines with their run functions.  The whole-batch calls and
// the node queue are api_batch.cpp, split-read scoring and read gathering api_reads.cpp; api_internal.hpp holds what they share.)
#include "api_internal.hpp"

extern "C" {

uint32_t manta_abi_version(void) { return MANTA_ABI_VERSION; }
uint64_t manta_batch_stats_size(void) { return sizeof(manta_batch_stats_t); }

int manta_ctx_create(int device_id, manta_ctx_t** out)
{
  if (!out) {
    g_createError = "manta_ctx_create: null output pointer";
    return MANTA_E_INVALID_ARG;
  }
  *out = nullptr;
  try {
    rt::init(device_id);
    if (!stringHashMatchesLibstdcxx()) {
      g_createError = "this libstdc++'s std::hash<std::string> is not the Murmur variant the exact repeat search restates";
      return MANTA_E_UNSUPPORTED;
    }
    manta_ctx_t* ctx = new manta_ctx();
    ctx->deviceName  = rt::deviceName();
    ctx->cuCount     = rt::cuCount();
    ctx->deviceId    = rt::currentDevice();
    *out             = ctx;
    return MANTA_OK;
  } catch (const std::exception& e) {
    g_createError = e.what();
    return MANTA_E_NO_DEVICE;
  }
}

void manta_ctx_destroy(manta_ctx_t* ctx)
{
  delete ctx;
}

const char* manta_last_error(const manta_ctx_t* ctx)
{
  return ctx ? ctx->lastError.c_str() : g_createError.c_str();
}

const char* manta_ctx_device_name(const manta_ctx_t* ctx)
{
  return ctx ? ctx->deviceName.c_str() : "";
}

int manta_align_batch(
    manta_ctx_t* ctx, int kind, const manta_align_scores_t* scores, int32_t extra_score, uint32_t n_tasks,
    const manta_align_task_t* tasks, const uint8_t* seq_arena, uint64_t seq_arena_bytes, manta_align_result_t* results,
    uint32_t* cigar_arena, uint64_t cigar_arena_cap, uint64_t* cigar_arena_used)
{
  if (!ctx) return MANTA_E_INVALID_ARG;
  if (!scores || (n_tasks && (!tasks || !seq_arena || !results || !cigar_arena)))
    return fail(ctx, MANTA_E_INVALID_ARG, "manta_align_batch: null argument");
  if (kind < 0 || kind > 2) return fail(ctx, MANTA_E_INVALID_ARG, "manta_align_batch: unknown aligner kind");
  if (kind == MANTA_ALIGNER_JUMP && scores->is_allow_edge_insertion)
    return fail(ctx, MANTA_E_INVALID_ARG, "GlobalJumpAligner does not support isAllowEdgeInsertion");
  if (cigar_arena_used) *cigar_arena_used = 0;
  if (n_tasks == 0) return MANTA_OK;

  try {
    rt::setDevice(ctx->deviceId);  // the calling thread may have another device current (multi-GPU hosts)
    rt::ScopedStream onStream(ctx->stream);
    // ---- validate + bucket by columns-per-lane (E) ----
    std::vector<AlignTaskDev>          dev(n_tasks);
    std::vector<std::vector<uint32_t>> buckets(kNumESet);
    std::vector<uint64_t>              bucketMaxRef(kNumESet, 0);
    uint64_t                           cigarDevWords = 0;
    int                                worst         = MANTA_OK;
    for (uint32_t i = 0; i < n_tasks; ++i) {
      const manta_align_task_t& t(tasks[i]);
      manta_align_result_t&     r(results[i]);
      std::memset(&r, 0, sizeof(r));
      const bool jump = (kind == MANTA_ALIGNER_JUMP);
      auto outside = [&](uint64_t off, uint64_t len) { return off > seq_arena_bytes || len > seq_arena_bytes - off; };
      if (outside(t.query_off, t.query_len) || outside(t.ref1_off, t.ref1_len) || (jump && outside(t.ref2_off, t.ref2_len)))
        return fail(ctx, MANTA_E_INVALID_ARG, "manta_align_batch: task " + std::to_string(i) + " outside the sequence arena");
      if (t.query_len == 0 || t.ref1_len == 0 || (jump && t.ref2_len == 0)) {
        r.status = MANTA_E_EMPTY_SEQ;  // GlobalJumpAlignerImpl.hpp:50-58, GlobalAlignerImpl.hpp:44-49
        worst    = MANTA_E_EMPTY_SEQ;
        continue;
      }
      const int eIdx = pickE(t.query_len);
      AlignTaskDev& d(dev[i]);
      d.query     = reinterpret_cast<const uint8_t*>(uintptr_t(t.query_off));  // rebased onto the device arena below
      d.ref1      = reinterpret_cast<const uint8_t*>(uintptr_t(t.ref1_off));
      d.ref2      = reinterpret_cast<const uint8_t*>(uintptr_t(jump ? t.ref2_off : 0));
      d.query_len = t.query_len;
      d.ref1_len  = t.ref1_len;
      d.ref2_len  = jump ? t.ref2_len : 0;
      d.cigar_off = uint32_t(cigarDevWords);
      cigarDevWords += 4ull * t.query_len + 16;
      if (cigarDevWords > 0xffffffffull) return fail(ctx, MANTA_E_UNSUPPORTED, "manta_align_batch: batch too large (cigar workspace)");
      buckets[eIdx].push_back(i);
      bucketMaxRef[eIdx] = std::max<uint64_t>(bucketMaxRef[eIdx], alignSlabRefLen(kind, kESet[eIdx], d.query_len, uint64_t(d.ref1_len) + d.ref2_len));
    }

    // ---- stage ----
    uint8_t*        dSeq     = ctx->dSeq.as<uint8_t>(seq_arena_bytes);
    AlignTaskDev*   dTasks   = ctx->dTasks.as<AlignTaskDev>(n_tasks);
    AlignResultDev* dResults = ctx->dResults.as<AlignResultDev>(n_tasks);
    uint32_t*       dCigar   = ctx->dCigar.as<uint32_t>(cigarDevWords + 1);
    uint32_t*       dIds     = ctx->dTaskIds.as<uint32_t>(n_tasks);
    uint32_t*       dCounter = ctx->dCounter.as<uint32_t>(kNumESet);
    for (AlignTaskDev& d : dev) {
      d.query = dSeq + uintptr_t(d.query);
      d.ref1  = dSeq + uintptr_t(d.ref1);
      d.ref2  = dSeq + uintptr_t(d.ref2);
    }
    rt::h2d(dSeq, seq_arena, seq_arena_bytes);
    rt::h2d(dTasks, dev.data(), sizeof(AlignTaskDev) * n_tasks);
    rt::dzero(dCounter, sizeof(uint32_t) * kNumESet);
    rt::dzero(dResults, sizeof(AlignResultDev) * n_tasks);

    const int    maxWaves  = std::max(1, ctx->cuCount * alignWavesPerCu());
    const size_t wsBudget  = workspaceBudget(size_t(24) << 30);
    size_t       idsCursor = 0;
    for (int b = 0; b < kNumESet; ++b) {
      if (buckets[b].empty()) continue;
      const bool     pair   = alignUsesPairs(kind, b, scores->match, scores->mismatch, scores->open, scores->extend, scores->off_edge, extra_score,
                                             scores->is_allow_edge_insertion ? 1 : 0, bucketMaxRef[b]);
      const uint64_t stride = ((pair ? 2 : 1) * alignPtrSlabBytes(kind, kESet[b], bucketMaxRef[b]) + 255) & ~uint64_t(255);
      int            grid   = int(std::min<size_t>(pair ? (buckets[b].size() + 1) / 2 : buckets[b].size(), size_t(maxWaves)));
      grid                  = int(std::max<size_t>(1, std::min<size_t>(size_t(grid), wsBudget / stride)));
      grid                  = rt::roundGrid(grid);
      uint8_t* dWs          = ctx->dPtrWs.as<uint8_t>(stride * grid);
      rt::h2d(dIds + idsCursor, buckets[b].data(), sizeof(uint32_t) * buckets[b].size());
      AlignParams P;
      P.tasks          = dTasks;
      P.results        = dResults;
      P.cigar          = dCigar;
      P.task_ids       = dIds + idsCursor;
      P.n_tasks        = uint32_t(buckets[b].size());
      P.n_tasks_dev    = nullptr;
      P.counter        = dCounter + b;
      P.ptr_ws         = dWs;
      P.ptr_ws_stride  = stride;
      P.match          = scores->match;
      P.mismatch       = scores->mismatch;
      P.open           = scores->open;
      P.extend         = scores->extend;
      P.off_edge       = scores->off_edge;
      P.allow_edge_ins = scores->is_allow_edge_insertion ? 1 : 0;
      P.extra          = extra_score;
      rt::Event e0, e1;
      e0.record();
      launchAlignKind(kind, b, grid, P, pair);
      e1.record();
      rt::sync();  // dPtrWs may be re-sized by the next bucket
      if (std::getenv("MANTA_AMD_DEBUG"))
        std::fprintf(stderr, "manta_amd: align_kernel kind %d E=%d %zu tasks %.3f ms\n", kind, kESet[b], buckets[b].size(), rt::elapsedMs(e0, e1));
      idsCursor += buckets[b].size();
    }

    // ---- fetch + compact cigars into the caller's arena ----
    std::vector<AlignResultDev> hres(n_tasks);
    std::vector<uint32_t>       hcig(cigarDevWords + 1);
    rt::d2h(hres.data(), dResults, sizeof(AlignResultDev) * n_tasks);
    rt::d2h(hcig.data(), dCigar, sizeof(uint32_t) * cigarDevWords);
    uint64_t used = 0;
    for (uint32_t i = 0; i < n_tasks; ++i) {
      manta_align_result_t& r(results[i]);
      if (r.status != MANTA_OK) continue;
      const AlignResultDev& h(hres[i]);
      if (h.status != 0) {
        r.status = MANTA_E_DEVICE_FAULT;
        worst    = MANTA_E_DEVICE_FAULT;
        continue;
      }
      const uint64_t n = uint64_t(h.cigar1_len) + h.cigar2_len;
      if (used + n > cigar_arena_cap) return fail(ctx, MANTA_E_CAPACITY, "manta_align_batch: cigar arena too small");
      std::memcpy(cigar_arena + used, hcig.data() + dev[i].cigar_off, sizeof(uint32_t) * n);
      r.score            = h.score;
      r.is_jumped        = h.is_jumped;
      r.begin_pos1       = h.begin1;
      r.begin_pos2       = h.begin2;
      r.jump_insert_size = h.jump_insert_size;
      r.jump_range       = h.jump_range;
      r.cigar1_len       = h.cigar1_len;
      r.cigar2_len       = h.cigar2_len;
      r.cigar1_off       = used;
      r.cigar2_off       = used + h.cigar1_len;
      used += n;
    }
    if (cigar_arena_used) *cigar_arena_used = used;
    if (worst != MANTA_OK) return fail(ctx, worst, "manta_align_batch: one or more tasks failed; see per-task status");
    return MANTA_OK;
  } catch (const std::exception& e) {
    return fail(ctx, MANTA_E_HIP, e.what());
  }
}


int manta_assemble_batch(
    manta_ctx_t* ctx, const manta_asm_options_t* opt, uint32_t n_loci, const uint8_t* bases, const uint64_t* read_off,
    const uint32_t* locus_read_begin, manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs, uint64_t contigs_cap,
    uint8_t* seq_arena, uint64_t seq_arena_cap, uint64_t* seq_arena_used, uint64_t* bits_arena, uint64_t bits_arena_cap,
    uint64_t* bits_arena_used)
{
  if (!ctx) return MANTA_E_INVALID_ARG;
  if (!opt || (n_loci && (!bases || !read_off || !locus_read_begin || !loci || !contigs || !seq_arena || !bits_arena)))
    return fail(ctx, MANTA_E_INVALID_ARG, "manta_assemble_batch: null argument");
  if (seq_arena_used) *seq_arena_used = 0;
  if (bits_arena_used) *bits_arena_used = 0;
  if (n_loci == 0) return MANTA_OK;
  try {
    rt::setDevice(ctx->deviceId);  // the calling thread may have another device current (multi-GPU hosts)
    rt::ScopedStream onStream(ctx->stream);
    AsmStage st(ctx);
    int      rc = st.plan(*opt, n_loci, read_off, locus_read_begin);
    if (rc != MANTA_OK) {
      // refused as a whole (a locus outside the envelope sizes the shared workspace): every record says so, nothing is left unwritten
      for (uint32_t l = 0; l < n_loci; ++l) {
        std::memset(&loci[l], 0, sizeof(loci[l]));
        loci[l].status = rc;
      }
      return rc;
    }
    st.upload(bases, read_off, locus_read_begin);
    rt::Event e0, e1;
    e0.record();
    st.launch();
    e1.record();
    uint64_t capacityFailures = 0;
    rt::d2h(&capacityFailures, st.dCnt + 3, sizeof(uint64_t));  // (waits for the kernel)
    st.rerunCapacityFailures(capacityFailures);
    if (std::getenv("MANTA_AMD_DEBUG"))
      std::fprintf(stderr, "manta_amd: assemble_kernel %u loci %.3f ms\n", n_loci, rt::elapsedMs(e0, e1));
    return st.fetch(loci, contigs, contigs_cap, seq_arena, seq_arena_cap, seq_arena_used, bits_arena, bits_arena_cap, bits_arena_used);
  } catch (const std::exception& e) {
    return fail(ctx, MANTA_E_HIP, e.what());
  }
}

int manta_debug_repeat_words(
    manta_ctx_t* ctx, const manta_asm_options_t* opt, uint32_t n_reads, const uint8_t* bases, const uint64_t* read_off, char* out,
    uint64_t out_cap, uint32_t* n_words)
{
  if (!ctx) return MANTA_E_INVALID_ARG;
  if (!opt || !bases || !read_off || !out || !n_words) return fail(ctx, MANTA_E_INVALID_ARG, "manta_debug_repeat_words: null argument");
  *n_words = 0;
  try {
    rt::setDevice(ctx->deviceId);
    rt::ScopedStream onStream(ctx->stream);
    manta_asm_options_t o = *opt;
    o.max_word_length     = o.min_word_length;  // one word length: its graph is what the workspace holds afterwards
    const uint32_t begin[2] = {0, n_reads};
    AsmStage       st(ctx);
    int            rc = st.plan(o, 1, read_off, begin);
    if (rc != MANTA_OK) return rc;
    st.useFast = false;  // the general kernel: its workspace slab is read back below
    st.upload(bases, read_off, begin);
    rt::dzero(st.dWs, st.stride * uint64_t(st.grid));
    st.launch();
    AsmLocusOut lo;
    rt::d2h(&lo, st.dLoci, sizeof(lo));
    if (lo.status != ASM_OK) return fail(ctx, asmStatusToAbi(lo.status), "manta_debug_repeat_words: the locus did not assemble");
    const uint32_t nNodes = lo.reserved & 0x3ffffffu, slab = lo.reserved >> 26, k = lo.final_word_length;
    const AsmWsLayout L = asmWorkspaceLayout(st.capSlots, st.capNodes, st.capWords, st.capReads, st.maxContigLen, st.wMax, o.max_assembly_count);
    const uint8_t*    ws = st.dWs + st.stride * uint64_t(slab);
    std::vector<uint32_t> flag(nNodes), key(nNodes), codes(st.capWords + 2);
    rt::d2h(flag.data(), ws + L.node_flag, sizeof(uint32_t) * nNodes);
    rt::d2h(key.data(), ws + L.node_key, sizeof(uint32_t) * nNodes);
    rt::d2h(codes.data(), ws + L.codes, sizeof(uint32_t) * codes.size());
    std::vector<std::string> words;
    for (uint32_t nd = 0; nd < nNodes; ++nd) {
      if (!(flag[nd] & NF_REPEAT)) continue;
      std::string w(k, '?');
      for (uint32_t i = 0; i < k; ++i) {
        const uint32_t pb = key[nd] + i;
        w[i]              = "ACGT"[(codes[pb >> 4] >> (30 - 2 * (pb & 15))) & 3];
      }
      words.push_back(w);
    }
    std::sort(words.begin(), words.end());
    uint64_t used = 0;
    for (const std::string& w : words) {
      if (used + w.size() + 1 > out_cap) return fail(ctx, MANTA_E_CAPACITY, "manta_debug_repeat_words: output buffer too small");
      std::memcpy(out + used, w.data(), w.size());
      used += w.size();
      out[used++] = '\n';
    }
    if (used < out_cap) out[used] = 0;
    *n_words = uint32_t(words.size());
    return MANTA_OK;
  } catch (const std::exception& e) {
    return fail(ctx, MANTA_E_HIP, e.what());
  }
}

int manta_small_assemble_batch(
    manta_ctx_t* ctx, const manta_small_asm_options_t* opt, uint32_t n_loci, const uint8_t* bases, const uint64_t* read_off,
    const uint32_t* locus_read_begin, manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs, uint64_t contigs_cap,
    uint8_t* seq_arena, uint64_t seq_arena_cap, uint64_t* seq_arena_used, uint64_t* bits_arena, uint64_t bits_arena_cap,
    uint64_t* bits_arena_used)
{
  if (!ctx) return MANTA_E_INVALID_ARG;
  if (!opt || (n_loci && (!bases || !read_off || !locus_read_begin || !loci || !contigs || !seq_arena || !bits_arena)))
    return fail(ctx, MANTA_E_INVALID_ARG, "manta_small_assemble_batch: null argument");
  if (opt->word_step_size == 0 || opt->min_word_length < 2 || opt->max_assembly_iterations > 31)
    return fail(ctx, MANTA_E_INVALID_ARG, "manta_small_assemble_batch: wordStepSize 0, minWordLength < 2 or more than 31 iterations");
  if (seq_arena_used) *seq_arena_used = 0;
  if (bits_arena_used) *bits_arena_used = 0;
  if (n_loci == 0) return MANTA_OK;
  try {
    rt::setDevice(ctx->deviceId);  // the calling thread may have another device current (multi-GPU hosts)
    rt::ScopedStream onStream(ctx->stream);
    AsmStage st(ctx);
    // the slab and the record slots are those of the iterative assembler: one slot per iteration's contig + the isFiltered record
    manta_asm_options_t o{};
    o.min_word_length           = opt->min_word_length;
    o.max_word_length           = opt->max_word_length;
    o.word_step_size            = opt->word_step_size;
    o.min_contig_length         = opt->min_contig_length;
    o.min_coverage              = opt->min_coverage;
    o.min_conservative_coverage = opt->min_conservative_coverage;
    o.min_unused_reads          = 0;
    o.min_support_reads         = 0;
    o.max_assembly_count        = opt->max_assembly_iterations + 1;
    st.smallMode                = true;
    st.smallMinSeedReads        = opt->min_seed_reads;
    st.smallMaxIterations       = opt->max_assembly_iterations;
    int rc = st.plan(o, n_loci, read_off, locus_read_begin);
    if (rc != MANTA_OK) {
      for (uint32_t l = 0; l < n_loci; ++l) {
        std::memset(&loci[l], 0, sizeof(loci[l]));
        loci[l].status = rc;
      }
      return rc;
    }
    st.upload(bases, read_off, locus_read_begin);
    st.launch();
    rt::sync();
    return st.fetch(loci, contigs, contigs_cap, seq_arena, seq_arena_cap, seq_arena_used, bits_arena, bits_arena_cap, bits_arena_used);
  } catch (const std::exception& e) {
    return fail(ctx, MANTA_E_HIP, e.what());
  }
}

// ------------------------------------------------------------------------------------------------------
// fused small-SV pipeline
// ------------------------------------------------------------------------------------------------------
int manta_smallsv_create(
    manta_ctx_t* ctx, const manta_asm_options_t* opt, const manta_align_scores_t* scores, int32_t large_indel_score,
    manta_smallsv_t** out)
{
  if (!ctx) return MANTA_E_INVALID_ARG;
  if (!opt || !scores || !out) return fail(ctx, MANTA_E_INVALID_ARG, "manta_smallsv_create: null argument");
  try {
    rt::setDevice(ctx->deviceId);  // (the pipeline's streams and events belong to the context's device)
    manta_smallsv* b = new manta_smallsv(ctx);
    b->opt           = *opt;
    b->scores        = *scores;
    b->largeIndel    = large_indel_score;
    *out             = b;
  } catch (const std::exception& e) {
    return fail(ctx, MANTA_E_HIP, e.what());
  }
  return MANTA_OK;
}

void manta_smallsv_destroy(manta_smallsv_t* b)
{
  delete b;
}

int manta_smallsv_upload(
    manta_smallsv_t* b, uint32_t n_loci, const uint8_t* bases, const uint64_t* read_off, const uint32_t* locus_read_begin,
    const uint8_t* refs, const uint64_t* ref_off, const manta_ref_cuts_t* cuts)
{
  if (!b) return MANTA_E_INVALID_ARG;
  manta_ctx_t* ctx = b->ctx;
  if (n_loci == 0 || !bases || !read_off || !locus_read_begin || !refs || !ref_off || !cuts)
    return fail(ctx, MANTA_E_INVALID_ARG, "manta_smallsv_upload: null argument or empty batch");
  try {
    rt::setDevice(ctx->deviceId);  // the calling thread may have another device current (multi-GPU hosts)
    rt::ScopedStream onStream(b->main);
    b->uploaded = false;
    const double tP0 = nowMs();
    const bool streamed = b->streamUploads && !std::getenv("MANTA_AMD_NO_STREAM_UPLOAD");
    // a streamed upload starts the DMA of the read bases before anything else: plan()'s pass over the read offsets runs beside it
    const bool noEarlyStream = std::getenv("MANTA_AMD_NO_EARLY_STREAM") != nullptr;  // A/B knob (read per call: the tests flip it)
    const bool        chunksInFlight = streamed && !noEarlyStream && b->asmStage.startStream(n_loci, bases, read_off, locus_read_begin, b->copy);
    struct DrainOnFailure {  // (a failed call must not leave DMA reads of the caller's buffers queued)
      manta_smallsv_t* b;
      bool             armed;
      ~DrainOnFailure() { if (armed) drainCopyStream(b); }
    } drain{b, chunksInFlight};
    int rc      = b->asmStage.plan(b->opt, n_loci, read_off, locus_read_begin);
    if (rc != MANTA_OK) return rc;
    const double tP1 = nowMs();
    b->nLoci    = n_loci;
    b->refBytes = ref_off[n_loci];
    b->maxRef   = 0;
    for (uint32_t l = 0; l < n_loci; ++l) {
      if (ref_off[l + 1] < ref_off[l]) return fail(ctx, MANTA_E_INVALID_ARG, "manta_smallsv_upload: ref_off not monotone");
      b->maxRef = std::max<uint64_t>(b->maxRef, ref_off[l + 1] - ref_off[l]);
      if (cuts[l].leading_cut < 0 || cuts[l].trailing_cut < 0 || cuts[l].max_leading_cut < 0 || cuts[l].max_trailing_cut < 0)
        return fail(ctx, MANTA_E_INVALID_ARG, "manta_smallsv_upload: negative reference cut");
    }
    uint8_t*  dRefs   = b->dRefs.as<uint8_t>(b->refBytes + 16);
    uint64_t* dRefOff = b->dRefOff.as<uint64_t>(n_loci + 1);
    auto*     dCuts   = b->dCuts.as<SmallSvCuts>(n_loci);
    static_assert(sizeof(SmallSvCuts) == sizeof(manta_ref_cuts_t), "cuts layout");
    b->refsOnCopy       = streamed;
    if (streamed) {
      // the assembler does not read the reference windows: they travel on the copy stream BEHIND the read bases and the
      // schedule kernel waits for them (smallsvRunImpl); nothing but the small per-read arrays is waited for here
      b->asmStage.uploadStreamed(bases, read_off, locus_read_begin, b->copy);  // syncs the small copies, returns with the bases in flight
      rt::ScopedStream onCopy(b->copy);
      rt::h2d(dRefs, refs, b->refBytes);
      rt::h2d(dRefOff, ref_off, sizeof(uint64_t) * (n_loci + 1));
      rt::h2d(dCuts, cuts, sizeof(SmallSvCuts) * n_loci);
      b->refsReady.record();
    } else {
      rt::h2d(dRefs, refs, b->refBytes);
      rt::h2d(dRefOff, ref_off, sizeof(uint64_t) * (n_loci + 1));
      rt::h2d(dCuts, cuts, sizeof(SmallSvCuts) * n_loci);
      b->asmStage.upload(bases, read_off, locus_read_begin);
      rt::sync();
    }
    if (std::getenv("MANTA_AMD_DEBUG_TIMING"))
      std::fprintf(stderr, "manta_amd: smallsv_upload plan %.2f ms, upload (%s) %.2f ms\n", tP1 - tP0, b->asmStage.streaming ? "streamed" : "blocking", nowMs() - tP1);
    drain.armed = false;
    b->uploaded = true;
    return MANTA_OK;
  } catch (const std::exception& e) {
    return fail(ctx, MANTA_E_HIP, e.what());
  }
}


int manta_smallsv_upload_piles(
    manta_smallsv_t* b, uint32_t n_loci, const manta_packed_piles_t* piles, const uint8_t* refs, const uint64_t* ref_off,
    const manta_ref_cuts_t* cuts)
{
  if (!b) return MANTA_E_INVALID_ARG;
  manta_ctx_t* ctx = b->ctx;
  if (n_loci == 0 || !refs || !ref_off || !cuts) return fail(ctx, MANTA_E_INVALID_ARG, "manta_smallsv_upload_piles: null argument or empty batch");
  int rc = checkPiles(ctx, piles, "manta_smallsv_upload_piles");
  if (rc != MANTA_OK) return rc;
  try {
    rt::setDevice(ctx->deviceId);
    rt::ScopedStream onStream(b->main);
    b->uploaded = false;
    b->refsOnCopy = false;
    rc          = b->asmStage.plan(b->opt, n_loci, nullptr, piles->locus_read_begin, piles->read_len);
    if (rc != MANTA_OK) return rc;
    b->nLoci    = n_loci;
    b->refBytes = ref_off[n_loci];
    b->maxRef   = 0;
    for (uint32_t l = 0; l < n_loci; ++l) {
      if (ref_off[l + 1] < ref_off[l]) return fail(ctx, MANTA_E_INVALID_ARG, "manta_smallsv_upload_piles: ref_off not monotone");
      b->maxRef = std::max<uint64_t>(b->maxRef, ref_off[l + 1] - ref_off[l]);
      if (cuts[l].leading_cut < 0 || cuts[l].trailing_cut < 0 || cuts[l].max_leading_cut < 0 || cuts[l].max_trailing_cut < 0)
        return fail(ctx, MANTA_E_INVALID_ARG, "manta_smallsv_upload_piles: negative reference cut");
    }
    uint8_t*  dRefs   = b->dRefs.as<uint8_t>(b->refBytes + 16);
    uint64_t* dRefOff = b->dRefOff.as<uint64_t>(n_loci + 1);
    auto*     dCuts   = b->dCuts.as<SmallSvCuts>(n_loci);
    b->refsOnCopy     = b->streamUploads && !std::getenv("MANTA_AMD_NO_STREAM_UPLOAD");
    if (b->refsOnCopy) {  // whole-batch call: piles chunk by chunk behind the running assembler, references behind them (manta_smallsv_upload)
      b->asmStage.uploadPilesStreamed(*piles, b->copy);
      rt::ScopedStream onCopy(b->copy);
      rt::h2d(dRefs, refs, b->refBytes);
      rt::h2d(dRefOff, ref_off, sizeof(uint64_t) * (n_loci + 1));
      rt::h2d(dCuts, cuts, sizeof(SmallSvCuts) * n_loci);
      b->refsReady.record();
    } else {
      b->asmStage.uploadPiles(*piles);
      rt::h2d(dRefs, refs, b->refBytes);
      rt::h2d(dRefOff, ref_off, sizeof(uint64_t) * (n_loci + 1));
      rt::h2d(dCuts, cuts, sizeof(SmallSvCuts) * n_loci);
      rt::sync();
    }
    b->uploaded = true;
    return MANTA_OK;
  } catch (const std::exception& e) {
    return fail(ctx, MANTA_E_HIP, e.what());
  }
}

}  // extern "C"

int smallsvRunImpl(manta_smallsv_t* b, StageGates* gates)
{
  if (!b) return MANTA_E_INVALID_ARG;
  manta_ctx_t* ctx = b->ctx;
  if (!b->uploaded) return fail(ctx, MANTA_E_INVALID_ARG, "manta_smallsv_run: nothing uploaded");
  try {
    rt::setDevice(ctx->deviceId);  // the calling thread may have another device current (multi-GPU hosts)
    rt::ScopedStream onStream(b->main);
    const uint32_t nLoci   = b->nLoci;
    const uint32_t maxAsm  = b->opt.max_assembly_count;
    const uint64_t nSlots  = uint64_t(nLoci) * maxAsm;
    AsmStage&      as(b->asmStage);
    // ---- per-run device state ----
    AlignTaskDev*    dTasks   = b->dTasks.as<AlignTaskDev>(nSlots);
    SmallSvTaskInfo* dInfo    = b->dInfo.as<SmallSvTaskInfo>(nSlots);
    AlignResultDev*  dResults = b->dResults.as<AlignResultDev>(nSlots);
    uint32_t*        dBuckets = b->dBucketIds.as<uint32_t>(nSlots * kNumESet);
    uint32_t*        dSmall   = b->dSmall.as<uint32_t>(64);  // [0..11] counts, [16..27] maxref, [32] sched counter, [34..35] cigar_used, [40..51] align counters
    const uint32_t   tableCap = nextPow2(2ull * as.maxContigLen);
    const int        schedWaves = std::getenv("MANTA_AMD_SCHED_WAVES_PER_CU") ? std::max(1, std::atoi(std::getenv("MANTA_AMD_SCHED_WAVES_PER_CU"))) : 12;
    const uint32_t   schedChunk = std::getenv("MANTA_AMD_SCHED_CHUNK") ? uint32_t(std::max(1, std::atoi(std::getenv("MANTA_AMD_SCHED_CHUNK")))) : 2u;
    const int        schedGrid = rt::roundGrid(int(std::min<uint64_t>((nLoci + schedChunk - 1) / schedChunk, uint64_t(std::max(1, ctx->cuCount * schedWaves)))));
    uint32_t*        dTable   = b->dTable.as<uint32_t>(uint64_t(tableCap) * schedGrid);
    rt::dzero(dSmall, sizeof(uint32_t) * 64);
    rt::dzero(dResults, sizeof(AlignResultDev) * nSlots);

    const bool dbg = std::getenv("MANTA_AMD_DEBUG") != nullptr;
    auto       stage = [&](const char* what) {
      if (!dbg) return;
      rt::sync();
      std::fprintf(stderr, "manta_amd: smallsv_run %s\n", what);
      std::fflush(stderr);
    };
    stage("start");
    uint64_t asmCnt[4];  // [1] text bytes, [3] capacity failures
    {
      GateLock only(gates, &StageGates::asmMu);
      // A streamed-upload assembler polls for chunks whose copies may need a free workgroup slot (launch() leaves some);
      // a second persistent assembler would take exactly those slots and both would spin until the kernels' time-out.
      // Per device, process-wide: never two streamed assemblers on a device at once.
      std::unique_lock<std::mutex> streamedOnly(streamedAsmMu(ctx), std::defer_lock);
      if (as.streaming) streamedOnly.lock();
      as.stageQueued = false;  // (a run that failed behind its queued staging must not leave the flag to the next one)
      as.earlyStaged = false;
      b->evStart.record();
      as.launch();
      b->evAsm.record();
      // loci that did not fit the typical-case workspace run again here (rare; the counter rides on a wait that is there anyway
      // in the whole-batch calls, the staged API pays one small read)
      rt::d2h(asmCnt, as.dCnt, sizeof(asmCnt));  // (also: the gate opens when the assembler has left the device)
      if (asmCnt[3]) {
        as.rerunCapacityFailures(asmCnt[3]);
        rt::d2h(asmCnt, as.dCnt, sizeof(asmCnt));  // the text arena grew
      }
    }
    stage("assembled");
    // whole-batch calls: the assembler's outputs leave for the host now, on the copy stream, beside the schedule kernel and the aligners
    const bool noEarlyStage = std::getenv("MANTA_AMD_NO_EARLY_STAGE") != nullptr;  // A/B knob (read per call: the tests flip it)
    const bool        earlyStage   = b->stageBehindRun && b->whileAligning && !noEarlyStage;
    if (earlyStage) as.stageEarly(b->copy, asmCnt, b->evEarlyStaged);
    // CIGAR scratch, sized from what the assembler produced (as in spanningRunImpl): a task takes 4 * contig length + 16 words
    // (smallsv_schedule_kernel), every contig is aligned once, and the text arena counter bounds the summed contig lengths.  A
    // worst case per slot would be tens of GB at 65536-locus blocks and overflow the 32-bit offsets of the task records.
    const uint64_t cigarCap = 8ull * std::min<uint64_t>(asmCnt[1], as.devSeqCap) + 64;  // (8 words per text byte: smallsv_schedule_kernel)
    if (cigarCap + 16 > 0xffffffffull)
      return fail(ctx, MANTA_E_UNSUPPORTED, "manta_smallsv_run: alignment scratch of this block exceeds 2^32 words; use smaller blocks "
                                         "(manta_smallsv_batch splits a batch into blocks, manta_batch_plan_t::block_loci)");
    uint32_t* dCigar = b->dCigar.as<uint32_t>(cigarCap + 16);
    GateLock alignOnly(gates, &StageGates::alignMu);
    if (b->refsOnCopy) rt::curStreamWaits(b->refsReady);  // reference windows of a streamed upload (manta_smallsv_upload)

    ScheduleParams S;
    S.loci               = as.dLoci;
    S.contigs            = as.dCont;
    S.seq_arena          = as.dSeq;
    S.n_loci             = nLoci;
    S.max_assembly_count = maxAsm;
    S.refs               = static_cast<const uint8_t*>(b->dRefs.p);
    S.ref_off            = static_cast<const uint64_t*>(b->dRefOff.p);
    S.cuts               = static_cast<const SmallSvCuts*>(b->dCuts.p);
    S.tasks              = dTasks;
    S.info               = dInfo;
    S.bucket_ids         = dBuckets;
    S.bucket_count       = dSmall;
    S.bucket_maxref      = dSmall + 16;
    S.cigar_used         = reinterpret_cast<unsigned long long*>(dSmall + 34);
    S.cigar_cap          = cigarCap;
    S.counter            = dSmall + 32;
    S.table_ws           = dTable;
    S.table_cap          = tableCap;
    S.n_e                = kNumESet;
    S.chunk              = schedChunk;
    for (int i = 0; i < kNumESet; ++i) S.e_set[i] = uint32_t(kESet[i]);
    rt::launch(smallsv_schedule_kernel, schedGrid, SCHED_LDS_BYTES, S);
    // the pair-eligible buckets by descending reference length (bucket_sort_kernel): their align kernels read the sorted lists
    uint32_t  pairMask       = 0;
    uint32_t* dBucketsSorted = nullptr;
    for (int k = 0; k < kNumESet; ++k)
      if (alignUsesPairs(MANTA_ALIGNER_LARGE_INDEL, k, b->scores.match, b->scores.mismatch, b->scores.open, b->scores.extend, b->scores.off_edge,
                         b->largeIndel, b->scores.is_allow_edge_insertion ? 1 : 0, b->maxRef))
        pairMask |= 1u << k;
    if (pairMask) {
      dBucketsSorted = b->dBucketIds2.as<uint32_t>(nSlots * kNumESet);
      BucketSortParams BS;
      BS.tasks        = dTasks;
      BS.loci         = as.dLoci;
      BS.info         = dInfo;
      BS.ids_out      = dBucketsSorted;
      BS.n_loci       = nLoci;
      BS.max_assembly_count = maxAsm;
      BS.total        = uint32_t(nSlots);
      BS.mask         = pairMask;
      rt::launchWG(bucket_sort_kernel, kNumESet, int(BS_WAVES), BS_LDS_BYTES, BS);
    }
    b->evSched.record();
    stage("scheduled");

    // Bucket sizes decide the alignment launches.  First run of a pipeline: one tiny D2H.  Later runs of a whole-batch call
    // do not wait for it: every bucket is launched with the task count read on the device (AlignParams::n_tasks_dev), its
    // grid sized from the previous run's counts (consecutive blocks look alike; the waves of a bucket pull tasks from a queue,
    // so a grid that is off costs time, never results) and its slabs from what the host knows (longest reference window,
    // longest possible contig).  The counts come back with the results.
    uint32_t hSmall[40];
    const int    maxWaves = std::max(1, ctx->cuCount * alignWavesPerCu());
    const size_t wsBudget = workspaceBudget(size_t(48) << 30);
    bool         fromHistory = b->bucketHistory && b->stageBehindRun && !std::getenv("MANTA_AMD_SYNC_BUCKETS");  // (no host wait)
    struct Launch {
      int      k, grid;
      uint64_t stride, slabOff;
      bool     pair;
    };
    std::vector<Launch> launches;
    uint64_t            slabBytes = 0;
    auto pairOf = [&](int k) {
      return alignUsesPairs(MANTA_ALIGNER_LARGE_INDEL, k, b->scores.match, b->scores.mismatch, b->scores.open, b->scores.extend, b->scores.off_edge,
                            b->largeIndel, b->scores.is_allow_edge_insertion ? 1 : 0, b->maxRef);
    };
    const bool batchCall = b->stageBehindRun && !std::getenv("MANTA_AMD_SYNC_BUCKETS");
    bool       haveCounts = false;
    if (batchCall && !fromHistory) {
      // first run of this pipeline: read the counts once, then launch exactly as the later runs will (same kernels, same slab
      // layout), so that the second run does not pay for a changed allocation or for kernels seen for the first time
      rt::d2h(hSmall, dSmall, sizeof(hSmall));
      std::memcpy(b->lastSmall, hSmall, sizeof(b->lastSmall));
      haveCounts  = true;
      fromHistory = true;
    }
    if (fromHistory) {
      for (int k = kNumESet - 1; k >= 0; --k) {
        const bool     pair   = pairOf(k);
        // (a bucket that was empty last time gets a token grid: tasks that do turn up are still aligned, just by few waves)
        const uint64_t tasks  = b->lastSmall[k] ? uint64_t(b->lastSmall[k]) + b->lastSmall[k] / 4 + 64 : 4;
        const uint64_t hint   = pair ? (tasks + 1) / 2 : tasks;  // work items: alignments, or pairs of them
        const uint64_t qBound = (kESet[k] == 32) ? std::max<uint64_t>(as.maxContigLen, 64ull * 32) : 64ull * uint64_t(kESet[k]);
        const uint64_t refLen = alignSlabRefLen(MANTA_ALIGNER_LARGE_INDEL, kESet[k], qBound, b->maxRef);
        const uint64_t stride = ((pair ? 2 : 1) * alignPtrSlabBytes(MANTA_ALIGNER_LARGE_INDEL, kESet[k], refLen) + 255) & ~uint64_t(255);
        const int      grid   = rt::roundGrid(int(std::min<uint64_t>(hint, uint64_t(maxWaves))));
        launches.push_back(Launch{k, grid, stride, slabBytes, pair});
        slabBytes += stride * uint64_t(grid);
      }
      if (slabBytes > wsBudget / 3) {  // too generous for this device right now: size from the real counts
        fromHistory = false;
        launches.clear();
        slabBytes = 0;
      }
    }
    if (!fromHistory) {
      if (!haveCounts) rt::d2h(hSmall, dSmall, sizeof(hSmall));
      for (int k = kNumESet - 1; k >= 0; --k) {  // widest (longest-running) buckets first
        const uint32_t cnt = hSmall[k];
        if (cnt == 0) continue;
        const bool     pair   = pairOf(k);
        const uint64_t stride = ((pair ? 2 : 1) * alignPtrSlabBytes(MANTA_ALIGNER_LARGE_INDEL, kESet[k], hSmall[16 + k]) + 255) & ~uint64_t(255);
        int            grid   = int(std::min<size_t>(pair ? (cnt + 1) / 2 : cnt, size_t(maxWaves)));
        grid                  = int(std::max<size_t>(1, std::min<size_t>(size_t(grid), (wsBudget / 3) / stride)));
        grid                  = rt::roundGrid(grid);
        launches.push_back(Launch{k, grid, stride, slabBytes, pair});
        slabBytes += stride * uint64_t(grid);
      }
    }
    b->stats.n_align_launches = 0;
    b->stats.n_alignments     = 0;
    b->stats.ptr_matrix_bytes = 0;
    // The E buckets are independent launches: they run on side streams so that the tail of one overlaps the others
    // (a launch's last alignments leave most of the device idle otherwise).  Each bucket gets its own slab region.
    {
      // The packed buckets share ONE launch and one queue (align_pair_multi_kernel: buckets with the fewest tasks first, so the
      // few long alignments of narrow buckets start at once and the bulk of the widest fills in behind them); every other
      // bucket is its own launch on a side stream.
      static const bool   noMerge = std::getenv("MANTA_AMD_NO_ALIGN_MERGE") != nullptr;  // A/B knob: one launch per packed bucket
      std::vector<Launch> merged;
      if (!noMerge) {
        std::vector<Launch> rest;
        for (const Launch& l : launches) (l.pair ? merged : rest).push_back(l);
        if (merged.size() < 2 || merged.size() > 6) {
          merged.clear();
        } else {
          launches.swap(rest);
          slabBytes = 0;
          for (Launch& l : launches) {
            l.slabOff = slabBytes;
            slabBytes += l.stride * uint64_t(l.grid);
          }
        }
      }
      uint64_t mergedStride = 0, mergedWaves = 0, mergedOff = slabBytes;
      const uint32_t* counts = fromHistory ? b->lastSmall : hSmall;
      if (!merged.empty()) {
        std::stable_sort(merged.begin(), merged.end(), [&](const Launch& x, const Launch& y) { return counts[x.k] < counts[y.k]; });
        for (const Launch& l : merged) {
          mergedStride = std::max(mergedStride, l.stride);
          mergedWaves += uint64_t(l.grid);
        }
        // (clamp first, then a whole number of workgroups -- rt::launch divides the grid by the workgroup's wave count -- and at least one)
        mergedWaves = std::min<uint64_t>(std::min<uint64_t>(mergedWaves, uint64_t(maxWaves)), (wsBudget / 3) / mergedStride);
        mergedWaves = std::max<uint64_t>(WV_WAVES_PER_WG, (mergedWaves / WV_WAVES_PER_WG) * WV_WAVES_PER_WG);
        slabBytes += mergedStride * mergedWaves;
      }
      uint8_t* dWsAll = b->dPtrWs.as<uint8_t>(slabBytes + 256);
      // (without the host read in between nothing else orders the side streams behind the schedule kernel)
      for (int i = 0; i < 3; ++i) rt::streamWaits(b->side[i], b->evSched);
      auto baseParams = [&]() {
        AlignParams P;
        P.tasks          = dTasks;
        P.results        = dResults;
        P.cigar          = dCigar;
        P.task_ids       = nullptr;
        P.n_tasks        = 0;
        P.n_tasks_dev    = nullptr;
        P.counter        = nullptr;
        P.ptr_ws         = nullptr;
        P.ptr_ws_stride  = 0;
        P.match          = b->scores.match;
        P.mismatch       = b->scores.mismatch;
        P.open           = b->scores.open;
        P.extend         = b->scores.extend;
        P.off_edge       = b->scores.off_edge;
        P.allow_edge_ins = b->scores.is_allow_edge_insertion ? 1 : 0;
        P.extra          = b->largeIndel;
        return P;
      };
      for (size_t i = 0; i < launches.size(); ++i) {
        const Launch& l(launches[i]);
        AlignParams   P  = baseParams();
        P.task_ids       = (l.pair ? dBucketsSorted : dBuckets) + uint64_t(l.k) * nSlots;
        P.n_tasks        = fromHistory ? uint32_t(nSlots) : hSmall[l.k];
        P.n_tasks_dev    = fromHistory ? dSmall + l.k : nullptr;
        P.counter        = dSmall + 40 + l.k;
        P.ptr_ws         = dWsAll + l.slabOff;
        P.ptr_ws_stride  = l.stride;
        {
          rt::ScopedStream onSide(b->side[i % 3]);  // (restores the pipeline's stream: nothing of this call runs on the null stream)
          launchAlignKind(MANTA_ALIGNER_LARGE_INDEL, l.k, l.grid, P, l.pair);
        }
        if (!fromHistory) {
          b->stats.n_align_launches++;
          b->stats.n_alignments += hSmall[l.k];
        }
      }
      // (the packed launch goes last: the buckets above are small or token grids that finish at once on an empty device; queued
      // behind a grid that fills every SIMD's registers they would sit in the dispatcher until its first waves retire)
      if (!merged.empty()) {
        PairMultiParams M;
        M.A               = baseParams();
        M.A.counter       = dSmall + 40 + merged[0].k;  // (the queue head of the first bucket serves the shared queue)
        M.A.ptr_ws        = dWsAll + mergedOff;
        M.A.ptr_ws_stride = mergedStride;
        M.bucket_ids      = dBucketsSorted;
        M.counts          = dSmall;
        M.n_slots         = uint32_t(nSlots);
        M.n_order         = uint32_t(merged.size());
        M.prio_work[0] = 2560u;  // (defaults from the config-2 measurement: a typical pair is E x (G + 64) ~ 5 x 384 = 1 920)
        M.prio_work[1] = 0u;
        M.prio_work[2] = 4096u;
        if (const char* e = std::getenv("MANTA_AMD_ALIGN_PRIO")) {  // experiments: "w1,w2,w3" (0 = level unused; "0,0,0" = no priorities)
          unsigned v[3] = {0, 0, 0};
          std::sscanf(e, "%u,%u,%u", &v[0], &v[1], &v[2]);
          for (int i = 0; i < 3; ++i) M.prio_work[i] = v[i];
        }
        for (size_t i = 0; i < 8; ++i) {
          M.order[i] = uint8_t(i < merged.size() ? merged[i].k : 0);
          M.e_of[i]  = uint8_t(i < merged.size() ? kESet[merged[i].k] : 0);
        }
        if (std::getenv("MANTA_AMD_DEBUG"))
          std::fprintf(stderr, "manta_amd: align_pair_multi_kernel: %llu waves over %zu packed buckets, slab stride %llu\n",
                       (unsigned long long)mergedWaves, merged.size(), (unsigned long long)mergedStride);
        rt::launch(align_pair_multi_kernel, int(mergedWaves), 0, M);  // on the pipeline's own stream
        if (!fromHistory) {
          b->stats.n_align_launches++;
          for (const Launch& l : merged) b->stats.n_alignments += hSmall[l.k];
        }
      }
      // the null stream (events, later copies) continues after every side stream has drained
      for (size_t i = 0; i < std::min<size_t>(launches.size(), 3); ++i) {
        b->sideDone[i].recordOn(b->side[i]);
        rt::curStreamWaits(b->sideDone[i]);
      }
    }
    launchPack(b, dTasks, nullptr, dResults, nullptr, dInfo, nullptr, dCigar, cigarCap);
    b->evAlign.record();
    uint32_t* hSmallPin = b->pSmall.as<uint32_t>(40);
    rt::d2hAsync(hSmallPin, dSmall, sizeof(uint32_t) * 40);
    if (b->stageBehindRun) pipeStageEnqueue(b);  // the results start for the host behind the last kernel: no wake-up in between
    if (earlyStage) {  // everything is queued and the aligners have ~2 ms to go: the host's share of the assembler's outputs
      b->evEarlyStaged.sync();
      as.finishEarly();
      b->whileAligning();
    }
    rt::sync();
    std::memcpy(b->lastSmall, hSmallPin, sizeof(b->lastSmall));
    b->bucketHistory = true;
    if (fromHistory)
      for (int k = 0; k < kNumESet; ++k)
        if (b->lastSmall[k]) {
          b->stats.n_align_launches++;
          b->stats.n_alignments += b->lastSmall[k];
        }
    alignOnly.release();
    if (as.streaming) {  // all chunks were consumed by the kernel, so this returns at once; it closes the stream's error state
      rt::ScopedStream onCopy(b->copy);
      rt::sync();
    }
    stage("aligned");
    b->stats.assemble_ms = rt::elapsedMs(b->evStart, b->evAsm);
    b->stats.schedule_ms = rt::elapsedMs(b->evAsm, b->evSched);
    b->stats.align_ms    = rt::elapsedMs(b->evSched, b->evAlign);
    b->stats.total_ms    = rt::elapsedMs(b->evStart, b->evAlign);
    b->ran               = true;
    return MANTA_OK;
  } catch (const std::exception& e) {
    drainCopyStream(b);
    return fail(ctx, MANTA_E_HIP, e.what());
  }
}

extern "C" {

int manta_smallsv_run(manta_smallsv_t* b) { return smallsvRunImpl(b, nullptr); }

int manta_smallsv_stats(const manta_smallsv_t* b, manta_smallsv_stats_t* stats)
{
  if (!b || !stats) return MANTA_E_INVALID_ARG;
  *stats = b->stats;
  return MANTA_OK;
}

int manta_smallsv_output_sizes(const manta_smallsv_t* b, uint64_t* contigs, uint64_t* seq_bytes, uint64_t* bits_words, uint64_t* cigar_words)
{
  if (!b || !contigs || !seq_bytes || !bits_words || !cigar_words) return MANTA_E_INVALID_ARG;
  if (!b->ran) return fail(b->ctx, MANTA_E_INVALID_ARG, "manta_smallsv_output_sizes: run first");
  try {
    rt::setDevice(b->ctx->deviceId);  // the calling thread may have another device current (multi-GPU hosts)
    rt::ScopedStream onStream(const_cast<manta_smallsv_t*>(b)->main);
    b->asmStage.outputSizes(*contigs, *seq_bytes, *bits_words);
    uint32_t hSmall[40];
    rt::d2h(hSmall, b->dSmall.p, sizeof(hSmall));
    uint64_t cig = 0;
    std::memcpy(&cig, hSmall + 34, sizeof(uint64_t));
    *cigar_words = cig + 64;
    return MANTA_OK;
  } catch (const std::exception& e) {
    return fail(b->ctx, MANTA_E_HIP, e.what());
  }
}

}  // extern "C"


extern "C" {

int manta_smallsv_download(
    manta_smallsv_t* b, manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs, manta_smallsv_alignment_t* alignments,
    uint64_t contigs_cap, uint8_t* seq_arena, uint64_t seq_arena_cap, uint64_t* seq_arena_used, uint64_t* bits_arena,
    uint64_t bits_arena_cap, uint64_t* bits_arena_used, uint32_t* cigar_arena, uint64_t cigar_arena_cap,
    uint64_t* cigar_arena_used)
{
  if (!b) return MANTA_E_INVALID_ARG;
  manta_ctx_t* ctx = b->ctx;
  if (!b->ran) return fail(ctx, MANTA_E_INVALID_ARG, "manta_smallsv_download: run first");
  if (!loci || !contigs || !alignments || !seq_arena || !bits_arena || !cigar_arena)
    return fail(ctx, MANTA_E_INVALID_ARG, "manta_smallsv_download: null argument");
  try {
    rt::setDevice(ctx->deviceId);  // the calling thread may have another device current (multi-GPU hosts)
    rt::ScopedStream onStream(b->main);
    pipeStage(b);
    return smallsvCompact(b, loci, contigs, alignments, 0, contigs_cap, seq_arena, seq_arena_cap, 0, seq_arena_used, bits_arena,
                          bits_arena_cap, 0, bits_arena_used, cigar_arena, cigar_arena_cap, 0, cigar_arena_used);
  } catch (const std::exception& e) {
    return fail(ctx, MANTA_E_HIP, e.what());
  }
}

// ------------------------------------------------------------------------------------------------------
// fused spanning pipeline
// ------------------------------------------------------------------------------------------------------
int manta_spanning_create(
    manta_ctx_t* ctx, const manta_asm_options_t* opt, const manta_align_scores_t* scores, int32_t jump_score, manta_spanning_t** out)
{
  if (!ctx) return MANTA_E_INVALID_ARG;
  if (!opt || !scores || !out) return fail(ctx, MANTA_E_INVALID_ARG, "manta_spanning_create: null argument");
  if (scores->is_allow_edge_insertion)
    return fail(ctx, MANTA_E_INVALID_ARG, "GlobalJumpAligner does not support isAllowEdgeInsertion");
  try {
    rt::setDevice(ctx->deviceId);  // (the pipeline's streams and events belong to the context's device)
    manta_spanning* b = new manta_spanning(ctx);
    b->opt            = *opt;
    b->scores         = *scores;
    b->jumpScore      = jump_score;
    *out              = b;
  } catch (const std::exception& e) {
    return fail(ctx, MANTA_E_HIP, e.what());
  }
  return MANTA_OK;
}

void manta_spanning_destroy(manta_spanning_t* b)
{
  delete b;
}

int manta_spanning_upload(
    manta_spanning_t* b, uint32_t n_loci, const uint8_t* bases, const uint64_t* read_off, const uint32_t* locus_read_begin,
    const uint8_t* refs1, const uint64_t* ref1_off, const uint8_t* refs2, const uint64_t* ref2_off, const manta_jump_cuts_t* cuts)
{
  if (!b) return MANTA_E_INVALID_ARG;
  manta_ctx_t* ctx = b->ctx;
  if (n_loci == 0 || !bases || !read_off || !locus_read_begin || !refs1 || !ref1_off || !refs2 || !ref2_off || !cuts)
    return fail(ctx, MANTA_E_INVALID_ARG, "manta_spanning_upload: null argument or empty batch");
  try {
    rt::setDevice(ctx->deviceId);  // the calling thread may have another device current (multi-GPU hosts)
    rt::ScopedStream onStream(b->main);
    b->uploaded = false;
    int rc      = b->asmStage.plan(b->opt, n_loci, read_off, locus_read_begin);
    if (rc != MANTA_OK) return rc;
    b->nLoci     = n_loci;
    b->ref1Bytes = ref1_off[n_loci];
    b->ref2Bytes = ref2_off[n_loci];
    for (uint32_t l = 0; l < n_loci; ++l) {
      if (ref1_off[l + 1] < ref1_off[l] || ref2_off[l + 1] < ref2_off[l])
        return fail(ctx, MANTA_E_INVALID_ARG, "manta_spanning_upload: reference offsets not monotone");
      const manta_jump_cuts_t& c(cuts[l]);
      if (c.align1_leading_cut < 0 || c.align1_trailing_cut < 0 || c.align2_leading_cut < 0 || c.align2_trailing_cut < 0)
        return fail(ctx, MANTA_E_INVALID_ARG, "manta_spanning_upload: negative reference cut");
    }
    static_assert(sizeof(JumpCuts) == sizeof(manta_jump_cuts_t), "cuts layout");
    uint8_t*  dRefs1   = b->dRefs1.as<uint8_t>(b->ref1Bytes + 16);
    uint64_t* dRef1Off = b->dRef1Off.as<uint64_t>(n_loci + 1);
    uint8_t*  dRefs2   = b->dRefs2.as<uint8_t>(b->ref2Bytes + 16);
    uint64_t* dRef2Off = b->dRef2Off.as<uint64_t>(n_loci + 1);
    JumpCuts* dCuts    = b->dCuts.as<JumpCuts>(n_loci);
    auto      copyRefs = [&] {
      rt::h2d(dRefs1, refs1, b->ref1Bytes);
      rt::h2d(dRef1Off, ref1_off, sizeof(uint64_t) * (n_loci + 1));
      rt::h2d(dRefs2, refs2, b->ref2Bytes);
      rt::h2d(dRef2Off, ref2_off, sizeof(uint64_t) * (n_loci + 1));
      rt::h2d(dCuts, cuts, sizeof(JumpCuts) * n_loci);
    };
    b->hostCuts.assign(reinterpret_cast<const JumpCuts*>(cuts), reinterpret_cast<const JumpCuts*>(cuts) + n_loci);
    b->refsOnCopy = b->streamUploads && !std::getenv("MANTA_AMD_NO_STREAM_UPLOAD");
    if (b->refsOnCopy) {  // whole-batch call: as manta_smallsv_upload -- bases in chunks behind the running assembler, references behind them
      b->asmStage.uploadStreamed(bases, read_off, locus_read_begin, b->copy);
      rt::ScopedStream onCopy(b->copy);
      copyRefs();
      b->refsReady.record();
    } else {
      b->asmStage.upload(bases, read_off, locus_read_begin);
      copyRefs();
      rt::sync();
    }
    b->uploaded = true;
    return MANTA_OK;
  } catch (const std::exception& e) {
    return fail(ctx, MANTA_E_HIP, e.what());
  }
}

int manta_spanning_upload_piles(
    manta_spanning_t* b, uint32_t n_loci, const manta_packed_piles_t* piles, const uint8_t* refs1, const uint64_t* ref1_off,
    const uint8_t* refs2, const uint64_t* ref2_off, const manta_jump_cuts_t* cuts)
{
  if (!b) return MANTA_E_INVALID_ARG;
  manta_ctx_t* ctx = b->ctx;
  if (n_loci == 0 || !refs1 || !ref1_off || !refs2 || !ref2_off || !cuts)
    return fail(ctx, MANTA_E_INVALID_ARG, "manta_spanning_upload_piles: null argument or empty batch");
  int rc = checkPiles(ctx, piles, "manta_spanning_upload_piles");
  if (rc != MANTA_OK) return rc;
  try {
    rt::setDevice(ctx->deviceId);
    rt::ScopedStream onStream(b->main);
    b->uploaded = false;
    b->refsOnCopy = false;
    rc          = b->asmStage.plan(b->opt, n_loci, nullptr, piles->locus_read_begin, piles->read_len);
    if (rc != MANTA_OK) return rc;
    b->asmStage.uploadPiles(*piles);
    b->nLoci     = n_loci;
    b->ref1Bytes = ref1_off[n_loci];
    b->ref2Bytes = ref2_off[n_loci];
    for (uint32_t l = 0; l < n_loci; ++l) {
      if (ref1_off[l + 1] < ref1_off[l] || ref2_off[l + 1] < ref2_off[l])
        return fail(ctx, MANTA_E_INVALID_ARG, "manta_spanning_upload_piles: reference offsets not monotone");
      const manta_jump_cuts_t& c(cuts[l]);
      if (c.align1_leading_cut < 0 || c.align1_trailing_cut < 0 || c.align2_leading_cut < 0 || c.align2_trailing_cut < 0)
        return fail(ctx, MANTA_E_INVALID_ARG, "manta_spanning_upload_piles: negative reference cut");
    }
    rt::h2d(b->dRefs1.as<uint8_t>(b->ref1Bytes + 16), refs1, b->ref1Bytes);
    rt::h2d(b->dRef1Off.as<uint64_t>(n_loci + 1), ref1_off, sizeof(uint64_t) * (n_loci + 1));
    rt::h2d(b->dRefs2.as<uint8_t>(b->ref2Bytes + 16), refs2, b->ref2Bytes);
    rt::h2d(b->dRef2Off.as<uint64_t>(n_loci + 1), ref2_off, sizeof(uint64_t) * (n_loci + 1));
    rt::h2d(b->dCuts.as<JumpCuts>(n_loci), cuts, sizeof(JumpCuts) * n_loci);
    b->hostCuts.assign(reinterpret_cast<const JumpCuts*>(cuts), reinterpret_cast<const JumpCuts*>(cuts) + n_loci);
    rt::sync();
    b->uploaded = true;
    return MANTA_OK;
  } catch (const std::exception& e) {
    return fail(ctx, MANTA_E_HIP, e.what());
  }
}

}  // extern "C"

int spanningRunImpl(manta_spanning_t* b, StageGates* gates)
{
  if (!b) return MANTA_E_INVALID_ARG;
  manta_ctx_t* ctx = b->ctx;
  if (!b->uploaded) return fail(ctx, MANTA_E_INVALID_ARG, "manta_spanning_run: nothing uploaded");
  try {
    rt::setDevice(ctx->deviceId);  // the calling thread may have another device current (multi-GPU hosts)
    rt::ScopedStream onStream(b->main);
    const uint32_t nLoci  = b->nLoci;
    const uint32_t maxAsm = b->opt.max_assembly_count;
    const uint64_t nSlots = uint64_t(nLoci) * maxAsm;
    AsmStage&      as(b->asmStage);
    AlignTaskDev*   dTasks    = b->dTasks.as<AlignTaskDev>(nSlots);
    AlignTaskDev*   dTasks2   = b->dTasks2.as<AlignTaskDev>(nSlots);
    SpanTaskInfo*   dInfo     = b->dInfo.as<SpanTaskInfo>(nSlots);
    AlignResultDev* dResults  = b->dResults.as<AlignResultDev>(nSlots);
    AlignResultDev* dResults2 = b->dResults2.as<AlignResultDev>(nSlots);
    uint32_t*       dBuckets  = b->dBucketIds.as<uint32_t>(nSlots * kNumESet);
    uint32_t*       dBuckets2 = b->dBucketIds2.as<uint32_t>(nSlots * kNumESet);
    // [0..15] counts, [16..31] maxref, [32..47] counts2, [48..63] maxref2, [64..65] cigar_used, [72..87] / [88..103] align counters
    uint32_t*      dSmall   = b->dSmall.as<uint32_t>(128);
    rt::dzero(dSmall, sizeof(uint32_t) * 128);
    rt::dzero(dResults, sizeof(AlignResultDev) * nSlots);
    rt::dzero(dResults2, sizeof(AlignResultDev) * nSlots);

    const bool dbg   = std::getenv("MANTA_AMD_DEBUG") != nullptr;
    auto       stage = [&](const char* what) {
      if (!dbg) return;
      rt::sync();
      std::fprintf(stderr, "manta_amd: spanning_run %s\n", what);
      std::fflush(stderr);
    };
    // ---- the alignment stage as a function of (which loci, on which streams): it runs once over every locus, or -- the early pass --
    // first over the loci that are final after the assembler's first word length, concurrently with the later word lengths, and then
    // over the rest.
    SpanParams S;
    S.loci               = as.dLoci;
    S.contigs            = as.dCont;
    S.seq_arena          = as.dSeq;
    S.n_loci             = nLoci;
    S.max_assembly_count = maxAsm;
    S.refs1              = static_cast<const uint8_t*>(b->dRefs1.p);
    S.ref1_off           = static_cast<const uint64_t*>(b->dRef1Off.p);
    S.refs2              = static_cast<const uint8_t*>(b->dRefs2.p);
    S.ref2_off           = static_cast<const uint64_t*>(b->dRef2Off.p);
    S.cuts               = static_cast<const JumpCuts*>(b->dCuts.p);
    S.tasks              = dTasks;
    S.tasks2             = dTasks2;
    S.info               = dInfo;
    S.bucket_ids         = dBuckets;
    S.bucket_count       = dSmall;
    S.bucket_maxref      = dSmall + 16;
    S.bucket_ids2        = dBuckets2;
    S.bucket_count2      = dSmall + 32;
    S.bucket_maxref2     = dSmall + 48;
    S.cigar_used         = reinterpret_cast<unsigned long long*>(dSmall + 64);
    S.cigar_cap          = 0;
    S.results            = dResults;
    S.cigar              = nullptr;
    S.n_e                = kNumESet;
    for (int i = 0; i < kNumESet; ++i) S.e_set[i] = uint32_t(kESet[i]);
    S.pass_mask  = nullptr;
    S.pass_value = 0;
    const int glueGrid = rt::roundGrid(int(std::min<uint64_t>((nSlots + 63) / 64, uint64_t(std::max(1, ctx->cuCount * 8)))));
    b->stats.n_align_launches = 0;
    b->stats.n_alignments     = 0;
    const size_t wsBudget = workspaceBudget(size_t(48) << 30);
    uint32_t*    dCigar   = nullptr;
    // CIGAR scratch, sized from what the assembler produced: a task takes 4 * contig length + 16 words (spanFileTask), every
    // contig is aligned at most twice (second round: spanning_realign_kernel), and the text arena counter bounds the summed
    // contig lengths.  (A fixed worst case of max_contig_len per slot is ~40x the real need at 200 x 250 bp loci.)
    // `keepWords` > 0: the buffer already holds that many words of the early pass -- a larger one takes them over.
    auto cigarScratch = [&](const uint64_t seqUsed, const uint64_t keepWords) -> uint64_t {
      const uint64_t need = 2 * (4ull * std::min<uint64_t>(seqUsed, as.devSeqCap) + 16ull * nSlots) + 64;
      if (keepWords && (need + 16) * sizeof(uint32_t) > b->dCigar.cap) {
        DevBuf bigger;
        uint32_t* q = bigger.as<uint32_t>(need + 16);
        rt::d2d(q, b->dCigar.p, sizeof(uint32_t) * keepWords);
        rt::sync();
        std::swap(bigger.p, b->dCigar.p);
        std::swap(bigger.cap, b->dCigar.cap);
      }
      dCigar = b->dCigar.as<uint32_t>(need + 16);
      return need;
    };
    auto alignRound = [&](rt::Stream* const* sides, const int maxWaves, const uint32_t* hCounts, const uint32_t* hMaxref, const AlignTaskDev* tasks,
                          AlignResultDev* results, const uint32_t* bucketIds, uint32_t* counters) {
      // buckets on side streams, one slab region each (see manta_smallsv_run)
      struct Launch {
        int      k, grid;
        uint64_t stride, slabOff;
        bool     pair;
      };
      std::vector<Launch> launches;
      uint64_t            slabBytes = 0;
      // (sharing the wave slots among the buckets by work -- tasks x columns x steps -- so that they would end together was measured in
      // round 5 and lost: 170 ms against 86 per 16 384 config-5 loci.  A bucket's waves are long dependent chains; capping its grid only
      // lengthens them.)
      for (int k = kNumESet - 1; k >= 0; --k) {
        const uint32_t cnt = hCounts[k];
        if (cnt == 0) continue;
        // (two alignments per wave in packed 16-bit arithmetic where the scores allow it: align_jump_pair.hpp)
        const bool     pair   = alignUsesPairs(MANTA_ALIGNER_JUMP, k, b->scores.match, b->scores.mismatch, b->scores.open, b->scores.extend, b->scores.off_edge,
                                               b->jumpScore, 0, hMaxref[k]);
        const uint64_t stride = ((pair ? 2 : 1) * alignPtrSlabBytes(MANTA_ALIGNER_JUMP, kESet[k], hMaxref[k]) + 255) & ~uint64_t(255);
        int            grid   = int(std::min<size_t>(pair ? (cnt + 1) / 2 : cnt, size_t(maxWaves)));
        grid                  = int(std::max<size_t>(1, std::min<size_t>(size_t(grid), (wsBudget / 3) / stride)));
        grid                  = rt::roundGrid(grid);
        launches.push_back(Launch{k, grid, stride, slabBytes, pair});
        slabBytes += stride * uint64_t(grid);
      }
      if (launches.empty()) return;
      uint8_t* dWsAll = b->dPtrWs.as<uint8_t>(slabBytes + 256);
      for (size_t i = 0; i < launches.size(); ++i) {
        const Launch& l(launches[i]);
        AlignParams   P;
        P.tasks          = tasks;
        P.results        = results;
        P.cigar          = dCigar;
        P.task_ids       = bucketIds + uint64_t(l.k) * nSlots;
        P.n_tasks        = hCounts[l.k];
        P.n_tasks_dev    = nullptr;
        P.counter        = counters + l.k;
        P.ptr_ws         = dWsAll + l.slabOff;
        P.ptr_ws_stride  = l.stride;
        P.match          = b->scores.match;
        P.mismatch       = b->scores.mismatch;
        P.open           = b->scores.open;
        P.extend         = b->scores.extend;
        P.off_edge       = b->scores.off_edge;
        P.allow_edge_ins = 0;
        P.extra          = b->jumpScore;
        {
          rt::ScopedStream onSide(*sides[i % 3]);
          launchAlignKind(MANTA_ALIGNER_JUMP, l.k, l.grid, P, l.pair);
        }
        b->stats.n_align_launches++;
        b->stats.n_alignments += hCounts[l.k];
      }
      for (size_t i = 0; i < std::min<size_t>(launches.size(), 3); ++i) {
        b->sideDone[i].recordOn(*sides[i]);
        rt::curStreamWaits(b->sideDone[i]);
      }
      rt::sync();  // the next stage reads the results and may re-size the slab buffer
    };
    /// schedule -> round 1 -> re-align rule -> round 2 over the loci of S.pass_mask / S.pass_value, on the current stream + `sides`
    auto alignPass = [&](rt::Stream* const* sides, const int maxWaves, const char* what) {
      S.cigar = dCigar;
      rt::launch(spanning_schedule_kernel, glueGrid, 0, S);
      uint32_t hSmall[64];
      rt::d2h(hSmall, dSmall, sizeof(hSmall));
      alignRound(sides, maxWaves, hSmall, hSmall + 16, dTasks, dResults, dBuckets, dSmall + 72);
      stage(what);
      rt::launch(spanning_realign_kernel, rt::roundGrid(int(std::min<uint64_t>((nLoci + 63) / 64, uint64_t(std::max(1, ctx->cuCount * 8))))), 0, S);
      rt::d2h(hSmall, dSmall, sizeof(hSmall));
      alignRound(sides, maxWaves, hSmall + 32, hSmall + 48, dTasks2, dResults2, dBuckets2, dSmall + 88);
    };
    rt::Stream* const plainSides[3] = {&b->side[0], &b->side[1], &b->side[2]};

    // The early pass: after the first word length 94 % of the config-4/5 loci are final, while the tandem piles go through up to ten more
    // word lengths -- launches that are latency-bound on small lists and leave most of the device idle.  The jump aligner of the final loci
    // runs beside them: a mark kernel on the assembler's stream freezes "who is final" (span_mark_kernel), the pass itself runs on `early`
    // + CU-masked side streams (a persistent aligner grid on every CU would keep graph_big_kernel -- one workgroup owns a CU -- from ever
    // starting; MANTA_AMD_EARLY_RESERVE_CUS, default a quarter of the CUs, stay free for the rounds), the rest of the loci follow when the
    // assembler is done.  Not with stage gates (pipelined workers hold one stage at a time).  MANTA_AMD_EARLY_ALIGN=0 switches it off.
    static const bool earlyOff = std::getenv("MANTA_AMD_EARLY_ALIGN") && std::atoi(std::getenv("MANTA_AMD_EARLY_ALIGN")) == 0;
    const bool        early    = !earlyOff && !gates && as.useFast && as.bigRounds > 1 && !as.bigIds.empty();
    uint8_t*          dMask    = early ? b->dPassMask.as<uint8_t>(nLoci) : nullptr;
    uint64_t asmCnt[4];
    b->earlyLoci = 0;
    {
      GateLock only(gates, &StageGates::asmMu);
      std::unique_lock<std::mutex> streamedOnly(streamedAsmMu(ctx), std::defer_lock);  // see smallsvRunImpl
      if (as.streaming) streamedOnly.lock();
      as.stageQueued = false;  // (a run that failed behind its queued staging must not leave the flag to the next one)
      as.earlyStaged = false;
      b->evStart.record();
      if (early) {
        as.afterFirstRound = [&] {
          SpanMarkParams K;
          K.loci    = as.dLoci;
          K.n_loci  = nLoci;
          K.mask    = dMask;
          K.counter = dSmall + 120;
          rt::launch(span_mark_kernel, rt::roundGrid(int(std::min<uint64_t>((nLoci + 63) / 64, uint64_t(std::max(1, ctx->cuCount * 4))))), 0, K);
          b->evRound0.record();
        };
      }
      struct HookReset {
        AsmStage& a;
        ~HookReset() { a.afterFirstRound = nullptr; }
      } hookReset{as};
      as.launch();
      b->evAsm.record();
      if (early && as.firstRoundHookRan) {
        // (everything below is queued behind evRound0 on `early`; `main` keeps running the word-length rounds)
        static const int reserveEnv = std::getenv("MANTA_AMD_EARLY_RESERVE_CUS") ? std::atoi(std::getenv("MANTA_AMD_EARLY_RESERVE_CUS")) : -1;
        const int reserve = std::max(8, std::min(ctx->cuCount - 8, reserveEnv >= 0 ? (reserveEnv / 8) * 8 : ((ctx->cuCount / 4) / 8) * 8));
        if (b->sideEarlyReserved != reserve) {
          b->early.reset(new rt::Stream(ctx->cuCount, reserve));
          for (int i = 0; i < 3; ++i) b->sideEarly[i].reset(new rt::Stream(ctx->cuCount, reserve));
          b->sideEarlyReserved = reserve;
          if (dbg) std::fprintf(stderr, "manta_amd: early alignment pass: %d of %d CUs reserved for the word-length rounds (CU mask %s)\n", reserve, ctx->cuCount,
                                b->sideEarly[0]->masked() ? "set" : "not available");
        }
        rt::ScopedStream onEarly(*b->early);
        rt::curStreamWaits(b->evRound0);
        if (b->refsOnCopy) rt::curStreamWaits(b->refsReady);
        rt::Stream* const earlySides[3] = {b->sideEarly[0].get(), b->sideEarly[1].get(), b->sideEarly[2].get()};
        uint64_t cnt0[4];
        uint32_t nEarly = 0;
        rt::d2h(cnt0, as.dCnt, sizeof(cnt0));  // (text arena in use: at least what the final loci wrote -- the rounds keep adding)
        rt::d2h(&nEarly, dSmall + 120, sizeof(nEarly));
        b->earlyLoci = nEarly;
        if (nEarly) {
          // (room for a quarter more text than is there now, so that the second pass seldom has to move the buffer)
          S.cigar_cap  = std::min<uint64_t>(cigarScratch(cnt0[1] + cnt0[1] / 4 + 65536, 0), 0xfffffff0ull);
          S.pass_mask  = dMask;
          S.pass_value = 1;
          static const int wavesEnv = std::getenv("MANTA_AMD_EARLY_WAVES_PER_CU") ? std::atoi(std::getenv("MANTA_AMD_EARLY_WAVES_PER_CU")) : 0;
          const int maxWavesEarly = std::max(1, (ctx->cuCount - (b->sideEarly[0]->masked() ? reserve : 0)) * (wavesEnv > 0 ? wavesEnv : alignWavesPerCu()));
          alignPass(earlySides, maxWavesEarly, "aligned round 1 (early pass)");
          stage("aligned round 2 (early pass)");
        }
      }
      rt::d2h(asmCnt, as.dCnt, sizeof(asmCnt));  // (waits for the assembler)
      if (asmCnt[3]) {  // loci that did not fit the typical-case workspace (see smallsvRunImpl)
        as.rerunCapacityFailures(asmCnt[3]);
        rt::d2h(asmCnt, as.dCnt, sizeof(asmCnt));  // the text arena grew
      }
    }
    stage("assembled");
    // whole-batch calls: the assembler's outputs leave for the host now, beside the (last) alignment pass (as smallsvRunImpl)
    const bool earlyStage = b->stageBehindRun && b->whileAligning && !std::getenv("MANTA_AMD_NO_EARLY_STAGE");
    if (earlyStage) as.stageEarly(b->copy, asmCnt, b->evEarlyStaged);
    GateLock alignOnly(gates, &StageGates::alignMu);
    if (b->refsOnCopy) rt::curStreamWaits(b->refsReady);  // reference windows of a streamed upload (manta_spanning_upload)
    uint64_t cigarCap = 0;
    {
      uint64_t keepWords = 0;
      if (b->earlyLoci) {
        // second pass: fresh bucket lists (the allocator of the CIGAR scratch goes on where the early pass stopped)
        rt::d2h(&keepWords, dSmall + 64, sizeof(keepWords));
        keepWords = std::min<uint64_t>(keepWords, S.cigar_cap);
        rt::dzero(dSmall, sizeof(uint32_t) * 64);
        rt::dzero(dSmall + 72, sizeof(uint32_t) * 32);
        S.pass_mask  = dMask;
        S.pass_value = 0;
      }
      cigarCap = cigarScratch(asmCnt[1], keepWords);
      if (cigarCap + 16 > 0xffffffffull)
        return fail(ctx, MANTA_E_UNSUPPORTED, "manta_spanning_run: alignment scratch of this block exceeds 2^32 words; use smaller blocks "
                                           "(manta_spanning_batch splits a batch into blocks, manta_batch_plan_t::block_loci)");
      S.cigar_cap = cigarCap;
    }
    b->evSched.record();
    const int maxWaves = std::max(1, ctx->cuCount * alignWavesPerCu());
    alignPass(plainSides, maxWaves, "aligned round 1");
    launchPack(b, dTasks, dTasks2, dResults, dResults2, nullptr, dInfo, dCigar, cigarCap);
    b->evAlign.record();
    if (b->stageBehindRun) pipeStageEnqueue(b);
    if (earlyStage) {
      b->evEarlyStaged.sync();
      as.finishEarly();
      b->whileAligning();
    }
    rt::sync();
    alignOnly.release();
    if (as.streaming) {  // every chunk was consumed by the kernel, so this returns at once; it closes the copy stream's error state
      rt::ScopedStream onCopy(b->copy);
      rt::sync();
    }
    stage("aligned round 2");
    b->stats.assemble_ms = rt::elapsedMs(b->evStart, b->evAsm);
    b->stats.schedule_ms = rt::elapsedMs(b->evAsm, b->evSched);
    b->stats.align_ms    = rt::elapsedMs(b->evSched, b->evAlign);
    b->stats.total_ms    = rt::elapsedMs(b->evStart, b->evAlign);
    b->ran               = true;
    return MANTA_OK;
  } catch (const std::exception& e) {
    drainCopyStream(b);
    return fail(ctx, MANTA_E_HIP, e.what());
  }
}

extern "C" {

int manta_spanning_run(manta_spanning_t* b) { return spanningRunImpl(b, nullptr); }

int manta_spanning_stats(const manta_spanning_t* b, manta_smallsv_stats_t* stats)
{
  if (!b || !stats) return MANTA_E_INVALID_ARG;
  *stats = b->stats;
  return MANTA_OK;
}

int manta_spanning_output_sizes(const manta_spanning_t* b, uint64_t* contigs, uint64_t* seq_bytes, uint64_t* bits_words, uint64_t* cigar_words)
{
  if (!b || !contigs || !seq_bytes || !bits_words || !cigar_words) return MANTA_E_INVALID_ARG;
  if (!b->ran) return fail(b->ctx, MANTA_E_INVALID_ARG, "manta_spanning_output_sizes: run first");
  try {
    rt::setDevice(b->ctx->deviceId);  // the calling thread may have another device current (multi-GPU hosts)
    rt::ScopedStream onStream(const_cast<manta_spanning_t*>(b)->main);
    b->asmStage.outputSizes(*contigs, *seq_bytes, *bits_words);
    uint32_t hSmall[72];
    rt::d2h(hSmall, b->dSmall.p, sizeof(hSmall));
    uint64_t cig = 0;
    std::memcpy(&cig, hSmall + 64, sizeof(uint64_t));
    *cigar_words = cig + 64;
    return MANTA_OK;
  } catch (const std::exception& e) {
    return fail(b->ctx, MANTA_E_HIP, e.what());
  }
}

}  // extern "C"


extern "C" {

int manta_spanning_download(
    manta_spanning_t* b, manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs, manta_spanning_alignment_t* alignments,
    uint64_t contigs_cap, uint8_t* seq_arena, uint64_t seq_arena_cap, uint64_t* seq_arena_used, uint64_t* bits_arena,
    uint64_t bits_arena_cap, uint64_t* bits_arena_used, uint32_t* cigar_arena, uint64_t cigar_arena_cap,
    uint64_t* cigar_arena_used)
{
  if (!b) return MANTA_E_INVALID_ARG;
  manta_ctx_t* ctx = b->ctx;
  if (!b->ran) return fail(ctx, MANTA_E_INVALID_ARG, "manta_spanning_download: run first");
  if (!loci || !contigs || !alignments || !seq_arena || !bits_arena || !cigar_arena)
    return fail(ctx, MANTA_E_INVALID_ARG, "manta_spanning_download: null argument");
  try {
    rt::setDevice(ctx->deviceId);  // the calling thread may have another device current (multi-GPU hosts)
    rt::ScopedStream onStream(b->main);
    pipeStage(b);
    return spanningCompact(b, loci, contigs, alignments, 0, contigs_cap, seq_arena, seq_arena_cap, 0, seq_arena_used, bits_arena,
                           bits_arena_cap, 0, bits_arena_used, cigar_arena, cigar_arena_cap, 0, cigar_arena_used);
  } catch (const std::exception& e) {
    return fail(ctx, MANTA_E_HIP, e.what());
  }
}

}  // extern "C"



manta_ctx::~manta_ctx()
{
  for (manta_smallsv* p : smallPool) delete p;
  for (manta_spanning* p : spanPool) delete p;
}

