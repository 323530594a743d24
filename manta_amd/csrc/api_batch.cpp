// C-ABI implementation, part 2 (see api.cpp, api_internal.hpp): per-locus word lengths, pinned host memory, the whole-batch calls
// (block queue + worker pipelines inside a device) and the node-wide queue over several devices.
#include "api_internal.hpp"

namespace {

int setWordLengths(manta_ctx_t* ctx, AsmStage& as, uint32_t n_loci, const uint32_t* minWl, const uint32_t* maxWl)
{
  as.locusMinWl.clear();
  as.locusMaxWl.clear();
  if (!minWl && !maxWl) return MANTA_OK;
  if (!minWl || !maxWl || n_loci == 0) return fail(ctx, MANTA_E_INVALID_ARG, "set_word_lengths: both arrays (or neither) must be given");
  as.locusMinWl.assign(minWl, minWl + n_loci);
  as.locusMaxWl.assign(maxWl, maxWl + n_loci);
  return MANTA_OK;
}

/// what the workers of one whole-batch call share: the block queue, the bump allocators over the caller's arenas, the
/// first fatal error and the statistics
struct BatchShared {
  std::vector<uint32_t> blockOrder;  // block indices, most expensive first
  std::atomic<uint32_t> next{0};
  std::atomic<uint64_t> contigsUsed{0}, seqUsed{0}, bitsUsed{0}, cigarUsed{0};
  std::mutex            mu;
  std::mutex            kernelMu;  // MANTA_BATCH_SERIAL_KERNELS
  bool                  serialKernels = false;
  std::vector<StageGates> gates;   // one per device; used whenever a device has more than one worker
  bool                  pipelineStages = false;
  uint32_t*             sharedQueue = nullptr;  // manta_batch_plan_t::shared_queue: the block counter of several processes
  std::vector<uint32_t> lociOfCtx;  // loci processed per context (device)
  /// next position of the block queue: this call's own counter, or the counter shared by the processes of the node
  uint32_t takeNext() { return sharedQueue ? __atomic_fetch_add(sharedQueue, 1u, __ATOMIC_RELAXED) : next.fetch_add(1); }
  int                   fatal = MANTA_OK, worst = MANTA_OK;
  std::string           msg;
  manta_batch_stats_t   st{};
  void                  error(int code, const std::string& m, bool isFatal)
  {
    std::lock_guard<std::mutex> g(mu);
    if (isFatal) {
      if (fatal == MANTA_OK) {
        fatal = code;
        msg   = m;
      }
    } else if (worst == MANTA_OK) {
      worst = code;
      msg   = m;
    }
  }
  bool stop()
  {
    std::lock_guard<std::mutex> g(mu);
    return fatal != MANTA_OK;
  }
};

/// default block size of a whole-batch call: one block per call is the measured optimum as long as the
/// device-side arenas stay moderate; beyond that, equal blocks of at most 65536 loci / 4 GiB of read bases
uint32_t autoBlockLoci(const uint32_t n_loci, const uint64_t totalBases)
{
  const uint64_t byLoci  = (uint64_t(n_loci) + 65535) / 65536;
  const uint64_t byBases = (totalBases + (uint64_t(4) << 30) - 1) / (uint64_t(4) << 30);
  const uint64_t nBlocks = std::max<uint64_t>(1, std::max(byLoci, byBases));
  return uint32_t((uint64_t(n_loci) + nBlocks - 1) / nBlocks);
}

/// several devices (or processes) on one queue: blocks small enough that the queue can balance uneven costs -- about eight
/// per puller would be ideal -- but not below what keeps a device efficient.  Measured on MI355X with the LDS assembler pipeline
/// (config-2 loci, one device): 10 000 loci as one block 971 k loci/s, as blocks of 5 000 778 k, of 4 096 633 k (fixed per-block
/// host work and kernel tails) -- hence a floor of 8 192.
uint32_t nodeBlockLoci(const uint32_t n_loci, const uint64_t totalBases)
{
  const uint32_t one = autoBlockLoci(n_loci, totalBases);
  return std::max<uint32_t>(1, std::min<uint32_t>(one, std::max<uint32_t>(8192, n_loci / 64)));
}

/// contiguous blocks of `blockLoci` loci, ordered by decreasing cost (reads x bases, the same estimate the kernels'
/// work queue uses): EdgeRetrieverBin.cpp:38-57 hands out contiguous edge ranges too, but statically
void planBlocks(BatchShared& sh, uint32_t n_loci, uint32_t blockLoci, const uint64_t* read_off, const uint32_t* locus_read_begin,
                const uint32_t* read_len = nullptr)
{
  const uint32_t        nBlocks = (n_loci + blockLoci - 1) / blockLoci;
  std::vector<uint64_t> cost(nBlocks, 0);
  for (uint32_t b = 0; b < nBlocks; ++b) {
    const uint32_t l0 = b * blockLoci, l1 = std::min(n_loci, l0 + blockLoci);
    for (uint32_t l = l0; l < l1; ++l) {
      const uint32_t rb = locus_read_begin[l], re = locus_read_begin[l + 1];
      uint64_t bases = 0;
      if (read_off)
        bases = read_off[re] - read_off[rb];
      else
        for (uint32_t r = rb; r < re; ++r) bases += read_len[r];
      cost[b] += bases * uint64_t(re - rb);
    }
  }
  sh.blockOrder.resize(nBlocks);
  for (uint32_t b = 0; b < nBlocks; ++b) sh.blockOrder[b] = b;
  std::stable_sort(sh.blockOrder.begin(), sh.blockOrder.end(), [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
}

/// offsets of a block relative to the block's first element.  A block that starts where the batch's offsets start at zero -- the one-block
/// call of the metric -- takes the caller's array as it is: no 6.4 MB copy of the read offsets into fresh (page-faulting, pageable) memory
/// in front of every upload; other blocks are rebased into `out`, which belongs to the worker's pipeline object and keeps its pages.
template <typename T>
const T* rebase(PinnedBuf& out, const T* src, size_t first, size_t count)
{
  const T base = src[first];
  if (base == 0) return src + first;
  T* dst = out.as<T>(count);  // (page-locked: the upload of a later block is a DMA as well)
  for (size_t i = 0; i < count; ++i) dst[i] = src[first + i] - base;
  return dst;
}

}  // namespace

extern "C" {

int manta_host_alloc(uint64_t bytes, void** out)
{
  if (!out) return MANTA_E_INVALID_ARG;
  try {
    *out = rt::hostAlloc(size_t(bytes));
    return MANTA_OK;
  } catch (const std::exception& e) {
    g_createError = e.what();
    *out          = nullptr;
    return MANTA_E_HIP;
  }
}

void manta_host_free(void* p)
{
  if (p) rt::hostFree(p);
}

int manta_smallsv_set_word_lengths(manta_smallsv_t* b, uint32_t n_loci, const uint32_t* min_word_length, const uint32_t* max_word_length)
{
  if (!b) return MANTA_E_INVALID_ARG;
  return setWordLengths(b->ctx, b->asmStage, n_loci, min_word_length, max_word_length);
}

int manta_spanning_set_word_lengths(manta_spanning_t* b, uint32_t n_loci, const uint32_t* min_word_length, const uint32_t* max_word_length)
{
  if (!b) return MANTA_E_INVALID_ARG;
  return setWordLengths(b->ctx, b->asmStage, n_loci, min_word_length, max_word_length);
}

namespace {
int smallsvBatchImpl(
    manta_ctx_t* const* ctxs, const uint32_t nCtx, uint32_t* lociPerDevice, const manta_asm_options_t* opt, const manta_align_scores_t* scores, int32_t large_indel_score, uint32_t n_loci,
    const uint8_t* bases, const uint64_t* read_off, const uint32_t* locus_read_begin, const manta_packed_piles_t* piles, const uint8_t* refs,
    const uint64_t* ref_off, const manta_ref_cuts_t* cuts, const uint32_t* locus_min_word_length, const uint32_t* locus_max_word_length,
    manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs, manta_smallsv_alignment_t* alignments, uint64_t contigs_cap,
    uint8_t* seq_arena, uint64_t seq_arena_cap, uint64_t* seq_arena_used, uint64_t* bits_arena, uint64_t bits_arena_cap,
    uint64_t* bits_arena_used, uint32_t* cigar_arena, uint64_t cigar_arena_cap, uint64_t* cigar_arena_used,
    const manta_batch_plan_t* plan, manta_batch_stats_t* stats)
{
  if (!ctxs || nCtx == 0 || !ctxs[0]) return MANTA_E_INVALID_ARG;
  manta_ctx_t* ctx = ctxs[0];  // (call-level errors are reported here)
  if (!opt || !scores || n_loci == 0 || (!piles && (!bases || !read_off)) || !locus_read_begin || !refs || !ref_off || !cuts || !loci ||
      !contigs || !alignments || !seq_arena || !bits_arena || !cigar_arena)
    return fail(ctx, MANTA_E_INVALID_ARG, "manta_smallsv_batch: null argument or empty batch");
  if ((locus_min_word_length == nullptr) != (locus_max_word_length == nullptr))
    return fail(ctx, MANTA_E_INVALID_ARG, "manta_smallsv_batch: per-locus word lengths need both arrays");
  for (uint32_t l = 0; l < n_loci; ++l)
    if (locus_read_begin[l + 1] < locus_read_begin[l] || ref_off[l + 1] < ref_off[l])
      return fail(ctx, MANTA_E_INVALID_ARG, "manta_smallsv_batch: offsets not monotone");
  const uint64_t totalBases = piles ? 0 : read_off[locus_read_begin[n_loci]] - read_off[locus_read_begin[0]];
  const bool     shared     = (plan && plan->shared_queue) || nCtx > 1;  // several devices pull from the queue
  const uint32_t blockLoci  = (plan && plan->block_loci) ? plan->block_loci : (shared ? nodeBlockLoci(n_loci, totalBases) : autoBlockLoci(n_loci, totalBases));
  BatchShared    sh;
  sh.serialKernels  = plan && (plan->flags & MANTA_BATCH_SERIAL_KERNELS);
  sh.sharedQueue    = plan ? plan->shared_queue : nullptr;
  sh.gates          = std::vector<StageGates>(nCtx);
  sh.lociOfCtx.assign(nCtx, 0);
  planBlocks(sh, n_loci, blockLoci, piles ? nullptr : read_off, locus_read_begin, piles ? piles->read_len : nullptr);
  const uint32_t nBlocks    = uint32_t(sh.blockOrder.size());
  const uint32_t perCtx     = std::max(1u, (plan && plan->n_workers) ? plan->n_workers : 1u);
  const uint32_t nWorkers   = std::max(1u, std::min(nBlocks, perCtx * nCtx));
  sh.pipelineStages         = perCtx > 1 && !std::getenv("MANTA_AMD_NO_STAGE_GATES");  // (experiments: concurrent workers without the stage gates)
  if (sh.sharedQueue)  // blocks another process of the node takes stay marked; the caller merges (bench.py: the final gather)
    for (uint32_t l = 0; l < n_loci; ++l) {
      std::memset(&loci[l], 0, sizeof(loci[l]));
      loci[l].status = MANTA_E_NOT_TAKEN;
    }
  try {
    for (uint32_t c = 0; c < nCtx; ++c) {
      rt::setDevice(ctxs[c]->deviceId);
      while (ctxs[c]->smallPool.size() < (nWorkers + nCtx - 1) / nCtx) ctxs[c]->smallPool.push_back(new manta_smallsv(ctxs[c]));
    }
  } catch (const std::exception& e) {
    return fail(ctx, MANTA_E_HIP, e.what());
  }
  const double tStart = nowMs();
  auto         worker = [&](const uint32_t w) {
    manta_ctx_t*   ctx = ctxs[w % nCtx];  // (shadows the call-level context: this worker's device)
    manta_smallsv* b   = ctx->smallPool[w / nCtx];
    try {
      rt::setDevice(ctx->deviceId);
      // a pooled pipeline that served other options or scores: its bucket counts say nothing about this call's contigs
      if (std::memcmp(&b->opt, opt, sizeof(*opt)) != 0 || std::memcmp(&b->scores, scores, sizeof(*scores)) != 0 || b->largeIndel != large_indel_score)
        b->bucketHistory = false;
      b->opt           = *opt;
      b->scores        = *scores;
      b->largeIndel    = large_indel_score;
      b->streamUploads = !(plan && (plan->flags & MANTA_BATCH_NO_STREAMED_UPLOAD));
      b->asmStage.wavesPerCuCap = sh.pipelineStages ? kPipelinedAsmWavesPerCu : 0;
      PinnedBuf &rOffBuf(b->hostOff[0]), &fOffBuf(b->hostOff[1]), &lBegBuf(b->hostBegin);
      while (!sh.stop()) {
        const uint32_t qi = sh.takeNext();
        if (qi >= nBlocks) break;
        const uint32_t blk = sh.blockOrder[qi];
        const uint32_t l0 = blk * blockLoci, l1 = std::min(n_loci, l0 + blockLoci), n = l1 - l0;
        const uint32_t r0 = locus_read_begin[l0], r1 = locus_read_begin[l1];
        const uint64_t* rOff = piles ? nullptr : rebase(rOffBuf, read_off, r0, size_t(r1 - r0) + 1);
        const uint32_t* lBeg = rebase(lBegBuf, locus_read_begin, l0, size_t(n) + 1);
        const uint64_t* fOff = rebase(fOffBuf, ref_off, l0, size_t(n) + 1);
        setWordLengths(ctx, b->asmStage, n, locus_min_word_length ? locus_min_word_length + l0 : nullptr,
                       locus_max_word_length ? locus_max_word_length + l0 : nullptr);
        const double t0 = nowMs();
        int          rc;
        if (piles) {
          manta_packed_piles_t pl = *piles;
          pl.read_len            = piles->read_len + r0;
          pl.read_code_off       = piles->read_code_off + r0;
          pl.read_mask_off       = piles->read_mask_off + r0;
          pl.locus_read_begin    = lBeg;
          rc                     = manta_smallsv_upload_piles(b, n, &pl, refs + ref_off[l0], fOff, cuts + l0);
        } else {
          rc = manta_smallsv_upload(b, n, bases + read_off[r0], rOff, lBeg, refs + ref_off[l0], fOff, cuts + l0);
        }
        if (perItemCode(rc)) {  // a locus outside the supported envelope: this block's loci carry the code, the batch goes on
          for (uint32_t l = l0; l < l1; ++l) {
            std::memset(&loci[l], 0, sizeof(loci[l]));
            loci[l].status = rc;
          }
          sh.error(rc, lastErrorOf(ctx), false);
          continue;
        }
        if (rc != MANTA_OK) {
          sh.error(rc, lastErrorOf(ctx), true);
          break;
        }
        const double t1 = nowMs();
        // compaction into the caller's arrays: a few contiguous locus ranges, one host thread each.  Pass 1 sizes the ranges,
        // the block then reserves its region of the caller's arenas, pass 2 writes every range at its own offset.  The assembler's
        // share of it (locus and contig records, contig text, read sets: most of the bytes) runs WHILE THE ALIGNERS DO, from the run's
        // hook (manta_smallsv::whileAligning); what is left for after the run are the alignment records and the CIGARs.
        struct Range {
          uint64_t nC = 0, nS = 0, nB = 0, nG = 0, cells = 0, ptrBytes = 0;
          uint64_t c0 = 0, s0 = 0, b0 = 0, g0 = 0;
          int      rc = MANTA_OK, rcAsm = MANTA_OK;
        };
        const unsigned     parts = hostParts(uint64_t(n) * 4);  // (a block of >= 1024 loci is worth the threads)
        std::vector<Range> rg(parts);
        uint64_t           nC = 0, nS = 0, nB = 0, nG = 0, cBase = 0, sBase = 0, bBase = 0;
        bool               asmDone = false, asmFits = true;
        double             hookMs = 0;
        // (contigAt: the per-slot contig records from the hook, the packed ones after the run -- the same records)
        auto asmShare = [&](auto contigAt) {
          hostParallel(n, parts, [&](unsigned t, uint64_t a, uint64_t z) { b->asmStage.rangeSizes(contigAt, uint32_t(a), uint32_t(z), rg[t].nC, rg[t].nS, rg[t].nB); });
          for (unsigned t = 0; t < parts; ++t) {
            rg[t].c0 = nC, rg[t].s0 = nS, rg[t].b0 = nB;
            nC += rg[t].nC, nS += rg[t].nS, nB += rg[t].nB;
          }
          cBase   = sh.contigsUsed.fetch_add(nC), sBase = sh.seqUsed.fetch_add(nS), bBase = sh.bitsUsed.fetch_add(nB);
          asmDone = true;
          asmFits = cBase + nC <= contigs_cap && sBase + nS <= seq_arena_cap && bBase + nB <= bits_arena_cap;
          if (!asmFits) return;
          hostParallel(n, parts, [&](unsigned t, uint64_t a, uint64_t z) {
            const Range& r(rg[t]);
            rg[t].rcAsm = b->asmStage.compact(contigAt, loci + l0, contigs + cBase + r.c0, r.nC, seq_arena + sBase + r.s0, r.nS, nullptr, bits_arena + bBase + r.b0, r.nB,
                                              nullptr, cBase + r.c0, sBase + r.s0, bBase + r.b0, uint32_t(a), uint32_t(z));
          });
        };
        b->whileAligning = [&] {
          const double h0 = nowMs();
          asmShare(AsmStage::SparseContigs{&b->asmStage});
          hookMs = nowMs() - h0;
        };
        {
          std::unique_lock<std::mutex> only(sh.kernelMu, std::defer_lock);
          if (sh.serialKernels) only.lock();
          b->stageBehindRun = true;
          rc = smallsvRunImpl(b, sh.pipelineStages ? &sh.gates[w % nCtx] : nullptr);
        }
        b->whileAligning = nullptr;
        if (rc != MANTA_OK) {
          sh.error(rc, lastErrorOf(ctx), true);
          break;
        }
        const double t2 = nowMs();
        {
          rt::ScopedStream onStream(b->main);
          pipeStage(b);
        }
        const double tStage = nowMs();
        if (!asmDone) asmShare(PackedContigs<manta_smallsv>{b});  // (the run did not call the hook: MANTA_AMD_NO_EARLY_STAGE)
        hostParallel(n, parts, [&](unsigned t, uint64_t a, uint64_t z) { rg[t].nG = smallsvCigarWords(b, uint32_t(a), uint32_t(z)); });
        for (unsigned t = 0; t < parts; ++t) {
          rg[t].g0 = nG;
          nG += rg[t].nG;
        }
        const double tSizes = nowMs();
        const uint64_t gBase = sh.cigarUsed.fetch_add(nG);
        if (!asmFits || gBase + nG > cigar_arena_cap) {
          sh.error(MANTA_E_CAPACITY, "manta_smallsv_batch: caller arenas too small", true);
          break;
        }
        hostParallel(n, parts, [&](unsigned t, uint64_t a, uint64_t z) {
          const Range& r(rg[t]);
          if (r.rcAsm != MANTA_OK && !perItemCode(r.rcAsm)) {
            rg[t].rc = r.rcAsm;
            return;
          }
          rg[t].rc = smallsvCompactAlign(b, loci + l0, alignments, cigar_arena + gBase + r.g0, r.nG, gBase + r.g0, nullptr, uint32_t(a), uint32_t(z), &rg[t].cells,
                                         &rg[t].ptrBytes, r.rcAsm);
        });
        rc = MANTA_OK;
        b->stats.dp_cells = b->stats.ptr_matrix_bytes = 0;
        for (const Range& r : rg) {
          if (r.rc != MANTA_OK && (rc == MANTA_OK || !perItemCode(r.rc))) rc = r.rc;
          b->stats.dp_cells += r.cells;
          b->stats.ptr_matrix_bytes += r.ptrBytes;
        }
        const double t3 = nowMs();
        if (std::getenv("MANTA_AMD_DEBUG_TIMING"))
          std::fprintf(stderr, "manta_amd: smallsv_batch block: upload %.2f ms, kernels %.2f (of which the assembler's share of the compaction, beside the aligners: %.2f), "
                               "stage-out %.2f, sizes %.2f, compact %.2f\n", t1 - t0, t2 - t1, hookMs, tStage - t2, tSizes - tStage, t3 - tSizes);
        if (rc != MANTA_OK) {
          sh.error(rc, lastErrorOf(ctx), !perItemCode(rc));
          if (!perItemCode(rc)) break;
        }
        std::lock_guard<std::mutex> g(sh.mu);
        sh.lociOfCtx[w % nCtx] += n;
        sh.st.h2d_ms += t1 - t0;
        sh.st.kernel_ms += t2 - t1;
        sh.st.d2h_ms += t3 - t2;
        sh.st.assemble_ms += b->stats.assemble_ms;
        sh.st.schedule_ms += b->stats.schedule_ms;
        sh.st.align_ms += b->stats.align_ms;
        sh.st.n_alignments += b->stats.n_alignments;
        sh.st.n_align_launches += b->stats.n_align_launches;
        sh.st.dp_cells += b->stats.dp_cells;
        sh.st.ptr_matrix_bytes += b->stats.ptr_matrix_bytes;
        sh.st.n_loci_lds_small += b->asmStage.fastIds.size();
        sh.st.n_loci_lds_big += b->asmStage.bigIds.size();
        sh.st.n_loci_handed_back += b->asmStage.ldsFallbacks;
        sh.st.n_loci_general += b->asmStage.useFast ? b->asmStage.genIds.size() : size_t(n);
        sh.st.h2d_bytes += (piles ? b->asmStage.plBytes : (read_off[r1] - read_off[r0]) + 8ull * (r1 - r0 + 1)) + (ref_off[l1] - ref_off[l0]) +
                           12ull * (n + 1) + 16ull * n;
        sh.st.d2h_bytes += pipeStagedBytes(b);
      }
    } catch (const std::exception& e) {
      sh.error(MANTA_E_HIP, e.what(), true);
    }
  };
  std::vector<std::thread> threads;
  for (uint32_t w = 1; w < nWorkers; ++w) threads.emplace_back(worker, w);
  worker(0);
  for (std::thread& t : threads) t.join();
  sh.st.wall_ms   = nowMs() - tStart;
  sh.st.n_blocks  = nBlocks;
  sh.st.n_workers = nWorkers;
  if (stats) *stats = sh.st;
  if (lociPerDevice) for (uint32_t c = 0; c < nCtx; ++c) lociPerDevice[c] = sh.lociOfCtx[c];
  if (seq_arena_used) *seq_arena_used = sh.seqUsed.load();
  if (bits_arena_used) *bits_arena_used = sh.bitsUsed.load();
  if (cigar_arena_used) *cigar_arena_used = sh.cigarUsed.load();
  if (sh.fatal != MANTA_OK) return fail(ctx, sh.fatal, sh.msg);
  if (sh.worst != MANTA_OK) return fail(ctx, sh.worst, sh.msg);
  return MANTA_OK;
}
}  // namespace

int manta_smallsv_batch(
    manta_ctx_t* ctx, const manta_asm_options_t* opt, const manta_align_scores_t* scores, int32_t large_indel_score, uint32_t n_loci,
    const uint8_t* bases, const uint64_t* read_off, const uint32_t* locus_read_begin, const uint8_t* refs, const uint64_t* ref_off,
    const manta_ref_cuts_t* cuts, const uint32_t* locus_min_word_length, const uint32_t* locus_max_word_length,
    manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs, manta_smallsv_alignment_t* alignments, uint64_t contigs_cap,
    uint8_t* seq_arena, uint64_t seq_arena_cap, uint64_t* seq_arena_used, uint64_t* bits_arena, uint64_t bits_arena_cap,
    uint64_t* bits_arena_used, uint32_t* cigar_arena, uint64_t cigar_arena_cap, uint64_t* cigar_arena_used,
    const manta_batch_plan_t* plan, manta_batch_stats_t* stats)
{
  return smallsvBatchImpl(&ctx, 1, nullptr, opt, scores, large_indel_score, n_loci, bases, read_off, locus_read_begin, nullptr, refs, ref_off, cuts,
                          locus_min_word_length, locus_max_word_length, loci, contigs, alignments, contigs_cap, seq_arena, seq_arena_cap,
                          seq_arena_used, bits_arena, bits_arena_cap, bits_arena_used, cigar_arena, cigar_arena_cap, cigar_arena_used, plan, stats);
}

int manta_smallsv_batch_piles(
    manta_ctx_t* ctx, const manta_asm_options_t* opt, const manta_align_scores_t* scores, int32_t large_indel_score, uint32_t n_loci,
    const manta_packed_piles_t* piles, const uint8_t* refs, const uint64_t* ref_off, const manta_ref_cuts_t* cuts,
    const uint32_t* locus_min_word_length, const uint32_t* locus_max_word_length, manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs,
    manta_smallsv_alignment_t* alignments, uint64_t contigs_cap, uint8_t* seq_arena, uint64_t seq_arena_cap, uint64_t* seq_arena_used,
    uint64_t* bits_arena, uint64_t bits_arena_cap, uint64_t* bits_arena_used, uint32_t* cigar_arena, uint64_t cigar_arena_cap,
    uint64_t* cigar_arena_used, const manta_batch_plan_t* plan, manta_batch_stats_t* stats)
{
  if (!ctx) return MANTA_E_INVALID_ARG;
  const int rc = checkPiles(ctx, piles, "manta_smallsv_batch_piles");
  if (rc != MANTA_OK) return rc;
  return smallsvBatchImpl(&ctx, 1, nullptr, opt, scores, large_indel_score, n_loci, nullptr, nullptr, piles->locus_read_begin, piles, refs, ref_off, cuts,
                          locus_min_word_length, locus_max_word_length, loci, contigs, alignments, contigs_cap, seq_arena, seq_arena_cap,
                          seq_arena_used, bits_arena, bits_arena_cap, bits_arena_used, cigar_arena, cigar_arena_cap, cigar_arena_used, plan, stats);
}

namespace {
int spanningBatchImpl(
    manta_ctx_t* const* ctxs, const uint32_t nCtx, uint32_t* lociPerDevice, const manta_asm_options_t* opt, const manta_align_scores_t* scores, int32_t jump_score, uint32_t n_loci,
    const uint8_t* bases, const uint64_t* read_off, const uint32_t* locus_read_begin, const manta_packed_piles_t* piles, const uint8_t* refs1, const uint64_t* ref1_off,
    const uint8_t* refs2, const uint64_t* ref2_off, const manta_jump_cuts_t* cuts, const uint32_t* locus_min_word_length,
    const uint32_t* locus_max_word_length, manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs,
    manta_spanning_alignment_t* alignments, uint64_t contigs_cap, uint8_t* seq_arena, uint64_t seq_arena_cap, uint64_t* seq_arena_used,
    uint64_t* bits_arena, uint64_t bits_arena_cap, uint64_t* bits_arena_used, uint32_t* cigar_arena, uint64_t cigar_arena_cap,
    uint64_t* cigar_arena_used, const manta_batch_plan_t* plan, manta_batch_stats_t* stats)
{
  if (!ctxs || nCtx == 0 || !ctxs[0]) return MANTA_E_INVALID_ARG;
  manta_ctx_t* ctx = ctxs[0];  // (call-level errors are reported here)
  if (!opt || !scores || n_loci == 0 || (!piles && (!bases || !read_off)) || !locus_read_begin || !refs1 || !ref1_off || !refs2 || !ref2_off || !cuts ||
      !loci || !contigs || !alignments || !seq_arena || !bits_arena || !cigar_arena)
    return fail(ctx, MANTA_E_INVALID_ARG, "manta_spanning_batch: null argument or empty batch");
  if (scores->is_allow_edge_insertion) return fail(ctx, MANTA_E_INVALID_ARG, "GlobalJumpAligner does not support isAllowEdgeInsertion");
  if ((locus_min_word_length == nullptr) != (locus_max_word_length == nullptr))
    return fail(ctx, MANTA_E_INVALID_ARG, "manta_spanning_batch: per-locus word lengths need both arrays");
  for (uint32_t l = 0; l < n_loci; ++l)
    if (locus_read_begin[l + 1] < locus_read_begin[l] || ref1_off[l + 1] < ref1_off[l] || ref2_off[l + 1] < ref2_off[l])
      return fail(ctx, MANTA_E_INVALID_ARG, "manta_spanning_batch: offsets not monotone");
  const uint64_t totalBases = piles ? 0 : read_off[locus_read_begin[n_loci]] - read_off[locus_read_begin[0]];
  const bool     shared     = (plan && plan->shared_queue) || nCtx > 1;
  const uint32_t blockLoci  = (plan && plan->block_loci) ? plan->block_loci : (shared ? nodeBlockLoci(n_loci, totalBases) : autoBlockLoci(n_loci, totalBases));
  BatchShared    sh;
  sh.serialKernels  = plan && (plan->flags & MANTA_BATCH_SERIAL_KERNELS);
  sh.sharedQueue    = plan ? plan->shared_queue : nullptr;
  sh.gates          = std::vector<StageGates>(nCtx);
  sh.lociOfCtx.assign(nCtx, 0);
  planBlocks(sh, n_loci, blockLoci, piles ? nullptr : read_off, locus_read_begin, piles ? piles->read_len : nullptr);
  const uint32_t nBlocks  = uint32_t(sh.blockOrder.size());
  const uint32_t perCtx   = std::max(1u, (plan && plan->n_workers) ? plan->n_workers : 1u);
  const uint32_t nWorkers = std::max(1u, std::min(nBlocks, perCtx * nCtx));
  sh.pipelineStages       = perCtx > 1 && !std::getenv("MANTA_AMD_NO_STAGE_GATES");  // (experiments: concurrent workers without the stage gates)
  if (sh.sharedQueue)
    for (uint32_t l = 0; l < n_loci; ++l) {
      std::memset(&loci[l], 0, sizeof(loci[l]));
      loci[l].status = MANTA_E_NOT_TAKEN;
    }
  try {
    for (uint32_t c = 0; c < nCtx; ++c) {
      rt::setDevice(ctxs[c]->deviceId);
      while (ctxs[c]->spanPool.size() < (nWorkers + nCtx - 1) / nCtx) ctxs[c]->spanPool.push_back(new manta_spanning(ctxs[c]));
    }
  } catch (const std::exception& e) {
    return fail(ctx, MANTA_E_HIP, e.what());
  }
  const double tStart = nowMs();
  auto         worker = [&](const uint32_t w) {
    manta_ctx_t*    ctx = ctxs[w % nCtx];  // (shadows the call-level context: this worker's device)
    manta_spanning* b   = ctx->spanPool[w / nCtx];
    try {
      rt::setDevice(ctx->deviceId);
      b->opt       = *opt;
      b->scores    = *scores;
      b->jumpScore = jump_score;
      b->streamUploads = !(plan && (plan->flags & MANTA_BATCH_NO_STREAMED_UPLOAD));
      b->asmStage.wavesPerCuCap = sh.pipelineStages ? kPipelinedAsmWavesPerCu : 0;
      PinnedBuf &rOffBuf(b->hostOff[0]), &f1OffBuf(b->hostOff[1]), &f2OffBuf(b->hostOff[2]), &lBegBuf(b->hostBegin);
      while (!sh.stop()) {
        const uint32_t qi = sh.takeNext();
        if (qi >= nBlocks) break;
        const uint32_t blk = sh.blockOrder[qi];
        const uint32_t l0 = blk * blockLoci, l1 = std::min(n_loci, l0 + blockLoci), n = l1 - l0;
        const uint32_t r0 = locus_read_begin[l0], r1 = locus_read_begin[l1];
        const uint64_t* rOff  = piles ? nullptr : rebase(rOffBuf, read_off, r0, size_t(r1 - r0) + 1);
        const uint32_t* lBeg  = rebase(lBegBuf, locus_read_begin, l0, size_t(n) + 1);
        const uint64_t* f1Off = rebase(f1OffBuf, ref1_off, l0, size_t(n) + 1);
        const uint64_t* f2Off = rebase(f2OffBuf, ref2_off, l0, size_t(n) + 1);
        setWordLengths(ctx, b->asmStage, n, locus_min_word_length ? locus_min_word_length + l0 : nullptr,
                       locus_max_word_length ? locus_max_word_length + l0 : nullptr);
        const double t0 = nowMs();
        int          rc;
        if (piles) {
          manta_packed_piles_t pl = *piles;
          pl.read_len            = piles->read_len + r0;
          pl.read_code_off       = piles->read_code_off + r0;
          pl.read_mask_off       = piles->read_mask_off + r0;
          pl.locus_read_begin    = lBeg;
          rc = manta_spanning_upload_piles(b, n, &pl, refs1 + ref1_off[l0], f1Off, refs2 + ref2_off[l0], f2Off, cuts + l0);
        } else {
          rc = manta_spanning_upload(b, n, bases + read_off[r0], rOff, lBeg, refs1 + ref1_off[l0], f1Off, refs2 + ref2_off[l0], f2Off, cuts + l0);
        }
        if (perItemCode(rc)) {  // a locus outside the supported envelope: this block's loci carry the code, the batch goes on
          for (uint32_t l = l0; l < l1; ++l) {
            std::memset(&loci[l], 0, sizeof(loci[l]));
            loci[l].status = rc;
          }
          sh.error(rc, lastErrorOf(ctx), false);
          continue;
        }
        if (rc != MANTA_OK) {
          sh.error(rc, lastErrorOf(ctx), true);
          break;
        }
        const double t1 = nowMs();
        // compaction in a few locus ranges, one host thread each; the assembler's share of it while the last alignment pass runs (as smallsvBatchImpl)
        struct Range {
          uint64_t nC = 0, nS = 0, nB = 0, nG = 0, cells = 0, ptrBytes = 0;
          uint64_t c0 = 0, s0 = 0, b0 = 0, g0 = 0;
          int      rc = MANTA_OK, rcAsm = MANTA_OK;
        };
        const unsigned     parts = hostParts(uint64_t(n) * 4);
        std::vector<Range> rg(parts);
        uint64_t           nC = 0, nS = 0, nB = 0, nG = 0, cBase = 0, sBase = 0, bBase = 0;
        bool               asmDone = false, asmFits = true;
        auto asmShare = [&](auto contigAt) {
          hostParallel(n, parts, [&](unsigned t, uint64_t a, uint64_t z) { b->asmStage.rangeSizes(contigAt, uint32_t(a), uint32_t(z), rg[t].nC, rg[t].nS, rg[t].nB); });
          for (unsigned t = 0; t < parts; ++t) {
            rg[t].c0 = nC, rg[t].s0 = nS, rg[t].b0 = nB;
            nC += rg[t].nC, nS += rg[t].nS, nB += rg[t].nB;
          }
          cBase   = sh.contigsUsed.fetch_add(nC), sBase = sh.seqUsed.fetch_add(nS), bBase = sh.bitsUsed.fetch_add(nB);
          asmDone = true;
          asmFits = cBase + nC <= contigs_cap && sBase + nS <= seq_arena_cap && bBase + nB <= bits_arena_cap;
          if (!asmFits) return;
          hostParallel(n, parts, [&](unsigned t, uint64_t a, uint64_t z) {
            const Range& r(rg[t]);
            rg[t].rcAsm = b->asmStage.compact(contigAt, loci + l0, contigs + cBase + r.c0, r.nC, seq_arena + sBase + r.s0, r.nS, nullptr, bits_arena + bBase + r.b0, r.nB,
                                              nullptr, cBase + r.c0, sBase + r.s0, bBase + r.b0, uint32_t(a), uint32_t(z));
          });
        };
        b->whileAligning = [&] { asmShare(AsmStage::SparseContigs{&b->asmStage}); };
        {
          std::unique_lock<std::mutex> only(sh.kernelMu, std::defer_lock);
          if (sh.serialKernels) only.lock();
          b->stageBehindRun = true;
          rc = spanningRunImpl(b, sh.pipelineStages ? &sh.gates[w % nCtx] : nullptr);
        }
        b->whileAligning = nullptr;
        if (rc != MANTA_OK) {
          sh.error(rc, lastErrorOf(ctx), true);
          break;
        }
        const double t2 = nowMs();
        {
          rt::ScopedStream onStream(b->main);
          pipeStage(b);
        }
        if (!asmDone) asmShare(PackedContigs<manta_spanning>{b});  // (the run did not call the hook: MANTA_AMD_NO_EARLY_STAGE)
        hostParallel(n, parts, [&](unsigned t, uint64_t a, uint64_t z) { rg[t].nG = spanningCigarWords(b, uint32_t(a), uint32_t(z)); });
        for (unsigned t = 0; t < parts; ++t) {
          rg[t].g0 = nG;
          nG += rg[t].nG;
        }
        const uint64_t gBase = sh.cigarUsed.fetch_add(nG);
        if (!asmFits || gBase + nG > cigar_arena_cap) {
          sh.error(MANTA_E_CAPACITY, "manta_spanning_batch: caller arenas too small", true);
          break;
        }
        hostParallel(n, parts, [&](unsigned t, uint64_t a, uint64_t z) {
          const Range& r(rg[t]);
          if (r.rcAsm != MANTA_OK && !perItemCode(r.rcAsm)) {
            rg[t].rc = r.rcAsm;
            return;
          }
          rg[t].rc = spanningCompactAlign(b, loci + l0, alignments, cigar_arena + gBase + r.g0, r.nG, gBase + r.g0, nullptr, uint32_t(a), uint32_t(z), &rg[t].cells,
                                          &rg[t].ptrBytes, r.rcAsm);
        });
        rc = MANTA_OK;
        b->stats.dp_cells = b->stats.ptr_matrix_bytes = 0;
        for (const Range& r : rg) {
          if (r.rc != MANTA_OK && (rc == MANTA_OK || !perItemCode(r.rc))) rc = r.rc;
          b->stats.dp_cells += r.cells;
          b->stats.ptr_matrix_bytes += r.ptrBytes;
        }
        const double t3 = nowMs();
        if (rc != MANTA_OK) {
          sh.error(rc, lastErrorOf(ctx), !perItemCode(rc));
          if (!perItemCode(rc)) break;
        }
        std::lock_guard<std::mutex> g(sh.mu);
        sh.lociOfCtx[w % nCtx] += n;
        sh.st.h2d_ms += t1 - t0;
        sh.st.kernel_ms += t2 - t1;
        sh.st.d2h_ms += t3 - t2;
        sh.st.assemble_ms += b->stats.assemble_ms;
        sh.st.schedule_ms += b->stats.schedule_ms;
        sh.st.align_ms += b->stats.align_ms;
        sh.st.n_alignments += b->stats.n_alignments;
        sh.st.n_align_launches += b->stats.n_align_launches;
        sh.st.dp_cells += b->stats.dp_cells;
        sh.st.ptr_matrix_bytes += b->stats.ptr_matrix_bytes;
        sh.st.n_loci_lds_small += b->asmStage.fastIds.size();
        sh.st.n_loci_lds_big += b->asmStage.bigIds.size();
        sh.st.n_loci_handed_back += b->asmStage.ldsFallbacks;
        sh.st.n_loci_general += b->asmStage.useFast ? b->asmStage.genIds.size() : size_t(n);
        sh.st.h2d_bytes += (piles ? b->asmStage.plBytes : (read_off[r1] - read_off[r0]) + 8ull * (r1 - r0 + 1)) + (ref1_off[l1] - ref1_off[l0]) + (ref2_off[l1] - ref2_off[l0]) +
                           20ull * (n + 1) + 16ull * n;
        sh.st.d2h_bytes += pipeStagedBytes(b);
      }
    } catch (const std::exception& e) {
      sh.error(MANTA_E_HIP, e.what(), true);
    }
  };
  std::vector<std::thread> threads;
  for (uint32_t w = 1; w < nWorkers; ++w) threads.emplace_back(worker, w);
  worker(0);
  for (std::thread& t : threads) t.join();
  sh.st.wall_ms   = nowMs() - tStart;
  sh.st.n_blocks  = nBlocks;
  sh.st.n_workers = nWorkers;
  if (stats) *stats = sh.st;
  if (lociPerDevice) for (uint32_t c = 0; c < nCtx; ++c) lociPerDevice[c] = sh.lociOfCtx[c];
  if (seq_arena_used) *seq_arena_used = sh.seqUsed.load();
  if (bits_arena_used) *bits_arena_used = sh.bitsUsed.load();
  if (cigar_arena_used) *cigar_arena_used = sh.cigarUsed.load();
  if (sh.fatal != MANTA_OK) return fail(ctx, sh.fatal, sh.msg);
  if (sh.worst != MANTA_OK) return fail(ctx, sh.worst, sh.msg);
  return MANTA_OK;
}
}  // namespace

int manta_spanning_batch(
    manta_ctx_t* ctx, const manta_asm_options_t* opt, const manta_align_scores_t* scores, int32_t jump_score, uint32_t n_loci,
    const uint8_t* bases, const uint64_t* read_off, const uint32_t* locus_read_begin, const uint8_t* refs1, const uint64_t* ref1_off,
    const uint8_t* refs2, const uint64_t* ref2_off, const manta_jump_cuts_t* cuts, const uint32_t* locus_min_word_length,
    const uint32_t* locus_max_word_length, manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs,
    manta_spanning_alignment_t* alignments, uint64_t contigs_cap, uint8_t* seq_arena, uint64_t seq_arena_cap, uint64_t* seq_arena_used,
    uint64_t* bits_arena, uint64_t bits_arena_cap, uint64_t* bits_arena_used, uint32_t* cigar_arena, uint64_t cigar_arena_cap,
    uint64_t* cigar_arena_used, const manta_batch_plan_t* plan, manta_batch_stats_t* stats)
{
  return spanningBatchImpl(&ctx, 1, nullptr, opt, scores, jump_score, n_loci, bases, read_off, locus_read_begin, nullptr, refs1, ref1_off, refs2, ref2_off,
                           cuts, locus_min_word_length, locus_max_word_length, loci, contigs, alignments, contigs_cap, seq_arena, seq_arena_cap,
                           seq_arena_used, bits_arena, bits_arena_cap, bits_arena_used, cigar_arena, cigar_arena_cap, cigar_arena_used, plan, stats);
}

/* the same with the read piles in packed form (what manta_read_piles_batch emits) */
int manta_spanning_batch_piles(
    manta_ctx_t* ctx, const manta_asm_options_t* opt, const manta_align_scores_t* scores, int32_t jump_score, uint32_t n_loci,
    const manta_packed_piles_t* piles, const uint8_t* refs1, const uint64_t* ref1_off, const uint8_t* refs2, const uint64_t* ref2_off,
    const manta_jump_cuts_t* cuts, const uint32_t* locus_min_word_length, const uint32_t* locus_max_word_length, manta_asm_locus_result_t* loci,
    manta_asm_contig_t* contigs, manta_spanning_alignment_t* alignments, uint64_t contigs_cap, uint8_t* seq_arena, uint64_t seq_arena_cap,
    uint64_t* seq_arena_used, uint64_t* bits_arena, uint64_t bits_arena_cap, uint64_t* bits_arena_used, uint32_t* cigar_arena,
    uint64_t cigar_arena_cap, uint64_t* cigar_arena_used, const manta_batch_plan_t* plan, manta_batch_stats_t* stats)
{
  if (!ctx) return MANTA_E_INVALID_ARG;
  const int rc = checkPiles(ctx, piles, "manta_spanning_batch_piles");
  if (rc != MANTA_OK) return rc;
  return spanningBatchImpl(&ctx, 1, nullptr, opt, scores, jump_score, n_loci, nullptr, nullptr, piles->locus_read_begin, piles, refs1, ref1_off, refs2,
                           ref2_off, cuts, locus_min_word_length, locus_max_word_length, loci, contigs, alignments, contigs_cap, seq_arena,
                           seq_arena_cap, seq_arena_used, bits_arena, bits_arena_cap, bits_arena_used, cigar_arena, cigar_arena_cap, cigar_arena_used,
                           plan, stats);
}

/* ------------------------------------------------------------------------------------------------------
 * manta_node_*: the GPUs of one node behind one block queue (include/manta_amd.h)
 * ---------------------------------------------------------------------------------------------------- */
struct manta_node {
  std::vector<manta_ctx_t*> ctxs;
  ~manta_node()
  {
    for (manta_ctx_t* c : ctxs) manta_ctx_destroy(c);
  }
};

int manta_node_create(const int32_t* device_ids, uint32_t n_devices, manta_node_t** out)
{
  if (!out || !device_ids || n_devices == 0) {
    g_createError = "manta_node_create: null argument or no device";
    return MANTA_E_INVALID_ARG;
  }
  *out = nullptr;
  std::unique_ptr<manta_node> node(new manta_node);
  for (uint32_t d = 0; d < n_devices; ++d) {
    manta_ctx_t* c  = nullptr;
    const int    rc = manta_ctx_create(device_ids[d], &c);
    if (rc != MANTA_OK) return rc;  // (the contexts created so far go with `node`)
    node->ctxs.push_back(c);
  }
  *out = node.release();
  return MANTA_OK;
}

void manta_node_destroy(manta_node_t* node) { delete node; }

uint32_t manta_node_device_count(const manta_node_t* node) { return node ? uint32_t(node->ctxs.size()) : 0u; }

const char* manta_node_last_error(const manta_node_t* node)
{
  return (node && !node->ctxs.empty()) ? manta_last_error(node->ctxs[0]) : g_createError.c_str();
}

int manta_node_smallsv_batch(
    manta_node_t* node, const manta_asm_options_t* opt, const manta_align_scores_t* scores, int32_t large_indel_score, uint32_t n_loci,
    const uint8_t* bases, const uint64_t* read_off, const uint32_t* locus_read_begin, const uint8_t* refs, const uint64_t* ref_off,
    const manta_ref_cuts_t* cuts, const uint32_t* locus_min_word_length, const uint32_t* locus_max_word_length,
    manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs, manta_smallsv_alignment_t* alignments, uint64_t contigs_cap,
    uint8_t* seq_arena, uint64_t seq_arena_cap, uint64_t* seq_arena_used, uint64_t* bits_arena, uint64_t bits_arena_cap,
    uint64_t* bits_arena_used, uint32_t* cigar_arena, uint64_t cigar_arena_cap, uint64_t* cigar_arena_used,
    const manta_batch_plan_t* plan, manta_batch_stats_t* stats, uint32_t* loci_per_device)
{
  if (!node) return MANTA_E_INVALID_ARG;
  return smallsvBatchImpl(node->ctxs.data(), uint32_t(node->ctxs.size()), loci_per_device, opt, scores, large_indel_score, n_loci, bases, read_off,
                          locus_read_begin, nullptr, refs, ref_off, cuts, locus_min_word_length, locus_max_word_length, loci, contigs, alignments,
                          contigs_cap, seq_arena, seq_arena_cap, seq_arena_used, bits_arena, bits_arena_cap, bits_arena_used, cigar_arena,
                          cigar_arena_cap, cigar_arena_used, plan, stats);
}

int manta_node_spanning_batch(
    manta_node_t* node, const manta_asm_options_t* opt, const manta_align_scores_t* scores, int32_t jump_score, uint32_t n_loci,
    const uint8_t* bases, const uint64_t* read_off, const uint32_t* locus_read_begin, const uint8_t* refs1, const uint64_t* ref1_off,
    const uint8_t* refs2, const uint64_t* ref2_off, const manta_jump_cuts_t* cuts, const uint32_t* locus_min_word_length,
    const uint32_t* locus_max_word_length, manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs,
    manta_spanning_alignment_t* alignments, uint64_t contigs_cap, uint8_t* seq_arena, uint64_t seq_arena_cap, uint64_t* seq_arena_used,
    uint64_t* bits_arena, uint64_t bits_arena_cap, uint64_t* bits_arena_used, uint32_t* cigar_arena, uint64_t cigar_arena_cap,
    uint64_t* cigar_arena_used, const manta_batch_plan_t* plan, manta_batch_stats_t* stats, uint32_t* loci_per_device)
{
  if (!node) return MANTA_E_INVALID_ARG;
  return spanningBatchImpl(node->ctxs.data(), uint32_t(node->ctxs.size()), loci_per_device, opt, scores, jump_score, n_loci, bases, read_off,
                           locus_read_begin, nullptr, refs1, ref1_off, refs2, ref2_off, cuts, locus_min_word_length, locus_max_word_length, loci, contigs,
                           alignments, contigs_cap, seq_arena, seq_arena_cap, seq_arena_used, bits_arena, bits_arena_cap, bits_arena_used,
                           cigar_arena, cigar_arena_cap, cigar_arena_used, plan, stats);
}

}  // extern "C"
