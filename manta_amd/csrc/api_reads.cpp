// C-ABI implementation, part 3 (see api.cpp, api_internal.hpp): split-read scoring and read gathering.
#include "api_internal.hpp"


// ------------------------------------------------------------------------------------------------------
// split-read scoring
// ------------------------------------------------------------------------------------------------------
extern "C" int manta_split_read_batch(
    manta_ctx_t* ctx, const double* ln_comp_error_prob, const double* ln_error_prob, uint32_t n_qscores, float ln_one_third,
    float ln_random_base, uint32_t n_tasks, const manta_split_task_t* tasks, const uint8_t* arena, uint64_t arena_bytes,
    manta_split_result_t* results)
{
  if (!ctx) return MANTA_E_INVALID_ARG;
  if (!ln_comp_error_prob || !ln_error_prob || n_qscores == 0 || (n_tasks && (!tasks || !arena || !results)))
    return fail(ctx, MANTA_E_INVALID_ARG, "manta_split_read_batch: null argument");
  if (n_tasks == 0) return MANTA_OK;
  try {
    rt::setDevice(ctx->deviceId);
    rt::ScopedStream onStream(ctx->stream);
    std::vector<SplitTaskDev> dev(n_tasks);
    uint8_t*                  dArena = ctx->dSeq.as<uint8_t>(arena_bytes + 16);
    auto outside = [&](uint64_t off, uint64_t len) { return off > arena_bytes || len > arena_bytes - off; };
    for (uint32_t i = 0; i < n_tasks; ++i) {
      const manta_split_task_t& t(tasks[i]);
      if (outside(t.query_off, t.query_len) || outside(t.qual_off, t.query_len) || outside(t.target_off, t.target_len))
        return fail(ctx, MANTA_E_INVALID_ARG, "manta_split_read_batch: task " + std::to_string(i) + " outside the arena");
      SplitTaskDev& d(dev[i]);
      d.query            = dArena + t.query_off;
      d.qual             = dArena + t.qual_off;
      d.target           = dArena + t.target_off;
      d.query_len        = t.query_len;
      d.target_len       = t.target_len;
      d.bp_begin         = t.bp_begin;
      d.bp_end           = t.bp_end;
      d.flank_score_size = t.flank_score_size;
      d.reserved         = 0;
    }
    SplitTaskDev*   dTasks = ctx->dSplitTasks.as<SplitTaskDev>(n_tasks);
    SplitResultDev* dRes   = ctx->dSplitResults.as<SplitResultDev>(n_tasks);
    double*         dTab   = ctx->dSplitTables.as<double>(2ull * n_qscores + 2);
    uint32_t*       dCount = ctx->dCounter.as<uint32_t>(kNumESet);
    rt::h2d(dArena, arena, arena_bytes);
    rt::h2d(dTasks, dev.data(), sizeof(SplitTaskDev) * n_tasks);
    rt::h2d(dTab, ln_comp_error_prob, sizeof(double) * n_qscores);
    rt::h2d(dTab + n_qscores, ln_error_prob, sizeof(double) * n_qscores);
    rt::dzero(dCount, sizeof(uint32_t));
    SplitParams P;
    P.tasks          = dTasks;
    P.results        = dRes;
    P.n_tasks        = n_tasks;
    P.n_q            = n_qscores;
    P.ln_comp_error  = dTab;
    P.ln_error       = dTab + n_qscores;
    P.ln_one_third   = ln_one_third;
    P.ln_random_base = ln_random_base;
    P.counter        = dCount;
    const int grid = rt::roundGrid(int(std::min<uint64_t>(n_tasks, uint64_t(std::max(1, ctx->cuCount * 32)))));
    rt::launch(split_read_kernel, grid, 0, P);
    std::vector<SplitResultDev> h(n_tasks);
    rt::d2h(h.data(), dRes, sizeof(SplitResultDev) * n_tasks);
    int worst = MANTA_OK;
    for (uint32_t i = 0; i < n_tasks; ++i) {
      manta_split_result_t& r(results[i]);
      std::memset(&r, 0, sizeof(r));
      const SplitResultDev& d(h[i]);
      r.status = (d.status == 0) ? MANTA_OK : (d.status == 3) ? MANTA_E_UNSUPPORTED : MANTA_E_INVALID_ARG;
      if (d.status == 1) r.status = MANTA_E_SPLIT_QUERY_NOT_SHORTER;
      if (d.status == 2) r.status = MANTA_E_SPLIT_EMPTY_SCAN;
      if (r.status != MANTA_OK) {
        worst = r.status;
        continue;
      }
      r.best_pos         = d.best_pos;
      r.best_ln_lhood    = d.best_ln_lhood;
      r.left_size        = d.left_size;
      r.hom_size         = d.hom_size;
      r.right_size       = d.right_size;
      r.left_mismatches  = d.left_mismatches;
      r.hom_mismatches   = d.hom_mismatches;
      r.right_mismatches = d.right_mismatches;
    }
    if (worst != MANTA_OK) return fail(ctx, worst, "manta_split_read_batch: one or more tasks failed; see per-task status");
    return MANTA_OK;
  } catch (const std::exception& e) {
    return fail(ctx, MANTA_E_HIP, e.what());
  }
}

extern "C" void manta_read_search_range(int32_t bp_begin, int32_t bp_end, int32_t* search_begin, int32_t* search_end)
{
  manta_read_scan_t sc;
  std::memset(&sc, 0, sizeof(sc));
  sc.bp_begin = bp_begin;
  sc.bp_end   = bp_end;
  int sb, se;
  ReadClass::searchRange(sc, sb, se);
  if (search_begin) *search_begin = sb;
  if (search_end) *search_end = se;
}

extern "C" int manta_read_piles_batch(
    manta_ctx_t* ctx, const manta_read_class_options_t* opt, uint32_t n_loci, const manta_read_locus_t* loci, uint32_t n_scans,
    const manta_read_scan_t* scans, uint32_t n_reads, const manta_bam_read_t* reads, const uint32_t* cigars, uint64_t n_cigar_words,
    const uint8_t* names, uint64_t names_bytes, const uint8_t* seqs, uint64_t seqs_bytes, const uint8_t* quals, uint64_t quals_bytes,
    const uint8_t* refs, uint64_t refs_bytes, uint8_t* decision, uint32_t* pile_index, manta_read_locus_result_t* results,
    uint32_t* codes, uint64_t codes_cap, uint64_t* codes_used, uint32_t* nmask, uint64_t mask_cap, uint64_t* mask_used,
    uint32_t* read_len, uint64_t* read_code_off, uint64_t* read_mask_off, uint32_t* pile_read, uint64_t reads_cap,
    uint64_t* reads_used, uint32_t* locus_read_begin)
{
  static const char* fn = "manta_read_piles_batch: ";
  if (!ctx) return MANTA_E_INVALID_ARG;
  // (the two offset arrays take their "one past the last read" entry whenever there is a candidate, also with no read at all)
  if (!opt || (n_loci && (!loci || !results || !locus_read_begin || !read_code_off || !read_mask_off)) || (n_scans && !scans) ||
      (n_reads && (!reads || !decision || !pile_index)) || (n_cigar_words && !cigars) || (names_bytes && !names) || (seqs_bytes && !seqs) ||
      (quals_bytes && !quals))
    return fail(ctx, MANTA_E_INVALID_ARG, std::string(fn) + "null argument");
  if (codes_used) *codes_used = 0;
  if (mask_used) *mask_used = 0;
  if (reads_used) *reads_used = 0;
  if (n_loci == 0) return MANTA_OK;
  // what the kernels index with must lie inside what was handed over
  uint64_t maxRange = 1, maxRecords = 1, codeBound = 0, maskBound = 0;
  std::vector<uint32_t> chunks;  // read_test_kernel's work list: 64 records of one query each (scan, first record, candidate)
  chunks.reserve(3 * (size_t(n_reads) / 64 + n_scans + 1));
  for (uint32_t l = 0; l < n_loci; ++l) {
    if (loci[l].scan_begin > loci[l].scan_end || loci[l].scan_end > n_scans)
      return fail(ctx, MANTA_E_INVALID_ARG, std::string(fn) + "candidate " + std::to_string(l) + ": scans outside the scan array");
    uint64_t recs = 0;
    for (uint32_t s = loci[l].scan_begin; s < loci[l].scan_end; ++s) {
      const manta_read_scan_t& sc(scans[s]);
      if (sc.read_begin > sc.read_end || sc.read_end > n_reads || sc.ref_off > refs_bytes || sc.ref_len > refs_bytes - sc.ref_off ||
          (sc.ref_len && !refs) || sc.bam_index >= (1u << 22))
        return fail(ctx, MANTA_E_INVALID_ARG, std::string(fn) + "scan " + std::to_string(s) + " outside the arrays");
      int sb, se;
      ReadClass::searchRange(sc, sb, se);
      maxRange = std::max<uint64_t>(maxRange, uint64_t(se > sb ? se - sb : 0) + 2);
      recs += sc.read_end - sc.read_begin;
      for (uint32_t base = sc.read_begin; base < sc.read_end; base += 64) {
        chunks.push_back(s);
        chunks.push_back(base);
        chunks.push_back(l);
      }
    }
    maxRecords = std::max(maxRecords, recs);
  }
  if (maxRange > (1ull << 26)) return fail(ctx, MANTA_E_UNSUPPORTED, std::string(fn) + "a breakend interval beyond 64 M bases");
  for (uint32_t i = 0; i < n_reads; ++i) {
    const manta_bam_read_t& r(reads[i]);
    const bool bad = r.cigar_off > n_cigar_words || r.n_cigar > n_cigar_words - r.cigar_off ||
                     ((r.tags & MANTA_READ_TAG_MC) && (r.mate_cigar_off > n_cigar_words || r.n_mate_cigar > n_cigar_words - r.mate_cigar_off)) ||
                     r.qname_off > names_bytes || r.qname_len > names_bytes - r.qname_off || r.seq_off > seqs_bytes ||
                     (uint64_t(r.read_len) + 1) / 2 > seqs_bytes - r.seq_off || r.qual_off > quals_bytes || r.read_len > quals_bytes - r.qual_off;
    if (bad) return fail(ctx, MANTA_E_INVALID_ARG, std::string(fn) + "record " + std::to_string(i) + " outside the arenas");
    codeBound += (uint64_t(r.read_len) + 15) / 16;
    maskBound += (uint64_t(r.read_len) + 31) / 32;
  }
  try {
    rt::setDevice(ctx->deviceId);
    rt::ScopedStream onStream(ctx->stream);
    uint64_t tableCap = 64;
    while (tableCap < 2 * maxRecords) tableCap *= 2;
    const uint64_t stride = 2 * maxRange + 1 + tableCap;
    // one workspace per wave, sized for the widest breakend interval of the batch: a multi-megabase interval shrinks the grid instead
    // of asking for (waves x that interval) bytes (the reference allocates searchRange.size() counters for the one candidate)
    const uint64_t wsFit  = std::max<uint64_t>(1, workspaceBudget(size_t(16) << 30) / (stride * 4));
    const int      grid   = rt::roundGrid(int(std::min<uint64_t>(std::min<uint64_t>(n_loci, wsFit), uint64_t(std::max(1, ctx->cuCount * 8)))));
    ReadClassParams P;
    std::memset(&P, 0, sizeof(P));
    P.opt    = *opt;
    P.n_loci = n_loci;
    auto up  = [&](DevBuf& b, const void* src, uint64_t bytes) {
      uint8_t* d = b.as<uint8_t>(bytes + 16);
      rt::h2d(d, src, bytes);
      return d;
    };
    P.loci   = reinterpret_cast<const manta_read_locus_t*>(up(ctx->dRc[0], loci, sizeof(manta_read_locus_t) * uint64_t(n_loci)));
    P.scans  = reinterpret_cast<const manta_read_scan_t*>(up(ctx->dRc[1], scans, sizeof(manta_read_scan_t) * uint64_t(n_scans)));
    P.reads  = reinterpret_cast<const manta_bam_read_t*>(up(ctx->dRc[2], reads, sizeof(manta_bam_read_t) * uint64_t(n_reads)));
    P.cigars = reinterpret_cast<const uint32_t*>(up(ctx->dRc[3], cigars, 4 * n_cigar_words));
    P.names  = up(ctx->dRc[4], names, names_bytes);
    P.seqs   = up(ctx->dRc[5], seqs, seqs_bytes);
    P.quals  = up(ctx->dRc[6], quals, quals_bytes);
    P.refs   = up(ctx->dRc[7], refs, refs_bytes);
    // outputs and workspace in one allocation each
    uint8_t* dOut = ctx->dRc[8].as<uint8_t>(uint64_t(n_reads) * 13 + uint64_t(n_loci) * (sizeof(manta_read_locus_result_t) + 16 + 32) + 256 +
                                            4 * (uint64_t(n_loci) + 1));
    P.pile_index   = reinterpret_cast<uint32_t*>(dOut);
    P.tmp          = P.pile_index + n_reads;
    P.results      = reinterpret_cast<manta_read_locus_result_t*>(P.tmp + n_reads);
    P.locus_counts = reinterpret_cast<uint32_t*>(P.results + n_loci);
    P.locus_base   = reinterpret_cast<unsigned long long*>(P.locus_counts + 4 * uint64_t(n_loci));
    P.locus_read_begin = reinterpret_cast<uint32_t*>(P.locus_base + 4 * (uint64_t(n_loci) + 1));
    P.counter      = P.locus_read_begin + n_loci + 1;  // four queue heads
    P.decision     = reinterpret_cast<uint8_t*>(P.counter + 4);
    P.pre          = P.decision + n_reads;
    P.chunks       = reinterpret_cast<const uint32_t*>(up(ctx->dRc[14], chunks.data(), 4 * chunks.size()));
    P.n_chunks     = uint32_t(chunks.size() / 3);
    P.ws           = ctx->dRc[9].as<uint32_t>(stride * uint64_t(grid));
    P.ws_stride    = stride;
    P.range_cap    = uint32_t(maxRange);
    P.table_cap    = uint32_t(tableCap);
    P.codes        = ctx->dRc[10].as<uint32_t>(codeBound + 4);
    P.nmask        = ctx->dRc[11].as<uint32_t>(maskBound + 4);
    P.read_len     = ctx->dRc[12].as<uint32_t>(2 * uint64_t(n_reads) + 4);
    P.pile_read    = P.read_len + n_reads + 1;
    P.read_code_off = ctx->dRc[13].as<unsigned long long>(2 * (uint64_t(n_reads) + 2));
    P.read_mask_off = P.read_code_off + n_reads + 1;
    rt::dzero(P.counter, 4 * sizeof(uint32_t));
    const int wide    = rt::roundGrid(std::max(1, ctx->cuCount * 16));
    auto      gridFor = [&](uint64_t items) { return std::min(wide, rt::roundGrid(int(std::max<uint64_t>(1, std::min<uint64_t>(items, 1u << 20))))); };
    rt::launch(read_test_kernel, gridFor(P.n_chunks), 0, P);
    rt::launch(read_class_kernel, grid, 0, P);
    rt::launch(read_pile_offsets_kernel, rt::roundGrid(1), 0, P);
    rt::launch(read_pile_pack_kernel, grid, 0, P);
    rt::launch(read_pile_bases_kernel, gridFor(uint64_t(n_reads) / 8 + 1), 0, P);
    unsigned long long totals[4] = {0, 0, 0, 0};
    rt::d2h(totals, P.locus_base + 4 * uint64_t(n_loci), sizeof(totals));  // (synchronizes)
    if (reads_used) *reads_used = totals[0];
    if (codes_used) *codes_used = totals[1];
    if (mask_used) *mask_used = totals[2];
    rt::d2hAsync(decision, P.decision, n_reads);
    rt::d2hAsync(pile_index, P.pile_index, 4 * uint64_t(n_reads));
    rt::d2hAsync(results, P.results, sizeof(manta_read_locus_result_t) * uint64_t(n_loci));
    rt::d2hAsync(locus_read_begin, P.locus_read_begin, 4 * (uint64_t(n_loci) + 1));
    const bool fits = totals[0] <= reads_cap && totals[1] <= codes_cap && totals[2] <= mask_cap &&
                      (totals[0] == 0 || (codes && nmask && read_len && pile_read));
    if (fits) {
      rt::d2hAsync(codes, P.codes, 4 * totals[1]);
      rt::d2hAsync(nmask, P.nmask, 4 * totals[2]);
      rt::d2hAsync(read_len, P.read_len, 4 * totals[0]);
      rt::d2hAsync(pile_read, P.pile_read, 4 * totals[0]);
      rt::d2hAsync(read_code_off, P.read_code_off, 8 * (totals[0] + 1));
      rt::d2hAsync(read_mask_off, P.read_mask_off, 8 * (totals[0] + 1));
    }
    rt::sync();
    if (!fits) return fail(ctx, MANTA_E_CAPACITY, std::string(fn) + "pile arrays too small (see *_used)");
    int worst = MANTA_OK;
    for (uint32_t l = 0; l < n_loci; ++l)
      if (results[l].status != MANTA_OK) worst = results[l].status;
    if (worst != MANTA_OK) return fail(ctx, worst, std::string(fn) + "one or more candidates failed; see per-candidate status");
    return MANTA_OK;
  } catch (const std::exception& e) {
    return fail(ctx, MANTA_E_HIP, e.what());
  }
}

#ifdef MANTA_WAVE_EMU
/// tests/emu only: speculation statistics of contig_kernel since the last call (loci done by it, walk rounds, walks,
/// accepted candidates, cache evictions)
extern "C" void manta_emu_fast_stats(unsigned long long* out)
{
  unsigned long long* v = manta_dev::fastStats();
  for (int i = 0; i < 8; ++i) {
    out[i] = v[i];
    v[i]   = 0;
  }
}
#endif

