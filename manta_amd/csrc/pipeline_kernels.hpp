// Glue kernel of the fused "small SV" locus pipeline (SVCandidateAssemblyRefiner::getSmallSVAssembly,
// applications/GenerateSVCandidates/SVCandidateAssemblyRefiner.cpp:1860-2038):
//   assemble_kernel  ->  [this file] per contig: 10-mer reference trim (:1984-2011), alignment task set-up and
//   bucketing by columns-per-lane  ->  align_kernel<LARGE_INDEL, E>  (:2032-2038).
// Contigs never leave the device between the stages.
#pragma once
#include "align_kernels.hpp"
#include "assemble_kernels.hpp"

namespace manta_dev {

static const int      SMALLSV_MER       = 10;    // SVCandidateAssemblyRefiner.cpp:1986
static const unsigned SCHED_TABLE_BYTES = 4096;  // per wavefront: 10-mer table of contigs up to 512 bp lives in LDS
static const unsigned SCHED_SEQ_BYTES   = 6144;  // ... and so do the contig and the reference window (staged with wide loads: the scans
                                                 // below read a byte per lane and 55 positions per round trip otherwise)
static const unsigned SCHED_LDS_BYTES   = SCHED_TABLE_BYTES + SCHED_SEQ_BYTES + 1024;  // (+ the pending bucket entries) 11 KB per wavefront: 12-14 waves per CU

struct SmallSvCuts {
  int32_t leadingCut, trailingCut, maxLeadingCut, maxTrailingCut;  // :1912-1915
};

struct SmallSvTaskInfo {
  int32_t status;  // 0 ok; else MANTA-level failure of the trim (contig shorter than a 10-mer, empty window)
  int32_t adj_leading_cut, adj_trailing_cut;
  int32_t bucket;  // index into the E set, -1 if not scheduled
};

struct ScheduleParams {
  // assembler outputs (device)
  const AsmLocusOut*  loci;
  const AsmContigOut* contigs;
  const uint8_t*      seq_arena;
  uint32_t            n_loci;
  uint32_t            max_assembly_count;
  // references
  const uint8_t*     refs;
  const uint64_t*    ref_off;  // n_loci + 1
  const SmallSvCuts* cuts;     // n_loci
  // outputs
  AlignTaskDev*    tasks;  // n_loci * max_assembly_count
  SmallSvTaskInfo* info;   // same
  uint32_t*        bucket_ids;     // n_buckets * (n_loci * max_assembly_count)
  uint32_t*        bucket_count;   // n_buckets
  uint32_t*        bucket_maxref;  // n_buckets
  unsigned long long* cigar_used;  // bump allocator (u32 units)
  uint64_t         cigar_cap;
  uint32_t*        counter;
  // per-workgroup 10-mer table
  uint32_t* table_ws;
  uint32_t  table_cap;  // power of two, >= 2 * longest contig
  // E set
  uint32_t e_set[16];
  uint32_t n_e;
  uint32_t chunk;  // loci per work-queue pop (smallsv_schedule_kernel)
};

WV_DEV unsigned merCode(const uint8_t* p, bool& valid)
{
  unsigned code = 0;
  valid         = true;
  for (int i = 0; i < SMALLSV_MER; ++i) {
    const unsigned c = baseCode(p[i]);
    if (c > 3) valid = false;
    code = (code << 2) | (c & 3u);
  }
  return code;
}

/// 10-mer codes of 55 consecutive positions with ONE byte load per lane: lane l loads base p0+l, then the codes of
/// bases l..l+9 are assembled by doubling with four lane shuffles (2, 4, 8, 10 bases).  Lane l (<= 54) returns the
/// code of the 10-mer starting at p0+l; `valid` is false if the 10-mer leaves [0,len) or holds a non-ACGT base.
/// All 64 lanes must call.
static const int MER_BATCH = 55;
/// the lane's base code (0..3, 4 = not ACGT / outside the sequence) for batch start p0: the only memory access of a batch
WV_DEV unsigned merLoadBase(const uint8_t* seq, const int len, const int p0)
{
  const int pos = p0 + wv::lane();
  return (pos >= 0 && pos < len) ? baseCode(seq[pos]) : 4u;
}
/// 10-mer codes of the batch from the per-lane base codes
WV_DEV unsigned merCodesFromBase(const unsigned c, bool& valid)
{
  const int      l   = wv::lane();
  unsigned bad = (c > 3) ? 1u : 0u;
  unsigned w1  = c & 3u;
  const unsigned w2 = (w1 << 2) | wv::shfl(w1, (l + 1) & 63);
  const unsigned b2 = bad | wv::shfl(bad, (l + 1) & 63);
  const unsigned w4 = (w2 << 4) | wv::shfl(w2, (l + 2) & 63);
  const unsigned b4 = b2 | wv::shfl(b2, (l + 2) & 63);
  const unsigned w8 = (w4 << 8) | wv::shfl(w4, (l + 4) & 63);
  const unsigned b8 = b4 | wv::shfl(b4, (l + 4) & 63);
  const unsigned w10 = (w8 << 4) | wv::shfl(w2, (l + 8) & 63);
  const unsigned b10 = b8 | wv::shfl(b2, (l + 8) & 63);
  valid = (l < MER_BATCH) && (b10 == 0);
  return w10;
}
WV_DEV unsigned merCodes55(const uint8_t* seq, const int len, const int p0, bool& valid)
{
  return merCodesFromBase(merLoadBase(seq, len, p0), valid);
}

WV_DEV bool merLookup(const uint32_t* table, const unsigned mask, const unsigned code)
{
  unsigned s = (code * 2654435761u) & mask;
  while (true) {
    const uint32_t v = table[s];
    if (v == 0) return false;
    if (v == code + 1) return true;
    s = (s + 1) & mask;
  }
}

/// One contig slot: 10-mer reference trim + alignment task set-up.  Returns the slot's SmallSvTaskInfo; every exit is
/// wave-uniform and the caller stores the record after a single reconvergence point.  (An earlier form that stored
/// from lane 0 and `continue`d straight to the work-queue pop was compiled into a loop whose lanes left at different
/// times -- 63 lanes then re-ran the pop's readfirstlane without lane 0 and spun forever on loci without contigs.)
WV_DEV SmallSvTaskInfo scheduleSlot(const ScheduleParams& P, const unsigned slot, const unsigned total, uint32_t* ltable, uint32_t* gtable)
{
  const unsigned    lane  = unsigned(wv::lane());
  const unsigned    locus = slot / P.max_assembly_count, ci = slot % P.max_assembly_count;
  const AsmLocusOut lo    = P.loci[locus];
  SmallSvTaskInfo   info  = {0, 0, 0, -1};
  if (lo.status != ASM_OK || ci >= lo.n_contigs) {
    info.status = (lo.status != ASM_OK) ? 1 : 0;
    return info;
  }
  const AsmContigOut co      = P.contigs[slot];
  const uint8_t*     contigG = P.seq_arena + co.seq_off;
  const unsigned     clen    = co.seq_len;
  const uint8_t*     refG    = P.refs + P.ref_off[locus];
  const int          refSize = int(P.ref_off[locus + 1] - P.ref_off[locus]);
  const SmallSvCuts  cuts    = P.cuts[locus];
  // contig and reference window into LDS (16 bytes per lane and load, all in flight together) when they fit; the scans then
  // run at LDS latency.  The task keeps the global pointers.
  const uint8_t* contig = contigG;
  const uint8_t* ref    = refG;
  {
    uint8_t* const  lseq = reinterpret_cast<uint8_t*>(ltable) + SCHED_TABLE_BYTES;
    const uintptr_t gc = reinterpret_cast<uintptr_t>(contigG), gr = reinterpret_cast<uintptr_t>(refG);
    const unsigned  leadC = unsigned(gc & 15), leadR = unsigned(gr & 15);
    const unsigned  bytesC = (leadC + clen + 15) & ~15u, bytesR = (leadR + unsigned(refSize > 0 ? refSize : 0) + 15) & ~15u;
    if (bytesC + bytesR <= SCHED_SEQ_BYTES) {
      const u32x4* sc = reinterpret_cast<const u32x4*>(gc - leadC);
      const u32x4* sr = reinterpret_cast<const u32x4*>(gr - leadR);
      u32x4*       dc = reinterpret_cast<u32x4*>(lseq);
      u32x4*       dr = reinterpret_cast<u32x4*>(lseq + bytesC);
      for (unsigned i = lane; i < bytesC / 16; i += 64) dc[i] = sc[i];
      for (unsigned i = lane; i < bytesR / 16; i += 64) dr[i] = sr[i];
      wv::sync();
      contig = lseq + leadC;
      ref    = lseq + bytesC + leadR;
    }
  }

  if (clen < unsigned(SMALLSV_MER) || 2 * clen > P.table_cap) {
    info.status = 2;
    return info;
  }
  // hash set of the contig's 10-mers (:1987-1991)
  unsigned tcap = 64;
  while (tcap < 2 * clen) tcap <<= 1;
  const unsigned mask  = tcap - 1;
  uint32_t*      table = (tcap * 4 <= SCHED_TABLE_BYTES) ? ltable : gtable;
  for (unsigned i = lane; i < tcap; i += 64) table[i] = 0;
  wv::sync();
  for (int p0 = 0; p0 + SMALLSV_MER <= int(clen); p0 += MER_BATCH) {
    bool           valid;
    const unsigned code = merCodes55(contig, int(clen), p0, valid);
    if (!valid) continue;
    unsigned s = (code * 2654435761u) & mask;
    while (true) {
      const uint32_t old = wv::atomic_cas(&table[s], 0u, code + 1);
      if (old == 0 || old == code + 1) break;
      s = (s + 1) & mask;
    }
  }
  wv::sync();
  if (table == gtable) wv::fence_acquire();

  const int minRefIndex    = cuts.leadingCut;
  const int maxRefIndex    = refSize - (cuts.trailingCut + SMALLSV_MER);
  const int maxFwdRefIndex = (cuts.maxLeadingCut < maxRefIndex) ? cuts.maxLeadingCut : maxRefIndex;
  // first hit scanning forward (:1997-2001) and last hit scanning backward (:2004-2008, windows of 55 start positions, highest
  // window first), one window of each per round: the two scans are independent, so their lane shuffles and table probes overlap
  int adjLead = maxFwdRefIndex + 1;
  if (adjLead < minRefIndex) adjLead = minRefIndex;  // empty scan range: the loop variable keeps its initial value
  const int minRevRefIndex = (minRefIndex > refSize - cuts.maxTrailingCut) ? minRefIndex : (refSize - cuts.maxTrailingCut);
  int       revIndex       = minRevRefIndex - 1;
  if (revIndex > maxRefIndex) revIndex = maxRefIndex;  // empty scan range
  {
    int  fBase = minRefIndex, bTop = maxRefIndex;
    bool fDone = fBase > maxFwdRefIndex, bDone = bTop < minRevRefIndex;
    while (!fDone || !bDone) {
      const int      bBase = bTop - (MER_BATCH - 1);
      const unsigned cf = merLoadBase(ref, refSize, fDone ? -1000 : fBase), cb = merLoadBase(ref, refSize, bDone ? -1000 : bBase);
      bool           vf, vb;
      const unsigned codeF = merCodesFromBase(cf, vf), codeB = merCodesFromBase(cb, vb);
      const int      iF = fBase + int(lane), iB = bBase + int(lane);
      const bool     hitF = !fDone && vf && iF <= maxFwdRefIndex && merLookup(table, mask, codeF);
      const bool     hitB = !bDone && vb && iB >= minRevRefIndex && iB <= bTop && merLookup(table, mask, codeB);
      const uint64_t mF = wv::ballot(hitF), mB = wv::ballot(hitB);
      if (!fDone) {
        if (mF) {
          adjLead = fBase + wv::ctz(mF);
          fDone   = true;
        } else {
          fBase += MER_BATCH;
          fDone = fBase > maxFwdRefIndex;
        }
      }
      if (!bDone) {
        if (mB) {
          revIndex = bBase + (63 - wv::clz(mB));
          bDone    = true;
        } else {
          bTop -= MER_BATCH;
          bDone = bTop < minRevRefIndex;
        }
      }
    }
  }
  const int adjTrail = refSize - (revIndex + SMALLSV_MER);
  const int winLen   = refSize - adjLead - adjTrail;
  info.adj_leading_cut  = adjLead;
  info.adj_trailing_cut = adjTrail;
  if (winLen <= 0 || adjLead < 0 || adjTrail < 0) {
    info.status = 3;
    return info;
  }
  int            bucket = -1;
  const unsigned need   = (clen + 63) / 64;
  for (unsigned b = 0; b < P.n_e; ++b)
    if (P.e_set[b] >= need) {
      bucket = int(b);
      break;
    }
  if (bucket < 0) bucket = int(P.n_e) - 1;  // longer than 64 x 32 columns: the widest kernel runs it in strips (align_kernels.hpp)
  // The task's CIGAR region needs no allocator: contig texts do not overlap in the text arena, so neither do regions of 8 words
  // per contig base starting at 8 x the text offset (a task takes 4 * length + 16 words, contigs here hold >= 10 bases).  (One
  // bump-allocator atomic per contig on ONE address, next to one per locus for the work queue and one per contig for the bucket
  // lists, was what this kernel's 0.9 ms consisted of: ~40 k same-address L2 atomics.)
  const unsigned long long cig = 8ull * co.seq_off;
  if (cig + 4ull * clen + 16 > P.cigar_cap) {
    info.status = 5;
    return info;
  }
  if (lane == 0) {
    AlignTaskDev t;
    t.query     = contigG;
    t.ref1      = refG + adjLead;
    t.ref2      = nullptr;
    t.query_len = clen;
    t.ref1_len  = unsigned(winLen);
    t.ref2_len  = 0;
    t.cigar_off = uint32_t(cig);
    P.tasks[slot] = t;
  }
  info.bucket = bucket;
  return info;
}

/// per-wave pending bucket entries (LDS): a bucket's list takes them eight at a time, with one atomic
static const unsigned SCHED_PEND = 8;
struct SchedPending {
  uint32_t count[16];
  uint32_t maxref[16];
  uint32_t slot[16][SCHED_PEND];
  uint32_t cigarWords;  ///< upper bound of the CIGAR words of this wave's tasks (manta_smallsv_output_sizes reports the kernel's total)
};
WV_DEV void schedFlush(const ScheduleParams& P, SchedPending* pend, const unsigned bucket, const unsigned total)
{
  const unsigned n = pend->count[bucket];
  if (n == 0) return;
  unsigned pos = 0;
  if (wv::lane() == 0) {
    pos = wv::atomic_add(&P.bucket_count[bucket], n);
    unsigned cur = wv::atomic_load(&P.bucket_maxref[bucket]);
    const unsigned want = pend->maxref[bucket];
    while (cur < want) {
      const unsigned old = wv::atomic_cas(&P.bucket_maxref[bucket], cur, want);
      if (old == cur) break;
      cur = old;
    }
  }
  pos = wv::first(pos);
  if (unsigned(wv::lane()) < n) P.bucket_ids[size_t(bucket) * total + pos + unsigned(wv::lane())] = pend->slot[bucket][wv::lane()];
  wv::sync();
  if (wv::lane() == 0) {
    pend->count[bucket]  = 0;
    pend->maxref[bucket] = 0;
  }
  wv::sync();
}

#if !MANTA_TU_DEFINES(MANTA_TU_GLUE)
WV_KERNEL void smallsv_schedule_kernel(const ScheduleParams P);
#else
WV_KERNEL void smallsv_schedule_kernel(const ScheduleParams P)
{
  const unsigned lane   = unsigned(wv::lane());
  uint32_t*      gtable = P.table_ws + size_t(wv::block()) * P.table_cap;
  uint32_t*      ltable = reinterpret_cast<uint32_t*>(wv::lds(SCHED_LDS_BYTES));
  SchedPending*  pend   = reinterpret_cast<SchedPending*>(reinterpret_cast<char*>(ltable) + SCHED_TABLE_BYTES + SCHED_SEQ_BYTES);
  const unsigned total  = P.n_loci * P.max_assembly_count;
  if (lane < 16) {
    pend->count[lane]  = 0;
    pend->maxref[lane] = 0;
  }
  if (lane == 0) pend->cigarWords = 0;
  wv::sync();
  // the work unit is a LOCUS (most contig slots are empty; one queue pop per slot made the queue head the bottleneck), popped
  // P.chunk at a time
  while (true) {
    unsigned base = 0;
    if (lane == 0) base = wv::atomic_add(P.counter, P.chunk);
    base = wv::first(base);
    if (base >= P.n_loci) break;
    const unsigned end = (base + P.chunk < P.n_loci) ? base + P.chunk : P.n_loci;
    for (unsigned locus = base; locus < end; ++locus) {
      const AsmLocusOut lo       = P.loci[locus];
      const unsigned    nContigs = (lo.status == ASM_OK) ? lo.n_contigs : 0u;
      // slots without a contig: one lane each
      if (lane >= nContigs && lane < P.max_assembly_count) {
        SmallSvTaskInfo none = {(lo.status != ASM_OK) ? 1 : 0, 0, 0, -1};
        P.info[locus * P.max_assembly_count + lane] = none;
      }
      for (unsigned ci = 0; ci < nContigs; ++ci) {
        const unsigned        slot = locus * P.max_assembly_count + ci;
        const SmallSvTaskInfo info = scheduleSlot(P, slot, total, ltable, gtable);
        wv::sync();  // single reconvergence point of every exit of scheduleSlot
        if (lane == 0) P.info[slot] = info;
        if (info.bucket >= 0) {
          const unsigned b = unsigned(info.bucket);
          if (lane == 0) {
            const AlignTaskDev& t = P.tasks[slot];
            const unsigned slabLen = unsigned(alignSlabRefLen(1, int(P.e_set[b]), t.query_len, t.ref1_len));
            pend->slot[b][pend->count[b]] = slot;
            pend->count[b] += 1;
            pend->cigarWords += 4 * t.query_len + 16;
            if (slabLen > pend->maxref[b]) pend->maxref[b] = slabLen;
          }
          wv::sync();
          if (pend->count[b] == SCHED_PEND) schedFlush(P, pend, b, total);
        }
        wv::sync();
      }
    }
  }
  for (unsigned b = 0; b < P.n_e; ++b) schedFlush(P, pend, b, total);
  if (lane == 0 && pend->cigarWords) wv::atomic_add(P.cigar_used, (unsigned long long)pend->cigarWords);
}
#endif

// ------------------------------------------------------------------------------------------------------------------
// bucket_sort_kernel: the tasks of an E bucket by descending reference length (counting sort over length / 8, one workgroup per
// bucket).  align_pair_kernel runs the tasks of a bucket two at a time and both sweep the LONGER reference's rows: the 10-mer
// trim leaves most windows at contig length + indel, but a spurious early 10-mer hit (one window in seven) leaves one three
// times that -- paired at random, such a window would cost its partner as well (measured: 1.6 x the VALU instructions of the
// bucket).  Longest first also keeps the launch's tail short.
// ------------------------------------------------------------------------------------------------------------------
static const unsigned BS_CLASSES = 512;
static const unsigned BS_WAVES   = 16;  // (four waves took 0.24 ms over the metric's 10 000 loci: a latency chain per 64 loci, so more waves, shorter shares)
static const unsigned BS_LDS_BYTES = 4 * (BS_WAVES * BS_CLASSES + BS_CLASSES + 8);
struct BucketSortParams {
  const AlignTaskDev*    tasks;
  const AsmLocusOut*     loci;          ///< contigs per locus: the slots worth looking at
  const SmallSvTaskInfo* info;          ///< per slot: which bucket the schedule kernel filed it in (-1: none)
  uint32_t*              ids_out;       ///< n_buckets * total: the sorted lists
  uint32_t               n_loci, max_assembly_count;
  uint32_t               total;
  uint32_t               mask;          ///< buckets to sort (one workgroup per bucket; the others return at once)
};
// The list is rebuilt from the per-slot records in a FIXED order -- the loci in order, 64 at a time, their first contigs, then their second
// ones, ... -- and the counting sort is stable, so the sorted order, and with it which two tasks share a wave in align_pair_kernel,
// depends on the batch alone.  (The schedule kernel's own bucket lists are in the order its waves' atomic appends happened to land: up
// to round 5 the pairing, and so the aligner's time to the last few percent, differed from run to run.)  Every wave takes a contiguous
// share of the loci and keeps its own class histogram; class bases are the exclusive scan over (class, wave); inside a step the lanes of
// a class rank themselves by ballots (asm_lds_big.hpp: radixPassIds).  Only the slots that hold a contig are read: n_loci records and
// ~1.5 slots per locus, not n_loci x maxAssemblyCount.
#if !MANTA_TU_DEFINES(MANTA_TU_GLUE)
WV_KERNEL_WG(16) void bucket_sort_kernel(const BucketSortParams P);
#else
WV_KERNEL_WG(16) void bucket_sort_kernel(const BucketSortParams P)
{
  const unsigned b = unsigned(wv::block_single());
  if (!((P.mask >> b) & 1u)) return;
  uint32_t*      hist  = reinterpret_cast<uint32_t*>(wv::lds_single());  // [wave][class]
  uint32_t*      dbase = hist + BS_WAVES * BS_CLASSES;                   // [class]
  const unsigned tw = unsigned(wv::wave_in_wg()), tn = unsigned(wv::wg_waves()), lane = unsigned(wv::lane());
  const unsigned tid = 64 * tw + lane, nt = 64 * tn;
  uint32_t*      out    = P.ids_out + size_t(b) * P.total;
  uint32_t*      myHist = hist + BS_CLASSES * tw;
  const unsigned chunk  = (((P.n_loci + tn - 1) / tn) + 63) & ~63u;
  const unsigned c0 = chunk * tw, c1 = (c0 + chunk < P.n_loci) ? (c0 + chunk) : P.n_loci;
  auto cls = [&](const unsigned id) {
    const unsigned c = P.tasks[id].ref1_len >> 3;
    return (BS_CLASSES - 1) - ((c < BS_CLASSES - 1) ? c : (BS_CLASSES - 1));
  };
  auto contigsOf = [&](const unsigned locus) -> unsigned {
    if (locus >= c1) return 0u;
    const AsmLocusOut lo = P.loci[locus];
    return (lo.status == ASM_OK) ? lo.n_contigs : 0u;
  };
  for (unsigned i = tid; i < BS_WAVES * BS_CLASSES + BS_CLASSES; i += nt) hist[i] = 0;
  wv::sync();
  wv::wg_barrier();
  for (unsigned i0 = c0; i0 < c1; i0 += 64) {
    const unsigned locus = i0 + lane, n = contigsOf(locus);
    for (unsigned ci = 0; ci < n; ++ci) {
      const unsigned slot = locus * P.max_assembly_count + ci;
      if (P.info[slot].bucket == int(b)) wv::atomic_add(&myHist[cls(slot)], 1u);
    }
  }
  wv::sync();
  wv::wg_barrier();
  for (unsigned d = tid; d < BS_CLASSES; d += nt) {  // per class: the waves' counts -> their offsets inside the class, and the class total
    unsigned run = 0;
    for (unsigned w = 0; w < BS_WAVES; ++w) {
      const unsigned c          = (w < tn) ? hist[BS_CLASSES * w + d] : 0u;
      hist[BS_CLASSES * w + d] = run;
      run += c;
    }
    dbase[d] = run;
  }
  wv::sync();
  wv::wg_barrier();
  if (tw == 0) {  // exclusive prefix over the classes: 8 per lane
    unsigned v[8], sum = 0;
    for (int j = 0; j < 8; ++j) {
      v[j] = dbase[8 * lane + j];
      sum += v[j];
    }
    unsigned inc = sum;
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned o = wv::shfl(inc, int(lane) - off);
      if (int(lane) >= off) inc += o;
    }
    unsigned run = inc - sum;
    for (int j = 0; j < 8; ++j) {
      dbase[8 * lane + j] = run;
      run += v[j];
    }
  }
  wv::sync();
  wv::wg_barrier();
  for (unsigned i0 = c0; i0 < c1; i0 += 64) {
    const unsigned locus = i0 + lane, n = contigsOf(locus);
    unsigned       nMax  = n;  // the most contigs of a locus of this step (wave-uniform loop bound)
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned o = wv::shfl(nMax, int(lane) ^ off);
      nMax             = (o > nMax) ? o : nMax;
    }
    for (unsigned ci = 0; ci < nMax; ++ci) {
      const unsigned slot  = locus * P.max_assembly_count + ci;
      const bool     valid = ci < n && P.info[slot].bucket == int(b);
      const unsigned d     = valid ? cls(slot) : 0u;
      uint64_t       peers = wv::ballot(valid);
      for (int bit = 0; bit < 9; ++bit) {
        const bool     on = (d >> bit) & 1u;
        const uint64_t m  = wv::ballot(valid && on);
        peers &= on ? m : ~m;
      }
      unsigned base = 0;
      if (valid) base = dbase[d] + myHist[d];
      wv::sync();
      if (valid) {
        out[base + unsigned(wv::popc(peers & ((uint64_t(1) << lane) - 1)))] = slot;
        if ((peers >> lane) == 1u) myHist[d] += unsigned(wv::popc(peers));  // (the class' highest lane of this step)
      }
      wv::sync();
    }
  }
}
#endif

// ------------------------------------------------------------------------------------------------------------------
// Fused "spanning" locus pipeline (SVCandidateAssemblyRefiner::getJumpAssembly -> alignJumpContigs, DNA branch,
// applications/GenerateSVCandidates/SVCandidateAssemblyRefiner.cpp:1525-1743):
//   assemble_kernel -> spanning_schedule_kernel (round 1: every contig against the CUT references, :1663-1670)
//   -> align_kernel<JUMP,E> -> spanning_realign_kernel (the re-align rule, :1672-1713) -> align_kernel<JUMP,E> (round 2).
// Both glue kernels are one LANE per item: a few loads, compares and atomics each.
// ------------------------------------------------------------------------------------------------------------------
struct JumpCuts {
  int32_t a1Lead, a1Trail, a2Lead, a2Trail;  // AlignData after the orientation swaps (:1533-1550)
};

struct SpanTaskInfo {
  int32_t status;   // 0 ok; 4 contig longer than the widest aligner bucket; 5 CIGAR workspace exhausted; 6 empty cut window
  int32_t bucket;   // round-1 E bucket (-1: no contig in this slot)
  int32_t bucket2;  // round-2 E bucket (-1: not re-aligned)
  int32_t is_uncut; // final alignment is the round-2 one (references without cuts, begin positions without offset)
};

struct SpanParams {
  const AsmLocusOut*  loci;
  const AsmContigOut* contigs;
  const uint8_t*      seq_arena;
  uint32_t            n_loci, max_assembly_count;
  const uint8_t*      refs1;
  const uint64_t*     ref1_off;  // n_loci + 1
  const uint8_t*      refs2;
  const uint64_t*     ref2_off;
  const JumpCuts*     cuts;
  AlignTaskDev*       tasks;   // round 1, n_loci * max_assembly_count
  AlignTaskDev*       tasks2;  // round 2
  SpanTaskInfo*       info;
  uint32_t*           bucket_ids;     // n_buckets * n_slots
  uint32_t*           bucket_count;   // n_buckets
  uint32_t*           bucket_maxref;  // n_buckets
  uint32_t*           bucket_ids2;
  uint32_t*           bucket_count2;
  uint32_t*           bucket_maxref2;
  unsigned long long* cigar_used;  // bump allocator (u32 units), shared by both rounds
  uint64_t            cigar_cap;
  const AlignResultDev* results;   // round-1 results (read by the re-align kernel)
  const uint32_t*       cigar;
  uint32_t            e_set[16];
  uint32_t            n_e;
  // two passes over one block (api.cpp: spanningRunImpl): a pass takes the loci with pass_mask[locus] == pass_value (nullptr: every
  // locus) -- the loci the assembler finished at its first word length are aligned while the later word lengths of the others still run
  const uint8_t*      pass_mask;
  uint32_t            pass_value;
};

/// written between the first word length's launches and the later ones, on the assembler's stream: which loci are final already
struct SpanMarkParams {
  const AsmLocusOut* loci;
  uint32_t           n_loci;
  uint8_t*           mask;     // out: 1 = the locus' record is final (status ASM_OK)
  uint32_t*          counter;  // out: how many
};
#if !MANTA_TU_DEFINES(MANTA_TU_GLUE)
WV_KERNEL void span_mark_kernel(const SpanMarkParams P);
#else
WV_KERNEL void span_mark_kernel(const SpanMarkParams P)
{
  for (unsigned base = unsigned(wv::block()) * 64u; base < P.n_loci; base += unsigned(wv::nblocks()) * 64u) {
    const unsigned locus = base + unsigned(wv::lane());
    const bool     done  = locus < P.n_loci && P.loci[locus].status == ASM_OK;
    if (locus < P.n_loci) P.mask[locus] = done ? 1 : 0;
    const unsigned n = unsigned(wv::popc(wv::ballot(done)));
    if (wv::lane() == 0 && n) wv::atomic_add(P.counter, n);
  }
}
#endif

WV_DEV void spanAtomicMax(uint32_t* p, const unsigned v)
{
  unsigned cur = wv::atomic_load(p);
  while (cur < v) {
    const unsigned old = wv::atomic_cas(p, cur, v);
    if (old == cur) break;
    cur = old;
  }
}

/// files one jump-alignment task; returns the E bucket or a negative status
WV_DEV int spanFileTask(
    const SpanParams& P, const unsigned slot, const unsigned total, const AsmContigOut& co, const uint8_t* r1, const int r1Len,
    const uint8_t* r2, const int r2Len, AlignTaskDev* tasks, uint32_t* bucketIds, uint32_t* bucketCount, uint32_t* bucketMaxref)
{
  if (r1Len <= 0 || r2Len <= 0) return -6;
  const unsigned need = (co.seq_len + 63) / 64;
  int            bucket = -1;
  for (unsigned b = 0; b < P.n_e; ++b)
    if (P.e_set[b] >= need) {
      bucket = int(b);
      break;
    }
  if (bucket < 0) bucket = int(P.n_e) - 1;  // strips on the widest kernel
  const unsigned long long words = 4ull * co.seq_len + 16;
  const unsigned long long cig   = wv::atomic_add(P.cigar_used, words);
  if (cig + words > P.cigar_cap) return -5;
  AlignTaskDev t;
  t.query     = P.seq_arena + co.seq_off;
  t.ref1      = r1;
  t.ref2      = r2;
  t.query_len = co.seq_len;
  t.ref1_len  = unsigned(r1Len);
  t.ref2_len  = unsigned(r2Len);
  t.cigar_off = uint32_t(cig);
  tasks[slot] = t;
  const unsigned pos = wv::atomic_add(&bucketCount[bucket], 1u);
  bucketIds[size_t(bucket) * total + pos] = slot;
  spanAtomicMax(&bucketMaxref[bucket], unsigned(alignSlabRefLen(2, int(P.e_set[bucket]), co.seq_len, unsigned(r1Len + r2Len))));
  return bucket;
}

#if !MANTA_TU_DEFINES(MANTA_TU_GLUE)
WV_KERNEL void spanning_schedule_kernel(const SpanParams P);
#else
WV_KERNEL void spanning_schedule_kernel(const SpanParams P)
{
  const unsigned total = P.n_loci * P.max_assembly_count;
  for (unsigned slot = unsigned(wv::block()) * 64u + unsigned(wv::lane()); slot < total; slot += unsigned(wv::nblocks()) * 64u) {
    const unsigned    locus = slot / P.max_assembly_count, ci = slot % P.max_assembly_count;
    if (P.pass_mask && P.pass_mask[locus] != P.pass_value) continue;
    const AsmLocusOut lo    = P.loci[locus];
    SpanTaskInfo      info  = {0, -1, -1, 0};
    if (lo.status == ASM_OK && ci < lo.n_contigs) {
      const JumpCuts c    = P.cuts[locus];
      const int      len1 = int(P.ref1_off[locus + 1] - P.ref1_off[locus]), len2 = int(P.ref2_off[locus + 1] - P.ref2_off[locus]);
      const int      b    = spanFileTask(P, slot, total, P.contigs[slot], P.refs1 + P.ref1_off[locus] + c.a1Lead, len1 - c.a1Lead - c.a1Trail,
                                         P.refs2 + P.ref2_off[locus] + c.a2Lead, len2 - c.a2Lead - c.a2Trail, P.tasks, P.bucket_ids,
                                         P.bucket_count, P.bucket_maxref);
      if (b >= 0)
        info.bucket = b;
      else
        info.status = -b;
    } else if (lo.status != ASM_OK) {
      info.status = 1;
    }
    P.info[slot] = info;
  }
}
#endif

/// the re-align rule (:1672-1713), one lane per locus: the first contig (in contig order) whose junction holds an
/// insertion while a breakend sits within 5 bases of a cut edge zeroes the cuts -- for itself and, because the
/// reference's AlignData is shared by the contig loop, for every later contig of the locus.
#if !MANTA_TU_DEFINES(MANTA_TU_GLUE)
WV_KERNEL void spanning_realign_kernel(const SpanParams P);
#else
WV_KERNEL void spanning_realign_kernel(const SpanParams P)
{
  const unsigned total = P.n_loci * P.max_assembly_count;
  for (unsigned locus = unsigned(wv::block()) * 64u + unsigned(wv::lane()); locus < P.n_loci; locus += unsigned(wv::nblocks()) * 64u) {
    if (P.pass_mask && P.pass_mask[locus] != P.pass_value) continue;
    const AsmLocusOut lo = P.loci[locus];
    if (lo.status != ASM_OK) continue;
    const JumpCuts c    = P.cuts[locus];
    const int      len1 = int(P.ref1_off[locus + 1] - P.ref1_off[locus]), len2 = int(P.ref2_off[locus + 1] - P.ref2_off[locus]);
    unsigned       firstUncut = lo.n_contigs;
    for (unsigned ci = 0; ci < lo.n_contigs; ++ci) {
      const unsigned slot = locus * P.max_assembly_count + ci;
      if (P.info[slot].status != 0) continue;
      const AlignResultDev r = P.results[slot];
      if (r.status != 0 || r.jump_insert_size == 0) continue;
      // reference length of align1's path: '=' (7), 'X' (8), 'D' (2), 'N' (3) segments
      const uint32_t* cig    = P.cigar + P.tasks[slot].cigar_off;
      int             refLen = 0;
      for (unsigned i = 0; i < r.cigar1_len; ++i) {
        const unsigned op = cig[i] & 15u;
        if (op == 7 || op == 8 || op == 2 || op == 3 || op == 0) refLen += int(cig[i] >> 4);
      }
      const int ref1EndPos   = len1 - c.a1Lead - c.a1Trail - 1;
      const int align1EndPos = r.begin1 + refLen;
      if ((ref1EndPos - align1EndPos < 5) || (r.begin2 < 5)) {
        firstUncut = ci;
        break;
      }
    }
    for (unsigned ci = firstUncut; ci < lo.n_contigs; ++ci) {
      const unsigned slot = locus * P.max_assembly_count + ci;
      SpanTaskInfo   info = P.info[slot];
      info.is_uncut       = 1;
      if (info.status == 0) {
        const int b = spanFileTask(P, slot, total, P.contigs[slot], P.refs1 + P.ref1_off[locus], len1, P.refs2 + P.ref2_off[locus], len2,
                                   P.tasks2, P.bucket_ids2, P.bucket_count2, P.bucket_maxref2);
        if (b >= 0) {
          info.bucket2 = b;
          info.status  = 0;
        } else {
          info.status = -b;
        }
      }
      P.info[slot] = info;
    }
  }
}
#endif

// ------------------------------------------------------------------------------------------------------
// Result packing: the last kernel of both pipelines.  The stages above keep one record per (locus, contig slot) --
// mostly empty -- and a CIGAR region of 4*Q+16 words per task; a result download of those would move ~10x the payload
// over PCIe.  One lane per locus gathers what the host needs into dense arrays: one PackedContigOut per contig and
// the CIGAR words back to back.
// ------------------------------------------------------------------------------------------------------
struct PackedContigOut {
  AsmContigOut contig;
  int32_t      info_status;  ///< the schedule kernel's per-slot status
  int32_t      bucket;       ///< E bucket of the alignment that counts (-1: none)
  int32_t      res_status;   ///< AlignResultDev::status of that alignment
  int32_t      score, is_jumped, begin1, begin2;
  uint32_t     jump_insert_size, jump_range, cigar1_len, cigar2_len;
  uint32_t     cigar_off;  ///< first word in the packed CIGAR arena
  int32_t      a, b;       ///< small SV: adjusted leading / trailing cut; spanning: is_uncut, 0
  uint32_t     query_len, ref_len;
};

struct PackParams {
  const AsmLocusOut*     loci;
  const AsmContigOut*    contigs;
  uint32_t               n_loci, max_assembly_count;
  const AlignTaskDev*    tasks;
  const AlignTaskDev*    tasks2;  ///< spanning round 2 (nullptr for small SV)
  const AlignResultDev*  results;
  const AlignResultDev*  results2;
  const SmallSvTaskInfo* info_small;  ///< exactly one of info_small / info_span is set
  const SpanTaskInfo*    info_span;
  const uint32_t*        cigar;
  uint32_t*              first;         ///< out: n_loci, index of the locus' first packed contig
  PackedContigOut*       packed;        ///< out
  uint32_t*              cigar_packed;  ///< out
  uint32_t*              counters;      ///< [0] packed contigs, [1] packed cigar words (zeroed before launch)
};

#if !MANTA_TU_DEFINES(MANTA_TU_GLUE)
WV_KERNEL void pack_results_kernel(const PackParams P);
#else
WV_KERNEL void pack_results_kernel(const PackParams P)
{
  for (unsigned locus = unsigned(wv::block()) * 64 + unsigned(wv::lane()); locus < P.n_loci; locus += unsigned(wv::nblocks()) * 64) {
    const AsmLocusOut lo = P.loci[locus];
    const unsigned    n  = (lo.status == ASM_OK) ? lo.n_contigs : 0u;
    const unsigned    base = n ? wv::atomic_add(&P.counters[0], n) : 0u;
    P.first[locus]         = base;
    for (unsigned c = 0; c < n; ++c) {
      const unsigned  slot = locus * P.max_assembly_count + c;
      PackedContigOut o;
      o.contig = P.contigs[slot];
      const AlignResultDev* res  = &P.results[slot];
      const AlignTaskDev*   task = &P.tasks[slot];
      if (P.info_small) {
        const SmallSvTaskInfo inf = P.info_small[slot];
        o.info_status             = inf.status;
        o.bucket                  = inf.bucket;
        o.a                       = inf.adj_leading_cut;
        o.b                       = inf.adj_trailing_cut;
      } else {
        const SpanTaskInfo inf = P.info_span[slot];
        const bool         uncut = inf.is_uncut != 0;
        o.info_status            = inf.status;
        o.bucket                 = uncut ? inf.bucket2 : inf.bucket;
        o.a                      = uncut ? 1 : 0;
        o.b                      = 0;
        if (uncut) {
          res  = &P.results2[slot];
          task = &P.tasks2[slot];
        }
      }
      const bool           ok = (o.info_status == 0 && o.bucket >= 0);
      const AlignResultDev r  = *res;
      o.res_status            = ok ? r.status : -1;
      o.score                 = r.score;
      o.is_jumped             = r.is_jumped;
      o.begin1                = r.begin1;
      o.begin2                = r.begin2;
      o.jump_insert_size      = r.jump_insert_size;
      o.jump_range            = r.jump_range;
      o.cigar1_len            = (ok && r.status == 0) ? r.cigar1_len : 0u;
      o.cigar2_len            = (ok && r.status == 0) ? r.cigar2_len : 0u;
      o.query_len             = ok ? task->query_len : 0u;
      o.ref_len               = ok ? (task->ref1_len + task->ref2_len) : 0u;
      const unsigned words    = o.cigar1_len + o.cigar2_len;
      o.cigar_off             = words ? wv::atomic_add(&P.counters[1], words) : 0u;
      const uint32_t* src     = P.cigar + task->cigar_off;
      for (unsigned i = 0; i < words; ++i) P.cigar_packed[o.cigar_off + i] = src[i];
      P.packed[base + c] = o;
    }
  }
}
#endif

/// how many loci ended with `code` (after assemble_kernel: ASM_E_TABLE_FULL -> AsmStage::rerunCapacityFailures).  A separate tiny
/// launch instead of a counter inside assemble_kernel: that kernel's code stays exactly what was measured.
struct CountStatusParams {
  const AsmLocusOut* loci;
  uint32_t           n_loci;
  int32_t            code;
  unsigned long long* counter;
};
#if !MANTA_TU_DEFINES(MANTA_TU_GLUE)
WV_KERNEL void count_status_kernel(const CountStatusParams P);
#else
WV_KERNEL void count_status_kernel(const CountStatusParams P)
{
  unsigned n = 0;
  for (unsigned l = unsigned(wv::block()) * 64u + unsigned(wv::lane()); l < P.n_loci; l += unsigned(wv::nblocks()) * 64u)
    n += (P.loci[l].status == P.code) ? 1u : 0u;
  for (int off = 1; off < 64; off <<= 1) n += wv::shfl(n, wv::lane() ^ off);
  if (wv::lane() == 0 && n) wv::atomic_add(P.counter, (unsigned long long)n);
}
#endif

}  // namespace manta_dev
