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
static const unsigned SCHED_LDS_BYTES   = 8192;  // per wavefront: 10-mer table of contigs up to 1024 bp lives in LDS

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
WV_DEV unsigned merCodes55(const uint8_t* seq, const int len, const int p0, bool& valid)
{
  const int      l   = wv::lane();
  const int      pos = p0 + l;
  unsigned       c   = 4;
  if (pos >= 0 && pos < len) c = baseCode(seq[pos]);
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
  const uint8_t*     contig  = P.seq_arena + co.seq_off;
  const unsigned     clen    = co.seq_len;
  const uint8_t*     ref     = P.refs + P.ref_off[locus];
  const int          refSize = int(P.ref_off[locus + 1] - P.ref_off[locus]);
  const SmallSvCuts  cuts    = P.cuts[locus];

  if (clen < unsigned(SMALLSV_MER) || 2 * clen > P.table_cap) {
    info.status = 2;
    return info;
  }
  // hash set of the contig's 10-mers (:1987-1991)
  unsigned tcap = 64;
  while (tcap < 2 * clen) tcap <<= 1;
  const unsigned mask  = tcap - 1;
  uint32_t*      table = (tcap * 4 <= SCHED_LDS_BYTES) ? ltable : gtable;
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
  // first hit scanning forward (:1997-2001)
  int adjLead = maxFwdRefIndex + 1;
  if (adjLead < minRefIndex) adjLead = minRefIndex;  // empty scan range: the loop variable keeps its initial value
  for (int base = minRefIndex; base <= maxFwdRefIndex; base += MER_BATCH) {
    bool           valid;
    const unsigned code = merCodes55(ref, refSize, base, valid);
    const int      i    = base + int(lane);
    const bool     hit  = valid && i <= maxFwdRefIndex && merLookup(table, mask, code);
    const uint64_t m    = wv::ballot(hit);
    if (m) {
      adjLead = base + wv::ctz(m);
      break;
    }
  }
  // last hit scanning backward (:2004-2008): windows of 55 start positions, highest window first
  const int minRevRefIndex = (minRefIndex > refSize - cuts.maxTrailingCut) ? minRefIndex : (refSize - cuts.maxTrailingCut);
  int       revIndex       = minRevRefIndex - 1;
  if (revIndex > maxRefIndex) revIndex = maxRefIndex;  // empty scan range
  for (int top = maxRefIndex; top >= minRevRefIndex; top -= MER_BATCH) {
    const int      base = top - (MER_BATCH - 1);
    bool           valid;
    const unsigned code = merCodes55(ref, refSize, base, valid);
    const int      i    = base + int(lane);
    const bool     hit  = valid && i >= minRevRefIndex && i <= top && merLookup(table, mask, code);
    const uint64_t m    = wv::ballot(hit);
    if (m) {
      revIndex = base + (63 - wv::clz(m));
      break;
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
  if (bucket < 0) {
    info.status = 4;
    return info;
  }
  // lane 0 claims CIGAR space and files the task; the verdict is broadcast so that `info` stays wave-uniform
  unsigned claimed = 0;
  if (lane == 0) {
    const unsigned long long cig = wv::atomic_add(P.cigar_used, (unsigned long long)(4ull * clen + 16));
    if (cig + 4ull * clen + 16 <= P.cigar_cap) {
      claimed = 1;
      AlignTaskDev t;
      t.query     = contig;
      t.ref1      = ref + adjLead;
      t.ref2      = nullptr;
      t.query_len = clen;
      t.ref1_len  = unsigned(winLen);
      t.ref2_len  = 0;
      t.cigar_off = uint32_t(cig);
      P.tasks[slot] = t;
      const unsigned pos = wv::atomic_add(&P.bucket_count[bucket], 1u);
      P.bucket_ids[size_t(bucket) * total + pos] = slot;
      // atomic max of the window length via CAS loop
      unsigned cur = wv::atomic_load(&P.bucket_maxref[bucket]);
      while (cur < unsigned(winLen)) {
        const unsigned old = wv::atomic_cas(&P.bucket_maxref[bucket], cur, unsigned(winLen));
        if (old == cur) break;
        cur = old;
      }
    }
  }
  claimed = wv::first(claimed);
  if (claimed)
    info.bucket = bucket;
  else
    info.status = 5;
  return info;
}

WV_KERNEL void smallsv_schedule_kernel(const ScheduleParams P)
{
  const unsigned lane   = unsigned(wv::lane());
  uint32_t*      gtable = P.table_ws + size_t(wv::block()) * P.table_cap;
  uint32_t*      ltable = reinterpret_cast<uint32_t*>(wv::lds(SCHED_LDS_BYTES));
  const unsigned total  = P.n_loci * P.max_assembly_count;
  while (true) {
    unsigned slot = 0;
    if (lane == 0) slot = wv::atomic_add(P.counter, 1u);
    slot = wv::first(slot);
    if (slot >= total) break;
    const SmallSvTaskInfo info = scheduleSlot(P, slot, total, ltable, gtable);
    wv::sync();  // single reconvergence point of every exit of scheduleSlot
    if (lane == 0) P.info[slot] = info;
    wv::sync();
  }
}

}  // namespace manta_dev
