// Host side of the device read gathering (second half): what SVCandidateAssembler's candidate loop hands to
// manta_read_piles_batch (include/manta_amd.h) instead of running getBreakendReads' per-read tests itself
// (manta/SVCandidateAssembler.cpp:271-659; paths relative to /root/reference/src/c++/lib).
//
// The BAM layer stays the reference's (bam_streamer / htslib).  Where the reference's loop body (:387-567) tests a record and
// calls insertAssemblyRead, the drop-in loop only copies the record's fields:
//
//     ReadGatherBatch batch;                                     // one per worker thread, reused
//     for each candidate:
//       batch.beginCandidate(isMaxDepth, maxDepth, maxLocalDepth, isSearchRemoteInsertionReads);
//       for each breakend, for each bam file (the order of :371):
//         batch.beginQuery(bp.interval.range.begin_pos(), ...end_pos(), bp.state, isLocusReversed, bamIndex, isTumor,
//                          bamIndex == 0, refSeq.get_offset(), refSeq.seq());
//         bamStream.resetRegion(tid, batch.searchBegin(), batch.searchEnd());
//         while (bamStream.next()) batch.addRecord(*bamStream.get_record_ptr()->get_data(), isSASplit, mateCigarOrNull);
//     batch.run(ctx, options);   // -> piles() in manta_packed_piles_t layout, decisions(), results()
//
// addRecord takes the htslib bam1_t fields as plain values so that this header does not depend on htslib.
#pragma once
#include <cctype>
#include <cstring>
#include <string>
#include <vector>

#include "manta_amd.hpp"

namespace manta_amd {

struct ReadGatherBatch {
  std::vector<manta_read_locus_t>        loci;
  std::vector<manta_read_scan_t>         scans;
  std::vector<manta_bam_read_t>          reads;
  std::vector<uint32_t>                  cigars;
  std::vector<uint8_t>                   names, seqs, quals, refs;
  // results
  std::vector<uint8_t>                   decision;
  std::vector<uint32_t>                  pileIndex, codes, nmask, readLen, pileRead, locusReadBegin;
  std::vector<uint64_t>                  codeOff, maskOff;
  std::vector<manta_read_locus_result_t> results;

  void clear()
  {
    loci.clear();
    scans.clear();
    reads.clear();
    cigars.clear();
    names.clear();
    seqs.clear();
    quals.clear();
    refs.clear();
  }

  void beginCandidate(const bool isMaxDepth, const float maxDepth, const float maxLocalDepthForRemoteReadRetrieval, const bool isSearchRemote)
  {
    manta_read_locus_t l;
    std::memset(&l, 0, sizeof(l));
    l.scan_begin = l.scan_end    = uint32_t(scans.size());
    l.is_max_depth               = isMaxDepth ? 1 : 0;
    l.search_remote              = isSearchRemote ? 1 : 0;
    l.max_depth                  = maxDepth;
    l.max_local_depth_remote     = maxLocalDepthForRemoteReadRetrieval;
    loci.push_back(l);
  }

  /// one bamStream.resetRegion of getBreakendReads (:381).  refSeq: the reference_contig_segment the refiner fetched.
  void beginQuery(
      const int32_t bpBegin, const int32_t bpEnd, const int32_t bpState, const bool isLocusReversed, const uint32_t bamIndex,
      const bool isTumor, const bool isFirstFileOfBreakend, const int32_t refOffset, const std::string& refSeq)
  {
    manta_read_scan_t s;
    std::memset(&s, 0, sizeof(s));
    s.read_begin = s.read_end = uint32_t(reads.size());
    s.bam_index         = bamIndex;
    s.is_tumor          = isTumor ? 1 : 0;
    s.is_locus_reversed = isLocusReversed ? 1 : 0;
    s.first_of_breakend = isFirstFileOfBreakend ? 1 : 0;
    s.bp_begin          = bpBegin;
    s.bp_end            = bpEnd;
    s.bp_state          = bpState;
    s.ref_begin         = refOffset;
    s.ref_len           = uint32_t(refSeq.size());
    s.ref_off           = refs.size();
    refs.insert(refs.end(), refSeq.begin(), refSeq.end());
    scans.push_back(s);
    loci.back().scan_end = uint32_t(scans.size());
    manta_read_search_range(bpBegin, bpEnd, &_searchBegin, &_searchEnd);
  }
  int32_t searchBegin() const { return _searchBegin; }
  int32_t searchEnd() const { return _searchEnd; }

  /// one record of the query, in file order: the bam1_core_t fields, bam_get_cigar / bam_get_qname / bam_get_seq / bam_get_qual,
  /// whether it carries an SA tag (bam_record::isSASplit) and its MC tag text (nullptr: none).  Records at or behind the search
  /// end may be passed or not (the scan stops at the first one, :397-398).
  void addRecord(
      const int32_t tid, const int32_t pos, const int32_t mtid, const int32_t mpos, const uint16_t flag, const uint8_t mapq,
      const uint32_t* cigar, const uint32_t nCigar, const char* qname, const uint8_t* seq4, const uint8_t* qual, const uint32_t lQseq,
      const bool hasSA, const char* mateCigar)
  {
    manta_bam_read_t r;
    std::memset(&r, 0, sizeof(r));
    r.tid       = tid;
    r.pos       = pos;
    r.mate_tid  = mtid;
    r.mate_pos  = mpos;
    r.flag      = flag;
    r.mapq      = mapq;
    r.tags      = uint8_t((hasSA ? MANTA_READ_TAG_SA : 0u) | (mateCigar ? MANTA_READ_TAG_MC : 0u));
    r.read_len  = lQseq;
    r.n_cigar   = nCigar;
    r.cigar_off = uint32_t(cigars.size());
    cigars.insert(cigars.end(), cigar, cigar + nCigar);
    r.mate_cigar_off = uint32_t(cigars.size());
    if (mateCigar) {  // cigar_to_apath (blt_util/align_path.cpp:66-95): P and zero-length operations are dropped
      static const char ops[] = "MIDNSHP=X";
      uint32_t          len   = 0;
      for (const char* c = mateCigar; *c; ++c) {
        if (std::isdigit(static_cast<unsigned char>(*c))) {
          len = len * 10 + uint32_t(*c - '0');
          continue;
        }
        const char* at = std::strchr(ops, *c);
        if (!at) throw GeneralException(std::string("can't parse unknown cigar string: ") + mateCigar);  // unknown_cigar_error
        const uint32_t op = uint32_t(at - ops);
        if (op != 6 && len != 0) cigars.push_back((len << 4) | op);
        len = 0;
      }
    }
    r.n_mate_cigar = uint32_t(cigars.size()) - r.mate_cigar_off;
    r.qname_len    = uint32_t(std::strlen(qname));
    r.qname_off    = uint32_t(names.size());
    names.insert(names.end(), qname, qname + r.qname_len);
    r.seq_off = seqs.size();
    seqs.insert(seqs.end(), seq4, seq4 + (lQseq + 1) / 2);
    r.qual_off = quals.size();
    quals.insert(quals.end(), qual, qual + lQseq);
    reads.push_back(r);
    scans.back().read_end = uint32_t(reads.size());
  }

  /// the reference's defaults (options/ReadScannerOptions.hpp, options/IterativeAssemblerOptions.hpp:33)
  static manta_read_class_options_t defaultOptions()
  {
    manta_read_class_options_t o;
    o.min_qval                      = 5;
    o.min_candidate_variant_size    = 10;
    o.min_singleton_mapq_candidates = 15;
    o.min_mapq                      = 15;
    o.use_overlap_pair_evidence     = 0;
    o.max_reads                     = 0;
    return o;
  }

  /// Builds the piles of every candidate added since clear().  Candidates whose kept reads hold the BAM code '=' come back with
  /// status MANTA_E_UNSUPPORTED in results() (their decisions are valid; the caller builds that pile as text); anything else
  /// that fails throws.
  void run(manta_ctx_t* ctx, const manta_read_class_options_t& opt)
  {
    const size_t n = reads.size();
    uint64_t     codeCap = 4, maskCap = 4;
    for (const manta_bam_read_t& r : reads) {
      codeCap += (uint64_t(r.read_len) + 15) / 16;
      maskCap += (uint64_t(r.read_len) + 31) / 32;
    }
    decision.assign(n + 1, 0);
    pileIndex.assign(n + 1, 0);
    codes.assign(codeCap, 0);
    nmask.assign(maskCap, 0);
    readLen.assign(n + 1, 0);
    pileRead.assign(n + 1, 0);
    codeOff.assign(n + 2, 0);
    maskOff.assign(n + 2, 0);
    locusReadBegin.assign(loci.size() + 1, 0);
    results.assign(loci.size() + 1, manta_read_locus_result_t());
    uint64_t codesUsed = 0, maskUsed = 0, readsUsed = 0;
    const uint8_t pad = 0;
    const int rc = manta_read_piles_batch(
        ctx, &opt, uint32_t(loci.size()), loci.data(), uint32_t(scans.size()), scans.data(), uint32_t(n), reads.data(), cigars.data(),
        cigars.size(), names.empty() ? &pad : names.data(), names.size(), seqs.empty() ? &pad : seqs.data(), seqs.size(),
        quals.empty() ? &pad : quals.data(), quals.size(), refs.empty() ? &pad : refs.data(), refs.size(), decision.data(), pileIndex.data(),
        results.data(), codes.data(), codeCap, &codesUsed, nmask.data(), maskCap, &maskUsed, readLen.data(), codeOff.data(), maskOff.data(),
        pileRead.data(), n, &readsUsed, locusReadBegin.data());
    if (rc != MANTA_OK && rc != MANTA_E_UNSUPPORTED) throw GeneralException(std::string("manta_amd read gathering: ") + manta_last_error(ctx), rc);
    _nPileReads = readsUsed;
  }

  manta_packed_piles_t piles() const
  {
    manta_packed_piles_t p;
    p.codes            = codes.data();
    p.nmask            = nmask.data();
    p.read_len         = readLen.data();
    p.read_code_off    = codeOff.data();
    p.read_mask_off    = maskOff.data();
    p.locus_read_begin = locusReadBegin.data();
    return p;
  }
  uint64_t nPileReads() const { return _nPileReads; }

  /// pile read `r` as the text the reference would have pushed into AssemblyReadInput
  std::string pileReadText(const uint64_t r) const
  {
    std::string s(readLen[r], 'N');
    for (uint32_t i = 0; i < readLen[r]; ++i)
      if (!((nmask[maskOff[r] + (i >> 5)] >> (i & 31)) & 1u)) s[i] = "ACGT"[(codes[codeOff[r] + (i >> 4)] >> (30 - 2 * (i & 15))) & 3u];
    return s;
  }

private:
  int32_t  _searchBegin = 0, _searchEnd = 0;
  uint64_t _nPileReads  = 0;
};

}  // namespace manta_amd
