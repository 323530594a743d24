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
//     batch.retrieveRemoteReads(fetcher, options);   // complex candidates searched with isSearchRemoteInsertionReads (:570-655)
//
// addRecord takes the htslib bam1_t fields as plain values so that this header does not depend on htslib.
//
// Remote mates (retrieveRemoteReads, :141-256).  The kernel flags the records whose mates the reference would look up
// (MANTA_READ_REMOTE_MATE) and reports isRetrieveRemoteReads per candidate; the look-ups are BAM seeks, i.e. the caller's
// bam_streamer, reached through RemoteMateFetcher.  Everything between the flags and the final pile is here: the target list per
// file (RemoteMateReadUtil.hpp:39-71), its sort and the merge into region queries (:154-185), the scan of a region with the
// (read number, name) match, the MAPQ-0 rule, the orientation rule and insertAssemblyRead's read index (:205-250) -- the mates go
// behind the candidate's local reads exactly as the reference appends them.
#pragma once
#include <algorithm>
#include <cctype>
#include <cstring>
#include <functional>
#include <string>
#include <unordered_set>
#include <vector>

#include "manta_amd.hpp"
#include "read_pile.hpp"

namespace manta_amd {

/// one record of a remote region query, as addRecord takes them
struct RemoteRecord {
  int32_t        pos  = 0;  ///< bam1_core_t::pos (0-based)
  uint16_t       flag = 0;
  uint8_t        mapq = 0;
  const char*    qname = nullptr;
  const uint8_t* seq4  = nullptr;
  const uint8_t* qual  = nullptr;
  uint32_t       lQseq = 0;
  bool           hasSA = false;  ///< bam_record::isSASplit
};

/// the caller's BAM layer: `bamStream.resetRegion(tid, begin, end); while (bamStream.next()) ...` on file `bamIndex`
/// (SVCandidateAssembler.cpp:205-207).  Hands the records to `sink` in file order and stops when it returns false.
struct RemoteMateFetcher {
  virtual ~RemoteMateFetcher() {}
  virtual void scanRegion(uint32_t bamIndex, int32_t tid, int32_t begin, int32_t end, const std::function<bool(const RemoteRecord&)>& sink) = 0;
};

/// what the reference keeps in its RemoteReadCache for PE scoring (:241)
struct RemoteReadCacheEntry {
  uint32_t    candidate;
  std::string qname;
  int         readNo;
  uint64_t    pileRead;  ///< index into finalPiles()
};

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
    _haveFinal  = false;
  }

  /// The reference's tail of getBreakendReads (:570-655) for every candidate added with isSearchRemote whose result says
  /// retrieve_remote.  Afterwards finalPiles() / finalPileReadText() / finalLocusReadBegin() describe the piles with the mates
  /// appended (identical to piles() when nothing was fetched); a candidate whose mate holds the BAM code '=' gets status
  /// MANTA_E_UNSUPPORTED like a local read would.
  void retrieveRemoteReads(RemoteMateFetcher& fetcher, const manta_read_class_options_t& opt)
  {
    const unsigned maxNumReads = opt.max_reads ? opt.max_reads : 1000u;  // :342
    _final.clear();
    remoteCache.clear();
    nRemoteTargets = nRemoteInserted = 0;
    struct Target {  // RemoteReadInfo
      std::string qname;
      int         readNo, tid, pos, readSize;
      bool        isFound;
    };
    for (size_t l = 0; l < loci.size(); ++l) {
      // the candidate's local reads, as the device packed them
      for (uint32_t r = locusReadBegin[l]; r < locusReadBegin[l + 1]; ++r) copyPileRead(r);
      if (loci[l].search_remote && results[l].retrieve_remote && results[l].status == MANTA_OK) {
        uint32_t nBam = 0;
        for (uint32_t s = loci[l].scan_begin; s < loci[l].scan_end; ++s) nBam = std::max(nBam, scans[s].bam_index + 1);
        std::unordered_set<std::string> readIndex;  // insertAssemblyRead's keys of what is in the pile (:108-119)
        bool                            haveIndex = false;
        for (uint32_t bamIndex = 0; bamIndex < nBam; ++bamIndex) {  // :620
          std::vector<Target> remotes;
          bool                isLocusReversed = false;
          for (uint32_t s = loci[l].scan_begin; s < loci[l].scan_end; ++s) {
            if (scans[s].bam_index != bamIndex) continue;
            isLocusReversed = scans[s].is_locus_reversed != 0;
            for (uint32_t i = scans[s].read_begin; i < scans[s].read_end; ++i) {
              if (!(decision[i] & MANTA_READ_REMOTE_MATE)) continue;
              const manta_bam_read_t& r(reads[i]);
              remotes.push_back(Target{std::string(reinterpret_cast<const char*>(names.data()) + r.qname_off, r.qname_len),
                                       (recordReadNo(r.flag) == 1) ? 2 : 1, r.mate_tid, r.mate_pos, int(r.read_len), false});
            }
          }
          if (remotes.empty()) continue;
          nRemoteTargets += remotes.size();
          if (!haveIndex) {
            for (uint32_t s = loci[l].scan_begin; s < loci[l].scan_end; ++s)
              for (uint32_t i = scans[s].read_begin; i < scans[s].read_end; ++i)
                if (decision[i] & MANTA_READ_IN_PILE) readIndex.insert(readKey(reads[i], scans[s].bam_index));
            haveIndex = true;
          }
          std::sort(remotes.begin(), remotes.end(), [](const Target& a, const Target& b) {  // RemoteMateReadUtil.hpp:53-60
            if (a.tid < b.tid) return true;
            if (a.tid == b.tid) return a.pos < b.pos;
            return false;
          });
          struct Region {
            int    tid, begin, end;
            size_t first, last;  // targets [first, last)
          };
          std::vector<Region> regions;
          int                 lastTid = -1, lastPos = -1;
          for (size_t t = 0; t < remotes.size(); ++t) {  // :162-185
            if (lastTid == remotes[t].tid && lastPos + remotes[t].readSize >= remotes[t].pos) {
              regions.back().end  = remotes[t].pos;
              regions.back().last = t + 1;
            } else {
              regions.push_back(Region{remotes[t].tid, remotes[t].pos, remotes[t].pos, t, t + 1});
            }
            lastTid = remotes[t].tid;
            lastPos = remotes[t].pos;
          }
          for (const Region& g : regions) {
            const int lastTargetPos = remotes[g.last - 1].pos;
            fetcher.scanRegion(bamIndex, g.tid, g.begin, g.end + 1, [&](const RemoteRecord& rec) -> bool {
              if (_final.nReads() - _final.locusBegin.back() >= maxNumReads) return false;  // :208
              if (rec.pos + 1 > lastTargetPos + 1) return false;                           // :219 (bam_record::pos() is 1-based)
              if ((rec.flag & 0x800u) || ((rec.flag & 0x100u) && rec.hasSA)) return true;  // isNonStrictSupplement
              for (size_t t = g.first; t < g.last; ++t) {
                Target& remote(remotes[t]);
                if (remote.isFound) continue;
                if (recordReadNo(rec.flag) != remote.readNo) continue;
                if (std::strcmp(rec.qname, remote.qname.c_str()) != 0) continue;
                remote.isFound = true;
                if (rec.mapq != 0) break;
                bool isReversed = isLocusReversed;
                if (!(rec.flag & 0x10u) == !(rec.flag & 0x20u)) isReversed = !isReversed;  // :231-234
                const std::string key = std::string(rec.qname) + "_" + ((rec.flag & 0x80u) ? '2' : '1') + "_" + std::to_string(bamIndex);
                if (!readIndex.insert(key).second) break;  // name collision (:112-119)
                if (!_final.addBamRead(rec.seq4, rec.qual, rec.lQseq, uint8_t(opt.min_qval), isReversed)) {
                  results[l].status = MANTA_E_UNSUPPORTED;
                  break;
                }
                remoteCache.push_back(RemoteReadCacheEntry{uint32_t(l), remote.qname, recordReadNo(rec.flag), uint64_t(_final.nReads() - 1)});
                ++nRemoteInserted;
                break;
              }
              return true;
            });
          }
        }
      }
      _final.endLocus();
    }
    _haveFinal = true;
  }

  /// piles with the remote mates appended (retrieveRemoteReads), else the device piles
  manta_packed_piles_t finalPiles() const { return _haveFinal ? _final.view() : piles(); }
  uint64_t             nFinalPileReads() const { return _haveFinal ? _final.nReads() : _nPileReads; }
  uint32_t             finalLocusReadBegin(const size_t l) const { return _haveFinal ? _final.locusBegin[l] : locusReadBegin[l]; }
  std::string          finalPileReadText(const uint64_t r) const
  {
    if (!_haveFinal) return pileReadText(r);
    std::string s(_final.readLen[r], 'N');
    for (uint32_t i = 0; i < _final.readLen[r]; ++i)
      if (!((_final.nmask[_final.maskOff[r] + (i >> 5)] >> (i & 31)) & 1u))
        s[i] = "ACGT"[(_final.codes[_final.codeOff[r] + (i >> 4)] >> (30 - 2 * (i & 15))) & 3u];
    return s;
  }
  std::vector<RemoteReadCacheEntry> remoteCache;
  uint64_t                          nRemoteTargets = 0, nRemoteInserted = 0;

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
  static int recordReadNo(const uint16_t flag) { return ((flag & 0x80u) && !(flag & 0x40u)) ? 2 : 1; }  // bam_record::read_no
  std::string readKey(const manta_bam_read_t& r, const uint32_t bamIndex) const
  {
    return std::string(reinterpret_cast<const char*>(names.data()) + r.qname_off, r.qname_len) + "_" + ((r.flag & 0x80u) ? '2' : '1') + "_" +
           std::to_string(bamIndex);
  }
  void copyPileRead(const uint64_t r)
  {
    _final.codes.insert(_final.codes.end(), codes.begin() + long(codeOff[r]), codes.begin() + long(codeOff[r] + (readLen[r] + 15) / 16));
    _final.nmask.insert(_final.nmask.end(), nmask.begin() + long(maskOff[r]), nmask.begin() + long(maskOff[r] + (readLen[r] + 31) / 32));
    _final.readLen.push_back(readLen[r]);
    _final.codeOff.push_back(_final.codes.size());
    _final.maskOff.push_back(_final.nmask.size());
  }
  int32_t         _searchBegin = 0, _searchEnd = 0;
  uint64_t        _nPileReads  = 0;
  ReadPileBuilder _final;
  bool            _haveFinal = false;
};

}  // namespace manta_amd
