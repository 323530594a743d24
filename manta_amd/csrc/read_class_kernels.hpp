// Read-pile construction on the device: the per-read tests of SVCandidateAssembler::getBreakendReads
// (manta/SVCandidateAssembler.cpp:271-659; paths relative to /root/reference/src/c++/lib) over decoded BAM records, and
// insertAssemblyRead's string handling (:102-136) straight into the packed pile layout (manta_packed_piles_t).
//
// The reference walks the records of a region query one by one and carries state from record to record: the depth buffer of
// the normal samples (:85-100, :404-425), the shadow finder's "last anchor" (ShadowReadFinder.cpp:33-113), the read index
// (:102-119) and the 1000-read cap (:387-393).  None of that state reaches far, so one wavefront takes a candidate and runs
// the scan as passes over 64 records at a time (lane = record):
//   A  per record: isReadFilteredCore, supplement, "adds to the depth estimate"; coverage differences by atomics
//   B  prefix sum -> coverage of this query.  The depth a record sees is the coverage of the earlier normal queries of the
//      breakend + this query's coverage - what later records of the query add at that position (records are position
//      sorted: those are the records after it that start at the same position)
//   C  per record: depth thresholds; indel test; semi-aligned test (per-lane walk over cigar and bases); remote-mate test;
//      shadow-anchor and shadow predicates
//   D  shadow = the previous record that reached ShadowReadFinder::check is an anchor, and the names match: a prefix "last
//      reaching record" per 64 records with a carry
//   E  read index: a hash table per candidate holds the smallest record index of every read key (atomic min); a record is the
//      first of its key iff it is that index
//   F  ordered count of the firsts -> pile positions; the cap cuts everything behind the 1000th insertion (those records are
//      never looked at by the reference: their decisions are cleared, their depth trips do not count)
// What in C depends on the record alone -- the expensive part: cigar walks, the semi-aligned base comparison, quality sums -- runs
// first and RECORD-PARALLEL over the whole batch (read_test_kernel: a wave per 64 records of a query, one byte of verdicts per
// record); the candidate-serial passes above then only combine those bits with the state they carry.
// Then one wave scans the per-candidate totals into offsets, a wave per candidate lays out its pile reads' rows, and the inserted
// reads are converted PILE-READ-PARALLEL (4 bit -> 2 bit + N bitmap, Q mask, reverse complement; eight lanes per read, 32 bases per
// lane step).  Five launches, no host turnaround.
#pragma once
#include "../../include/manta_amd.h"
#include "wave.hpp"

namespace manta_dev {

// temp word per record: low bits = flags below, bits 8.. = bam_index << 1 | is_second (the read key's other parts)
enum {
  RC_VALID  = 1u,   // before the end of the scan (:398)
  RC_REACH  = 2u,   // reaches ShadowReadFinder::check (not filtered, not a supplement, not over depth)
  RC_ANCHOR = 4u,   // isShadowAnchor
  RC_GOOD   = 8u,   // isGoodShadow but for the name comparison
  RC_KEEP   = 16u,  // a keeper
  RC_TRIP   = 32u,  // over the local depth threshold for remote retrieval
  RC_MATEREV = 64u  // shadow keeper whose mate is on the forward strand: orientation flips (:558-563)
};
static const unsigned RC_EMPTY = 0xffffffffu;
// read_test_kernel's byte per record
enum { PRE_REMOTE = 1u, PRE_INDEL = 2u, PRE_SEMI = 4u, PRE_ANCHOR = 8u, PRE_GOOD = 16u };

struct ReadClassParams {
  manta_read_class_options_t       opt;
  uint32_t                         n_loci;
  const manta_read_locus_t*        loci;
  const manta_read_scan_t*         scans;
  const manta_bam_read_t*          reads;
  const uint32_t*                  cigars;
  const uint8_t *                  names, *seqs, *quals, *refs;
  uint8_t*                         decision;
  uint32_t*                        pile_index;
  manta_read_locus_result_t*       results;
  uint32_t*                        tmp;        // [n_reads]; after read_class_kernel: the candidate of pile read r
  uint8_t*                         pre;        // [n_reads] PRE_* (read_test_kernel)
  const uint32_t*                  chunks;     // [n_chunks][3]: scan, first record, candidate -- 64 records of one query each
  uint32_t                         n_chunks;
  uint32_t*                        ws;         // per wave: cov_prev[range_cap] | cov_cur[range_cap + 1] | table[table_cap]
  uint64_t                         ws_stride;  // dwords
  uint32_t                         range_cap, table_cap;
  uint32_t*                        counter;
  uint32_t*                        locus_counts;  // [n_loci][4]: pile reads, code dwords, mask dwords, -
  unsigned long long*              locus_base;    // [n_loci + 1][4]: exclusive scans of the same
  uint32_t *                       codes, *nmask, *read_len, *pile_read, *locus_read_begin;
  unsigned long long *             read_code_off, *read_mask_off;
};

struct ReadClass {
  const ReadClassParams& P;
  WV_DEV explicit ReadClass(const ReadClassParams& p) : P(p) {}

  // ---- record fields (htsapi/bam_record.hpp) ----
  WV_DEV static bool paired(const manta_bam_read_t& r) { return r.flag & 0x1; }
  WV_DEV static bool unmapped(const manta_bam_read_t& r) { return r.flag & 0x4; }
  WV_DEV static bool mateUnmapped(const manta_bam_read_t& r) { return r.flag & 0x8; }
  WV_DEV static bool fwd(const manta_bam_read_t& r) { return !(r.flag & 0x10); }
  WV_DEV static bool mateFwd(const manta_bam_read_t& r) { return !(r.flag & 0x20); }
  WV_DEV static bool saSplit(const manta_bam_read_t& r) { return r.tags & MANTA_READ_TAG_SA; }
  WV_DEV static bool nonStrictSupplement(const manta_bam_read_t& r) { return (r.flag & 0x800) || ((r.flag & 0x100) && saSplit(r)); }  // :139-144
  WV_DEV static bool filteredCore(const manta_bam_read_t& r)  // manta/ReadFilter.cpp:32-50
  {
    if (r.flag & (0x200 | 0x400)) return true;
    if ((r.flag & 0x800) && !saSplit(r)) return true;
    if ((r.flag & 0x100) && !saSplit(r)) return true;
    return false;
  }

  // ---- alignment path (blt_util/align_path.hpp:89-138; segment type = 1 + BAM operation) ----
  WV_DEV static bool opReadLen(const unsigned op) { return op == 0 || op == 1 || op == 4 || op == 7 || op == 8; }
  WV_DEV static bool opRefLen(const unsigned op) { return op == 0 || op == 2 || op == 3 || op == 7 || op == 8; }
  WV_DEV static bool opMatch(const unsigned op) { return op == 0 || op == 7 || op == 8; }
  WV_DEV static bool opIndel(const unsigned op) { return op == 1 || op == 2; }
  struct PathView {
    const uint32_t* c;
    unsigned        n;
    bool            whole;   // the faked mate alignment: one match of `len`
    unsigned        len;
    WV_DEV unsigned size() const { return whole ? 1u : n; }
    WV_DEV unsigned op(const unsigned i) const { return whole ? 0u : (c[i] & 15u); }
    WV_DEV unsigned length(const unsigned i) const { return whole ? len : (c[i] >> 4); }
  };
  WV_DEV PathView pathOf(const manta_bam_read_t& r) const { return PathView{P.cigars + r.cigar_off, r.n_cigar, false, 0}; }
  WV_DEV PathView matePathOf(const manta_bam_read_t& r) const  // SimpleAlignment_bam_util.cpp:43-61
  {
    if (r.tags & MANTA_READ_TAG_MC) return PathView{P.cigars + r.mate_cigar_off, r.n_mate_cigar, false, 0};
    return PathView{nullptr, 0, true, r.read_len};
  }
  WV_DEV static unsigned refLength(const PathView& p)  // align_path.cpp:126-134
  {
    unsigned v = 0;
    for (unsigned i = 0; i < p.size(); ++i)
      if (opRefLen(p.op(i))) v += p.length(i);
    return v;
  }
  WV_DEV static unsigned softClipLeft(const PathView& p)  // :169-182
  {
    unsigned v = 0;
    for (unsigned i = 0; i < p.size(); ++i) {
      if (p.op(i) == 5) continue;
      if (p.op(i) != 4) break;
      v += p.length(i);
    }
    return v;
  }
  WV_DEV static unsigned softClipRight(const PathView& p)  // :184-198
  {
    unsigned v = 0;
    for (unsigned i = p.size(); i-- > 0;) {
      if (p.op(i) == 5) continue;
      if (p.op(i) != 4) break;
      v += p.length(i);
    }
    return v;
  }
  WV_DEV static void matchEdges(const PathView& p, unsigned& first, unsigned& last)  // :542-556
  {
    first = last = p.size();
    bool seen    = false;
    for (unsigned i = 0; i < p.size(); ++i)
      if (opMatch(p.op(i))) {
        if (!seen) first = i;
        seen = true;
        last = i;
      }
  }

  WV_DEV unsigned codeAt(const manta_bam_read_t& r, const int i) const  // htsapi/bam_seq.hpp:159-164
  {
    if (i < 0 || i >= int(r.read_len)) return 15u;
    return (P.seqs[r.seq_off + unsigned(i >> 1)] >> (4u * (1u - (unsigned(i) & 1u)))) & 0xfu;
  }
  /// get_bam_seq_char (:41-59) as one of '=', A, C, G, T, N
  WV_DEV static char charOfCode(const unsigned c) { return c == 0 ? '=' : c == 1 ? 'A' : c == 2 ? 'C' : c == 4 ? 'G' : c == 8 ? 'T' : 'N'; }
  WV_DEV static char refAt(const manta_read_scan_t& sc, const uint8_t* text, const int pos)  // reference_contig_segment.hpp:42-46
  {
    return (pos < sc.ref_begin || pos >= sc.ref_begin + int(sc.ref_len)) ? 'N' : char(text[pos - sc.ref_begin]);
  }
  WV_DEV static bool baseMatch(const char a, const char b) { return a == 'N' || b == 'N' || a == b; }  // SVLocusScannerSemiAligned.cpp:45-49

  WV_DEV bool overlappingPair(const manta_bam_read_t& r, const PathView& path) const  // bam_record_util.cpp:84-108
  {
    if (!paired(r) || unmapped(r) || mateUnmapped(r)) return false;
    if (r.tid != r.mate_tid) return false;
    if (fwd(r) == mateFwd(r)) return false;
    int posDiff = (r.pos + 1) - (r.mate_pos + 1);
    if (!fwd(r)) posDiff = -posDiff;
    if (posDiff > int(r.read_len)) return false;
    if (fwd(r)) return (r.pos + int(refLength(path)) - (r.mate_pos + 1)) >= 0;
    return (r.pos - (r.mate_pos + int(refLength(matePathOf(r))))) <= 0;
  }
  WV_DEV bool adapterPair(const manta_bam_read_t& r, const PathView& aln) const  // :54-82
  {
    if (saSplit(r)) return false;
    if (r.tags & MANTA_READ_TAG_MC) {
      const PathView mate = matePathOf(r);
      if (fwd(r)) return (unsigned(r.pos) + refLength(aln) + softClipRight(aln)) > (unsigned(r.mate_pos) + refLength(mate) + softClipRight(mate));
      return (unsigned(r.pos) - softClipLeft(aln)) < (unsigned(r.mate_pos) - softClipLeft(mate));
    }
    return (fwd(r) ? softClipRight(aln) : softClipLeft(aln)) > 0;
  }

  /// getSVBreakendCandidateSemiAligned (SVLocusScannerSemiAligned.cpp:216-316 with :52-168).  The soft clips outside the
  /// outermost match segments count as matches (matchifyEdgeSoftClip, blt_util/SimpleAlignment.cpp:33-75): walked in place.
  WV_DEV void semiAligned(const manta_bam_read_t& r, const manta_read_scan_t& sc, unsigned& leading, unsigned& trailing) const
  {
    leading = trailing    = 0;
    const PathView aln    = pathOf(r);
    const bool     overlap = overlappingPair(r, aln);
    if (overlap && (!P.opt.use_overlap_pair_evidence || adapterPair(r, aln))) return;
    const uint8_t* refText = P.refs + sc.ref_off;
    const unsigned readSize = r.read_len;
    unsigned       first, last;
    matchEdges(aln, first, last);
    // the matchified alignment starts before pos by the leading soft clips, and its reference length adds the edge clips
    int      mpos = r.pos;
    unsigned mRefLen = 0;
    for (unsigned i = 0; i < aln.size(); ++i) {
      const unsigned op = aln.op(i), len = aln.length(i);
      const bool     edge = (i < first || i > last) && op == 4;
      if (edge && i < first) mpos -= int(len);
      if (edge || opRefLen(op)) mRefLen += len;
    }
    auto asMatch = [&](const unsigned i) { return opMatch(aln.op(i)) || ((i < first || i > last) && aln.op(i) == 4); };
    unsigned leadTmp = 0, trailTmp = 0;
    {  // leadingEdgePoorAlignmentLength (:52-98)
      int      readIndex = 0, refIndex = mpos;
      unsigned run  = 0;
      bool     done = false;
      for (unsigned i = 0; i < aln.size() && !done; ++i) {
        const unsigned op = aln.op(i), len = aln.length(i);
        const bool     m  = asMatch(i);
        if (m) {
          for (unsigned s = 0; s < len; ++s) {
            if (baseMatch(charOfCode(codeAt(r, readIndex + int(s))), refAt(sc, refText, refIndex + int(s)))) {
              if (++run >= 5) {
                leadTmp = unsigned((readIndex + int(s)) - int(run - 1));
                done    = true;
                break;
              }
            } else {
              run = 0;
            }
          }
        } else if (opIndel(op)) {
          run = 0;
        }
        if (m || opReadLen(op)) readIndex += int(len);
        if (m || opRefLen(op)) refIndex += int(len);
      }
      if (!done) leadTmp = unsigned(readIndex);
    }
    {  // trailingEdgePoorAlignmentLength (:101-151)
      int      readIndex = int(readSize) - 1, refIndex = mpos + int(mRefLen) - 1;
      unsigned run  = 0;
      bool     done = false;
      for (unsigned i = aln.size(); i-- > 0 && !done;) {
        const unsigned op = aln.op(i), len = aln.length(i);
        const bool     m  = asMatch(i);
        if (m) {
          for (unsigned s = 0; s < len; ++s) {
            if (baseMatch(charOfCode(codeAt(r, readIndex - int(s))), refAt(sc, refText, refIndex - int(s)))) {
              if (++run >= 5) {
                trailTmp = unsigned((int(readSize) - (readIndex - int(s))) - int(run));
                done     = true;
                break;
              }
            } else {
              run = 0;
            }
          }
        } else if (opIndel(op)) {
          run = 0;
        }
        if (m || opReadLen(op)) readIndex -= int(len);
        if (m || opRefLen(op)) refIndex -= int(len);
      }
      if (!done) trailTmp = unsigned(int(readSize) - (readIndex + 1));
    }
    if (leadTmp + trailTmp >= readSize) return;  // :259
    const uint8_t* q = P.quals + r.qual_off;
    // (count / length >= 0.75f in float is the same as 4 count >= 3 length for lengths below 2^16: the quotient is never
    // within half an ulp of 0.75 unless it is 0.75)
    if (leadTmp != 0 && (!overlap || saSplit(r) || fwd(r))) {  // :267-285
      unsigned hq = 0;
      for (unsigned p = 0; p < leadTmp && p < readSize; ++p) hq += (q[p] >= 20) ? 1u : 0u;
      if (4ull * hq >= 3ull * leadTmp) leading = leadTmp;
    }
    if (trailTmp != 0 && (!overlap || saSplit(r) || !fwd(r))) {  // :287-305
      unsigned hq = 0;
      for (unsigned p = 0; p < trailTmp && p < readSize; ++p) hq += (q[readSize - p - 1] >= 20) ? 1u : 0u;
      if (4ull * hq >= 3ull * trailTmp) trailing = trailTmp;
    }
  }

  // ---- read key (:110-111): name, read number, alignment file ----
  WV_DEV bool nameEq(const manta_bam_read_t& a, const manta_bam_read_t& b) const
  {
    if (a.qname_len != b.qname_len) return false;
    const uint8_t *x = P.names + a.qname_off, *y = P.names + b.qname_off;
    for (unsigned i = 0; i < a.qname_len; ++i)
      if (x[i] != y[i]) return false;
    return true;
  }
  WV_DEV bool keyEq(const unsigned i, const unsigned j) const
  {
    if ((P.tmp[i] >> 8) != (P.tmp[j] >> 8)) return false;
    return nameEq(P.reads[i], P.reads[j]);
  }
  WV_DEV uint32_t keyHash(const unsigned i) const
  {
    const manta_bam_read_t& r = P.reads[i];
    uint32_t                h = 0x811C9DC5u ^ (P.tmp[i] >> 8);
    const uint8_t*          x = P.names + r.qname_off;
    for (unsigned k = 0; k < r.qname_len; ++k) h = (h ^ x[k]) * 16777619u;
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    h ^= h >> 12;
    return h;
  }

  WV_HD static void searchRange(const manta_read_scan_t& sc, int& sb, int& se)  // :285-303
  {
    const unsigned size = unsigned(sc.bp_end - sc.bp_begin > 0 ? sc.bp_end - sc.bp_begin : 0);
    if (size >= 400) {
      sb = sc.bp_begin;
      se = sc.bp_end;
    } else {
      const unsigned wobble = (400 - size) / 2;
      sb                    = int(unsigned(sc.bp_begin) - wobble);
      se                    = int(unsigned(sc.bp_end) + wobble);
    }
  }

  /// The tests of the scan loop's body that read nothing but the record, its query and the options (:443-509 and the shadow
  /// finder's predicates, ShadowReadFinder.cpp:33-100): run record-parallel over the whole batch by read_test_kernel, before the
  /// candidate-serial passes, which only combine the bits with the state they carry.
  WV_DEV unsigned recordTests(const manta_read_locus_t& loc, const manta_read_scan_t& sc, const manta_bam_read_t& r) const
  {
    int sb, se;
    searchRange(sc, sb, se);
    const unsigned minIndel  = P.opt.min_candidate_variant_size / 2;  // :317
    const bool     rightOpen = (sc.bp_state != 2), leftOpen = (sc.bp_state != 1);  // :320-329
    unsigned       pre       = 0;
    const PathView path = pathOf(r);
    if (loc.search_remote) {  // :443-470, RemoteMateReadUtil.cpp:29-55
      bool cand = paired(r) && !unmapped(r) && !mateUnmapped(r) && r.mapq >= P.opt.min_mapq && r.tid >= 0 && r.mate_tid >= 0;
      if (cand && r.tid == r.mate_tid) {
        const int dd = r.pos - r.mate_pos;
        cand         = (dd < 0 ? -dd : dd) >= 10000;
      }
      if (cand) {  // matchifyEdgeSoftClipRefRange (SimpleAlignment.cpp:77-108) against the flanks (:302-303)
        int      rb = r.pos, re = r.pos;
        unsigned first, last;
        matchEdges(path, first, last);
        for (unsigned k = 0; k < path.size(); ++k) {
          const unsigned op = path.op(k), len = path.length(k);
          if (k < first) {
            if (opReadLen(op)) rb -= int(len);
          } else if (k > last) {
            if (opReadLen(op)) re += int(len);
          } else if (opRefLen(op)) {
            re += int(len);
          }
        }
        const bool hitsLeft = (re > sb) && (rb < sc.bp_begin), hitsRight = (re > sc.bp_end) && (rb < se);
        const bool leftMate = leftOpen && !hitsLeft, rightMate = rightOpen && !hitsRight;
        if (!((!leftMate && !fwd(r)) || (!rightMate && fwd(r)))) pre |= PRE_REMOTE;
      }
    }
    if (!unmapped(r)) {
      for (unsigned k = 0; k < path.size(); ++k)  // :473-483: the FIRST indel segment decides
        if (opIndel(path.op(k))) {
          if (path.length(k) >= minIndel) pre |= PRE_INDEL;
          break;
        }
      unsigned lead, trail;  // :486-509
      semiAligned(r, sc, lead, trail);
      if ((rightOpen && trail >= 4) || (leftOpen && lead >= 4)) pre |= PRE_SEMI;
    }
    // shadow finder predicates (ShadowReadFinder.cpp:33-100)
    if (paired(r) && !unmapped(r) && mateUnmapped(r) && !(!leftOpen && !fwd(r)) && !(!rightOpen && fwd(r)) &&
        r.mapq >= P.opt.min_singleton_mapq_candidates)
      pre |= PRE_ANCHOR;
    if (paired(r) && unmapped(r) && !mateUnmapped(r)) {
      unsigned       sum = 0;
      const uint8_t* q   = P.quals + r.qual_off;
      for (unsigned p = 0; p < r.read_len; ++p) sum += q[p];
      if ((r.read_len ? sum / r.read_len : 0u) >= 25u) pre |= PRE_GOOD;  // get_avg_quality (bam_record_util.cpp:110-122)
    }
    return pre;
  }

  /// passes A-D over one region query.  Returns false if the search range does not fit the workspace.
  WV_DEV bool scanQuery(const manta_read_locus_t& loc, const manta_read_scan_t& sc, uint32_t* covPrev, uint32_t* covCur)
  {
    const unsigned lane = unsigned(wv::lane());
    int            sb, se;
    searchRange(sc, sb, se);
    const unsigned range = unsigned(se > sb ? se - sb : 0);
    if (range + 1 > P.range_cap) return false;
    const bool depthOn = loc.is_max_depth != 0;
    if (sc.first_of_breakend)
      for (unsigned x = lane; x < range; x += 64) covPrev[x] = 0;
    for (unsigned x = lane; x <= range; x += 64) covCur[x] = 0;
    // end of the scan: the first record at or behind the search end (:397-398)
    unsigned firstBeyond = sc.read_end;
    for (unsigned base = sc.read_begin; base < sc.read_end; base += 64) {
      const unsigned i = base + lane;
      const uint64_t m = wv::ballot(i < sc.read_end && P.reads[i < sc.read_end ? i : sc.read_begin].pos >= se);
      if (m) {
        firstBeyond = base + unsigned(wv::ctz(m));
        break;
      }
    }
    wv::sync();
    const unsigned keyBits = (sc.bam_index << 1);
    // A
    for (unsigned base = sc.read_begin; base < sc.read_end; base += 64) {
      const unsigned i = base + lane;
      if (i >= sc.read_end) continue;
      P.decision[i]   = 0;
      P.pile_index[i] = RC_EMPTY;
      unsigned t      = 0;
      if (i < firstBeyond) {
        const manta_bam_read_t r = P.reads[i];
        t                        = RC_VALID | ((keyBits | ((r.flag & 0x80) ? 1u : 0u)) << 8);
        const bool filt          = filteredCore(r);
        if (!filt && depthOn && !sc.is_tumor && !unmapped(r)) {  // addReadToDepthEst (:85-100): read_size bases from pos, clipped
          const long long b = (r.pos > sb) ? r.pos : sb, e = ((long long)r.pos + r.read_len < se) ? (long long)r.pos + r.read_len : se;
          if (b < e) {
            wv::atomic_add(&covCur[unsigned(b - sb)], 1u);
            wv::atomic_add(&covCur[unsigned(e - sb)], 0xffffffffu);
          }
        }
      }
      P.tmp[i] = t;
    }
    wv::sync();
    // B
    if (depthOn) {
      unsigned carry = 0;
      for (unsigned x0 = 0; x0 <= range; x0 += 64) {
        const unsigned x = x0 + lane;
        unsigned       v = (x <= range) ? covCur[x] : 0u;
        for (int off = 1; off < 64; off <<= 1) {
          const unsigned o = wv::shfl(v, int(lane) - off);
          if (int(lane) >= off) v += o;
        }
        v += carry;
        if (x <= range) covCur[x] = v;
        carry = wv::shfl(v, 63);
      }
      wv::sync();
    }
    // C
    for (unsigned base = sc.read_begin; base < firstBeyond; base += 64) {
      const unsigned i = base + lane;
      if (i >= firstBeyond) continue;
      const manta_bam_read_t r = P.reads[i];
      unsigned               t = P.tmp[i];
      unsigned               d = 0;
      if (filteredCore(r) || nonStrictSupplement(r)) continue;  // :402, :416
      if (depthOn) {  // :418-427
        const int off = r.pos - sb;
        if (off >= 0) {
          unsigned depth = covPrev[off] + covCur[off];
          // what the records behind this one add at its position has not happened yet for the reference
          if (!sc.is_tumor)
            for (unsigned j = i + 1; j < firstBeyond; ++j) {
              const manta_bam_read_t& o = P.reads[j];
              if (o.pos > r.pos) break;
              if (!filteredCore(o) && !unmapped(o) && o.pos <= r.pos && (long long)o.pos + o.read_len > r.pos) depth--;
            }
          if (float(depth) > loc.max_local_depth_remote) t |= RC_TRIP;
          if (float(depth) > loc.max_depth) {
            P.tmp[i]      = t;
            P.decision[i] = MANTA_READ_DEPTH_FILTERED;
            continue;
          }
        }
      }
      t |= RC_REACH;
      const unsigned pre = P.pre[i];  // read_test_kernel: what depends on the record alone
      if (pre & PRE_REMOTE) d |= MANTA_READ_REMOTE_MATE;
      if (pre & PRE_INDEL) d |= MANTA_READ_INDEL;
      if (pre & PRE_SEMI) d |= MANTA_READ_SEMI_ALIGNED;
      if (pre & PRE_ANCHOR) t |= RC_ANCHOR;
      if (pre & PRE_GOOD) t |= RC_GOOD;
      P.tmp[i]      = t;
      P.decision[i] = uint8_t(d);
    }
    wv::sync();
    // D
    unsigned prevReach = RC_EMPTY;
    for (unsigned base = sc.read_begin; base < firstBeyond; base += 64) {
      const unsigned i = base + lane;
      const unsigned t = (i < firstBeyond) ? P.tmp[i] : 0u;
      const uint64_t m = wv::ballot((t & RC_REACH) != 0);
      const uint64_t b = m & ((uint64_t(1) << lane) - 1);
      const unsigned p = b ? (base + 63u - unsigned(wv::clz(b))) : prevReach;
      if (t & RC_REACH) {
        unsigned d = P.decision[i], t2 = t;
        if ((t & RC_GOOD) && p != RC_EMPTY && (P.tmp[p] & RC_ANCHOR) && nameEq(P.reads[i], P.reads[p])) {
          d |= MANTA_READ_SHADOW;
          if (mateFwd(P.reads[i])) t2 |= RC_MATEREV;
        }
        if (d & (MANTA_READ_INDEL | MANTA_READ_SEMI_ALIGNED | MANTA_READ_SHADOW)) t2 |= RC_KEEP;
        P.decision[i] = uint8_t(d);
        P.tmp[i]      = t2;
      }
      if (m) prevReach = base + 63u - unsigned(wv::clz(m));
      wv::sync();  // (tmp[p] of the next chunk's carry is this chunk's)
    }
    // this query's coverage joins the breakend's
    if (depthOn && !sc.is_tumor)
      for (unsigned x = lane; x < range; x += 64) covPrev[x] += covCur[x];
    wv::sync();
    return true;
  }

  WV_DEV void runLocus(const unsigned l, uint32_t* wsBase)
  {
    const unsigned            lane = unsigned(wv::lane());
    const manta_read_locus_t  loc  = P.loci[l];
    uint32_t*                 covPrev = wsBase;
    uint32_t*                 covCur  = wsBase + P.range_cap;
    uint32_t*                 table   = wsBase + 2 * size_t(P.range_cap) + 1;
    const unsigned            maxReads = P.opt.max_reads ? P.opt.max_reads : 1000u;  // :342
    manta_read_locus_result_t res;
    res.status          = MANTA_OK;
    res.n_pile_reads    = 0;
    res.retrieve_remote = 1;
    res.reserved        = 0;
    bool     ok        = true;
    unsigned nRecords  = 0;
    for (unsigned s = loc.scan_begin; s < loc.scan_end; ++s) {
      const manta_read_scan_t sc = P.scans[s];
      nRecords += sc.read_end - sc.read_begin;
      if (ok && !scanQuery(loc, sc, covPrev, covCur)) ok = false;
    }
    if (!ok || 2ull * nRecords > P.table_cap) {
      res.status = MANTA_E_DEVICE_FAULT;  // workspace sized wrongly by the host
      if (lane == 0) {
        P.results[l] = res;
        for (int k = 0; k < 4; ++k) P.locus_counts[4 * size_t(l) + k] = 0;
      }
      return;
    }
    // E: the smallest record index of every read key among the keepers
    for (unsigned x = lane; x < P.table_cap; x += 64) table[x] = RC_EMPTY;
    wv::sync();
    const unsigned mask = P.table_cap - 1;
    for (unsigned s = loc.scan_begin; s < loc.scan_end; ++s) {
      const manta_read_scan_t sc = P.scans[s];
      for (unsigned base = sc.read_begin; base < sc.read_end; base += 64) {
        const unsigned i = base + lane;
        if (i >= sc.read_end || !(P.tmp[i] & RC_KEEP)) continue;
        unsigned h = keyHash(i) & mask;
        while (true) {
          unsigned e = wv::atomic_load(&table[h]);
          if (e == RC_EMPTY) {
            e = wv::atomic_cas(&table[h], RC_EMPTY, i);
            if (e == RC_EMPTY) break;
          }
          if (keyEq(e, i)) {
            wv::atomic_min(&table[h], i);
            break;
          }
          h = (h + 1) & mask;
        }
      }
    }
    wv::sync();
    // F: pile positions in record order; the cap cuts the rest
    unsigned count = 0, codeDw = 0, maskDw = 0;
    bool     trip  = false;
    for (unsigned s = loc.scan_begin; s < loc.scan_end; ++s) {
      const manta_read_scan_t sc = P.scans[s];
      for (unsigned base = sc.read_begin; base < sc.read_end; base += 64) {
        const unsigned i     = base + lane;
        const unsigned t     = (i < sc.read_end) ? P.tmp[i] : 0u;
        bool           first = false;
        if (t & RC_KEEP) {
          unsigned h = keyHash(i) & mask;
          while (true) {
            const unsigned e = wv::atomic_load(&table[h]);
            if (e == RC_EMPTY) break;  // (cannot happen: every keeper was entered)
            if (keyEq(e, i)) {
              first = (e == i);
              break;
            }
            h = (h + 1) & mask;
          }
        }
        const uint64_t mf      = wv::ballot(first);
        const unsigned before  = count + unsigned(wv::popc(mf & ((uint64_t(1) << lane) - 1)));
        const bool     looked  = (t & RC_VALID) && before < maxReads;  // the reference got as far as this record (:388-393)
        const bool     ins     = first && looked;
        if (i < sc.read_end) {
          if (!looked) {
            P.decision[i] = 0;
          } else if (t & RC_KEEP) {
            unsigned d = P.decision[i];
            if (ins) {
              const bool rev = (sc.is_locus_reversed != 0) != ((t & RC_MATEREV) != 0);  // :557-563
              d |= MANTA_READ_IN_PILE | (rev ? MANTA_READ_REVERSED : 0u);
              P.pile_index[i] = before;
            } else {
              d |= MANTA_READ_DUPLICATE_KEY;
            }
            P.decision[i] = uint8_t(d);
          }
        }
        if (wv::any(looked && (t & RC_TRIP))) trip = true;
        const unsigned len = ins ? P.reads[i].read_len : 0u;
        unsigned       c = ins ? (len + 15) / 16 : 0u, mk = ins ? (len + 31) / 32 : 0u;
        for (int off = 1; off < 64; off <<= 1) {
          c += wv::shfl(c, wv::lane() ^ off);
          mk += wv::shfl(mk, wv::lane() ^ off);
        }
        codeDw += c;
        maskDw += mk;
        count += unsigned(wv::popc(wv::ballot(ins)));
      }
    }
    res.n_pile_reads    = count;
    res.retrieve_remote = trip ? 0u : 1u;
    if (lane == 0) {
      P.results[l]                      = res;
      P.locus_counts[4 * size_t(l) + 0] = count;
      P.locus_counts[4 * size_t(l) + 1] = codeDw;
      P.locus_counts[4 * size_t(l) + 2] = maskDw;
      P.locus_counts[4 * size_t(l) + 3] = 0;
    }
  }

  /// pile reads of one candidate: their rows of the packed layout (length, record, code / mask offsets) from the candidate's bases
  WV_DEV void packOffsets(const unsigned l)
  {
    const unsigned           lane = unsigned(wv::lane());
    const manta_read_locus_t loc  = P.loci[l];
    const unsigned long long rb = P.locus_base[4 * size_t(l) + 0], cb = P.locus_base[4 * size_t(l) + 1], mb = P.locus_base[4 * size_t(l) + 2];
    unsigned                 cRun = 0, mRun = 0;
    for (unsigned s = loc.scan_begin; s < loc.scan_end; ++s) {
      const manta_read_scan_t sc = P.scans[s];
      for (unsigned base = sc.read_begin; base < sc.read_end; base += 64) {
        const unsigned i   = base + lane;
        const bool     ins = (i < sc.read_end) && (P.decision[i] & MANTA_READ_IN_PILE);
        const unsigned len = ins ? P.reads[i].read_len : 0u;
        const unsigned c = (len + 15) / 16, mk = (len + 31) / 32;
        unsigned       ci = c, mi = mk;
        for (int off = 1; off < 64; off <<= 1) {
          const unsigned oc = wv::shfl(ci, int(lane) - off), om = wv::shfl(mi, int(lane) - off);
          if (int(lane) >= off) {
            ci += oc;
            mi += om;
          }
        }
        if (ins) {
          const unsigned long long r = rb + P.pile_index[i];
          P.read_len[r]              = len;
          P.pile_read[r]             = i;
          P.read_code_off[r]         = cb + cRun + ci - c;
          P.read_mask_off[r]         = mb + mRun + mi - mk;
          P.tmp[r]                   = l;  // (the per-record words are done with: pile read -> candidate, for packBases)
        }
        cRun += wv::shfl(ci, 63);
        mRun += wv::shfl(mi, 63);
      }
    }
  }

  /// insertAssemblyRead (:121-135) for eight pile reads [8 g, 8 g + 8) of the batch: text of the 4-bit codes, Q mask, reverse
  /// complement -- eight lanes per read, 32 output bases per lane step
  WV_DEV void packBases(const unsigned g, const unsigned long long nPileAll)
  {
    const unsigned lane = unsigned(wv::lane());
    bool           equals = false;
    {
      const unsigned long long r = 8ull * g + (lane >> 3);
      if (r >= nPileAll) return;
      const unsigned           i   = P.pile_read[r];
      const manta_bam_read_t   rec = P.reads[i];
      const bool               rev = (P.decision[i] & MANTA_READ_REVERSED) != 0;
      const unsigned           len = rec.read_len;
      const uint8_t*           q   = P.quals + rec.qual_off;
      uint32_t*                co  = P.codes + P.read_code_off[r];
      uint32_t*                mo  = P.nmask + P.read_mask_off[r];
      for (unsigned w = (lane & 7); w < (len + 31) / 32; w += 8) {
        uint32_t c0 = 0, c1 = 0, nm = 0;
        for (unsigned b = 0; b < 32; ++b) {
          const unsigned o = w * 32 + b;
          if (o >= len) break;
          const unsigned src = rev ? (len - 1 - o) : o;
          const unsigned c4  = codeAt(rec, int(src));
          unsigned       c2  = (c4 == 1) ? 0u : (c4 == 2) ? 1u : (c4 == 4) ? 2u : (c4 == 8) ? 3u : 4u;
          if (c4 == 0) equals = true;
          if (q[src] < P.opt.min_qval) c2 = 4;
          if (c2 == 4) {
            nm |= 1u << b;
          } else {
            if (rev) c2 = 3 - c2;
            if (b < 16)
              c0 |= c2 << (30 - 2 * b);
            else
              c1 |= c2 << (30 - 2 * (b - 16));
          }
        }
        mo[w]     = nm;
        co[2 * w] = c0;
        if (2 * w + 1 < (len + 15) / 16) co[2 * w + 1] = c1;
      }
      if (equals) P.results[P.tmp[r]].status = MANTA_E_UNSUPPORTED;
    }
  }
};

/// record-parallel: the per-record tests of every record of the batch (a wave per 64 records of one query)
#if !MANTA_TU_DEFINES(MANTA_TU_GLUE)
WV_KERNEL void read_test_kernel(const ReadClassParams P);
#else
WV_KERNEL void read_test_kernel(const ReadClassParams P)
{
  ReadClass rc(P);
  while (true) {
    unsigned c = 0;
    if (wv::lane() == 0) c = wv::atomic_add(&P.counter[2], 1u);
    c = wv::first(c);
    if (c >= P.n_chunks) break;
    const manta_read_scan_t sc = P.scans[P.chunks[3 * size_t(c)]];
    const unsigned          i  = P.chunks[3 * size_t(c) + 1] + unsigned(wv::lane());
    if (i < sc.read_end) P.pre[i] = uint8_t(rc.recordTests(P.loci[P.chunks[3 * size_t(c) + 2]], sc, P.reads[i]));
    wv::sync();
  }
}
#endif

/// persistent waves: one candidate at a time
#if !MANTA_TU_DEFINES(MANTA_TU_GLUE)
WV_KERNEL void read_class_kernel(const ReadClassParams P);
#else
WV_KERNEL void read_class_kernel(const ReadClassParams P)
{
  uint32_t* wsBase = P.ws + uint64_t(wv::block()) * P.ws_stride;
  ReadClass rc(P);
  while (true) {
    unsigned l = 0;
    if (wv::lane() == 0) l = wv::atomic_add(&P.counter[0], 1u);
    l = wv::first(l);
    if (l >= P.n_loci) break;
    rc.runLocus(l, wsBase);
    wv::sync();
  }
}
#endif

/// one wave: exclusive scans of the per-candidate totals (pile reads, code dwords, mask dwords)
#if !MANTA_TU_DEFINES(MANTA_TU_GLUE)
WV_KERNEL void read_pile_offsets_kernel(const ReadClassParams P);
#else
WV_KERNEL void read_pile_offsets_kernel(const ReadClassParams P)
{
  if (wv::block() != 0) return;
  const unsigned     lane = unsigned(wv::lane());
  unsigned long long run[3] = {0, 0, 0};
  for (unsigned l0 = 0; l0 < P.n_loci; l0 += 64) {
    const unsigned l = l0 + lane;
    for (int k = 0; k < 3; ++k) {
      const unsigned long long mine = (l < P.n_loci) ? P.locus_counts[4 * size_t(l) + k] : 0ull;
      unsigned long long       v    = mine;
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = wv::shfl(uint64_t(v), int(lane) - off);
        if (int(lane) >= off) v += o;
      }
      if (l < P.n_loci) P.locus_base[4 * size_t(l) + k] = run[k] + v - mine;
      run[k] += wv::shfl(uint64_t(v), 63);
    }
    if (l < P.n_loci) P.locus_read_begin[l] = uint32_t(P.locus_base[4 * size_t(l)]);
  }
  if (lane == 0) {
    for (int k = 0; k < 3; ++k) P.locus_base[4 * size_t(P.n_loci) + k] = run[k];
    P.locus_read_begin[P.n_loci] = uint32_t(run[0]);
    P.read_code_off[run[0]]      = run[1];
    P.read_mask_off[run[0]]      = run[2];
  }
}
#endif

#if !MANTA_TU_DEFINES(MANTA_TU_GLUE)
WV_KERNEL void read_pile_pack_kernel(const ReadClassParams P);
#else
WV_KERNEL void read_pile_pack_kernel(const ReadClassParams P)
{
  ReadClass rc(P);
  while (true) {
    unsigned l = 0;
    if (wv::lane() == 0) l = wv::atomic_add(&P.counter[1], 1u);
    l = wv::first(l);
    if (l >= P.n_loci) break;
    rc.packOffsets(l);
    wv::sync();
  }
}
#endif

/// pile-read-parallel: eight reads per wave step
#if !MANTA_TU_DEFINES(MANTA_TU_GLUE)
WV_KERNEL void read_pile_bases_kernel(const ReadClassParams P);
#else
WV_KERNEL void read_pile_bases_kernel(const ReadClassParams P)
{
  ReadClass                rc(P);
  const unsigned long long nPileAll = P.locus_base[4 * size_t(P.n_loci)];
  while (true) {
    unsigned g = 0;
    if (wv::lane() == 0) g = wv::atomic_add(&P.counter[3], 1u);
    g = wv::first(g);
    if (8ull * g >= nPileAll) break;
    rc.packBases(g, nPileAll);
    wv::sync();
  }
}
#endif

}  // namespace manta_dev
