// TEST INFRASTRUCTURE ONLY (see oracle/FORMAT.md): CPU restatement of the read gathering of one assembly,
//   SVCandidateAssembler::getBreakendReads, manta/SVCandidateAssembler.cpp:271-659 (paths relative to /root/reference/src/c++/lib),
// record by record and in the reference's order, over decoded BAM records in the layout of include/manta_amd.h
// (manta_bam_read_t ...).  Pinned on the unmodified reference (oracle/_ref/libmanta_ref_bam.so: the real SVCandidateAssembler.cpp
// over real BAM files through htslib) by tests/test_read_class.py: the demo BAMs of src/demo/data and synthetic BAM files.
// The product (manta_amd/csrc/read_class_kernels.hpp) never links or calls this file.
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../include/manta_amd.h"

namespace {

// blt_util/align_path.hpp:36-50 (align_t) through htsapi/align_path_bam_util.cpp:30-39: type = 1 + BAM op
enum { NONE = 0, MATCH, INSERT, DELETE, SKIP, SOFT_CLIP, HARD_CLIP, PAD, SEQ_MATCH, SEQ_MISMATCH };
struct Seg {
  unsigned type, length;
};
typedef std::vector<Seg> Path;

bool isReadLength(unsigned t) { return t == MATCH || t == INSERT || t == SOFT_CLIP || t == SEQ_MATCH || t == SEQ_MISMATCH; }  // align_path.hpp:89-101
bool isRefLength(unsigned t) { return t == MATCH || t == DELETE || t == SKIP || t == SEQ_MATCH || t == SEQ_MISMATCH; }        // :103-115
bool isAlignMatch(unsigned t) { return t == MATCH || t == SEQ_MATCH || t == SEQ_MISMATCH; }                                   // :117-127
bool isIndel(unsigned t) { return t == INSERT || t == DELETE; }                                                                // :129-138

Path toPath(const uint32_t* cigar, unsigned n)
{
  Path p(n);
  for (unsigned i = 0; i < n; ++i) {
    p[i].length = cigar[i] >> 4;
    p[i].type   = 1 + (cigar[i] & 15);
  }
  return p;
}
unsigned refLength(const Path& p)  // align_path.cpp:126-134
{
  unsigned v = 0;
  for (const Seg& s : p)
    if (isRefLength(s.type)) v += s.length;
  return v;
}
unsigned softClipLeft(const Path& p)  // :169-182
{
  unsigned v = 0;
  for (const Seg& s : p) {
    if (s.type == HARD_CLIP) continue;
    if (s.type != SOFT_CLIP) break;
    v += s.length;
  }
  return v;
}
unsigned softClipRight(const Path& p)  // :184-198
{
  unsigned v = 0;
  for (size_t i = p.size(); i-- > 0;) {
    if (p[i].type == HARD_CLIP) continue;
    if (p[i].type != SOFT_CLIP) break;
    v += p[i].length;
  }
  return v;
}
void matchEdgeSegments(const Path& p, unsigned& first, unsigned& last)  // :542-556
{
  first = last  = unsigned(p.size());
  bool isFirst = false;
  for (unsigned i = 0; i < p.size(); ++i)
    if (isAlignMatch(p[i].type)) {
      if (!isFirst) first = i;
      isFirst = true;
      last    = i;
    }
}

struct Rec {
  const manta_bam_read_t& r;
  const uint32_t*         cigars;
  const uint8_t *         names, *seqs, *quals;
  bool paired() const { return r.flag & 0x1; }
  bool unmapped() const { return r.flag & 0x4; }
  bool mateUnmapped() const { return r.flag & 0x8; }
  bool fwd() const { return !(r.flag & 0x10); }
  bool mateFwd() const { return !(r.flag & 0x20); }
  bool second() const { return r.flag & 0x80; }
  bool secondary() const { return r.flag & 0x100; }
  bool filter() const { return r.flag & 0x200; }
  bool dup() const { return r.flag & 0x400; }
  bool supplementary() const { return r.flag & 0x800; }
  bool saSplit() const { return r.tags & MANTA_READ_TAG_SA; }
  bool nonStrictSupplement() const { return supplementary() || (secondary() && saSplit()); }  // bam_record.hpp:139-144
  int  pos1() const { return r.pos + 1; }                                                      // bam_record::pos()
  int  matePos1() const { return r.mate_pos + 1; }
  std::string qname() const { return std::string(reinterpret_cast<const char*>(names + r.qname_off), r.qname_len); }
  unsigned    code(int i) const  // htsapi/bam_seq.hpp:159-164
  {
    if (i < 0 || i >= int(r.read_len)) return 15;
    return (seqs[r.seq_off + (i / 2)] >> (4 * (1 - (i % 2)))) & 0xf;
  }
  char          base(int i) const  // get_bam_seq_char, :41-59: '=' A C G T, everything else N
  {
    switch (code(i)) {
    case 0: return '=';
    case 1: return 'A';
    case 2: return 'C';
    case 4: return 'G';
    case 8: return 'T';
    default: return 'N';
    }
  }
  const uint8_t* qual() const { return quals + r.qual_off; }
  Path           path() const { return toPath(cigars + r.cigar_off, r.n_cigar); }
  Path           matePath() const  // htsapi/SimpleAlignment_bam_util.cpp:43-61
  {
    if (r.tags & MANTA_READ_TAG_MC) return toPath(cigars + r.mate_cigar_off, r.n_mate_cigar);
    Path p(1);
    p[0].type   = MATCH;
    p[0].length = r.read_len;
    return p;
  }
};

bool isReadFilteredCore(const Rec& b)  // manta/ReadFilter.cpp:32-50
{
  if (b.filter()) return true;
  if (b.dup()) return true;
  if (b.supplementary() && !b.saSplit()) return true;
  if (b.secondary() && !b.saSplit()) return true;
  return false;
}

bool isOverlappingPair(const Rec& b, const Path& path)  // htsapi/bam_record_util.cpp:84-108
{
  if (!b.paired() || b.unmapped() || b.mateUnmapped()) return false;
  if (b.r.tid != b.r.mate_tid) return false;
  if (b.fwd() == b.mateFwd()) return false;
  const int reverseOrientDist = int(b.r.read_len);
  int       posDiff           = b.pos1() - b.matePos1();
  if (!b.fwd()) posDiff *= -1;
  if (posDiff > reverseOrientDist) return false;
  if (b.fwd()) {
    const int alignEnd = b.r.pos + int(refLength(path));
    return (alignEnd - b.matePos1()) >= 0;
  }
  const int mateEnd = b.r.mate_pos + int(refLength(b.matePath()));
  return (b.r.pos - mateEnd) <= 0;
}

bool isAdapterPair(const Rec& b)  // :54-82
{
  if (b.saSplit()) return false;
  const Path aln = b.path();
  if (b.r.tags & MANTA_READ_TAG_MC) {
    const Path mate = b.matePath();
    if (b.fwd()) {
      const unsigned endpos       = unsigned(b.r.pos) + refLength(aln) + softClipRight(aln);
      const unsigned mateStartPos = unsigned(b.r.mate_pos) + refLength(mate) + softClipRight(mate);
      return endpos > mateStartPos;
    }
    const unsigned endpos       = unsigned(b.r.pos) - softClipLeft(aln);
    const unsigned mateStartPos = unsigned(b.r.mate_pos) - softClipLeft(mate);
    return endpos < mateStartPos;
  }
  return (b.fwd() ? softClipRight(aln) : softClipLeft(aln)) > 0;
}

struct RefSeg {
  int            begin;
  unsigned       len;
  const uint8_t* text;
  char           get(int pos) const { return (pos < begin || pos >= begin + int(len)) ? 'N' : char(text[pos - begin]); }  // reference_contig_segment.hpp:42-46
};

bool baseMatch(char a, char b) { return a == 'N' || b == 'N' || a == b; }  // SVLocusScannerSemiAligned.cpp:45-49

/// getSVBreakendCandidateSemiAligned, SVLocusScannerSemiAligned.cpp:216-316 (with :52-168)
void semiAligned(const Rec& b, const RefSeg& ref, bool useOverlapPairEvidence, unsigned& leading, unsigned& trailing)
{
  static const unsigned contiguousMatchCount = 5;
  leading = trailing = 0;
  const Path aln     = b.path();
  const bool overlap = isOverlappingPair(b, aln);
  if (overlap && (!useOverlapPairEvidence || isAdapterPair(b))) return;
  const unsigned readSize = b.r.read_len;
  // matchifyEdgeSoftClip (blt_util/SimpleAlignment.cpp:33-75): soft clips outside the outermost match segments become matches
  Path     m;
  int      mpos = b.r.pos;
  unsigned first, last;
  matchEdgeSegments(aln, first, last);
  for (unsigned i = 0; i < aln.size(); ++i) {
    const Seg& ps     = aln[i];
    const bool isLead = i < first, isTrail = i > last;
    const bool target = (isLead || isTrail) && ps.type == SOFT_CLIP;
    if (target && isLead) mpos -= int(ps.length);
    if (target || ps.type == MATCH) {
      if (!m.empty() && m.back().type == MATCH)
        m.back().length += ps.length;
      else
        m.push_back(Seg{MATCH, ps.length});
    } else {
      m.push_back(ps);
    }
  }
  // leadingEdgePoorAlignmentLength (:52-98)
  unsigned leadTmp = 0, trailTmp = 0;
  {
    int      readIndex = 0, refIndex = mpos;
    unsigned matchLength = 0;
    bool     done        = false;
    for (const Seg& ps : m) {
      if (isAlignMatch(ps.type)) {
        for (unsigned s = 0; s < ps.length; ++s) {
          if (baseMatch(b.base(readIndex + int(s)), ref.get(refIndex + int(s)))) {
            if (++matchLength >= contiguousMatchCount) {
              leadTmp = unsigned((readIndex + int(s)) - int(matchLength - 1));
              done    = true;
              break;
            }
          } else {
            matchLength = 0;
          }
        }
        if (done) break;
      } else if (isIndel(ps.type)) {
        matchLength = 0;
      }
      if (isReadLength(ps.type)) readIndex += int(ps.length);
      if (isRefLength(ps.type)) refIndex += int(ps.length);
    }
    if (!done) leadTmp = unsigned(readIndex);
  }
  // trailingEdgePoorAlignmentLength (:101-151)
  {
    int      readIndex = int(readSize) - 1, refIndex = mpos + int(refLength(m)) - 1;
    unsigned matchLength = 0;
    bool     done        = false;
    for (size_t k = m.size(); k-- > 0;) {
      const Seg& ps = m[k];
      if (isAlignMatch(ps.type)) {
        for (unsigned s = 0; s < ps.length; ++s) {
          if (baseMatch(b.base(readIndex - int(s)), ref.get(refIndex - int(s)))) {
            if (++matchLength >= contiguousMatchCount) {
              trailTmp = unsigned((int(readSize) - (readIndex - int(s))) - int(matchLength));
              done     = true;
              break;
            }
          } else {
            matchLength = 0;
          }
        }
        if (done) break;
      } else if (isIndel(ps.type)) {
        matchLength = 0;
      }
      if (isReadLength(ps.type)) readIndex -= int(ps.length);
      if (isRefLength(ps.type)) refIndex -= int(ps.length);
    }
    if (!done) trailTmp = unsigned(int(readSize) - (readIndex + 1));
  }
  if (leadTmp + trailTmp >= readSize) return;  // :259
  const uint8_t* q = b.qual();
  if (leadTmp != 0 && (!overlap || b.saSplit() || b.fwd())) {  // :267-285
    unsigned hq = 0;
    for (unsigned p = 0; p < leadTmp; ++p)
      if (q[p] >= 20) ++hq;
    if (float(hq) / float(leadTmp) >= 0.75f) leading = leadTmp;
  }
  if (trailTmp != 0 && (!overlap || b.saSplit() || !b.fwd())) {  // :287-305
    unsigned hq = 0;
    for (unsigned p = 0; p < trailTmp; ++p)
      if (q[readSize - p - 1] >= 20) ++hq;
    if (float(hq) / float(trailTmp) >= 0.75f) trailing = trailTmp;
  }
}

}  // namespace

extern "C" __attribute__((visibility("default"))) void oracle_read_search_range(int32_t bpBegin, int32_t bpEnd, int32_t* sb, int32_t* se)
{
  // :285-303 (size_t arithmetic folded back into pos_t)
  const unsigned size = unsigned(bpEnd - bpBegin > 0 ? bpEnd - bpBegin : 0);
  if (size >= 400) {
    *sb = bpBegin;
    *se = bpEnd;
  } else {
    const unsigned wobble = (400 - size) / 2;
    *sb                   = int32_t(uint32_t(bpBegin) - wobble);
    *se                   = int32_t(uint32_t(bpEnd) + wobble);
  }
}

/// the piles of n_loci candidates as text-free arrays: decision / pile_index per record, results per candidate, and the pile
/// reads as strings in `pileText` (one '\n'-terminated line per pile read, candidates in order) when non-null
extern "C" __attribute__((visibility("default"))) int oracle_read_piles(
    const manta_read_class_options_t* opt, uint32_t n_loci, const manta_read_locus_t* loci, const manta_read_scan_t* scans,
    const manta_bam_read_t* reads, const uint32_t* cigars, const uint8_t* names, const uint8_t* seqs, const uint8_t* quals,
    const uint8_t* refs, uint8_t* decision, uint32_t* pile_index, manta_read_locus_result_t* results, char* pileText, uint64_t pileTextCap,
    uint64_t* pileTextUsed)
{
  const unsigned maxNumReads = opt->max_reads ? opt->max_reads : 1000;  // :342
  std::string    text;
  for (uint32_t l = 0; l < n_loci; ++l) {
    const manta_read_locus_t&                loc = loci[l];
    std::map<std::string, unsigned>          readIndex;  // :102-119
    std::vector<std::string>                 pile;
    std::vector<unsigned>                    depth;
    int                                      depthBegin = 0;
    bool                                     localDepthTriggered = false;
    bool                                     hasEquals = false;
    for (uint32_t s = loc.scan_begin; s < loc.scan_end; ++s) {
      const manta_read_scan_t& sc = scans[s];
      int32_t                  searchBegin, searchEnd;
      oracle_read_search_range(sc.bp_begin, sc.bp_end, &searchBegin, &searchEnd);
      const int leftB = searchBegin, leftE = sc.bp_begin, rightB = sc.bp_end, rightE = searchEnd;  // :302-303
      const unsigned minAssembleIndelSize = opt->min_candidate_variant_size / 2;                  // :317
      const bool     rightOpen = (sc.bp_state != 2), leftOpen = (sc.bp_state != 1);              // :320-329
      if (sc.first_of_breakend) {                                                                   // :340
        depth.assign(size_t(searchEnd > searchBegin ? searchEnd - searchBegin : 0), 0u);
        depthBegin = searchBegin;
      }
      const RefSeg ref{sc.ref_begin, sc.ref_len, refs + sc.ref_off};
      bool         isLastSet = false;  // ShadowReadFinder (:383)
      std::string  lastQname;
      for (uint32_t i = sc.read_begin; i < sc.read_end; ++i) {
        decision[i]   = 0;
        pile_index[i] = 0xffffffffu;
      }
      for (uint32_t i = sc.read_begin; i < sc.read_end; ++i) {
        if (pile.size() >= maxNumReads) break;  // :388-393
        const Rec b{reads[i], cigars, names, seqs, quals};
        const int refPos = b.r.pos;            // :397
        if (refPos >= searchEnd) break;
        if (isReadFilteredCore(b)) continue;  // :402
        if (loc.is_max_depth && !sc.is_tumor && !b.unmapped()) {  // :406-414, addReadToDepthEst :85-100
          const int endPos = depthBegin + int(depth.size());
          for (int ri = (depthBegin - refPos > 0 ? depthBegin - refPos : 0); ri < int(b.r.read_len); ++ri) {
            const int p = refPos + ri;
            if (p >= endPos) break;
            depth[size_t(p - depthBegin)]++;
          }
        }
        if (b.nonStrictSupplement()) continue;  // :416
        if (loc.is_max_depth) {                 // :418-427
          const int off = refPos - searchBegin;
          if (off >= 0 && float(depth[size_t(off)]) > loc.max_local_depth_remote) localDepthTriggered = true;
          if (off >= 0 && float(depth[size_t(off)]) > loc.max_depth) {
            decision[i] |= MANTA_READ_DEPTH_FILTERED;
            continue;
          }
        }
        const Path path = b.path();
        if (loc.search_remote) {  // :443-470 with RemoteMateReadUtil.cpp:29-55
          bool cand = b.paired() && !b.nonStrictSupplement() && !b.unmapped() && !b.mateUnmapped() && b.r.mapq >= opt->min_mapq &&
                      b.r.tid >= 0 && b.r.mate_tid >= 0;
          if (cand && b.r.tid == b.r.mate_tid) {
            const int d = b.pos1() - b.matePos1();
            cand        = (d < 0 ? -d : d) >= 10000;
          }
          if (cand) {
            // matchifyEdgeSoftClipRefRange (blt_util/SimpleAlignment.cpp:77-108)
            int      rb = b.r.pos, re = b.r.pos;
            unsigned first, last;
            matchEdgeSegments(path, first, last);
            for (unsigned k = 0; k < path.size(); ++k) {
              const bool lead = k < first, trail = k > last;
              if (lead || trail) {
                if (isReadLength(path[k].type)) {
                  if (lead)
                    rb -= int(path[k].length);
                  else
                    re += int(path[k].length);
                }
              } else if (isRefLength(path[k].type)) {
                re += int(path[k].length);
              }
            }
            const bool hitsLeft = (re > leftB) && (rb < leftE), hitsRight = (re > rightB) && (rb < rightE);  // known_pos_range2.hpp:87-90
            const bool leftMate = leftOpen && !hitsLeft, rightMate = rightOpen && !hitsRight;
            bool       c2 = true;
            if (!leftMate && !b.fwd()) c2 = false;
            if (!rightMate && b.fwd()) c2 = false;
            if (c2) decision[i] |= MANTA_READ_REMOTE_MATE;
          }
        }
        bool indelKeeper = false;  // :473-483
        if (!b.unmapped())
          for (const Seg& ps : path)
            if (isIndel(ps.type)) {
              if (ps.length >= minAssembleIndelSize) indelKeeper = true;
              break;
            }
        bool semiKeeper = false;  // :486-509
        if (!b.unmapped()) {
          unsigned lead, trail;
          semiAligned(b, ref, opt->use_overlap_pair_evidence != 0, lead, trail);
          if (rightOpen && trail >= 4) semiKeeper = true;
          if (leftOpen && lead >= 4) semiKeeper = true;
        }
        // ShadowReadFinder::check (ShadowReadFinder.hpp:52-57, .cpp:33-113)
        bool shadowKeeper = false;
        if (isLastSet) {
          isLastSet = false;
          bool good = b.paired() && !b.nonStrictSupplement() && b.unmapped() && !b.mateUnmapped();
          if (good) {
            unsigned sum = 0;
            for (unsigned p = 0; p < b.r.read_len; ++p) sum += b.qual()[p];
            const unsigned avg = b.r.read_len ? sum / b.r.read_len : 0;  // bam_record_util.cpp:110-122
            if (avg < 25) good = false;
          }
          if (good && b.qname() != lastQname) good = false;
          shadowKeeper = good;
        }
        if (!shadowKeeper) {
          bool anchor = b.paired() && !b.unmapped() && b.mateUnmapped();
          if (anchor && !leftOpen && !b.fwd()) anchor = false;
          if (anchor && !rightOpen && b.fwd()) anchor = false;
          if (anchor && b.r.mapq < opt->min_singleton_mapq_candidates) anchor = false;
          if (anchor) {
            lastQname = b.qname();
            isLastSet = true;
          }
        }
        if (indelKeeper) decision[i] |= MANTA_READ_INDEL;
        if (semiKeeper) decision[i] |= MANTA_READ_SEMI_ALIGNED;
        if (shadowKeeper) decision[i] |= MANTA_READ_SHADOW;
        if (!(indelKeeper || semiKeeper || shadowKeeper)) continue;  // :543
        bool isReversed = sc.is_locus_reversed != 0;                  // :557-563
        if (shadowKeeper && b.mateFwd()) isReversed = !isReversed;
        // insertAssemblyRead (:102-136)
        const std::string key = b.qname() + "_" + (b.second() ? '2' : '1') + "_" + std::to_string(sc.bam_index);
        if (readIndex.count(key)) {
          decision[i] |= MANTA_READ_DUPLICATE_KEY;
          continue;
        }
        readIndex[key] = unsigned(pile.size());
        std::string rd(b.r.read_len, 'N');
        for (unsigned p = 0; p < b.r.read_len; ++p) {
          rd[p] = b.base(int(p));
          if (rd[p] == '=') hasEquals = true;
        }
        for (unsigned p = 0; p < b.r.read_len; ++p)
          if (b.qual()[p] < opt->min_qval) rd[p] = 'N';
        if (isReversed) {  // reverseCompStr, blt_util/seq_util.hpp:150-204 ('=' is a fatal base_error there: status UNSUPPORTED here)
          std::string rc(rd.rbegin(), rd.rend());
          for (char& ch : rc) {
            switch (ch) {
            case 'A': ch = 'T'; break;
            case 'C': ch = 'G'; break;
            case 'G': ch = 'C'; break;
            case 'T': ch = 'A'; break;
            default: ch = 'N'; break;
            }
          }
          rd = rc;
        }
        pile_index[i] = unsigned(pile.size());
        decision[i] |= MANTA_READ_IN_PILE | (isReversed ? MANTA_READ_REVERSED : 0u);
        pile.push_back(rd);
      }
    }
    results[l].status          = hasEquals ? MANTA_E_UNSUPPORTED : MANTA_OK;
    results[l].n_pile_reads    = unsigned(pile.size());
    results[l].retrieve_remote = localDepthTriggered ? 0u : 1u;
    results[l].reserved        = 0;
    for (const std::string& r : pile) text += r + "\n";
  }
  if (pileTextUsed) *pileTextUsed = text.size();
  if (pileText && pileTextCap > text.size()) std::memcpy(pileText, text.c_str(), text.size() + 1);
  return 0;
}
