// Pile dump ("manta-pile-dump v1"): one record per candidate as it reaches the assembler + aligner -- the read pile, the reference
// window(s) in alignment order and orientation, the cuts, the assembler options and the aligner's scores.  This is the boundary the
// whole-batch ABI calls take (manta_smallsv_batch_piles / manta_spanning_batch_piles), so a dump of a production run can be replayed
// through the device path and compared candidate by candidate (tools/replay_piles.py); SURVEY section 8(d) "C1/C3/C4".
//
// Text, line oriented (gzip it: read piles compress 4-5x):
//   #manta-pile-dump v1
//   S id=<n> reads=<R> opt=<minWordLength,maxWordLength,wordStepSize,minContigLength,minCoverage,minConservativeCoverage,minUnusedReads,
//        minSupportReads,maxAssemblyCount> scores=<match,mismatch,open,extend,offEdge,isAllowEdgeInsertion> extra=<largeGapOpenScore>
//        cuts=<leadingCut,trailingCut,maxLeadingCut,maxTrailingCut>                       -- getSmallSVAssembly, SVCandidateAssemblyRefiner.cpp:1860-2038
//   W <reference window (bp1ref, uncut)>
//   r <read>                                                                               -- R lines, in pile order (AssemblyReadInput)
//   J id=<n> reads=<R> opt=<...> scores=<...> extra=<jumpScore> cuts=<align1LeadingCut,align1TrailingCut,align2LeadingCut,align2TrailingCut>
//   W <reference 1>   W <reference 2>   r <read> ...                                       -- alignJumpContigs, :1525-1743: the two references
//                                                                                             AFTER the orientation step (:1533-1550)
// The refiner of this repository writes it when asked (SVCandidateAssemblyRefiner::setPileDump); INTEGRATION.md shows the lines an
// instrumented reference build needs to write the same records.
#pragma once
#include <mutex>
#include <ostream>
#include <string>
#include <vector>

namespace manta_amd {

class PileDumpWriter {
public:
  explicit PileDumpWriter(std::ostream& os) : _os(os) { _os << "#manta-pile-dump v1\n"; }

  template <typename AsmOpt, typename Scores>
  void small(const AsmOpt& o, const Scores& sc, const int largeGapOpenScore, const std::vector<std::string>& reads, const std::string& ref,
             const int leadingCut, const int trailingCut, const int maxLeadingCut, const int maxTrailingCut)
  {
    std::lock_guard<std::mutex> g(_mu);
    head('S', o, sc, largeGapOpenScore, reads.size());
    _os << " cuts=" << leadingCut << ',' << trailingCut << ',' << maxLeadingCut << ',' << maxTrailingCut << "\nW " << ref << '\n';
    for (const std::string& r : reads) _os << "r " << r << '\n';
  }
  template <typename AsmOpt, typename Scores>
  void spanning(const AsmOpt& o, const Scores& sc, const int jumpScore, const std::vector<std::string>& reads, const std::string& ref1,
                const std::string& ref2, const int a1Lead, const int a1Trail, const int a2Lead, const int a2Trail)
  {
    std::lock_guard<std::mutex> g(_mu);
    head('J', o, sc, jumpScore, reads.size());
    _os << " cuts=" << a1Lead << ',' << a1Trail << ',' << a2Lead << ',' << a2Trail << "\nW " << ref1 << "\nW " << ref2 << '\n';
    for (const std::string& r : reads) _os << "r " << r << '\n';
  }
  uint64_t count() const { return _n; }

private:
  template <typename AsmOpt, typename Scores>
  void head(const char kind, const AsmOpt& o, const Scores& sc, const int extra, const size_t nReads)
  {
    _os << kind << " id=" << _n++ << " reads=" << nReads << " opt=" << o.minWordLength << ',' << o.maxWordLength << ',' << o.wordStepSize << ','
        << o.minContigLength << ',' << o.minCoverage << ',' << o.minConservativeCoverage << ',' << o.minUnusedReads << ',' << o.minSupportReads << ','
        << o.maxAssemblyCount << " scores=" << sc.match << ',' << sc.mismatch << ',' << sc.open << ',' << sc.extend << ',' << sc.offEdge << ','
        << (sc.isAllowEdgeInsertion ? 1 : 0) << " extra=" << extra;
  }
  std::ostream& _os;
  std::mutex    _mu;
  uint64_t      _n = 0;
};

}  // namespace manta_amd
