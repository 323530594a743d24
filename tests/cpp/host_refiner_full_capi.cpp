// Test-only C surface over manta_amd/host/refiner.hpp (the product's SVCandidateAssemblyRefiner) with the same POD
// input and the same text dump as ref_get_candidate_assembly_data in oracle/ref_refiner_driver.cpp (the reference's
// own refiner run in memory).  Linked against the emulator build (CPU tier) or libmanta_amd.so (GPU tier).
#include <fstream>
#include <memory>
#include <cstdint>
#include <cstring>
#include <sstream>

#include "vcf_candidate.hpp"

using namespace manta_amd;
#define MINE_EXPORT extern "C" __attribute__((visibility("default")))

struct ref_refine_input_t {
  int32_t            n_chrom;
  const char* const* chrom_seq;
  int32_t            bp_state[2];
  int32_t            bp_tid[2];
  int32_t            bp_begin[2], bp_end[2];
  int32_t            is_find_large_insertions;
  int32_t            n_reads;
  const char* const* reads;
  int32_t            small_word[3];
  int32_t            spanning_word[3];
  int32_t            n_calls;
};

namespace {
int emit(const std::string& s, char* out, int cap)
{
  const int n = static_cast<int>(s.size());
  if (out != nullptr && cap > 0) {
    const int m = (n < cap - 1) ? n : (cap - 1);
    std::memcpy(out, s.data(), m);
    out[m] = '\0';
  }
  return n;
}
std::string segText(const std::vector<std::pair<unsigned, unsigned>>& segs)
{
  std::ostringstream os;
  for (size_t i = 0; i < segs.size(); ++i) os << (i ? "," : "") << segs[i].first << "-" << segs[i].second;
  return os.str();
}
void dumpAlign(std::ostream& os, const Alignment& a) { os << a.beginPos << ":" << ALIGNPATH::apath_to_cigar(a.apath); }
std::string dumpAssemblyData(const SVCandidateAssemblyData& d)
{
  std::ostringstream os;
  os << "isCandidateSpanning=" << d.isCandidateSpanning << " isSpanning=" << d.isSpanning << " isOverlapSkip=" << d.isOverlapSkip
     << " best=" << d.bestAlignmentIndex << " orient=" << d.bporient.isBp2AlignedFirst << d.bporient.isBp1Reversed
     << d.bporient.isBp2Reversed << d.bporient.isBp1First << "\n";
  os << "bp1ref=" << d.bp1ref.get_offset() << "+" << d.bp1ref.seq().size() << " bp2ref=" << d.bp2ref.get_offset() << "+"
     << d.bp2ref.seq().size() << "\n";
  for (size_t i = 0; i < d.contigs.size(); ++i) {
    const AssembledContig& c(d.contigs[i]);
    os << "contig " << i << " " << c.seq << " seed=" << c.seedReadCount << " cons=" << c.conservativeRange.begin_pos() << ","
       << c.conservativeRange.end_pos() << " sup=";
    for (const unsigned r : c.supportReads) os << r << ",";
    os << "\n";
  }
  for (size_t i = 0; i < d.smallSVAlignments.size(); ++i) {
    const auto& a(d.smallSVAlignments[i]);
    os << "small " << i << " score=" << a.score << " jumped=" << a.isJumped << " ";
    dumpAlign(os, a.align);
    os << " seg=" << (i < d.smallSVSegments.size() ? segText(d.smallSVSegments[i]) : std::string("-"));
    if (i < d.largeInsertInfo.size()) {
      const LargeInsertionInfo& li(d.largeInsertInfo[i]);
      os << " li=" << li.isLeftCandidate << li.isRightCandidate << "," << li.contigOffset << "," << li.refOffset << "," << li.score;
    }
    os << "\n";
  }
  for (size_t i = 0; i < d.spanningAlignments.size(); ++i) {
    const auto& a(d.spanningAlignments[i]);
    os << "span " << i << " score=" << a.score << " ins=" << a.jumpInsertSize << " range=" << a.jumpRange << " ";
    dumpAlign(os, a.align1);
    os << " ";
    dumpAlign(os, a.align2);
    os << "\n";
  }
  for (size_t i = 0; i < d.extendedContigs.size(); ++i) os << "ext " << i << " " << d.extendedContigs[i] << "\n";
  for (size_t i = 0; i < d.svs.size(); ++i) {
    const SVCandidate& sv(d.svs[i]);
    os << "sv " << i << " imprecise=" << sv.isImprecise() << " align=" << sv.assemblyAlignIndex << "/" << sv.assemblySegmentIndex
       << " bp1=" << SVBreakendState::label(sv.bp1.state) << ":" << sv.bp1.interval.tid << ":" << sv.bp1.interval.range.begin_pos() << "-"
       << sv.bp1.interval.range.end_pos() << " bp2=" << SVBreakendState::label(sv.bp2.state) << ":" << sv.bp2.interval.tid << ":"
       << sv.bp2.interval.range.begin_pos() << "-" << sv.bp2.interval.range.end_pos() << " insertSeq=" << sv.insertSeq
       << " insertAlignment=" << ALIGNPATH::apath_to_cigar(sv.insertAlignment) << " unknownSizeInsertion=" << sv.isUnknownSizeInsertion
       << " L=" << sv.unknownSizeInsertionLeftSeq << " R=" << sv.unknownSizeInsertionRightSeq << "\n";
  }
  return os.str();
}

/// in-memory chromosomes; read piles keyed by the first breakend that asks for them
struct MemorySource : RefinerInputSource {
  std::vector<std::string> chroms;
  struct Pile {
    int32_t           tid;
    pos_t             pos;
    AssemblyReadInput reads;
  };
  std::vector<Pile> piles;
  void getReferenceSeq(const std::string& chrom, pos_t beginPos, pos_t endPos, std::string& seq) override
  {
    seq = chroms.at(std::stoul(chrom)).substr(size_t(beginPos), size_t(endPos - beginPos + 1));
  }
  void getBreakendReads(const SVBreakend& bp, bool, const reference_contig_segment&, AssemblyReadInput& reads) override
  {
    if (!reads.empty()) return;  // the pile holds both breakends' reads already, in final order
    for (const Pile& p : piles)
      if (p.tid == bp.interval.tid && p.pos >= bp.interval.range.begin_pos() && p.pos < bp.interval.range.end_pos()) {
        reads = p.reads;
        return;
      }
  }
};

SVCandidate makeSV(const ref_refine_input_t& in)
{
  SVCandidate sv;
  sv.bp1.state    = static_cast<SVBreakendState::index_t>(in.bp_state[0]);
  sv.bp1.interval = GenomeInterval(in.bp_tid[0], in.bp_begin[0], in.bp_end[0]);
  sv.bp2.state    = static_cast<SVBreakendState::index_t>(in.bp_state[1]);
  sv.bp2.interval = GenomeInterval(in.bp_tid[1], in.bp_begin[1], in.bp_end[1]);
  return sv;
}
void setOptions(const ref_refine_input_t& in, GSCOptions& options)
{
  auto setWords = [](IterativeAssemblerOptions& o, const int32_t* w) {
    if (w[0] > 0) o.minWordLength = unsigned(w[0]);
    if (w[1] > 0) o.maxWordLength = unsigned(w[1]);
    if (w[2] > 0) o.wordStepSize = unsigned(w[2]);
  };
  setWords(options.refineOpt.smallSVAssembleOpt, in.small_word);
  setWords(options.refineOpt.spanningAssembleOpt, in.spanning_word);
}
}  // namespace

thread_local SVCandidateAssemblyRefiner::Stats g_stats;

// pile dump of every candidate the following calls send to the device (pile_dump.hpp; tools/replay_piles.py)
static std::unique_ptr<std::ofstream>  g_dumpFile;
static std::unique_ptr<PileDumpWriter> g_dump;
/// path: start a dump file (truncates); nullptr: close it.  Returns the number of records written so far.
MINE_EXPORT uint64_t mine_set_pile_dump(const char* path)
{
  const uint64_t n = g_dump ? g_dump->count() : 0;
  g_dump.reset();
  g_dumpFile.reset();
  if (path) {
    g_dumpFile.reset(new std::ofstream(path));
    g_dump.reset(new PileDumpWriter(*g_dumpFile));
  }
  return n;
}

/// n inputs that share chromosomes/options (those of inputs[0]); is_batched == 1 -> ONE getCandidateAssemblyDataBatch
/// call, 0 -> consecutive single calls on the same refiner object; > 1 / -1 -> one batched call with that many plan threads / one
/// (setPlanThreads) and per-candidate errors.  Text = the dumps in order.
MINE_EXPORT int mine_get_candidate_assembly_data_multi(const ref_refine_input_t* inputs, int n, int is_batched, char* out, int cap)
{
  try {
    MemorySource    source;
    bam_header_info header;
    for (int i = 0; i < inputs[0].n_chrom; ++i) {
      source.chroms.emplace_back(inputs[0].chrom_seq[i]);
      header.chrom_data.emplace_back(std::to_string(i).c_str(), unsigned(source.chroms.back().size()));
    }
    GSCOptions options;
    setOptions(inputs[0], options);
    std::vector<SVCandidate> svs;
    for (int k = 0; k < n; ++k) {
      const ref_refine_input_t& in(inputs[k]);
      MemorySource::Pile        pile;
      pile.tid = in.bp_tid[0];
      pile.pos = in.bp_begin[0];
      for (int i = 0; i < in.n_reads; ++i) pile.reads.emplace_back(in.reads[i]);
      source.piles.push_back(pile);
      for (int c = 0; c < std::max(1, in.n_calls); ++c) svs.push_back(makeSV(in));
    }
    const SVCandidateAssemblyRefiner refiner(options, header, source);
    const_cast<SVCandidateAssemblyRefiner&>(refiner).setPileDump(g_dump.get());
    const bool                       large = inputs[0].is_find_large_insertions != 0;
    std::string                      text;
    if (is_batched < 0 || is_batched > 1) {
      // per-candidate error isolation, with the input callbacks of the candidates on |is_batched| host threads (-1: one thread, the
      // sequential plan): a candidate whose processing throws reports its exception, every other one its result
      SVCandidateAssemblyRefiner& r(const_cast<SVCandidateAssemblyRefiner&>(refiner));
      r.setPlanThreads(is_batched < 0 ? 1u : unsigned(is_batched));
      std::vector<SVCandidateAssemblyData> data;
      std::vector<std::exception_ptr>      errors;
      refiner.getCandidateAssemblyDataBatch(svs, large, data, &errors);
      for (size_t i = 0; i < data.size(); ++i) {
        if (errors[i]) {
          try {
            std::rethrow_exception(errors[i]);
          } catch (const std::exception& e) {
            text += std::string("EXCEPTION ") + e.what() + "\n";
          }
        } else {
          text += dumpAssemblyData(data[i]);
        }
      }
    } else if (is_batched) {
      std::vector<SVCandidateAssemblyData> data;
      refiner.getCandidateAssemblyDataBatch(svs, large, data);
      for (const auto& d : data) text += dumpAssemblyData(d);
    } else {
      for (const SVCandidate& sv : svs) {
        SVCandidateAssemblyData data;
        refiner.getCandidateAssemblyData(sv, large, data);
        text += dumpAssemblyData(data);
      }
    }
    g_stats = refiner.stats();
    return emit(text, out, cap);
  } catch (const std::exception& e) {
    return emit(std::string("EXCEPTION ") + e.what(), out, cap);
  }
}

/// counters of the last call: small loci, spanning loci, contig alignments, re-aligned contigs, large-insertion alignments
MINE_EXPORT void mine_last_stats(uint64_t* out)
{
  out[0] = g_stats.smallLoci;
  out[1] = g_stats.spanningLoci;
  out[2] = g_stats.contigAlignments;
  out[3] = g_stats.realignedContigs;
  out[4] = g_stats.largeInsertionAlignments;
}

/// candidateSV.vcf records of the refined candidates of one call (same fixed evidence counts / edge ids as the reference driver)
MINE_EXPORT int mine_candidate_vcf_records(const ref_refine_input_t* in, char* out, int cap)
{
  try {
    MemorySource    source;
    bam_header_info header;
    for (int i = 0; i < in->n_chrom; ++i) {
      source.chroms.emplace_back(in->chrom_seq[i]);
      header.chrom_data.emplace_back(std::to_string(i).c_str(), unsigned(source.chroms.back().size()));
    }
    GSCOptions options;
    setOptions(*in, options);
    MemorySource::Pile pile;
    pile.tid = in->bp_tid[0];
    pile.pos = in->bp_begin[0];
    for (int i = 0; i < in->n_reads; ++i) pile.reads.emplace_back(in->reads[i]);
    source.piles.push_back(pile);
    SVCandidate sv(makeSV(*in));
    sv.bp1.pairCount      = 7;
    sv.bp1.localPairCount = 3;
    sv.bp2.pairCount      = 7;
    sv.bp2.localPairCount = 5;
    sv.candidateIndex     = 4;
    const SVCandidateAssemblyRefiner refiner(options, header, source);
    SVCandidateAssemblyData          data;
    refiner.getCandidateAssemblyData(sv, in->is_find_large_insertions != 0, data);
    std::ostringstream         vcf;
    const VcfWriterCandidateSV writer(source, header, vcf);
    const JunctionIdGenerator  idgen;
    EdgeInfo                   edge;
    edge.locusIndex = 11;
    edge.nodeIndex1 = 2;
    edge.nodeIndex2 = 3;
    for (const SVCandidate& refined : data.svs) {
      SVId svId;
      idgen.getId(edge, refined, false, svId);
      writer.writeSV(refined, svId);
    }
    return emit(vcf.str(), out, cap);
  } catch (const std::exception& e) {
    return emit(std::string("EXCEPTION ") + e.what(), out, cap);
  }
}

MINE_EXPORT int mine_get_candidate_assembly_data(const ref_refine_input_t* in, char* out, int cap)
{
  return mine_get_candidate_assembly_data_multi(in, 1, 0, out, cap);
}

/// same signature as ref_candidate_vcf_header; the fileDate line is written as the placeholder the test strips on both sides
MINE_EXPORT int mine_candidate_vcf_header(
    int nChrom, const char* const* chromLabels, const unsigned* chromLengths, const char* referenceFilename, int isOutputContig,
    const char* progName, const char* progVersion, int nSamples, const char* const* sampleNames, char* out, int cap)
{
  try {
    MemorySource    source;
    bam_header_info header;
    for (int i = 0; i < nChrom; ++i) header.chrom_data.emplace_back(chromLabels[i], chromLengths[i]);
    std::ostringstream         os;
    const VcfWriterCandidateSV writer(source, header, os, isOutputContig != 0);
    std::vector<std::string>   samples;
    for (int i = 0; i < nSamples; ++i) samples.emplace_back(sampleNames[i]);
    writer.writeHeader(progName, progVersion, referenceFilename, "DATE", samples);
    return emit(os.str(), out, cap);
  } catch (const std::exception& e) {
    return emit(std::string("EXCEPTION ") + e.what(), out, cap);
  }
}
