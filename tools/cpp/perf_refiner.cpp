// Whole-call throughput of manta_amd::SVCandidateAssemblyRefiner::getCandidateAssemblyDataBatch (host glue included) on
// config-2 shaped complex candidates and config-5 shaped breakend pairs.  Build: see tools/perf_refiner.sh
#include <chrono>
#include <cstdio>
#include <random>
#include <thread>

#include "refiner.hpp"

using namespace manta_amd;

struct Source : RefinerInputSource {
  std::vector<std::string> chroms;
  struct Pile {
    int32_t           tid;
    pos_t             pos;
    AssemblyReadInput reads;
  };
  std::vector<Pile> piles;  // sorted by pos (all on tid 0) in generation order
  void getReferenceSeq(const std::string& chrom, pos_t b, pos_t e, std::string& seq) override
  {
    seq = chroms[std::stoul(chrom)].substr(size_t(b), size_t(e - b + 1));
  }
  /// answers from memory without any state of its own: safe to call from several host threads (setPlanThreads)
  void getBreakendReads(const SVBreakend& bp, bool, const reference_contig_segment&, AssemblyReadInput& reads) override
  {
    if (!reads.empty()) return;
    size_t lo = 0, hi = piles.size();  // first pile at or behind the interval's begin
    while (lo < hi) {
      const size_t mid = (lo + hi) / 2;
      if (piles[mid].pos < bp.interval.range.begin_pos()) lo = mid + 1; else hi = mid;
    }
    if (lo < piles.size() && piles[lo].tid == bp.interval.tid && piles[lo].pos < bp.interval.range.end_pos()) reads = piles[lo].reads;
  }
};

static std::string randSeq(std::mt19937& g, size_t n)
{
  std::string s(n, 'A');
  for (char& c : s) c = "ACGT"[g() & 3];
  return s;
}

int main(int argc, char** argv)
{
  const int    n       = argc > 1 ? atoi(argv[1]) : 2000;
  const bool   span    = argc > 2 && atoi(argv[2]) != 0;
  std::mt19937 g(12345);
  Source       src;
  const size_t spacing = 4000;
  src.chroms.push_back(randSeq(g, size_t(n) * spacing + 8000));
  src.chroms.push_back(randSeq(g, size_t(n) * spacing + 8000));
  std::vector<SVCandidate> svs;
  for (int i = 0; i < n; ++i) {
    const pos_t      pos = pos_t(2000 + size_t(i) * spacing);
    Source::Pile     pile;
    pile.tid = 0;
    pile.pos = pos;
    SVCandidate sv;
    std::string hap;
    size_t      junction;
    if (!span) {
      const int d = 10 + int(g() % 50);
      hap         = src.chroms[0].substr(size_t(pos) - 400, 400) + src.chroms[0].substr(size_t(pos) + d, 400);
      junction    = 400;
      sv.bp1.state    = SVBreakendState::COMPLEX;
      sv.bp1.interval = GenomeInterval(0, pos - 20, pos + 20);
      sv.bp2.state    = SVBreakendState::UNKNOWN;
      sv.bp2.interval = sv.bp1.interval;
    } else {
      const pos_t p2 = pos + 137;
      hap            = src.chroms[0].substr(size_t(pos) - 450, 450) + src.chroms[1].substr(size_t(p2), 450);
      junction       = 450;
      sv.bp1.state    = SVBreakendState::RIGHT_OPEN;
      sv.bp1.interval = GenomeInterval(0, pos - 30, pos + 30);
      sv.bp2.state    = SVBreakendState::LEFT_OPEN;
      sv.bp2.interval = GenomeInterval(1, p2 - 30, p2 + 30);
    }
    const int nReads = span ? 200 : 80, readLen = span ? 250 : 150;
    for (int r = 0; r < nReads; ++r) {
      const size_t lo = junction - size_t(readLen) + 15, hi = junction - 15;
      const size_t s  = lo + g() % (hi - lo);
      std::string  rd = hap.substr(s, size_t(readLen));
      for (char& c : rd)
        if (g() % 333 == 0) c = "ACGT"[g() & 3];
      pile.reads.push_back(rd);
    }
    src.piles.push_back(pile);
    svs.push_back(sv);
  }
  bam_header_info header;
  header.chrom_data.emplace_back("0", unsigned(src.chroms[0].size()));
  header.chrom_data.emplace_back("1", unsigned(src.chroms[1].size()));
  GSCOptions opt;
  if (!span) opt.refineOpt.smallSVAssembleOpt.minWordLength = 31;
  SVCandidateAssemblyRefiner           refiner(opt, header, src);
  const unsigned hostThreads = argc > 3 ? unsigned(atoi(argv[3])) : std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
  refiner.setHostThreads(hostThreads);
  std::vector<SVCandidateAssemblyData> out;
  struct Best {
    double       dt = 1e30;
    RefinerTimes t;
    size_t       svs = 0, contigs = 0;
  };
  // Two plans.  `threaded`: the reference / read callbacks of the candidates run on the host threads -- ONLY for a source that answers
  // concurrently (this one answers from memory; a BAM / FASTA-backed source with one handle per worker cannot).  `sequential`: the
  // callbacks one candidate after the other, as the reference's worker calls them -- what any source supports.
  auto measure = [&](const unsigned planThreads, const char* what) {
    refiner.setPlanThreads(planThreads);
    Best b;
    for (int rep = 0; rep < 3; ++rep) {
      const auto t0 = std::chrono::steady_clock::now();
      refiner.getCandidateAssemblyDataBatch(svs, false, out);
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      size_t       nsv = 0, ncontig = 0;
      for (const auto& d : out) {
        nsv += d.svs.size();
        ncontig += d.contigs.size();
      }
      std::printf("%s (%s plan) n=%d  %.3f s  %.0f candidates/s  (refined SVs %zu, contigs %zu)\n", span ? "spanning" : "complex", what, n, dt, n / dt, nsv, ncontig);
      const RefinerTimes& t(refiner.times());
      std::printf("   plan(host, incl. read/ref callbacks) %.3f  pack %.3f  device(upload+run+download) %.3f  post(host glue) %.3f\n", t.plan, t.pack,
                  t.device, t.post);
      if (dt < b.dt) {
        b.dt      = dt;
        b.t       = t;
        b.svs     = nsv;
        b.contigs = ncontig;
      }
    }
    return b;
  };
  const Best seq = measure(1, "sequential");
  const Best thr = measure(hostThreads, "threaded");
  // (the best of the three calls per plan as one JSON line: bench.py's `refiner_batch` key)
  std::printf("{\"call\": \"SVCandidateAssemblyRefiner::getCandidateAssemblyDataBatch\", \"shape\": \"%s\", \"candidates\": %d, \"seconds\": %.4f, "
              "\"candidates_per_s\": %.1f, \"plan\": \"threaded: the candidates' reference / read callbacks on the host threads -- concurrent in-memory source only\", "
              "\"host_threads\": %u, \"refined_svs\": %zu, \"contigs\": %zu, "
              "\"times_s\": {\"plan\": %.4f, \"pack\": %.4f, \"device\": %.4f, \"post\": %.4f}, "
              "\"sequential_plan\": {\"candidates_per_s\": %.1f, \"seconds\": %.4f, \"note\": \"setPlanThreads(1): callbacks one candidate after the other, as the "
              "reference's worker calls them -- what a BAM / FASTA-backed source supports\", \"times_s\": {\"plan\": %.4f, \"pack\": %.4f, \"device\": %.4f, \"post\": %.4f}}}\n",
              span ? "config-5 breakend pairs" : "config-2 complex candidates", n, thr.dt, n / thr.dt, hostThreads, thr.svs, thr.contigs, thr.t.plan,
              thr.t.pack, thr.t.device, thr.t.post, n / seq.dt, seq.dt, seq.t.plan, seq.t.pack, seq.t.device, seq.t.post);
  return 0;
}
