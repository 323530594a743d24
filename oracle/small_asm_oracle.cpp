// TEST INFRASTRUCTURE ONLY -- CPU restatement of Manta's SmallAssembler, written the way small_assemble_kernel computes
// (dense node ids, per-node read bitsets, link tables, bitmap seed / edge sets), every step citing the reference:
//   /root/reference/src/c++/lib/assembly/SmallAssembler.cpp
//     runSmallAssembler :622-685, buildContigs :465-620, getKmerCounts :396-455, walk :143-391
// It is pinned against the unmodified reference (oracle/_ref/libmanta_ref.so: ref_small_assemble) by
// tests/test_small_assembler.py; only tests/ may call it.  No reference source is copied here.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <vector>

namespace {

struct Opts {
  unsigned minWordLength, maxWordLength, wordStepSize, minCoverage, minConservativeCoverage, minSeedReads, maxAssemblyIterations;
};

typedef std::vector<uint64_t> ReadSet;  // bit r = read r

ReadSet emptySet(size_t nReads) { return ReadSet((nReads + 63) / 64 + 1, 0); }
bool     has(const ReadSet& s, unsigned r) { return (s[r >> 6] >> (r & 63)) & 1; }
void     put(ReadSet& s, unsigned r) { s[r >> 6] |= uint64_t(1) << (r & 63); }
unsigned size(const ReadSet& s)
{
  unsigned n = 0;
  for (uint64_t w : s) n += unsigned(__builtin_popcountll(w));
  return n;
}
ReadSet both(const ReadSet& a, const ReadSet& b)
{
  ReadSet r(a);
  for (size_t i = 0; i < r.size(); ++i) r[i] &= b[i];
  return r;
}
void add(ReadSet& a, const ReadSet& b)
{
  for (size_t i = 0; i < a.size(); ++i) a[i] |= b[i];
}
void addUnless(ReadSet& a, const ReadSet& b, const ReadSet& unless)
{
  for (size_t i = 0; i < a.size(); ++i) a[i] |= b[i] & ~unless[i];
}
void drop(ReadSet& a, const ReadSet& b)
{
  for (size_t i = 0; i < a.size(); ++i) a[i] &= ~b[i];
}

const char SYM[4] = {'A', 'C', 'G', 'T'};
const int  NONE   = -1;

struct Graph {
  unsigned                   k = 0;
  std::vector<std::string>   word;
  std::vector<ReadSet>       reads;  // wordSupportReads (:452)
  std::vector<unsigned>      count;  // wordCount (:450)
  std::vector<int>           succ, pred;
  std::map<std::string, int> index;
  int find(const std::string& w) const
  {
    const auto it = index.find(w);
    return it == index.end() ? NONE : it->second;
  }
};

struct Contig {
  std::string seq;
  unsigned    seedReadCount = 0;
  ReadSet     support, reject;
  int         consBegin = 0, consEnd = 0;
};

/// getKmerCounts (:396-455) over the unused reads; returns the reads that hold a word twice
std::vector<unsigned> buildGraph(const std::vector<std::string>& reads, const std::vector<bool>& used, const unsigned k, Graph& g)
{
  g   = Graph();
  g.k = k;
  std::vector<unsigned> repeatReads;
  for (unsigned r = 0; r < reads.size(); ++r) {
    if (used[r]) continue;  // :413
    const std::string& seq(reads[r]);
    if (seq.size() < k) continue;  // :419
    std::set<std::string> mine;
    bool                  twice = false;
    for (size_t j = 0; j + k <= seq.size(); ++j) {
      const std::string w(seq.substr(j, k));
      if (w.find('N') != std::string::npos) continue;  // :428
      if (!mine.insert(w).second) {                    // :430-441
        twice = true;
        break;
      }
    }
    if (twice) repeatReads.push_back(r);
    for (const std::string& w : mine) {  // :448-453 (the words seen before the break are counted too; the caller discards the table then)
      int n = g.find(w);
      if (n == NONE) {
        n = int(g.word.size());
        g.word.push_back(w);
        g.reads.push_back(emptySet(reads.size()));
        g.count.push_back(0);
        g.index[w] = n;
      }
      g.count[n]++;
      put(g.reads[n], r);
    }
  }
  const size_t nn = g.word.size();
  g.succ.assign(nn * 4, NONE);
  g.pred.assign(nn * 4, NONE);
  for (size_t n = 0; n < nn; ++n)
    for (int c = 0; c < 4; ++c) {
      g.succ[n * 4 + c] = g.find(g.word[n].substr(1) + SYM[c]);
      g.pred[n * 4 + c] = g.find(SYM[c] + g.word[n].substr(0, k - 1));
    }
  return repeatReads;
}

/// walk (:143-391); seenEdge = the words met on this walk (seenEdgeBefore)
Contig walk(const Opts& opt, const Graph& g, const int seed, const size_t nReads, std::vector<bool>& seenEdge)
{
  const unsigned k = g.k;
  Contig         c;
  c.support = g.reads[seed];  // :158
  c.reject  = emptySet(nReads);
  c.seq     = g.word[seed];
  for (int s = 0; s < 4; ++s) {  // :162-185: siblings of the seed (same first k-1 bases)
    if (SYM[s] == g.word[seed][k - 1]) continue;
    const int n = g.find(g.word[seed].substr(0, k - 1) + SYM[s]);
    if (n != NONE) add(c.reject, g.reads[n]);
  }
  std::fill(seenEdge.begin(), seenEdge.end(), false);
  seenEdge[seed] = true;
  std::set<std::string> seenTrunk;  // seenVertexBefore (:196)
  for (int mode = 0; mode < 2; ++mode) {
    const bool isEnd = (mode == 0);
    unsigned   consOffset = 0;
    int        cur        = seed;  // the word at the growing end
    while (true) {
      const std::string trunk = isEnd ? g.word[cur].substr(1) : g.word[cur].substr(0, k - 1);  // :201-202
      if (!seenTrunk.insert(trunk).second) break;                                              // :212-219
      unsigned maxBaseCount = 0, maxShared = 0;
      int      maxNode = NONE, maxSym = 0;
      ReadSet  maxWordReads = emptySet(nReads), maxSharedReads = emptySet(nReads), remove2 = emptySet(nReads), rejectAdd = emptySet(nReads);
      for (int s = 0; s < 4; ++s) {  // :230-282
        const int n = isEnd ? g.succ[size_t(cur) * 4 + s] : g.pred[size_t(cur) * 4 + s];
        if (n == NONE) continue;
        const ReadSet  shared = both(c.support, g.reads[n]);
        const unsigned cnt    = size(shared);
        if (cnt == 0) continue;  // :259
        if (cnt > maxShared) {
          add(remove2, maxSharedReads);  // :265-266
          add(rejectAdd, maxWordReads);  // :269
          maxWordReads   = g.reads[n];
          maxShared      = cnt;
          maxSharedReads = shared;
          maxBaseCount   = g.count[n];
          maxSym         = s;
          maxNode        = n;
        } else {
          add(remove2, shared);  // :277-278
          add(rejectAdd, g.reads[n]);
        }
      }
      if (maxBaseCount < opt.minCoverage) break;  // :289
      if (maxBaseCount == 0) break;               // :298
      seenEdge[maxNode] = true;                   // :301-303
      c.seq             = isEnd ? c.seq + SYM[maxSym] : SYM[maxSym] + c.seq;
      if (consOffset != 0 || maxBaseCount < opt.minConservativeCoverage) consOffset++;  // :309-311
      // :319-345 -- previousWordReads is declared inside the loop body (:228): the test is "a word was chosen"
      for (int s = 0; s < 4; ++s) {
        const int n = isEnd ? g.pred[size_t(maxNode) * 4 + s] : g.succ[size_t(maxNode) * 4 + s];
        if (n == NONE || n == cur) continue;  // the selected branch (:324)
        add(rejectAdd, g.reads[n]);
      }
      add(c.reject, rejectAdd);                     // :359-361
      addUnless(c.support, maxWordReads, c.reject);  // :374-379
      drop(c.support, remove2);                     // :387-389
      cur = maxNode;
    }
    (isEnd ? c.consEnd : c.consBegin) = int(consOffset);  // :393-397
  }
  c.consEnd = int(c.seq.size()) - c.consEnd;  // :403
  return c;
}

std::string run(const Opts& opt, const std::vector<std::string>& reads)
{
  const size_t        nReads = reads.size();
  std::vector<bool>   used(nReads, false), filtered(nReads, false);
  std::vector<int>    contigOf(nReads, NONE);
  std::vector<Contig> contigs;
  unsigned            unusedReads = unsigned(nReads);
  for (unsigned it = 0; it < opt.maxAssemblyIterations; ++it) {  // :641
    if (unusedReads < opt.minSeedReads) break;                   // :642
    const unsigned before = unusedReads;
    for (unsigned k = opt.minWordLength; k <= opt.maxWordLength; k += opt.wordStepSize) {
      const bool isLastWord = (k + opt.wordStepSize > opt.maxWordLength);  // :647
      // ---- buildContigs (:465-620) ----
      Graph                       g;
      const std::vector<unsigned> repeatReads = buildGraph(reads, used, k, g);
      if (!repeatReads.empty()) {  // :494-505
        if (isLastWord)
          for (const unsigned r : repeatReads) {
            used[r] = filtered[r] = true;
            unusedReads--;
          }
        continue;
      }
      unsigned maxCount = 0;
      for (const unsigned c : g.count) maxCount = std::max(maxCount, c);
      if (maxCount < opt.minCoverage) continue;  // :537-542
      std::vector<bool> alive(g.word.size(), false), seenEdge(g.word.size(), false);
      for (size_t n = 0; n < g.word.size(); ++n) alive[n] = (g.count[n] == maxCount);
      Contig best;
      best.support = best.reject = emptySet(nReads);
      bool haveSeed = false;
      while (true) {  // :545-560: seeds in std::set<std::string> order = the ordered index
        int seed = NONE;
        for (const auto& kv : g.index)
          if (alive[kv.second]) {
            seed = kv.second;
            break;
          }
        if (seed == NONE) break;
        alive[seed] = false;
        haveSeed    = true;
        const Contig c = walk(opt, g, seed, nReads, seenEdge);
        if (c.seq.size() > best.seq.size()) best = c;  // :551-553
        for (size_t n = 0; n < alive.size(); ++n)
          if (seenEdge[n]) alive[n] = false;  // :556
      }
      best.seedReadCount = haveSeed ? maxCount : 0;         // :577-580 (counted for the LAST seed; every seed has the maximal count)
      if (best.seedReadCount < opt.minSeedReads) continue;  // :586-591
      for (unsigned r = 0; r < nReads; ++r)                 // :594-606
        if (!used[r] && has(best.support, r)) {
          used[r]     = true;
          contigOf[r] = int(contigs.size());
          unusedReads--;
        }
      contigs.push_back(best);
      break;  // isAssemblySuccess (:650)
    }
    if (unusedReads == before) break;  // :657
  }
  // the canonical text of oracle/ref_driver.cpp::assemblyText
  std::ostringstream os;
  auto               join = [&](const ReadSet& s) {
    bool first = true;
    for (unsigned r = 0; r < nReads; ++r)
      if (has(s, r)) {
        os << (first ? "" : ",") << r;
        first = false;
      }
  };
  os << "contigs " << contigs.size() << '\n';
  for (size_t i = 0; i < contigs.size(); ++i) {
    const Contig& c(contigs[i]);
    os << "contig " << i << " seq=" << c.seq << " seed=" << c.seedReadCount << " cons=" << c.consBegin << ',' << c.consEnd << " support=";
    join(c.support);
    os << " reject=";
    join(c.reject);
    os << '\n';
  }
  os << "reads " << nReads << " normal " << nReads << '\n';
  for (unsigned r = 0; r < nReads; ++r) {
    os << "read " << r << " used=" << used[r] << " filtered=" << filtered[r] << " pseudo=0 ids=";
    if (contigOf[r] != NONE) os << contigOf[r];
    os << '\n';
  }
  return os.str();
}

}  // namespace

/// opts = {minWordLength,maxWordLength,wordStepSize,minCoverage,minConservativeCoverage,minSeedReads,maxAssemblyIterations};
/// returns the length of the canonical text (may exceed cap)
extern "C" int orc_small_assemble(const uint32_t* o, int n_reads, const char* const* reads, const uint32_t* read_lens, char* out, int cap)
{
  const Opts               opt{o[0], o[1], o[2], o[3], o[4], o[5], o[6]};
  std::vector<std::string> in;
  for (int i = 0; i < n_reads; ++i) in.emplace_back(reads[i], read_lens[i]);
  const std::string s = run(opt, in);
  if (cap > 0) {
    const size_t n = std::min<size_t>(s.size(), size_t(cap) - 1);
    std::memcpy(out, s.data(), n);
    out[n] = 0;
  }
  return int(s.size());
}
