// TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of Manta's assemble+align hot path.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
// as the checker.  The product path (manta_amd/csrc) never links, loads or calls anything in oracle/.
//
// Parity status: PINNED.  Every function below is checked (tests/test_oracle_vs_ref.py) against
//   (a) the reference's own golden vectors, transcribed in tests/golden/ (SURVEY.md section 8c), and
//   (b) the UNMODIFIED reference sources compiled into oracle/_ref/libmanta_ref.so (oracle/ref_driver.cpp)
//       on seeded random inputs, including repeat-rich read piles.
//
// This is a restatement, not a copy: the assembler is expressed the way the HIP kernels compute it
// (dense node ids, successor/predecessor links, read-support bitsets) and the aligners as one generic
// state-transition table; each function cites the reference lines whose behaviour it reproduces
// (paths relative to /root/reference/src/c++/lib).
//
// Results are rendered as canonical text (same format as oracle/ref_driver.cpp, see oracle/FORMAT.md).

#include "bench_harness.hpp"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace orc {

// ------------------------------------------------------------------------------------------------
// options  (options/IterativeAssemblerOptions.hpp:26-59; alphabet is the fixed "ACGT")
// ------------------------------------------------------------------------------------------------
struct AsmOpts {
  unsigned minWordLength, maxWordLength, wordStepSize, minContigLength;
  unsigned minCoverage, minConservativeCoverage, minUnusedReads, minSupportReads, maxAssemblyCount;
};
static const char ALPHABET[4] = {'A', 'C', 'G', 'T'};

// ------------------------------------------------------------------------------------------------
// read-index bitsets (stand-in for the std::set<unsigned> of assembly/AssembledContig.hpp:47-48)
// ------------------------------------------------------------------------------------------------
struct Bits {
  std::vector<uint64_t> w;
  explicit Bits(size_t nbits = 0) : w((nbits + 63) / 64, 0) {}
  void set(unsigned i) { w[i >> 6] |= (uint64_t(1) << (i & 63)); }
  bool test(unsigned i) const { return (w[i >> 6] >> (i & 63)) & 1; }
  bool any() const
  {
    for (uint64_t x : w)
      if (x) return true;
    return false;
  }
  unsigned count() const
  {
    unsigned c = 0;
    for (uint64_t x : w) c += __builtin_popcountll(x);
    return c;
  }
  bool operator==(const Bits& o) const { return w == o.w; }
  std::vector<unsigned> members() const
  {
    std::vector<unsigned> v;
    for (size_t i = 0; i < w.size() * 64; ++i)
      if (test(i)) v.push_back(i);
    return v;
  }
};
static Bits band(const Bits& a, const Bits& b)
{
  Bits r(a);
  for (size_t i = 0; i < r.w.size(); ++i) r.w[i] &= b.w[i];
  return r;
}
static Bits bandnot(const Bits& a, const Bits& b)
{
  Bits r(a);
  for (size_t i = 0; i < r.w.size(); ++i) r.w[i] &= ~b.w[i];
  return r;
}
static void bor(Bits& a, const Bits& b)
{
  for (size_t i = 0; i < a.w.size(); ++i) a.w[i] |= b.w[i];
}

struct Contig {
  std::string seq;
  unsigned    seedReadCount = 0;  // never written by the reference (assembly/AssembledContig.hpp:45)
  Bits        support, reject;
  int         consBegin = 0, consEnd = 0;  // known_pos_range2 default = [0,0)
};

// ------------------------------------------------------------------------------------------------
// libstdc++ std::unordered_map<std::string,...> iteration-order emulation.
//
// The reference seeds its repeat-k-mer DFS by iterating a std::unordered_map (assembly/IterativeAssembler.cpp
// :630-641), so its output on cyclic k-mer graphs depends on libstdc++'s node order (SURVEY.md hard part 1).
// The order is a pure function of (key insertion sequence, hash, bucket-count growth schedule):
//   * a new node goes to the FRONT of its bucket's run if the bucket is non-empty, otherwise to the front of
//     the whole list  (bits/hashtable.h _M_insert_bucket_begin)
//   * a rehash re-inserts all nodes, in current list order, into the new bucket array with the same rule
//     (bits/hashtable.h _M_rehash_aux(unique keys))
//   * hash = std::hash<std::string> = _Hash_bytes(ptr,len,0xc70f6907) (libsupc++/hash_bytes.cc, Murmur-style)
//   * the growth schedule is libstdc++'s _Prime_rehash_policy; instead of restating its prime table we
//     record it once from a live std::unordered_map<int,int> of the same libstdc++ (growthSchedule()).
// ------------------------------------------------------------------------------------------------
static inline uint64_t shiftMix(uint64_t v)
{
  return v ^ (v >> 47);
}

uint64_t libstdcxxHashBytes(const char* buf, size_t len)
{
  static const uint64_t mul  = (uint64_t(0xc6a4a793UL) << 32) + uint64_t(0x5bd1e995UL);
  const uint64_t        seed = 0xc70f6907UL;
  const size_t          lenAligned = len & ~size_t(7);
  uint64_t              hash       = seed ^ (len * mul);
  for (size_t p = 0; p < lenAligned; p += 8) {
    uint64_t data;
    std::memcpy(&data, buf + p, 8);
    data = shiftMix(data * mul) * mul;
    hash ^= data;
    hash *= mul;
  }
  if (len & 7) {
    uint64_t data = 0;
    for (int n = int(len & 7) - 1; n >= 0; --n) data = (data << 8) + uint8_t(buf[lenAligned + n]);
    hash ^= data;
    hash *= mul;
  }
  hash = shiftMix(hash) * mul;
  hash = shiftMix(hash);
  return hash;
}

/// (element count that triggers the growth, new bucket count) pairs, recorded from the live libstdc++
struct GrowthStep {
  size_t sizeBefore;  ///< map.size() just before the insertion that rehashes
  size_t buckets;     ///< bucket_count() after it
};
const std::vector<GrowthStep>& growthSchedule()
{
  static const std::vector<GrowthStep> sched = []() {
    std::vector<GrowthStep>      s;
    std::unordered_map<int, int> m;
    size_t                       last = m.bucket_count();
    for (int i = 0; i < 1200000; ++i) {
      m[i] = 0;
      if (m.bucket_count() != last) {
        s.push_back({size_t(i), m.bucket_count()});
        last = m.bucket_count();
      }
    }
    return s;
  }();
  return sched;
}

/// Given keys in insertion order (all distinct), return their indices in libstdc++ iteration order.
std::vector<unsigned> unorderedMapOrder(const std::vector<std::string>& keys)
{
  const size_t          n = keys.size();
  std::vector<uint64_t> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = libstdcxxHashBytes(keys[i].data(), keys[i].size());

  const std::vector<GrowthStep>& sched(growthSchedule());
  // singly linked list over node indices, NIL-terminated
  const int        NIL = -1;
  std::vector<int> next(n, NIL);
  int              head = NIL;
  // bucket -> node BEFORE the bucket's first node; HEAD_SENTINEL means "the list head is the first node"
  const int        EMPTY = -2, HEAD_SENTINEL = -3;
  size_t           nb = 1;
  std::vector<int> before(nb, EMPTY);
  size_t           schedPos = 0;

  auto insertNode = [&](int node) {
    const size_t b = h[node] % nb;
    if (before[b] != EMPTY) {
      if (before[b] == HEAD_SENTINEL) {
        next[node] = head;
        head       = node;
      } else {
        next[node]       = next[before[b]];
        next[before[b]] = node;
      }
    } else {
      next[node] = head;
      head       = node;
      if (next[node] != NIL) before[h[next[node]] % nb] = node;
      before[b] = HEAD_SENTINEL;
    }
  };

  for (size_t i = 0; i < n; ++i) {
    if (schedPos < sched.size() && sched[schedPos].sizeBefore == i) {
      // rehash: re-insert every node in current list order
      nb = sched[schedPos].buckets;
      ++schedPos;
      std::vector<int> order;
      for (int p = head; p != NIL; p = next[p]) order.push_back(p);
      before.assign(nb, EMPTY);
      head = NIL;
      for (int p : order) insertNode(p);
    }
    insertNode(int(i));
  }
  std::vector<unsigned> out;
  out.reserve(n);
  for (int p = head; p != NIL; p = next[p]) out.push_back(unsigned(p));
  return out;
}

// ------------------------------------------------------------------------------------------------
// k-mer graph for one word length
// ------------------------------------------------------------------------------------------------
static const int NONE = -1;

struct KmerGraph {
  unsigned                   k = 0;
  std::vector<std::string>   key;      ///< node -> k-mer, in first-insertion order (= wordCount insertion order)
  std::vector<Bits>          support;  ///< node -> reads containing it        (IterativeAssembler.cpp:547)
  std::vector<unsigned>      count;    ///< node -> wordCount                  (:545)
  std::vector<int>           succ, pred;  ///< 4 per node; NONE if that k-mer does not exist
  std::map<std::string, int> index;    ///< ordered: lexicographic scans for seed selection (:689-696)

  int find(const std::string& w) const
  {
    auto it = index.find(w);
    return (it == index.end()) ? NONE : it->second;
  }
};

/// assembly/IterativeAssembler.cpp:506-550 (getKmerCounts) + link table used in place of the per-step hash
/// lookups of walk() (:242-251, :384-392)
static void buildGraph(
    const AsmOpts& opt, const std::vector<std::string>& reads, const std::vector<bool>& isPseudo, const unsigned k,
    KmerGraph& g)
{
  g              = KmerGraph();
  g.k            = k;
  const size_t nr = reads.size();
  for (unsigned r = 0; r < nr; ++r) {
    const std::string& seq(reads[r]);
    if (seq.size() < k) continue;  // :522
    std::set<std::string> words;   // per-read de-dup, sorted (:525-534)
    for (size_t j = 0; j + k <= seq.size(); ++j) {
      const std::string w(seq.substr(j, k));
      if (w.find('N') != std::string::npos) continue;  // :531
      words.insert(w);
    }
    const unsigned add = isPseudo[r] ? opt.minCoverage : 1;  // :541
    for (const std::string& w : words) {
      int n = g.find(w);
      if (n == NONE) {
        n = int(g.key.size());
        g.key.push_back(w);
        g.support.emplace_back(nr);
        g.count.push_back(0);
        g.index[w] = n;
      }
      g.count[n] += add;
      g.support[n].set(r);
    }
  }
  const size_t nn = g.key.size();
  g.succ.assign(nn * 4, NONE);
  g.pred.assign(nn * 4, NONE);
  for (size_t n = 0; n < nn; ++n) {
    const std::string& w(g.key[n]);
    for (int c = 0; c < 4; ++c) {
      g.succ[n * 4 + c] = g.find(w.substr(1) + ALPHABET[c]);
      g.pred[n * 4 + c] = g.find(ALPHABET[c] + w.substr(0, k - 1));
    }
  }
}

/// assembly/IterativeAssembler.cpp:555-642 (searchRepeats + getRepeatKmers), recursion unrolled.
/// DFS roots are visited in the emulated unordered_map order of `wordIndices`, which is itself filled by
/// iterating `wordCount` (:631-633), i.e. order2 = f(order1), order1 = f(first-insertion order).
static void repeatNodes(const KmerGraph& g, std::vector<bool>& isRepeat)
{
  const size_t nn = g.key.size();
  isRepeat.assign(nn, false);
  if (nn == 0) return;

  std::vector<unsigned> order1 = unorderedMapOrder(g.key);
  std::vector<std::string> keys2(nn);
  for (size_t i = 0; i < nn; ++i) keys2[i] = g.key[order1[i]];
  std::vector<unsigned> order2 = unorderedMapOrder(keys2);
  std::vector<unsigned> rootOrder(nn);
  for (size_t i = 0; i < nn; ++i) rootOrder[i] = order1[order2[i]];

  std::vector<unsigned> idx(nn, 0), low(nn, 0);
  std::vector<bool>     onStack(nn, false);
  std::vector<int>      stack;
  struct Frame {
    int node;
    int sym;
  };
  std::vector<Frame> frames;
  unsigned           nextIndex = 1;

  for (unsigned root : rootOrder) {
    if (idx[root] != 0) continue;
    frames.push_back({int(root), 0});
    idx[root] = low[root] = nextIndex++;
    stack.push_back(root);
    onStack[root] = true;
    while (!frames.empty()) {
      Frame&    f = frames.back();
      const int n = f.node;
      if (f.sym < 4) {
        const int c = f.sym++;
        const int s = g.succ[size_t(n) * 4 + c];
        if (s == n) {  // homopolymer (:574-577)
          isRepeat[n] = true;
          continue;
        }
        if (s == NONE) continue;  // :580
        if (idx[s] == 0) {        // unvisited: recurse (:583-590)
          idx[s] = low[s] = nextIndex++;
          stack.push_back(s);
          onStack[s] = true;
          frames.push_back({s, 0});
        } else if (onStack[s]) {  // :592-598
          low[n] = std::min(low[n], idx[s]);
        }
        continue;
      }
      // all successors done: root test (:603-622)
      if (low[n] == idx[n]) {
        const int last = stack.back();
        if (last == n) {
          stack.pop_back();
          onStack[n] = false;
        } else {
          const bool isSmallCircle((idx[last] - idx[n]) <= 50);
          while (true) {
            const int w = stack.back();
            if (isSmallCircle) isRepeat[w] = true;
            stack.pop_back();
            onStack[w] = false;
            if (w == n) break;
          }
        }
      }
      frames.pop_back();
      if (!frames.empty()) {  // the caller's lowlink update after the recursive call returns (:588-590)
        const int p = frames.back().node;
        low[p]      = std::min(low[p], low[n]);
      }
    }
  }
  if (std::getenv("MANTA_ORACLE_DUMP_REPEATS")) {  // developer aid: compare with the emulator build's MANTA_EMU_DUMP_REPEATS
    std::fprintf(stderr, "ORC repeats k=%u:", g.k);
    for (size_t n = 0; n < g.key.size(); ++n)
      if (isRepeat[n]) std::fprintf(stderr, " %s", g.key[n].c_str());
    std::fprintf(stderr, "\n");
  }
}

/// assembly/IterativeAssembler.cpp:149-501 (walk) in node-id / bitset form
static bool walk(
    const AsmOpts& opt, const KmerGraph& g, const int seed, const std::vector<bool>& isRepeat, std::vector<bool>& isUnused,
    const size_t nbits, Contig& contig)
{
  const unsigned k = g.k;
  contig.support   = g.support[seed];  // :168
  contig.reject    = Bits(nbits);
  contig.seq       = g.key[seed];
  isUnused[seed]   = false;  // :170

  if (isRepeat[seed]) {  // :172-179
    contig.consBegin = 0;
    contig.consEnd   = int(k);
    return true;
  }
  std::vector<bool> inContig(g.key.size(), false);  // wordsInContig (:182)
  inContig[seed] = true;

  // reads of the unselected siblings of the seed become rejecting reads (:185-210)
  {
    const std::string trunk(g.key[seed].substr(0, k - 1));
    for (int c = 0; c < 4; ++c) {
      if (ALPHABET[c] == g.key[seed][k - 1]) continue;
      const int n = g.find(trunk + ALPHABET[c]);
      if (n == NONE) continue;
      bor(contig.reject, g.support[n]);
    }
  }

  bool isRepeatFound = false;
  for (unsigned mode = 0; mode < 2; ++mode) {
    const bool isEnd(mode == 0);
    unsigned   consOffset = 0;
    int        cur        = seed;  // node of the contig's last (mode 0) / first (mode 1) word
    std::string left;               // bases prepended in mode 1, most recent first
    while (true) {
      unsigned maxBaseCount = 0, maxCnt = 0;
      int      maxNode = NONE, maxSym = 0;
      Bits     maxWordReads(nbits), maxCW(nbits), rm(nbits), add(nbits);
      bool     haveMax = false;
      for (int c = 0; c < 4; ++c) {  // :241-336
        const int n = isEnd ? g.succ[size_t(cur) * 4 + c] : g.pred[size_t(cur) * 4 + c];
        if (n == NONE) continue;
        const Bits& cw(g.support[n]);
        const Bits  CW(band(contig.support, cw));
        const Bits  SH(band(maxCW, cw));
        if (!CW.any()) continue;  // :280
        const unsigned cnt = CW.count();
        if (cnt > maxCnt) {  // :283-316
          if (maxCW.any()) bor(rm, bandnot(maxCW, SH));
          if (haveMax) bor(add, bandnot(maxWordReads, SH));
          maxWordReads = cw;
          haveMax      = true;
          maxCnt       = cnt;
          maxCW        = CW;
          maxBaseCount = g.count[n];
          maxSym       = c;
          maxNode      = n;
        } else {  // :317-335
          bor(rm, bandnot(CW, SH));
          bor(add, bandnot(cw, SH));
        }
      }
      if (maxBaseCount < opt.minCoverage) break;  // :343
      if (inContig[maxNode]) {                    // :352-358
        isRepeatFound = true;
        break;
      }
      if (isEnd)
        contig.seq.push_back(ALPHABET[maxSym]);  // :363
      else
        contig.seq.insert(contig.seq.begin(), ALPHABET[maxSym]);
      if ((consOffset != 0) || (maxBaseCount < opt.minConservativeCoverage)) consOffset += 1;  // :368-369

      // "walk backwards one step": previousWordReads is re-declared empty every iteration (:237), so this block
      // runs at every extension (:377-427)
      {
        const char skipSym = isEnd ? g.key[cur][0] : g.key[cur][k - 1];  // :378
        for (int c = 0; c < 4; ++c) {
          if (ALPHABET[c] == skipSym) continue;
          const int n = isEnd ? g.pred[size_t(maxNode) * 4 + c] : g.succ[size_t(maxNode) * 4 + c];  // :384
          if (n == maxNode) continue;                                                                  // :389
          if (n == NONE) continue;
          const Bits upd(bandnot(g.support[n], band(maxCW, g.support[n])));  // :400-414
          bor(add, upd);
          bor(rm, upd);
        }
      }
      bor(contig.reject, add);                                     // :440-442
      bor(contig.support, bandnot(maxWordReads, contig.reject));   // :458-464
      contig.support = bandnot(contig.support, rm);                // :471-473
      isUnused[maxNode] = false;                                   // :482
      inContig[maxNode] = true;                                    // :484
      cur               = maxNode;
    }
    if (mode == 0)
      contig.consEnd = int(consOffset);  // :488-491
    else
      contig.consBegin = int(consOffset);
  }
  contig.consEnd = int(contig.seq.size()) - contig.consEnd;  // :498
  return isRepeatFound;
}

/// assembly/IterativeAssembler.cpp:644-720
static bool buildContigs(
    const AsmOpts& opt, const std::vector<std::string>& reads, const std::vector<bool>& isPseudo, const unsigned k,
    std::vector<Contig>& contigs)
{
  contigs.clear();
  KmerGraph g;
  buildGraph(opt, reads, isPseudo, k, g);
  std::vector<bool> isRepeat;
  repeatNodes(g, isRepeat);

  const size_t      nn = g.key.size();
  std::vector<bool> isUnused(nn, false);
  size_t            unusedCount = 0;
  for (size_t n = 0; n < nn; ++n) {
    if (g.count[n] >= opt.minCoverage) {  // :681
      isUnused[n] = true;
    }
  }
  bool isAssemblySuccess = true;
  while (contigs.size() < 2 * size_t(opt.maxAssemblyCount)) {  // :685
    // highest count, ties -> lexicographically smallest (ordered scan with strict '>', :689-696)
    int      seed = NONE;
    unsigned best = 0;
    unusedCount   = 0;
    for (const auto& kv : g.index) {
      if (!isUnused[kv.second]) continue;
      ++unusedCount;
      if (g.count[kv.second] > best) {
        best = g.count[kv.second];
        seed = kv.second;
      }
    }
    if (unusedCount == 0) break;
    Contig contig;
    if (seed == NONE) {
      // all remaining unused words have count 0 (only possible with minCoverage==0): the reference would walk from
      // the empty string; not reachable with supported options
      break;
    }
    if (walk(opt, g, seed, isRepeat, isUnused, reads.size(), contig)) isAssemblySuccess = false;  // :700-702
    contigs.push_back(contig);
  }
  return isAssemblySuccess;
}

struct ReadInfo {
  bool                  isUsed = false, isFiltered = false, isPseudo = false;
  std::vector<unsigned> contigIds;
};

static Bits resized(const Bits& b, size_t nbits)
{
  Bits r(nbits);
  for (size_t i = 0; i < std::min(r.w.size(), b.w.size()); ++i) r.w[i] = b.w[i];
  return r;
}

/// assembly/IterativeAssembler.cpp:722-842
///
/// Stale pseudo-read indices: when the LAST word length also hits a repeat, the reference truncates `reads`/`readInfo`
/// (:882-893) and appends the new pseudo reads (:897-910) AFTER the final contigs were built, so contig support sets can
/// hold indices >= readInfo.size() (or indices that now name a different pseudo read).  `readInfo[rd]` at :776/:828 is
/// then an out-of-bounds read in the reference (undefined behaviour).  De facto (libstdc++ vector::erase keeps the
/// storage, the stale slots still hold destroyed pseudo-read entries) such an index behaves as "a pseudo read whose
/// readInfo update is invisible"; that is what is restated here and in the HIP path: index >= normalReadCount <=> pseudo.
static void selectContigs(
    const AsmOpts& opt, std::vector<ReadInfo>& readInfo, const unsigned normalReadCount, std::vector<Contig> candidates,
    std::vector<Contig>& finalContigs)
{
  finalContigs.clear();
  size_t nbits = readInfo.size();
  for (const Contig& c : candidates) nbits = std::max(nbits, c.support.w.size() * 64);
  Bits used(nbits), usedPseudo(nbits), pseudoMask(nbits);
  for (size_t r = normalReadCount; r < nbits; ++r) pseudoMask.set(r);
  for (Contig& c : candidates) {
    c.support = resized(c.support, nbits);
    c.reject  = resized(c.reject, nbits);
  }
  unsigned finalCount = 0;
  while (!candidates.empty() && finalCount < opt.maxAssemblyCount) {
    const unsigned usedNormal   = used.count() - usedPseudo.count();
    const unsigned unusedNormal = normalReadCount - usedNormal;
    if (unusedNormal < opt.minUnusedReads) return;  // :750

    std::vector<bool> remove(candidates.size(), false);
    int               selected   = NONE;
    unsigned          maxSupport = 0, maxLength = 0;
    for (size_t ci = 0; ci < candidates.size(); ++ci) {
      const Bits     fresh(bandnot(candidates[ci].support, used));
      const unsigned freshNormal = bandnot(fresh, pseudoMask).count();
      if (freshNormal < opt.minSupportReads) {  // :779-788
        remove[ci] = true;
        continue;
      }
      const unsigned sup = fresh.count();
      const unsigned len = candidates[ci].seq.size();
      if ((sup > maxSupport) || ((sup == maxSupport) && (len > maxLength))) {  // :794-801
        selected   = int(ci);
        maxSupport = sup;
        maxLength  = len;
      }
    }
    if (maxSupport == 0) break;  // :807
    const Contig chosen(candidates[selected]);
    finalContigs.push_back(chosen);
    remove[selected] = true;
    for (size_t ci = candidates.size(); ci-- > 0;)
      if (remove[ci]) candidates.erase(candidates.begin() + ci);  // :817-820
    for (unsigned r : chosen.support.members()) {  // :826-834
      used.set(r);
      if (r < readInfo.size()) {
        readInfo[r].isUsed = true;
        readInfo[r].contigIds.push_back(finalCount);
      }
      if (r >= normalReadCount) usedPseudo.set(r);
    }
    finalCount++;
  }
}

/// assembly/IterativeAssembler.cpp:844-931
void runIterativeAssembler(
    const AsmOpts& opt, std::vector<std::string>& reads, std::vector<ReadInfo>& readInfo, std::vector<Contig>& contigs)
{
  const unsigned normalReadCount = reads.size();
  readInfo.assign(reads.size(), ReadInfo());
  std::vector<Contig> iterative;
  for (unsigned k = opt.minWordLength; k <= opt.maxWordLength; k += opt.wordStepSize) {
    std::vector<bool> isPseudo(reads.size());
    for (size_t r = 0; r < reads.size(); ++r) isPseudo[r] = readInfo[r].isPseudo;
    const bool ok = buildContigs(opt, reads, isPseudo, k, iterative);
    if (ok) break;  // :872-877
    for (size_t r = 0; r < reads.size(); ++r) {  // :882-893
      if (readInfo[r].isPseudo) {
        reads.resize(r);
        readInfo.resize(r);
        break;
      }
    }
    for (const Contig& c : iterative) {  // :897-910
      if (c.seq.size() > (k + opt.wordStepSize)) {
        reads.push_back(c.seq);
        ReadInfo ri;
        ri.isPseudo = true;
        readInfo.push_back(ri);
      }
    }
  }
  selectContigs(opt, readInfo, normalReadCount, iterative, contigs);
}

// ------------------------------------------------------------------------------------------------
// aligners
// ------------------------------------------------------------------------------------------------
enum State { MATCH = 0, DELETE = 1, INSERT = 2, JUMP = 3, JUMPINS = 4 };  // alignment/Alignment.hpp:47-56
enum Seg { S_NONE, S_MATCH, S_INSERT, S_DELETE, S_SKIP, S_SOFT_CLIP, S_HARD_CLIP, S_PAD, S_SEQ_MATCH, S_SEQ_MISMATCH };
static const char SEGCODE[] = {'?', 'M', 'I', 'D', 'N', 'S', 'H', 'P', '=', 'X'};  // blt_util/align_path.hpp:35-62

struct Scores {
  int  match, mismatch, open, extend, offEdge;
  bool isAllowEdgeInsertion;
};
struct PathSeg {
  int      type;
  unsigned length;
};
typedef std::vector<PathSeg> Path;

static std::string cigar(const Path& p)
{
  std::ostringstream os;
  for (const PathSeg& s : p) os << s.length << SEGCODE[s.type];
  return os.str();
}
static unsigned readLength(const Path& p)
{
  unsigned v = 0;
  for (const PathSeg& s : p)
    if (s.type == S_MATCH || s.type == S_INSERT || s.type == S_SOFT_CLIP || s.type == S_SEQ_MATCH || s.type == S_SEQ_MISMATCH)
      v += s.length;
  return v;
}
static unsigned refLength(const Path& p)
{
  unsigned v = 0;
  for (const PathSeg& s : p)
    if (s.type == S_MATCH || s.type == S_DELETE || s.type == S_SKIP || s.type == S_SEQ_MATCH || s.type == S_SEQ_MISMATCH)
      v += s.length;
  return v;
}

/// blt_util/align_path_impl.hpp:33-72
static void addSeqMatch(const char* q, const char* qEnd, const char* r, const char* rEnd, Path& path)
{
  Path out;
  for (const PathSeg& ps : path) {
    if (ps.type == S_MATCH || ps.type == S_SEQ_MATCH || ps.type == S_SEQ_MISMATCH) {
      for (unsigned i = 0; i < ps.length; ++i) {
        if (q >= qEnd) throw std::runtime_error("apath_add_seqmatch: past end of query\n");
        if (r >= rEnd) throw std::runtime_error("apath_add_seqmatch: past end of reference\n");
        bool same = (*q == *r);
        if (*q == 'N' || *r == 'N') same = false;
        const int t = same ? S_SEQ_MATCH : S_SEQ_MISMATCH;
        if (!out.empty() && out.back().type == t)
          out.back().length++;
        else
          out.push_back({t, 1});
        ++q;
        ++r;
      }
    } else {
      out.push_back(ps);
      if (ps.type == S_INSERT || ps.type == S_SOFT_CLIP) q += ps.length;
      if (ps.type == S_DELETE || ps.type == S_SKIP) r += ps.length;
    }
  }
  path = out;
}

/// first-max-wins argmax (alignment/AlignerBase.hpp:46-59, JumpAlignerBase.hpp:93-111,
/// GlobalLargeIndelAligner.hpp:124-151): strict '>' scanning in state order
static int argmaxFirst(const int* v, int n, int& best)
{
  best  = v[0];
  int p = 0;
  for (int i = 1; i < n; ++i)
    if (v[i] > best) {
      best = v[i];
      p    = i;
    }
  return p;
}

static const int BAD = -10000;  // finite sentinel that takes part in arithmetic (GlobalJumpAlignerImpl.hpp:68)

struct BackTrace {  // alignment/AlignerUtil.hpp:41-67
  int      max   = 0;
  int      state = MATCH;
  unsigned q = 0, r = 0;
  bool     isInit = false;
  void     update(int v, unsigned ref, unsigned query, int st = MATCH)
  {
    if (!isInit || v > max) {
      max    = v;
      r      = ref;
      q      = query;
      isInit = true;
      state  = st;
    }
  }
};

struct AlignResult {
  int      score = 0;
  bool     isJumped = false;
  int      begin1 = 0, begin2 = 0;
  Path     path1, path2;
  unsigned jumpInsertSize = 0, jumpRange = 0;
};

static void pushSeg(Path& path, PathSeg& ps, int type)  // AlignerUtil::updatePath (AlignerUtil.hpp:31-38)
{
  if (ps.type == type) return;
  if (ps.type != S_NONE) path.push_back(ps);
  ps.type   = type;
  ps.length = 0;
}

/// Single-reference aligners: kind 0 = GlobalAligner (GlobalAlignerImpl.hpp:29-181),
/// kind 1 = GlobalLargeIndelAligner (GlobalLargeIndelAlignerImpl.hpp:35-225); shared traceback
/// SingleRefAlignerSharedImpl.hpp:75-168.
static void alignSingleRef(
    const int kind, const Scores& sc, const int L, const std::string& query, const std::string& ref, AlignResult& res)
{
  res = AlignResult();
  const size_t Q = query.size(), R = ref.size();
  if (Q == 0) throw std::runtime_error("Unexpected empty query sequence");
  if (R == 0) throw std::runtime_error("Unexpected empty reference sequence");
  const int NS = (kind == 0) ? 3 : 5;
  struct Cell {
    int v[5];
  };
  std::vector<Cell>    prev(Q + 1), cur(Q + 1);
  std::vector<uint8_t> ptr((Q + 1) * (R + 1) * 5, 0);
  auto                 P = [&](size_t q, size_t r, int s) -> uint8_t& { return ptr[((r * (Q + 1)) + q) * 5 + s]; };

  for (size_t q = 0; q <= Q; ++q) {  // row 0
    Cell& c(cur[q]);
    for (int s = 0; s < 5; ++s) c.v[s] = BAD;
    c.v[MATCH] = int(unsigned(q) * unsigned(sc.offEdge));
    if (sc.isAllowEdgeInsertion) {
      P(q, 0, INSERT) = INSERT;
      c.v[INSERT]     = sc.open + int(unsigned(q) * unsigned(sc.extend));
    }
  }
  BackTrace bt;
  for (size_t r = 1; r <= R; ++r) {
    std::swap(prev, cur);
    {
      Cell& c(cur[0]);
      for (int s = 0; s < 5; ++s) c.v[s] = BAD;
      c.v[MATCH] = 0;
    }
    for (size_t q = 1; q <= Q; ++q) {
      Cell&       h(cur[q]);
      const Cell& diag(prev[q - 1]);
      const Cell& up(prev[q]);
      const Cell& left(cur[q - 1]);
      int         best;
      {  // match
        const int v[5]  = {diag.v[MATCH], diag.v[DELETE], diag.v[INSERT], diag.v[JUMP], diag.v[JUMPINS]};
        P(q, r, MATCH)  = argmaxFirst(v, NS, best);
        h.v[MATCH]      = best + ((query[q - 1] == ref[r - 1]) ? sc.match : sc.mismatch);
      }
      if (kind == 0) {
        {
          const int v[3]  = {up.v[MATCH] + sc.open, up.v[DELETE], up.v[INSERT]};
          P(q, r, DELETE) = argmaxFirst(v, 3, best);
          h.v[DELETE]     = best + sc.extend;
          if (q == 1) h.v[DELETE] = BAD;
        }
        {
          const int v[3]  = {left.v[MATCH] + sc.open, BAD, left.v[INSERT]};
          P(q, r, INSERT) = argmaxFirst(v, 3, best);
          h.v[INSERT]     = best + sc.extend;
          if (q == 1) h.v[INSERT] = BAD;
        }
      } else {
        {
          const int v[5]  = {up.v[MATCH] + sc.open, up.v[DELETE], up.v[INSERT], BAD, up.v[JUMPINS]};
          P(q, r, DELETE) = argmaxFirst(v, 5, best);
          h.v[DELETE]     = best + sc.extend;
          if (q == 1) h.v[DELETE] = BAD;
        }
        {
          const int v[5]  = {left.v[MATCH] + sc.open, BAD, left.v[INSERT], BAD, BAD};
          P(q, r, INSERT) = argmaxFirst(v, 5, best);
          h.v[INSERT]     = best + sc.extend;
          if (q == 1) h.v[INSERT] = BAD;
        }
        {
          const int v[5] = {up.v[MATCH] + L, BAD, up.v[INSERT] + L - sc.open, up.v[JUMP], up.v[JUMPINS] + L};
          P(q, r, JUMP)  = argmaxFirst(v, 5, best);
          h.v[JUMP]      = best;
          if (q == 1) h.v[JUMP] = BAD;
        }
        {
          const int v[5]   = {left.v[MATCH] + L, BAD, BAD, BAD, left.v[JUMPINS]};
          P(q, r, JUMPINS) = argmaxFirst(v, 5, best);
          h.v[JUMPINS]     = best;
          if (q == 1) h.v[JUMPINS] = BAD;
        }
      }
    }
    bt.update(cur[Q].v[MATCH], unsigned(r), unsigned(Q));
  }
  if (sc.isAllowEdgeInsertion) bt.update(cur[Q].v[INSERT], unsigned(R), unsigned(Q), INSERT);
  // off-edge candidates: q<Q for GlobalAligner (GlobalAlignerImpl.hpp:165), q<=Q for LargeIndel (:211)
  const size_t qLimit = (kind == 0) ? Q : Q + 1;
  for (size_t q = 0; q < qLimit; ++q) {
    bt.update(cur[q].v[MATCH] + int(unsigned(Q - q) * unsigned(sc.offEdge)), unsigned(R), unsigned(q));
  }

  // traceback
  res.score = bt.max;
  Path&   path(res.path1);
  PathSeg ps{S_NONE, 0};
  if (bt.q < Q) ps = {S_SOFT_CLIP, unsigned(Q - bt.q)};
  while (true) {
    const int nextState = P(bt.q, bt.r, bt.state);
    if (bt.state == MATCH) {
      if (bt.q < 1 || bt.r < 1) break;
      pushSeg(path, ps, S_MATCH);
      bt.q--;
      bt.r--;
    } else if (bt.state == DELETE || bt.state == JUMP) {
      if (bt.r < 1) break;
      pushSeg(path, ps, S_DELETE);
      bt.r--;
    } else {
      if (bt.q < 1) break;
      pushSeg(path, ps, S_INSERT);
      bt.q--;
    }
    if (bt.state == JUMP || bt.state == JUMPINS) res.isJumped = true;
    bt.state = nextState;
    ps.length++;
  }
  if (ps.type != S_NONE) path.push_back(ps);
  if (bt.q != 0) path.push_back({S_SOFT_CLIP, bt.q});
  res.begin1 = int(bt.r);
  std::reverse(path.begin(), path.end());
  addSeqMatch(query.data(), query.data() + Q, ref.data() + res.begin1, ref.data() + R, path);
}

/// GlobalJumpAligner (GlobalJumpAlignerImpl.hpp:33-333) + traceback/jumpRange (JumpAlignerBaseImpl.hpp:86-242)
static void alignJump(
    const Scores& sc, const int J, const std::string& query, const std::string& ref1, const std::string& ref2,
    AlignResult& res)
{
  res = AlignResult();
  const size_t Q = query.size(), R1 = ref1.size(), R2 = ref2.size();
  if (Q == 0) throw std::runtime_error("Unexpected empty query sequence");
  if (R1 == 0) throw std::runtime_error("Unexpected empty reference1 sequence");
  if (R2 == 0) throw std::runtime_error("Unexpected empty reference2 sequence");
  struct Cell {
    int v[4];
  };
  std::vector<Cell>    prev(Q + 1), cur(Q + 1);
  std::vector<uint8_t> ptr((Q + 1) * (R1 + R2 + 2) * 4, 0);
  // rows 0..R1 = matrix 1, rows R1+1 .. R1+R2+1 = matrix 2 (its own row 0 first)
  auto P1 = [&](size_t q, size_t r, int s) -> uint8_t& { return ptr[((r * (Q + 1)) + q) * 4 + s]; };
  auto P2 = [&](size_t q, size_t r, int s) -> uint8_t& { return ptr[(((R1 + 1 + r) * (Q + 1)) + q) * 4 + s]; };

  for (size_t q = 0; q <= Q; ++q) {
    Cell& c(cur[q]);
    c.v[MATCH]  = int(unsigned(q) * unsigned(sc.offEdge));
    c.v[DELETE] = c.v[INSERT] = c.v[JUMP] = BAD;
  }
  BackTrace bt;
  for (size_t r = 1; r <= R1; ++r) {
    std::swap(prev, cur);
    cur[0].v[MATCH]  = 0;
    cur[0].v[DELETE] = cur[0].v[INSERT] = cur[0].v[JUMP] = BAD;
    for (size_t q = 1; q <= Q; ++q) {
      Cell&       h(cur[q]);
      const Cell& diag(prev[q - 1]);
      const Cell& up(prev[q]);
      const Cell& left(cur[q - 1]);
      int         best;
      {
        const int v[3]  = {diag.v[MATCH], diag.v[DELETE], diag.v[INSERT]};
        P1(q, r, MATCH) = argmaxFirst(v, 3, best);
        h.v[MATCH]      = best + ((query[q - 1] == ref1[r - 1]) ? sc.match : sc.mismatch);
      }
      {
        const int v[3]   = {up.v[MATCH] + sc.open, up.v[DELETE], up.v[INSERT]};
        P1(q, r, DELETE) = argmaxFirst(v, 3, best);
        h.v[DELETE]      = best + sc.extend;
        if (q == 1) h.v[DELETE] = BAD;
      }
      {
        const int v[3]   = {left.v[MATCH] + sc.open, BAD, left.v[INSERT]};
        P1(q, r, INSERT) = argmaxFirst(v, 3, best);
        h.v[INSERT]      = best + sc.extend;
        if (q == 1) h.v[INSERT] = BAD;
      }
      {  // uses THIS cell's final match/ins (:153-161)
        const int v[4] = {h.v[MATCH] + J, BAD, h.v[INSERT] + J, up.v[JUMP]};
        P1(q, r, JUMP) = argmaxFirst(v, 4, best);
        h.v[JUMP]      = best;
      }
    }
    bt.update(cur[Q].v[MATCH], unsigned(r), unsigned(Q));
  }
  for (size_t q = 0; q < Q; ++q) {  // :181-186
    bt.update(cur[q].v[MATCH] + int(unsigned(Q - q) * unsigned(sc.offEdge)), unsigned(R1), unsigned(q));
  }
  for (size_t q = 0; q <= Q; ++q) {  // seam: jump preserved (:197-204)
    cur[q].v[MATCH]  = int(unsigned(q) * unsigned(sc.offEdge));
    cur[q].v[DELETE] = cur[q].v[INSERT] = BAD;
  }
  for (size_t r = 1; r <= R2; ++r) {
    std::swap(prev, cur);
    cur[0].v[MATCH]  = 0;
    cur[0].v[DELETE] = cur[0].v[INSERT] = cur[0].v[JUMP] = BAD;
    for (size_t q = 1; q <= Q; ++q) {
      Cell&       h(cur[q]);
      const Cell& diag(prev[q - 1]);
      const Cell& up(prev[q]);
      const Cell& left(cur[q - 1]);
      int         best;
      {
        const int v[4]  = {diag.v[MATCH], diag.v[DELETE], diag.v[INSERT], diag.v[JUMP]};
        P2(q, r, MATCH) = argmaxFirst(v, 4, best);
        h.v[MATCH]      = best + ((query[q - 1] == ref2[r - 1]) ? sc.match : sc.mismatch);
      }
      {
        const int v[3]   = {up.v[MATCH] + sc.open, up.v[DELETE], up.v[INSERT]};
        P2(q, r, DELETE) = argmaxFirst(v, 3, best);
        h.v[DELETE]      = best + sc.extend;
      }
      {
        const int v[4]   = {left.v[MATCH] + sc.open, BAD, left.v[INSERT], left.v[JUMP]};
        P2(q, r, INSERT) = argmaxFirst(v, 4, best);
        h.v[INSERT]      = best + sc.extend;
      }
      P2(q, r, JUMP) = JUMP;
      h.v[JUMP]      = up.v[JUMP];
    }
    bt.update(cur[Q].v[MATCH], unsigned(R1 + r), unsigned(Q));
  }
  for (size_t q = 0; q < Q; ++q) {
    bt.update(cur[q].v[MATCH] + int(unsigned(Q - q) * unsigned(sc.offEdge)), unsigned(R1 + R2), unsigned(q));
  }

  // traceback (JumpAlignerBaseImpl.hpp:120-198)
  res.score = bt.max;
  PathSeg ps{S_NONE, 0};
  if (bt.q < Q) ps = {S_SOFT_CLIP, unsigned(Q - bt.q)};
  bool isRef2End = false;
  while (bt.q > 0 && bt.r > 0) {
    if (isRef2End) break;
    const bool     isRef1(bt.r <= R1);
    Path&          path(isRef1 ? res.path1 : res.path2);
    const unsigned rx = bt.r - (isRef1 ? 0 : unsigned(R1));
    const int      nextState = isRef1 ? P1(bt.q, rx, bt.state) : P2(bt.q, rx, bt.state);
    if (bt.state == MATCH) {
      if (!isRef1 && rx == 1 && nextState == MATCH) isRef2End = true;
      pushSeg(path, ps, S_MATCH);
      bt.q--;
      bt.r--;
    } else if (bt.state == DELETE) {
      pushSeg(path, ps, S_DELETE);
      bt.r--;
    } else if (bt.state == INSERT) {
      pushSeg(path, ps, S_INSERT);
      bt.q--;
    } else {  // JUMP
      if (ps.type != S_NONE) {
        res.begin2 = int(bt.r - R1);
        if (ps.type == S_INSERT) {
          res.jumpInsertSize += ps.length;
          ps = {S_NONE, 0};
        } else {
          pushSeg(res.path2, ps, S_NONE);
        }
      } else {
        if (nextState == JUMP) bt.r--;
      }
    }
    bt.state = nextState;
    ps.length++;
  }
  const bool isRef1(bt.r < R1);
  Path&      path(isRef1 ? res.path1 : res.path2);
  if (ps.type != S_NONE) path.push_back(ps);
  if (bt.q != 0) path.push_back({S_SOFT_CLIP, bt.q});
  if (isRef1)
    res.begin1 = int(bt.r);
  else
    res.begin2 = int(bt.r - R1);
  std::reverse(res.path1.begin(), res.path1.end());
  std::reverse(res.path2.begin(), res.path2.end());

  if (!res.path1.empty() && !res.path2.empty()) {  // jumpRange (:204-230)
    size_t   i1 = size_t(res.begin1) + refLength(res.path1);
    size_t   i2 = size_t(res.begin2);
    size_t   iq = readLength(res.path1);
    unsigned insCount = res.jumpInsertSize;
    while (true) {
      if (i1 >= R1) break;  // (path 1 can overrun ref1 when offEdge is 0; the reference reads past its string there: taken as a mismatch)
      if (insCount > 0) {
        if (iq == Q) break;
        if (ref1[i1] != query[iq]) break;
      } else {
        if (i2 == R2) break;
        if (ref1[i1] != ref2[i2]) break;
      }
      res.jumpRange++;
      i1++;
      if (insCount > 0) {
        insCount--;
        iq++;
      } else {
        i2++;
      }
    }
  }
  addSeqMatch(query.data(), query.data() + Q, ref1.data() + res.begin1, ref1.data() + R1, res.path1);
  const unsigned qoff = readLength(res.path1) + res.jumpInsertSize;
  addSeqMatch(query.data() + qoff, query.data() + Q, ref2.data() + res.begin2, ref2.data() + R2, res.path2);
}

// ------------------------------------------------------------------------------------------------
// 10-mer reference trim of getSmallSVAssembly (applications/GenerateSVCandidates/SVCandidateAssemblyRefiner.cpp:1984-2011)
// ------------------------------------------------------------------------------------------------
void trimRefTo10merHits(
    const std::string& contig, const std::string& ref, const int leadingCut, const int trailingCut, const int maxLeadingCut,
    const int maxTrailingCut, int& adjLead, int& adjTrail)
{
  static const int      merSize = 10;
  std::set<std::string> mers;
  const unsigned        csize = contig.size();
  for (unsigned i = 0; i < (csize - (merSize - 1)); ++i) mers.insert(contig.substr(i, merSize));
  const int refSize        = int(ref.size());
  const int minRefIndex    = leadingCut;
  const int maxRefIndex    = refSize - (trailingCut + merSize);
  const int maxFwdRefIndex = std::min(maxLeadingCut, maxRefIndex);
  int       i;
  for (i = minRefIndex; i <= maxFwdRefIndex; ++i)
    if (mers.count(ref.substr(i, merSize))) break;
  adjLead                 = i;
  const int minRevRefIndex = std::max(minRefIndex, refSize - maxTrailingCut);
  for (i = maxRefIndex; i >= minRevRefIndex; --i)
    if (mers.count(ref.substr(i, merSize))) break;
  adjTrail = refSize - (i + merSize);
}

// ------------------------------------------------------------------------------------------------
// canonical text
// ------------------------------------------------------------------------------------------------
static void joinBits(std::ostream& os, const Bits& b)
{
  bool first = true;
  for (unsigned v : b.members()) {
    if (!first) os << ',';
    os << v;
    first = false;
  }
}

std::string assemblyText(
    const unsigned normalReadCount, const std::vector<std::string>& reads, const std::vector<ReadInfo>& readInfo,
    const std::vector<Contig>& contigs)
{
  std::ostringstream os;
  os << "contigs " << contigs.size() << '\n';
  for (size_t i = 0; i < contigs.size(); ++i) {
    const Contig& c(contigs[i]);
    os << "contig " << i << " seq=" << c.seq << " seed=" << c.seedReadCount << " cons=" << c.consBegin << ',' << c.consEnd
       << " support=";
    joinBits(os, c.support);
    os << " reject=";
    joinBits(os, c.reject);
    os << '\n';
  }
  os << "reads " << readInfo.size() << " normal " << normalReadCount << '\n';
  for (size_t i = 0; i < readInfo.size(); ++i) {
    const ReadInfo& r(readInfo[i]);
    os << "read " << i << " used=" << r.isUsed << " filtered=" << r.isFiltered << " pseudo=" << r.isPseudo << " ids=";
    for (size_t j = 0; j < r.contigIds.size(); ++j) {
      if (j) os << ',';
      os << r.contigIds[j];
    }
    os << '\n';
  }
  for (size_t i = normalReadCount; i < reads.size(); ++i) os << "pseudo " << i << " seq=" << reads[i] << '\n';
  return os.str();
}

std::string alignText(const AlignResult& r)
{
  std::ostringstream os;
  os << "score=" << r.score << " jumped=" << r.isJumped << " begin=" << r.begin1 << " cigar=" << cigar(r.path1) << '\n';
  return os.str();
}
std::string jumpText(const AlignResult& r)
{
  std::ostringstream os;
  os << "score=" << r.score << " jumpInsertSize=" << r.jumpInsertSize << " jumpRange=" << r.jumpRange << " begin1=" << r.begin1
     << " cigar1=" << cigar(r.path1) << " begin2=" << r.begin2 << " cigar2=" << cigar(r.path2) << '\n';
  return os.str();
}

std::string smallSvLocus(
    const AsmOpts& opt, const Scores& sc, const int L, std::vector<std::string>& reads, const std::string& ref,
    const int leadingCut, const int trailingCut, const int maxLeadingCut, const int maxTrailingCut, const bool wantText)
{
  const unsigned        normalReadCount = reads.size();
  std::vector<ReadInfo> info;
  std::vector<Contig>   contigs;
  runIterativeAssembler(opt, reads, info, contigs);
  std::ostringstream os;
  if (wantText) os << assemblyText(normalReadCount, reads, info, contigs);
  for (size_t ci = 0; ci < contigs.size(); ++ci) {
    int adjLead, adjTrail;
    trimRefTo10merHits(contigs[ci].seq, ref, leadingCut, trailingCut, maxLeadingCut, maxTrailingCut, adjLead, adjTrail);
    AlignResult res;
    alignSingleRef(1, sc, L, contigs[ci].seq, ref.substr(adjLead, ref.size() - adjLead - adjTrail), res);
    res.begin1 += adjLead;
    if (wantText) os << "align " << ci << " lead=" << adjLead << " trail=" << adjTrail << ' ' << alignText(res);
  }
  return os.str();
}

}  // namespace orc

// ------------------------------------------------------------------------------------------------
// C entry points (same shapes as oracle/ref_driver.cpp, prefix orc_)
// ------------------------------------------------------------------------------------------------
namespace {
int emit(const std::string& s, char* out, int cap)
{
  const int n = int(s.size());
  if (out && cap > 0) {
    const int m = (n < cap - 1) ? n : (cap - 1);
    std::memcpy(out, s.data(), m);
    out[m] = '\0';
  }
  return n;
}
orc::AsmOpts makeOpt(const uint32_t* o)
{
  return orc::AsmOpts{o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[8]};
}
orc::Scores makeScores(const int32_t* s)
{
  return orc::Scores{s[0], s[1], s[2], s[3], s[4], s[5] != 0};
}
}  // namespace

extern "C" {

int orc_assemble(
    const uint32_t* opts, int n_reads, const char* const* reads, const uint32_t* read_lens, char* out, int cap)
{
  try {
    std::vector<std::string> in;
    for (int i = 0; i < n_reads; ++i) in.emplace_back(reads[i], read_lens[i]);
    std::vector<orc::ReadInfo> info;
    std::vector<orc::Contig>   contigs;
    orc::runIterativeAssembler(makeOpt(opts), in, info, contigs);
    return emit(orc::assemblyText(unsigned(n_reads), in, info, contigs), out, cap);
  } catch (const std::exception& e) {
    emit(std::string("EXCEPTION ") + e.what(), out, cap);
    return -1;
  }
}

int orc_align(
    int kind, const int32_t* scores, int32_t extra, const char* q, int qlen, const char* r1, int r1len, const char* r2,
    int r2len, char* out, int cap)
{
  try {
    const orc::Scores sc(makeScores(scores));
    const std::string query(q, qlen), ref1(r1, r1len), ref2(r2 ? r2 : "", r2 ? r2len : 0);
    orc::AlignResult  res;
    if (kind == 0 || kind == 1) {
      orc::alignSingleRef(kind, sc, extra, query, ref1, res);
      return emit(orc::alignText(res), out, cap);
    } else if (kind == 2) {
      orc::alignJump(sc, extra, query, ref1, ref2, res);
      return emit(orc::jumpText(res), out, cap);
    }
    emit("EXCEPTION unknown aligner kind", out, cap);
    return -1;
  } catch (const std::exception& e) {
    emit(std::string("EXCEPTION ") + e.what(), out, cap);
    return -1;
  }
}

int orc_small_sv_locus(
    const uint32_t* opts, const int32_t* scores, int32_t largeIndelScore, int n_reads, const char* const* reads,
    const uint32_t* read_lens, const char* ref, int ref_len, int leadingCut, int trailingCut, int maxLeadingCut,
    int maxTrailingCut, char* out, int cap)
{
  try {
    std::vector<std::string> in;
    for (int i = 0; i < n_reads; ++i) in.emplace_back(reads[i], read_lens[i]);
    return emit(
        orc::smallSvLocus(
            makeOpt(opts), makeScores(scores), largeIndelScore, in, std::string(ref, ref_len), leadingCut, trailingCut,
            maxLeadingCut, maxTrailingCut, true),
        out, cap);
  } catch (const std::exception& e) {
    emit(std::string("EXCEPTION ") + e.what(), out, cap);
    return -1;
  }
}

double orc_bench_small_sv(
    const uint32_t* opts, const int32_t* scores, int32_t largeIndelScore, int n_loci, const char* bases,
    const uint64_t* read_off, const uint32_t* locus_read_begin, const char* refs, const uint64_t* ref_off, int leadingCut,
    int trailingCut, int maxLeadingCut, int maxTrailingCut, int n_threads, uint64_t* n_done)
{
  const orc::AsmOpts    opt(makeOpt(opts));
  const orc::Scores     sc(makeScores(scores));
  std::atomic<int>      next(0);
  std::atomic<uint64_t> done(0);
  auto                  worker = [&]() {
    while (true) {
      const int li = next.fetch_add(1);
      if (li >= n_loci) break;
      std::vector<std::string> in;
      for (uint32_t r = locus_read_begin[li]; r < locus_read_begin[li + 1]; ++r)
        in.emplace_back(bases + read_off[r], read_off[r + 1] - read_off[r]);
      const std::string ref(refs + ref_off[li], ref_off[li + 1] - ref_off[li]);
      orc::smallSvLocus(opt, sc, largeIndelScore, in, ref, leadingCut, trailingCut, maxLeadingCut, maxTrailingCut, false);
      done += 1;
    }
  };
  const auto               t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> pool;
  for (int t = 1; t < n_threads; ++t) pool.emplace_back(worker);
  worker();
  for (auto& th : pool) th.join();
  const auto t1 = std::chrono::steady_clock::now();
  if (n_done) *n_done = done.load();
  return std::chrono::duration<double>(t1 - t0).count();
}

/// orc_bench_small_sv with the thread harness of bench_harness.hpp (see ref_driver.cpp: ref_bench_small_sv_timed)
double orc_bench_small_sv_timed(
    const uint32_t* opts, const int32_t* scores, int32_t largeIndelScore, int n_loci, const char* bases,
    const uint64_t* read_off, const uint32_t* locus_read_begin, const char* refs, const uint64_t* ref_off, int leadingCut,
    int trailingCut, int maxLeadingCut, int maxTrailingCut, int n_threads, uint64_t loci_total, double max_seconds, int pin, uint64_t* n_done)
{
  const orc::AsmOpts opt(makeOpt(opts));
  const orc::Scores  sc(makeScores(scores));
  auto makeWorker = [&]() {
    return [&](const int li) {
      std::vector<std::string> in;
      for (uint32_t r = locus_read_begin[li]; r < locus_read_begin[li + 1]; ++r) in.emplace_back(bases + read_off[r], read_off[r + 1] - read_off[r]);
      const std::string ref(refs + ref_off[li], ref_off[li + 1] - ref_off[li]);
      orc::smallSvLocus(opt, sc, largeIndelScore, in, ref, leadingCut, trailingCut, maxLeadingCut, maxTrailingCut, false);
    };
  };
  return bench_harness::run(n_threads, loci_total, max_seconds, pin != 0, n_loci, makeWorker, n_done);
}

/// test hook: emulated libstdc++ unordered_map<string,...> iteration order for distinct keys given in insertion order
void orc_unordered_order(int n, const char* const* keys, uint32_t* out)
{
  std::vector<std::string> k;
  for (int i = 0; i < n; ++i) k.emplace_back(keys[i]);
  const std::vector<unsigned> o(orc::unorderedMapOrder(k));
  for (int i = 0; i < n; ++i) out[i] = o[i];
}

/// test hook: the real thing, from the libstdc++ this library was built against
void orc_unordered_order_real(int n, const char* const* keys, uint32_t* out)
{
  std::unordered_map<std::string, unsigned> m;
  for (int i = 0; i < n; ++i) m[std::string(keys[i])] = unsigned(i);
  int j = 0;
  for (const auto& kv : m) out[j++] = kv.second;
}

uint64_t orc_hash_bytes(const char* p, uint64_t len)
{
  return orc::libstdcxxHashBytes(p, len);
}
uint64_t orc_hash_bytes_real(const char* p, uint64_t len)
{
  return std::hash<std::string>()(std::string(p, len));
}

}  // extern "C"
