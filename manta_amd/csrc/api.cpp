// C-ABI implementation (include/manta_amd.h).  Compiled by hipcc for gfx950 into manta_amd/libmanta_amd.so.
// Host code here only stages buffers, buckets work and launches the kernels; all arithmetic of the hot path
// runs in the HIP kernels of align_kernels.hpp / assemble_kernels.hpp.
#include "../../include/manta_amd.h"

#include <algorithm>
#include <atomic>
#include <unistd.h>
#include <functional>
#include <condition_variable>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "align_kernels.hpp"
#include "align_jump_pair.hpp"
#include "assemble_kernels.hpp"
#include "asm_lds.hpp"
#include "small_asm.hpp"
#include "pipeline_kernels.hpp"
#include "split_kernels.hpp"
#include "read_class_kernels.hpp"
#include <unordered_map>
#include "rt.hpp"

using namespace manta_dev;

namespace {

thread_local std::string g_createError;

/// grow-only device buffer, reused across calls of one context
struct DevBuf {
  void*  p   = nullptr;
  size_t cap = 0;
  ~DevBuf() { release(); }
  void release()
  {
    if (p) rt::dfree(p);
    p   = nullptr;
    cap = 0;
  }
  void* need(size_t n)
  {
    if (n > cap) {
      release();
      const size_t want = n + n / 4 + 256;
      p                 = rt::dmalloc(want);
      cap               = want;
    }
    return p;
  }
  template <typename T>
  T* as(size_t count)
  {
    return static_cast<T*>(need(count * sizeof(T)));
  }
};

/// grow-only page-locked host buffer: device -> host staging of the pipelines (DMA needs pinned memory to run
/// asynchronously on the pipeline's stream)
struct PinnedBuf {
  void*  p   = nullptr;
  size_t cap = 0;
  ~PinnedBuf()
  {
    if (p) rt::hostFree(p);
  }
  /// keep: the bytes held so far survive a growth (a second staging round appends to the first)
  template <typename T>
  T* as(size_t count, bool keep = false)
  {
    const size_t n = count * sizeof(T);
    if (n > cap) {
      const size_t want = n + n / 4 + 256;
      void*        q    = rt::hostAlloc(want);
      if (p && keep) std::memcpy(q, p, cap);
      if (p) rt::hostFree(p);
      cap = want;
      p   = q;
    }
    return static_cast<T*>(p);
  }
};

double nowMs()
{
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

/// live contexts/pipelines of this process share the device's free memory (per-wave workspaces are sized from it)
std::atomic<int> g_liveWorkspaces{0};

}  // namespace

struct manta_smallsv;
struct manta_spanning;

struct manta_ctx {
  std::mutex  errMu;  // the workers of a whole-batch call report through the same context
  std::string lastError;
  std::string deviceName;
  int         cuCount = 0;
  // align scratch
  DevBuf dSeq, dTasks, dResults, dCigar, dTaskIds, dCounter, dPtrWs;
  DevBuf dSplitTasks, dSplitResults, dSplitTables;
  DevBuf dRc[16];  // manta_read_piles_batch: inputs, workspace, outputs
  rt::Stream stream;  // the context's own stream (manta_align_batch / manta_assemble_batch run on it)
  int        deviceId = 0;
  // worker pipelines of the whole-batch calls (manta_smallsv_batch / manta_spanning_batch), kept across calls
  std::vector<manta_smallsv*>  smallPool;
  std::vector<manta_spanning*> spanPool;
  ~manta_ctx();
  std::vector<uint32_t> growthSize, growthBuckets;  // libstdc++ bucket growth schedule (see repeat_exact.hpp)
  std::once_flag         growthOnce;
};

namespace {

int fail(manta_ctx_t* ctx, int code, const std::string& msg)
{
  if (ctx) {
    std::lock_guard<std::mutex> g(ctx->errMu);
    ctx->lastError = msg;
  }
  return code;
}

/// failures of single loci / alignments: reported in the per-item status, never fatal for the batch
bool perItemCode(int rc)
{
  return rc == MANTA_E_UNSUPPORTED || rc == MANTA_E_DEVICE_FAULT || rc == MANTA_E_EMPTY_SEQ;
}

std::string lastErrorOf(manta_ctx_t* ctx)
{
  std::lock_guard<std::mutex> g(ctx->errMu);
  return ctx->lastError;
}

const int kESet[]  = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 24, 32};
const int kNumESet = sizeof(kESet) / sizeof(kESet[0]);

int pickE(uint32_t qlen)
{
  const uint32_t need = (qlen + 63) / 64;
  for (int i = 0; i < kNumESet; ++i)
    if (uint32_t(kESet[i]) >= need) return i;
  return kNumESet - 1;  // longer than 64 x 32 columns: the widest kernel runs it in strips (align_kernels.hpp)
}

template <int KIND, int E>
void launchAlign(int grid, const AlignParams& P)
{
  rt::launch(align_kernel<KIND, E>, grid, 0, P);
}

template <int KIND>
void launchAlignE(int eIdx, int grid, const AlignParams& P)
{
  switch (kESet[eIdx]) {
  case 1: launchAlign<KIND, 1>(grid, P); break;
  case 2: launchAlign<KIND, 2>(grid, P); break;
  case 3: launchAlign<KIND, 3>(grid, P); break;
  case 4: launchAlign<KIND, 4>(grid, P); break;
  case 5: launchAlign<KIND, 5>(grid, P); break;
  case 6: launchAlign<KIND, 6>(grid, P); break;
  case 8: launchAlign<KIND, 8>(grid, P); break;
  case 10: launchAlign<KIND, 10>(grid, P); break;
  case 12: launchAlign<KIND, 12>(grid, P); break;
  case 16: launchAlign<KIND, 16>(grid, P); break;
  case 24: launchAlign<KIND, 24>(grid, P); break;
  case 32: launchAlign<KIND, 32>(grid, P); break;
  default: throw rt::Error("internal: unsupported E");
  }
}

/// GlobalLargeIndelAligner buckets of short queries run two alignments per wave in packed 16-bit arithmetic (align_pair.hpp) when
/// the scores leave the margin pairEligible() asks for.  The caller sizes the slabs for it: a wave's cell pairs take twice the
/// single-alignment slab, and a bucket needs half as many work items.
/// `maxRef`: the longest reference of the bucket -- the packed kernel keeps a traceback start's row in 16 bits (align_pair.hpp: rowKey), so a
/// bucket with a reference of 65 536 rows or more runs on align_kernel<1, E>
bool alignUsesPairs(int kind, int eIdx, int match, int mismatch, int open, int extend, int offEdge, int extra, int allowEdgeIns, uint64_t maxRef)
{
  static const bool off = std::getenv("MANTA_AMD_NO_ALIGN_PAIRS") != nullptr;  // A/B knob
  static const bool offJump = std::getenv("MANTA_AMD_NO_JUMP_PAIRS") != nullptr;
  if (off || maxRef > 0xfffeu) return false;
  if (kind == MANTA_ALIGNER_LARGE_INDEL) return pairEligible(kESet[eIdx], match, mismatch, open, extend, offEdge, extra, allowEdgeIns);
  // GlobalJumpAligner: align_jump_pair.hpp (maxRef: both references together -- the combined rows of a task)
  if (kind == MANTA_ALIGNER_JUMP) return !offJump && !allowEdgeIns && jumpPairEligible(kESet[eIdx], match, mismatch, open, extend, offEdge, extra);
  return false;
}

void launchJumpPair(int eIdx, int grid, const AlignParams& P)
{
  if (std::getenv("MANTA_AMD_DEBUG")) std::fprintf(stderr, "manta_amd: align_jump_pair_kernel<%d>: %d waves (two alignments each)\n", kESet[eIdx], grid);
  switch (kESet[eIdx]) {
  case 1: rt::launch(align_jump_pair_kernel<1>, grid, 0, P); break;
  case 2: rt::launch(align_jump_pair_kernel<2>, grid, 0, P); break;
  case 3: rt::launch(align_jump_pair_kernel<3>, grid, 0, P); break;
  case 4: rt::launch(align_jump_pair_kernel<4>, grid, 0, P); break;
  case 5: rt::launch(align_jump_pair_kernel<5>, grid, 0, P); break;
  case 6: rt::launch(align_jump_pair_kernel<6>, grid, 0, P); break;
  case 8: rt::launch(align_jump_pair_kernel<8>, grid, 0, P); break;
  default: throw rt::Error("internal: no packed jump aligner for this E");
  }
}

void launchAlignPair(int eIdx, int grid, const AlignParams& P)
{
  if (std::getenv("MANTA_AMD_DEBUG")) std::fprintf(stderr, "manta_amd: align_pair_kernel<%d>: %d waves (two alignments each)\n", kESet[eIdx], grid);
  switch (kESet[eIdx]) {
  case 1: rt::launch(align_pair_kernel<1>, grid, 0, P); break;
  case 2: rt::launch(align_pair_kernel<2>, grid, 0, P); break;
  case 3: rt::launch(align_pair_kernel<3>, grid, 0, P); break;
  case 4: rt::launch(align_pair_kernel<4>, grid, 0, P); break;
  case 5: rt::launch(align_pair_kernel<5>, grid, 0, P); break;
  case 6: rt::launch(align_pair_kernel<6>, grid, 0, P); break;
  default: throw rt::Error("internal: no packed aligner for this E");
  }
}

/// `pair`: the launch was sized for align_pair_kernel (alignUsesPairs)
void launchAlignKind(int kind, int eIdx, int grid, const AlignParams& P, bool pair = false)
{
  if (kind == MANTA_ALIGNER_GLOBAL)
    launchAlignE<0>(eIdx, grid, P);
  else if (kind == MANTA_ALIGNER_LARGE_INDEL) {
    if (pair)
      launchAlignPair(eIdx, grid, P);
    else
      launchAlignE<1>(eIdx, grid, P);
  } else if (pair)
    launchJumpPair(eIdx, grid, P);
  else
    launchAlignE<2>(eIdx, grid, P);
}

int alignWavesPerCu()
{
  return std::getenv("MANTA_AMD_ALIGN_WAVES_PER_CU") ? std::atoi(std::getenv("MANTA_AMD_ALIGN_WAVES_PER_CU")) : 16;
}

/// bucket_count() transitions of the libstdc++ this library is linked against, recorded from a live
/// std::unordered_map (the reference's repeat search iterates such maps: assembly/IterativeAssembler.cpp:630-641)
void recordGrowthSchedule(std::vector<uint32_t>& sizes, std::vector<uint32_t>& buckets, const uint32_t upTo)
{
  std::unordered_map<int, int> m;
  size_t                       last = m.bucket_count();
  for (uint32_t i = 0; i < upTo; ++i) {
    m[int(i)] = 0;
    if (m.bucket_count() != last) {
      sizes.push_back(i);
      buckets.push_back(uint32_t(m.bucket_count()));
      last = m.bucket_count();
    }
  }
}

/// host twin of manta_dev::libstdcxxStringHash, used once per context to verify that the murmur restatement the
/// kernels use matches the std::hash<std::string> of the libstdc++ this process runs against
bool stringHashMatchesLibstdcxx()
{
  auto mix = [](uint64_t v) { return v ^ (v >> 47); };
  const char* probes[] = {"A", "ACGTACG", "ACGTACGT", "ACGTTGCAAGCTTGACCATGGTACCAGTCAGT", "TTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTGACGATCGATCGTAGCTAGCTAGCTAGCTAGCTAGTCG"};
  for (const char* p : probes) {
    const std::string s(p);
    const uint64_t    mul  = (uint64_t(0xc6a4a793UL) << 32) + uint64_t(0x5bd1e995UL);
    uint64_t          hash = uint64_t(0xc70f6907UL) ^ (uint64_t(s.size()) * mul);
    const size_t      al   = s.size() & ~size_t(7);
    for (size_t i = 0; i < al; i += 8) {
      uint64_t d = 0;
      for (int b = 0; b < 8; ++b) d |= uint64_t(uint8_t(s[i + b])) << (8 * b);
      d = mix(d * mul) * mul;
      hash ^= d;
      hash *= mul;
    }
    if (s.size() & 7) {
      uint64_t d = 0;
      for (size_t b = 0; b < (s.size() & 7); ++b) d |= uint64_t(uint8_t(s[al + b])) << (8 * b);
      hash ^= d;
      hash *= mul;
    }
    hash = mix(hash) * mul;
    hash = mix(hash);
    if (hash != uint64_t(std::hash<std::string>()(s))) return false;
  }
  return true;
}

/// Device memory one pipeline may take for its per-wave workspaces: an equal share of half of what is free ON ITS DEVICE
/// among the pipelines that live on that device, capped.  Process-wide bookkeeping per device (a node context keeps
/// pipelines on several GPUs; a reading of another device's free memory would be the wrong number).
static const int kMaxDevices = 64;
std::atomic<int> g_livePerDevice[kMaxDevices];
size_t workspaceBudget(const size_t capBytes)
{
  // hipMemGetInfo is a driver round trip (tenths of a millisecond) and this is called on every upload and run: the
  // budget is a soft bound (half of the free memory), so a reading that is a fraction of a second old is good enough
  static std::mutex                            mu;
  static size_t                                cachedFree[kMaxDevices] = {0};
  static std::chrono::steady_clock::time_point stamp[kMaxDevices];
  const int                                    dev = std::min(kMaxDevices - 1, std::max(0, rt::currentDevice()));
  size_t                                       freeNow;
  {
    std::lock_guard<std::mutex> g(mu);
    const auto                  now = std::chrono::steady_clock::now();
    if (cachedFree[dev] == 0 || now - stamp[dev] > std::chrono::milliseconds(250)) {
      cachedFree[dev] = rt::freeBytes();
      stamp[dev]      = now;
    }
    freeNow = cachedFree[dev];
  }
  const int live = std::max(1, g_livePerDevice[dev].load());
  return std::min<size_t>(freeNow / 2 / size_t(live), capBytes);
}

/// Host loops over a whole batch (validation scan of the offset arrays, compaction of the results) split over a few
/// threads: fn(part, begin, end) for `parts` contiguous ranges of [0, n); the caller's thread takes part 0.
unsigned hostParts(const uint64_t n)
{
  static const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  if (const char* forced = std::getenv("MANTA_AMD_HOST_PARTS"))  // tests: take the multi-range paths on small batches too
    return unsigned(std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(n, 4), uint64_t(std::max(1, std::atoi(forced))))));
  if (n < 4096) return 1;
  return std::min(4u, hw);
}
/// three helper threads per process, parked on a condition variable between jobs (starting std::threads per call costs more
/// than the loops they would share on a 256-core host)
class HostPool {
 public:
  static HostPool& get()
  {
    static HostPool pool;
    return pool;
  }
  /// fn(part, begin, end) for `parts` (<= 4) contiguous ranges of [0, n); part 0 runs on the caller's thread.  One job at a
  /// time: concurrent callers (workers of a batch call) queue up behind runMu.
  template <typename F>
  void run(const uint64_t n, const unsigned parts, F fn)
  {
    auto begin = [&](unsigned t) { return n * t / parts; };
    if (parts <= 1 || getpid() != owner) {  // (a fork()ed child has no helper threads: it runs the loop itself)
      fn(0u, uint64_t(0), n);
      return;
    }
    std::lock_guard<std::mutex> only(runMu);
    std::function<void(unsigned)> job = [&](unsigned t) { fn(t, begin(t), begin(t + 1)); };
    {
      std::lock_guard<std::mutex> g(mu);
      current = &job;
      wanted  = parts - 1;
      pending = parts - 1;
      ++generation;
    }
    cv.notify_all();
    struct WaitForHelpers {  // also when fn throws on the caller's part: the helpers still hold a pointer to `job`
      HostPool& p;
      ~WaitForHelpers()
      {
        std::unique_lock<std::mutex> g(p.mu);
        p.done.wait(g, [&] { return p.pending == 0; });
        p.current = nullptr;
      }
    } waitForHelpers{*this};
    fn(0u, begin(0), begin(1));
  }

 private:
  HostPool() : owner(getpid())
  {
    for (unsigned i = 0; i < 3; ++i) threads.emplace_back([this, i] { loop(i + 1); });
  }
  ~HostPool()
  {
    if (getpid() != owner) {  // fork()ed child: the threads do not exist here
      for (std::thread& t : threads) t.detach();
      return;
    }
    {
      std::lock_guard<std::mutex> g(mu);
      stop = true;
    }
    cv.notify_all();
    for (std::thread& t : threads) t.join();
  }
  void loop(const unsigned id)
  {
    uint64_t seen = 0;
    while (true) {
      std::function<void(unsigned)>* job = nullptr;
      {
        std::unique_lock<std::mutex> g(mu);
        cv.wait(g, [&] { return stop || generation != seen; });
        if (stop) return;
        seen = generation;
        if (id <= wanted) job = current;
      }
      if (job) {
        (*job)(id);
        std::lock_guard<std::mutex> g(mu);
        if (--pending == 0) done.notify_one();
      }
    }
  }
  const pid_t                    owner;
  std::mutex                     mu, runMu;
  std::condition_variable        cv, done;
  std::vector<std::thread>       threads;
  std::function<void(unsigned)>* current = nullptr;
  unsigned                       wanted = 0, pending = 0;
  uint64_t                       generation = 0;
  bool                           stop = false;
};
template <typename F>
void hostParallel(const uint64_t n, const unsigned parts, F fn)
{
  HostPool::get().run(n, parts, fn);
}

uint32_t nextPow2(uint64_t v)
{
  uint64_t p = 1;
  while (p < v) p <<= 1;
  return uint32_t(p);
}

int asmStatusToAbi(int st)
{
  switch (st) {
  case ASM_OK: return MANTA_OK;
  case ASM_E_ALPHABET:
  case ASM_E_WORD_TOO_LONG:
  case ASM_E_TOO_MANY_READS: return MANTA_E_UNSUPPORTED;
  // (ASM_E_OUT_CAPACITY = the library's own device arena ran out: not something the caller's arenas can fix)
  default: return MANTA_E_DEVICE_FAULT;
  }
}


/// One batch of loci through the assembler: sizing, staging, launch, fetch.
struct AsmStage {
  manta_ctx_t* ctx;
  explicit AsmStage(manta_ctx_t* c) : ctx(c)
  {
    g_liveWorkspaces++;
    g_livePerDevice[std::min(kMaxDevices - 1, std::max(0, c->deviceId))]++;
  }
  ~AsmStage()
  {
    g_liveWorkspaces--;
    g_livePerDevice[std::min(kMaxDevices - 1, std::max(0, ctx->deviceId))]--;
    if (dChunksDone) rt::dfree(dChunksDone);
  }
  AsmStage(const AsmStage&) = delete;
  AsmStage& operator=(const AsmStage&) = delete;
  // device -> host staging (pinned), filled by stageOut()
  PinnedBuf     pLoci, pCont, pSeq, pBits, pCnt;
  AsmLocusOut*  hLoci = nullptr;
  AsmContigOut* hCont = nullptr;
  uint8_t*      hSeq  = nullptr;
  uint64_t*     hBits = nullptr;
  uint64_t*     hCnt  = nullptr;
  uint64_t      seqUsedDev = 0, bitsUsedDev = 0, nContigsOut = 0, pseudoBytesOut = 0, pseudoCountOut = 0;
  bool          staged = false;
  uint32_t      ldsFallbacks = 0;  // loci the LDS pipeline handed to the general kernel (valid after stageOut)

  manta_asm_options_t opt{};
  uint32_t            nLoci = 0, nReadsTotal = 0, maxContigLen = 0, wMax = 0, capWords = 0, capReads = 0, capNodes = 0, capSlots = 0;
  uint64_t            nBases = 0, stride = 0, devSeqCap = 0, devBitsCap = 0;
  int                 grid = 1;
  DevBuf              bBases, bReadOff, bLocusBegin, bLoci, bContigs, bSeqArena, bBitsArena, bCounters, bWs, bGrowth, bWl, bOrder;
  // optional per-locus word lengths of the NEXT batch (manta_*_set_word_lengths); empty = the option block's values
  std::vector<uint32_t> locusMinWl, locusMaxWl;
  std::vector<uint32_t> order;  // loci by decreasing estimated cost: the work queue hands out the long ones first
  uint32_t              maxWordLen = 0;
  uint32_t              capWords2 = 0, capNodes2 = 0, capSlots2 = 0;  // worst-case capacities (rerunCapacityFailures)
  uint64_t              stride2 = 0;
  DevBuf                bWs2, bFailIds;
  uint32_t              nRerun = 0;  // loci of the last launch that needed the worst-case workspace
  AsmParams             lastParams{};
  bool                  smallMode = false;  // small_assemble_kernel (SmallAssembler) instead of the iterative assembler
  uint32_t              smallMinSeedReads = 0, smallMaxIterations = 0;
  int                   wavesPerCuCap = 0;  // > 0: leave wave slots free for another block's aligners (pipelined batch calls)
  bool                  useFast = false;  // the LDS pipeline (graph_kernel -> contig_kernel, asm_lds.hpp); what it does not cover goes to assemble_kernel
  int                   gridFast = 1;      // graph_kernel workgroups
  int                   gridContig[manta_dev::LG_CLASSES] = {0, 0, 0, 0};  // contig_kernel workgroups per LDS size class
  uint32_t              classBytes[manta_dev::LG_CLASSES] = {0, 0, 0, 0};
  uint64_t              lgArenaCap = 0, cwsStride = 0;
  std::vector<uint32_t> fastIds, genIds;  // cost-ordered work lists of the two paths
  // the pipeline's big class (graph_big_kernel -> contig_big_kernel, asm_lds_big.hpp): piles of up to 256 reads
  std::vector<uint32_t> bigIds;
  int                   gridBig = 1;
  int                   gridContigBig[manta_dev::LGL_CLASSES] = {0, 0};
  uint32_t              classBytesBig[manta_dev::LGL_CLASSES] = {0, 0};
  uint64_t              cwsStrideBig = 0;
  DevBuf                bLgClassIdsBig, bCwsBig;
  // ... and its word-length rounds (IterativeAssembler.cpp:856-910 on the pipeline: graph_big -> repeat_big -> contig_big per word length)
  uint32_t              bigRounds = 0;     // rounds launched (0: rounds off -- a repeat hit / a cyclic graph is handed back to assemble_kernel)
  // launch(): called on the assembler's stream between the first word length's launches (the small class' contig_kernel included) and the
  // later ones -- only when there are later ones (bigRounds > 1).  What is ASM_OK in the locus records at that point of the stream is final:
  // a pipeline starts aligning those contigs while the remaining word lengths of the tandem piles run (spanningRunImpl).
  std::function<void()> afterFirstRound;
  bool                  firstRoundHookRan = false;
  int                   gridRepeat = 0;    // repeat_big_kernel wavefronts
  uint64_t              pseudoArenaDw = 0, rwsStride = 0;
  DevBuf                bLgIter, bLgPseudo, bLgNext, bLgCyc, bLgRounds, bRws, bGws;
  DevBuf                bPunt, bLgArena, bLgOff, bLgClassIds, bLgCnt, bCws;
  uint32_t*             dPunt = nullptr;   // the general kernel's list: genIds, then the loci the LDS pipeline punted
  // packed piles of the uploaded batch (manta_packed_piles_t), device side; dPlCodes == nullptr: 1 byte per base input
  DevBuf                bPlCodes, bPlMask, bPlLen, bPlCodeOff, bPlMaskOff;
  uint32_t *            dPlCodes = nullptr, *dPlMask = nullptr, *dPlLen = nullptr;
  uint64_t *            dPlCodeOff = nullptr, *dPlMaskOff = nullptr;
  uint32_t*             dMinWl = nullptr;
  uint32_t*             dMaxWl = nullptr;
  uint32_t*             dOrder = nullptr;
  uint8_t*            dBases = nullptr;
  uint64_t*           dOff   = nullptr;
  uint32_t*           dBegin = nullptr;
  AsmLocusOut*        dLoci  = nullptr;
  AsmContigOut*       dCont  = nullptr;
  uint8_t*            dSeq   = nullptr;
  uint64_t*           dBits  = nullptr;
  uint64_t*           dCnt   = nullptr;
  uint8_t*            dWs    = nullptr;
  uint32_t*           dGrowth = nullptr;

  /// exactly one of read_off (1 byte per base input) / read_len (packed piles) is set
  int plan(const manta_asm_options_t& o, uint32_t n_loci, const uint64_t* read_off, const uint32_t* locus_read_begin,
           const uint32_t* read_len = nullptr)
  {
    if (o.min_word_length == 0 || o.word_step_size == 0 || o.min_coverage == 0 || o.max_assembly_count == 0)
      return fail(ctx, MANTA_E_INVALID_ARG, "assembler options: word length, step, minCoverage and maxAssemblyCount must be >= 1");
    if (o.max_word_length > 16u * ASM_MAX_KW) return fail(ctx, MANTA_E_UNSUPPORTED, "word lengths above 128 are not supported");
    if (2 * o.max_assembly_count > ASM_MAX_CAND) return fail(ctx, MANTA_E_UNSUPPORTED, "maxAssemblyCount above 32 is not supported");
    opt         = o;
    nLoci       = n_loci;
    for (uint32_t l = 0; l < n_loci; ++l)  // validate before the last elements are trusted as totals
      if (locus_read_begin[l + 1] < locus_read_begin[l]) return fail(ctx, MANTA_E_INVALID_ARG, "locus_read_begin not monotone");
    nReadsTotal = locus_read_begin[n_loci];
    maxWordLen  = o.max_word_length;
    if (!locusMinWl.empty()) {
      if (locusMinWl.size() != n_loci || locusMaxWl.size() != n_loci)
        return fail(ctx, MANTA_E_INVALID_ARG, "per-locus word lengths were set for a different number of loci");
      for (uint32_t l = 0; l < n_loci; ++l) {
        if (locusMinWl[l] == 0 || locusMaxWl[l] > 16u * ASM_MAX_KW) return fail(ctx, MANTA_E_UNSUPPORTED, "per-locus word length outside 1..128");
        maxWordLen = std::max(maxWordLen, locusMaxWl[l]);
      }
    }
    nBases      = read_off ? read_off[nReadsTotal] : 0;
    const double tPlan0 = nowMs();
    uint64_t maxLocusBases = 0, maxLocusWords = 0, bitsBound = 0;
    uint32_t maxLocusReads = 0, maxReadLen = 0;
    std::vector<uint64_t> cost(n_loci);
    uint32_t              ldsFit = 0, ldsFitBig = 0;  // loci small enough for the LDS pipeline's small / big class
    {
      // one pass over every read offset (6.4 MB for the 800 k reads of the metric's batch): a few host threads
      struct Part {
        uint64_t maxLocusBases = 0, maxLocusWords = 0, bitsBound = 0;
        uint32_t maxLocusReads = 0, maxReadLen = 0, ldsFit = 0, ldsFitBig = 0;
        int      bad = 0;  // 1 = locus_read_begin, 2 = read_off
      };
      const unsigned    parts = hostParts(nReadsTotal);
      std::vector<Part> part(parts);
      const uint32_t    maxAsm = opt.max_assembly_count;
      hostParallel(n_loci, parts, [&](unsigned t, uint64_t l0, uint64_t l1) {
        Part& p(part[t]);
        for (uint64_t l = l0; l < l1; ++l) {
          const uint32_t rb = locus_read_begin[l], re = locus_read_begin[l + 1];
          if (re < rb || re > nReadsTotal) {
            p.bad = 1;
            return;
          }
          // branch-free inner loops (they vectorise): a negative step shows up as a huge unsigned length in `longest`
          uint64_t b = 0, w = 0, longest = 0;
          if (read_off) {
            const uint64_t* o = read_off + rb;
            for (uint32_t i = 0; i < re - rb; ++i) {
              const uint64_t len = o[i + 1] - o[i];
              w += (len + 15) >> 4;
              longest = std::max(longest, len);
            }
            b = o[re - rb] - o[0];
          } else {
            const uint32_t* o = read_len + rb;
            for (uint32_t i = 0; i < re - rb; ++i) {
              const uint64_t len = o[i];
              b += len;
              w += (len + 15) >> 4;
              longest = std::max(longest, len);
            }
          }
          w += re - rb;
          if (longest > 0xffffffffull) {  // (also: a single read of 4 G bases is not a read)
            p.bad = 2;
            return;
          }
          p.maxReadLen = std::max<uint32_t>(p.maxReadLen, uint32_t(longest));
          cost[l] = b * uint64_t(re - rb);
          if ((re - rb) + 2 * maxAsm <= manta_dev::LG_MAX_READS && w + 2 <= manta_dev::LG_MAX_PILE) {
            p.ldsFit++;
            cost[l] |= uint64_t(1) << 63;  // (marks the locus for the split below; reads x bases stays far below 2^62)
          } else if ((re - rb) + 2 * maxAsm <= manta_dev::LGL_MAX_READS && w + 2 <= manta_dev::LGL_MAX_PILE + 2 &&
                     (!read_off || b + 64 <= manta_dev::LGL_STAGE_BYTES)) {
            p.ldsFitBig++;
            cost[l] |= uint64_t(1) << 62;  // the pipeline's big class
          }
          p.maxLocusBases = std::max(p.maxLocusBases, b);
          p.maxLocusWords = std::max(p.maxLocusWords, w);
          p.maxLocusReads = std::max(p.maxLocusReads, re - rb);
          const uint64_t W = ((re - rb) + 2 * maxAsm + 63) / 64;
          p.bitsBound += uint64_t(maxAsm) * 2 * W + 2 * maxAsm;
        }
      });
      for (const Part& p : part) {
        if (p.bad == 1) return fail(ctx, MANTA_E_INVALID_ARG, "locus_read_begin not monotone");
        if (p.bad == 2) return fail(ctx, MANTA_E_INVALID_ARG, "read_off not monotone");
        maxLocusBases = std::max(maxLocusBases, p.maxLocusBases);
        maxLocusWords = std::max(maxLocusWords, p.maxLocusWords);
        maxLocusReads = std::max(maxLocusReads, p.maxLocusReads);
        maxReadLen    = std::max(maxReadLen, p.maxReadLen);
        ldsFit += p.ldsFit;
        ldsFitBig += p.ldsFitBig;
        bitsBound += p.bitsBound;
      }
    }
    // work-queue order: most expensive loci first (reads x bases is what the table pass and the walks scale with), so
    // that the long ones are not the last to start
    const double tPlan1 = nowMs();
    {
      std::vector<std::pair<uint64_t, uint32_t>> keyed(n_loci);  // (inverted cost, locus): ascending = most expensive first, ties in locus order
      for (uint32_t l = 0; l < n_loci; ++l) keyed[l] = std::make_pair(~(cost[l] & ~(uint64_t(3) << 62)), l);
      std::sort(keyed.begin(), keyed.end());
      order.resize(n_loci);
      for (uint32_t l = 0; l < n_loci; ++l) order[l] = keyed[l].second;
    }
    // The LDS pipeline (asm_lds.hpp: graph_kernel -> contig_kernel) is the default for the loci whose pile fits its envelope;
    // the rest -- and whatever it punts: cycles, next word length, graphs that do not fit -- goes to the general kernel.
    // MANTA_AMD_ASM_PATH=general switches it off (A/B runs, the tests of the general kernel).
    {
      const char*       pathEnv = std::getenv("MANTA_AMD_ASM_PATH");
      const std::string path    = pathEnv ? pathEnv : "";
      const bool        useBig  = !(std::getenv("MANTA_AMD_LG_BIG") && std::atoi(std::getenv("MANTA_AMD_LG_BIG")) == 0);  // A/B runs: the big class off
      useFast                   = !smallMode && (ldsFit > 0 || (useBig && ldsFitBig > 0)) && path != "general";
      fastIds.clear();
      bigIds.clear();
      genIds.clear();
      if (useFast) {
        for (uint32_t i = 0; i < n_loci; ++i) {
          const uint64_t c = cost[order[i]];
          ((c >> 63) ? fastIds : ((useBig && ((c >> 62) & 1u)) ? bigIds : genIds)).push_back(order[i]);
        }
        // (Ordering the big class' list by first word length -- so that the tandem piles with most word lengths ahead of them reach the
        //  general kernel's queue first -- was measured: 1155 vs 1111 ms per 65 536 spanning loci.  The cost order stays.)
      }
    }
    const double   tPlan2   = nowMs();
    const uint32_t nCandMax = 2 * opt.max_assembly_count;
    wMax                    = uint32_t((maxLocusReads + nCandMax + 63) / 64);
    if (wMax > ASM_MAX_W) return fail(ctx, MANTA_E_UNSUPPORTED, "more than ~1000 reads in one locus");
    maxContigLen               = uint32_t(std::min<uint64_t>(maxLocusBases, 32768) + maxWordLen + 16);
    const uint64_t pseudoLen   = std::min<uint64_t>(maxContigLen, 3ull * maxReadLen + maxWordLen);
    const uint64_t pseudoBases = uint64_t(nCandMax) * pseudoLen;
    capWords                   = uint32_t(maxLocusWords + pseudoBases / 16 + 2 * nCandMax + 8);
    capReads                   = maxLocusReads + nCandMax + 1;
    capNodes                   = uint32_t(maxLocusBases + pseudoBases + 64);
    if (const char* e = std::getenv("MANTA_AMD_ASM_NODE_DIV")) {  // experiment: typical-case node capacity = bases / div (overflows run again, rerunCapacityFailures)
      const uint64_t div = uint64_t(std::max(1, std::atoi(e)));
      capNodes           = uint32_t(std::max<uint64_t>(1024, (maxLocusBases + pseudoBases) / div + 64));
    }
    if (capNodes >= LINK_NONE21) return fail(ctx, MANTA_E_UNSUPPORTED, "a locus with more than ~2M read bases is not supported");
    capSlots                   = nextPow2(2ull * capNodes);
    const AsmWsLayout L = asmWorkspaceLayout(capSlots, capNodes, capWords, capReads, maxContigLen, wMax, opt.max_assembly_count);
    stride              = (L.total + 255) & ~uint64_t(255);
    {
      // worst case for the few loci the typical-case capacities above turn out too small for (rerunCapacityFailures): every
      // candidate contig at full length comes back as a pseudo read
      const uint64_t pseudoWorst = uint64_t(nCandMax) * maxContigLen;
      capWords2                  = uint32_t(maxLocusWords + pseudoWorst / 16 + 2 * nCandMax + 8);
      capNodes2                  = uint32_t(std::min<uint64_t>(maxLocusBases + pseudoWorst + 64, LINK_NONE21 - 1));
      capSlots2                  = nextPow2(2ull * capNodes2);
      const AsmWsLayout L2 = asmWorkspaceLayout(capSlots2, capNodes2, capWords2, capReads, maxContigLen, wMax, opt.max_assembly_count);
      stride2              = (L2.total + 255) & ~uint64_t(255);
    }
    // per-wave workspaces: at most half of the free HBM, 64 GiB by default (MANTA_AMD_WS_BUDGET_GB lowers it for callers
    // that keep several batches resident at once)
    const size_t wsCapGb  = std::getenv("MANTA_AMD_WS_BUDGET_GB") ? size_t(std::max(1, std::atoi(std::getenv("MANTA_AMD_WS_BUDGET_GB")))) : size_t(64);
    const size_t wsBudget = workspaceBudget(wsCapGb << 30);
    int wavesPerCu = std::getenv("MANTA_AMD_ASM_WAVES_PER_CU") ? std::atoi(std::getenv("MANTA_AMD_ASM_WAVES_PER_CU")) : 16;
    if (wavesPerCuCap > 0) wavesPerCu = std::min(wavesPerCu, wavesPerCuCap);
    grid                  = int(std::min<uint64_t>(n_loci, uint64_t(std::max(1, ctx->cuCount * wavesPerCu))));
    grid                  = rt::roundGrid(int(std::max<uint64_t>(1, std::min<uint64_t>(uint64_t(grid), wsBudget / stride))));
    if (useFast) {
      using namespace manta_dev;
      // graph_kernel: two workgroups of LG_WAVES wavefronts per CU (LG_BUDGET bytes of LDS each).  contig_kernel: one launch per
      // LDS size class; a class of B bytes runs floor(160 KB / B) single-wave workgroups per CU (asked of the runtime).
      gridFast = int(std::max<uint64_t>(1, std::min<uint64_t>(fastIds.size(), uint64_t(ctx->cuCount) * (163840 / LG_BUDGET))));
      static const uint32_t kClassDefault[LG_CLASSES] = {20480, 54272, 0, 0};  // 8 / 3 workgroups per CU (measured: every further class costs a launch tail)
      for (unsigned c = 0; c < LG_CLASSES; ++c) classBytes[c] = kClassDefault[c];
      if (const char* e = std::getenv("MANTA_AMD_LG_CLASSES")) {  // experiments: up to four ascending byte counts, comma separated
        unsigned c = 0;
        for (const char* q = e; *q && c < LG_CLASSES; ++c) {
          classBytes[c] = uint32_t(std::strtoul(q, nullptr, 10)) & ~511u;
          q             = std::strchr(q, ',');
          if (!q) {
            ++c;
            break;
          }
          ++q;
        }
        for (; c < LG_CLASSES; ++c) classBytes[c] = 0;
      }
      int maxGrid = 1;
      for (unsigned c = 0; c < LG_CLASSES; ++c) {
        gridContig[c] = 0;
        if (!classBytes[c]) continue;
        static const int wgCap = std::getenv("MANTA_AMD_CONTIG_WG_CAP") ? std::atoi(std::getenv("MANTA_AMD_CONTIG_WG_CAP")) : 8;  // experiments
        const int perCu = std::max(1, std::min(wgCap, rt::blocksPerCu(contig_kernel, 64, classBytes[c], int(163840 / classBytes[c]))));
        gridContig[c]   = int(std::max<uint64_t>(1, std::min<uint64_t>(fastIds.size(), uint64_t(ctx->cuCount) * perCu)));
        maxGrid         = std::max(maxGrid, gridContig[c]);
      }
      if (std::getenv("MANTA_AMD_DEBUG"))
        for (unsigned c = 0; c < LG_CLASSES; ++c)
          if (classBytes[c]) std::fprintf(stderr, "manta_amd: contig_kernel class %u: %u bytes of LDS, %d workgroups (%d per CU by the runtime's count)\n", c, classBytes[c], gridContig[c], rt::blocksPerCu(contig_kernel, 64, classBytes[c], -1));
      cwsStride  = ckWorkspaceLayout().total;
      // the big class: one graph workgroup per CU; contig_big_kernel in two LDS classes (two loci / one locus per CU)
      gridBig = int(std::max<uint64_t>(1, std::min<uint64_t>(bigIds.size(), uint64_t(ctx->cuCount))));
      static const uint32_t kClassBig[LGL_CLASSES] = {81920, 163840};
      for (unsigned c = 0; c < LGL_CLASSES; ++c) {
        classBytesBig[c] = kClassBig[c];
        gridContigBig[c] = int(std::max<uint64_t>(1, std::min<uint64_t>(bigIds.size(), uint64_t(ctx->cuCount) * (163840 / kClassBig[c]))));
      }
      cwsStrideBig = ckWorkspaceLayout(LgL::SETW).total;
      bigRounds    = 0;
      // The rounds keep a pile with a tandem repeat on the pipeline through all its word lengths: ~15 ms of launches per word length for
      // the whole block instead of ~20 ms of ONE wave of assemble_kernel per locus and word length.  The gain shrinks as the device fills
      // (contig_big_kernel holds a CU per cyclic graph, assemble_kernel a sixteenth of one) -- config-5 loci, ms per block with / without
      // the rounds: 16 384: 342 / 431, 65 536: 1 043 / 1 062 -- so blocks beyond MANTA_AMD_BIG_ROUNDS_MAX big-class loci (not measured)
      // hand their cyclic graphs to assemble_kernel as before; MANTA_AMD_BIG_ROUNDS = 0 / 1 forces the rounds off / on.
      const char*    re       = std::getenv("MANTA_AMD_BIG_ROUNDS");
      const uint64_t roundMax = std::getenv("MANTA_AMD_BIG_ROUNDS_MAX") ? std::strtoull(std::getenv("MANTA_AMD_BIG_ROUNDS_MAX"), nullptr, 10) : uint64_t(65536);
      const bool     roundsOn = re ? (std::atoi(re) != 0) : (bigIds.size() <= roundMax);
      if (roundsOn && !bigIds.empty() && opt.max_assembly_count <= 20) {
        for (const uint32_t l : bigIds) {
          const uint32_t lo = locusMinWl.empty() ? opt.min_word_length : locusMinWl[l], hi = locusMaxWl.empty() ? opt.max_word_length : locusMaxWl[l];
          if (hi >= lo) bigRounds = std::max<uint32_t>(bigRounds, (hi - lo) / opt.word_step_size + 1);
        }
        bigRounds     = std::min<uint32_t>(bigRounds, LGL_MAX_ROUNDS);
        gridRepeat    = rt::roundGrid(int(std::min<uint64_t>(uint64_t(ctx->cuCount) * 4 * MANTA_RPB_OCC, std::max<uint64_t>(64, bigIds.size() / 4))));
        rwsStride     = rpbWorkspaceLayout().total;
        pseudoArenaDw = std::max<uint64_t>(uint64_t(4) << 20, uint64_t(bigIds.size()) * 512);  // dwords
      }
      lgArenaCap   = std::min<uint64_t>(uint64_t(fastIds.size()) * lgSlabBytes(LG_MAX_NODES, LG_MAX_NODES, LG_MAX_PILE + 2) +
                                          uint64_t(bigIds.size() + (bigRounds ? 3 * std::min<size_t>(bigIds.size(), 512) : 0)) * lgSlabL(LGL_MAX_NODES, LGL_POOL_CAP + LGL_POOL_OVF, LGL_MAX_PILE_ALL + 4).total + 4096,
                                      wsBudget / 2);
      (void)maxGrid;
    }
    // contig + pseudo-read text one locus can emit at worst; the arena holds the typical case for every locus plus one
    // worst case, so a single-locus call (the runIterativeAssembler adapter) can never exhaust it
    const uint64_t worstLocusSeq = uint64_t(opt.max_assembly_count) * maxContigLen + uint64_t(nCandMax) * pseudoLen;
    devSeqCap  = uint64_t(n_loci) * std::min<uint64_t>(worstLocusSeq, 65536) + worstLocusSeq + 4096;
    devBitsCap = bitsBound + 64;
    if (std::getenv("MANTA_AMD_DEBUG_TIMING"))
      std::fprintf(stderr, "manta_amd: plan: scan %.2f ms, order %.2f, sizing %.2f\n", tPlan1 - tPlan0, tPlan2 - tPlan1, nowMs() - tPlan2);
    std::call_once(ctx->growthOnce, [&] { recordGrowthSchedule(ctx->growthSize, ctx->growthBuckets, 4u << 20); });  // (workers of a batch call plan concurrently)
    return MANTA_OK;
  }

  /// packed piles: offsets are rebased so that the first read of this batch starts at dword 0
  void uploadPiles(const manta_packed_piles_t& pl)
  {
    const uint64_t c0 = pl.read_code_off[0], c1 = pl.read_code_off[nReadsTotal], m0 = pl.read_mask_off[0], m1 = pl.read_mask_off[nReadsTotal];
    dPlCodes   = bPlCodes.as<uint32_t>(c1 - c0 + 4);
    dPlMask    = bPlMask.as<uint32_t>(m1 - m0 + 4);
    dPlLen     = bPlLen.as<uint32_t>(nReadsTotal + 1);
    dPlCodeOff = bPlCodeOff.as<uint64_t>(nReadsTotal + 1);
    dPlMaskOff = bPlMaskOff.as<uint64_t>(nReadsTotal + 1);
    rt::h2d(dPlCodes, pl.codes + c0, sizeof(uint32_t) * (c1 - c0));
    rt::h2d(dPlMask, pl.nmask + m0, sizeof(uint32_t) * (m1 - m0));
    rt::h2d(dPlLen, pl.read_len, sizeof(uint32_t) * nReadsTotal);
    if (c0 == 0 && m0 == 0) {
      rt::h2d(dPlCodeOff, pl.read_code_off, sizeof(uint64_t) * (nReadsTotal + 1));
      rt::h2d(dPlMaskOff, pl.read_mask_off, sizeof(uint64_t) * (nReadsTotal + 1));
    } else {
      plRebased.resize(2 * (size_t(nReadsTotal) + 1));
      for (uint32_t r = 0; r <= nReadsTotal; ++r) {
        plRebased[r]                   = pl.read_code_off[r] - c0;
        plRebased[nReadsTotal + 1 + r] = pl.read_mask_off[r] - m0;
      }
      rt::h2d(dPlCodeOff, plRebased.data(), sizeof(uint64_t) * (nReadsTotal + 1));
      rt::h2d(dPlMaskOff, plRebased.data() + nReadsTotal + 1, sizeof(uint64_t) * (nReadsTotal + 1));
    }
    plBytes = 4 * (c1 - c0) + 4 * (m1 - m0) + 20ull * nReadsTotal;
    upload(nullptr, nullptr, pl.locus_read_begin);
  }
  std::vector<uint64_t> plRebased;
  uint64_t              plBytes = 0;

  // ---- streamed upload (whole-batch calls): the read bases arrive chunk by chunk on `copyStream` while assemble_kernel,
  // launched right away on the pipeline's stream, works through the loci whose chunk has landed (AsmParams::upload_*)
  static const uint32_t kStreamChunks = 8;
  DevBuf                bPlShift;  // streamed packed piles: three shifts per chunk
  uint64_t*             dPlShift = nullptr;
  bool                  streamingPiles = false;
  DevBuf                bStream;  // chunk shifts
  uint32_t*             dStream = nullptr;
  uint32_t*             dChunksDone = nullptr;  // fine-grained device word the copy engine bumps after every chunk
  PinnedBuf             pChunkIds;              // the values 0..kStreamChunks it is bumped to (DMA sources)
  uint32_t              chunkLoci = 0;
  bool                  streaming = false;

  /// like upload(), but only ENQUEUES the copy of the read bases (in kStreamChunks pieces, each followed by its completion
  /// signal) on copyStream and returns; launch() passes the counters to the kernel.  The caller keeps `bases` alive and
  /// synchronises copyStream before it touches them again.
  void uploadStreamed(const uint8_t* bases, const uint64_t* read_off, const uint32_t* locus_read_begin, rt::Stream& copyStream)
  {
    chunkLoci = std::max<uint32_t>(1, (nLoci + kStreamChunks - 1) / kStreamChunks);
    const uint32_t nChunks = (nLoci + chunkLoci - 1) / chunkLoci;
    std::vector<uint32_t> host(1 + kStreamChunks, 0u);
    std::vector<uint64_t> hostBegin(nChunks + 1), devBegin(nChunks + 1);
    uint64_t              cursor = 0;
    for (uint32_t c = 0; c < nChunks; ++c) {
      const uint32_t l0 = c * chunkLoci, l1 = std::min(nLoci, l0 + chunkLoci);
      hostBegin[c]      = read_off[locus_read_begin[l0]];
      const uint64_t len = read_off[locus_read_begin[l1]] - hostBegin[c];
      devBegin[c]        = cursor;
      host[1 + c]        = uint32_t(devBegin[c] - hostBegin[c]);  // modulo 2^32: the kernel adds it in 32-bit arithmetic to a 64-bit offset
      cursor             = (cursor + len + 64 + 255) & ~uint64_t(255);
    }
    hostBegin[nChunks] = read_off[nReadsTotal];
    // shifts must be exact in 64 bits: keep them small by construction (device offsets only grow by the padding)
    for (uint32_t c = 0; c < nChunks; ++c) host[1 + c] = uint32_t(devBegin[c] - hostBegin[c]);
    const uint64_t savedBases = nBases;
    nBases                    = cursor + 64;  // device arena incl. the per-chunk padding
    upload(nullptr, nullptr, locus_read_begin);  // allocations + the small arrays (order, word lengths, growth schedule, locus begins)
    nBases  = savedBases;
    dPlCodes = nullptr;
    rt::h2d(dOff, read_off, sizeof(uint64_t) * (nReadsTotal + 1));
    dStream = bStream.as<uint32_t>(1 + kStreamChunks);
    rt::h2d(dStream, host.data(), sizeof(uint32_t) * (1 + kStreamChunks));
    if (!dChunksDone) dChunksDone = static_cast<uint32_t*>(rt::dmallocFine(64));
    uint32_t* ids = pChunkIds.as<uint32_t>(kStreamChunks + 1);
    for (uint32_t c = 0; c <= kStreamChunks; ++c) ids[c] = c;
    {  // a call that failed half way may have left counter bumps queued on the copy stream: none may land after the reset
      rt::ScopedStream onCopy(copyStream);
      rt::sync();
    }
    rt::h2d(dChunksDone, ids, sizeof(uint32_t));  // = 0
    rt::sync();  // the counter is zero and the small arrays are in place before the first chunk can land
    {
      // one copy per chunk, each followed by a stream-ordered 32-bit write of the counter (command processor; a 4-byte copy
      // if the runtime refuses): the counter says c+1 only after chunk c is in HBM.  Nothing here needs a workgroup slot --
      // the persistent assembler, or another process' kernels, may own every one of them.
      rt::ScopedStream onCopy(copyStream);
      for (uint32_t c = 0; c < nChunks; ++c) {
        rt::h2d(dBases + devBegin[c], bases + hostBegin[c], hostBegin[c + 1] - hostBegin[c]);
        if (!rt::streamWrite32(dChunksDone, c + 1)) rt::h2d(dChunksDone, ids + c + 1, sizeof(uint32_t));
      }
    }
    streaming = true;
  }

  /// uploadStreamed() for packed piles: per chunk the slices of the five pile arrays (codes, N masks, read lengths and the two
  /// per-read offset arrays) are copied to line-aligned device positions, then the counter is bumped; the kernel finds a
  /// locus' slices through three per-chunk shifts (AsmParams::pl_chunk_shift).  Only the locus table goes first.
  void uploadPilesStreamed(const manta_packed_piles_t& pl, rt::Stream& copyStream)
  {
    chunkLoci = std::max<uint32_t>(1, (nLoci + kStreamChunks - 1) / kStreamChunks);
    const uint32_t nChunks = (nLoci + chunkLoci - 1) / chunkLoci;
    struct Piece {
      uint32_t r0, r1;
      uint64_t c0, c1, m0, m1, dr, dc, dm;
    };
    std::vector<Piece>    pc(nChunks);
    std::vector<uint64_t> shifts(3 * kStreamChunks, 0);
    uint64_t              curR = 0, curC = 0, curM = 0;
    auto                  lineUp = [](uint64_t v) { return (v + 63) & ~uint64_t(63); };  // 64 elements >= one 128-byte line for every array
    for (uint32_t c = 0; c < nChunks; ++c) {
      const uint32_t l0 = c * chunkLoci, l1 = std::min(nLoci, l0 + chunkLoci);
      Piece&         q(pc[c]);
      q.r0 = pl.locus_read_begin[l0], q.r1 = pl.locus_read_begin[l1];
      q.c0 = pl.read_code_off[q.r0], q.c1 = pl.read_code_off[q.r1];
      q.m0 = pl.read_mask_off[q.r0], q.m1 = pl.read_mask_off[q.r1];
      q.dr = curR, q.dc = curC, q.dm = curM;
      shifts[3 * c + 0] = q.dr - q.r0;  // modulo 2^64 on purpose
      shifts[3 * c + 1] = q.dc - q.c0;
      shifts[3 * c + 2] = q.dm - q.m0;
      curR = lineUp(curR + (q.r1 - q.r0) + 1);  // +1: the offset arrays hold one more entry than there are reads
      curC = lineUp(curC + (q.c1 - q.c0) + 4);
      curM = lineUp(curM + (q.m1 - q.m0) + 4);
    }
    dPlCodes   = bPlCodes.as<uint32_t>(curC + 64);
    dPlMask    = bPlMask.as<uint32_t>(curM + 64);
    dPlLen     = bPlLen.as<uint32_t>(curR + 64);
    dPlCodeOff = bPlCodeOff.as<uint64_t>(curR + 64);
    dPlMaskOff = bPlMaskOff.as<uint64_t>(curR + 64);
    plBytes    = 0;
    upload(nullptr, nullptr, pl.locus_read_begin);  // allocations + order, word lengths, growth schedule, locus table
    dPlShift = bPlShift.as<uint64_t>(3 * kStreamChunks);
    rt::h2d(dPlShift, shifts.data(), sizeof(uint64_t) * 3 * kStreamChunks);
    if (!dChunksDone) dChunksDone = static_cast<uint32_t*>(rt::dmallocFine(64));
    uint32_t* ids = pChunkIds.as<uint32_t>(kStreamChunks + 1);
    for (uint32_t c = 0; c <= kStreamChunks; ++c) ids[c] = c;
    {  // a call that failed half way may have left counter bumps queued on the copy stream: none may land after the reset
      rt::ScopedStream onCopy(copyStream);
      rt::sync();
    }
    rt::h2d(dChunksDone, ids, sizeof(uint32_t));  // = 0
    rt::sync();
    {
      rt::ScopedStream onCopy(copyStream);
      for (uint32_t c = 0; c < nChunks; ++c) {
        const Piece&   q(pc[c]);
        const uint64_t nR = q.r1 - q.r0;
        rt::h2d(dPlLen + q.dr, pl.read_len + q.r0, sizeof(uint32_t) * nR);
        rt::h2d(dPlCodeOff + q.dr, pl.read_code_off + q.r0, sizeof(uint64_t) * (nR + 1));
        rt::h2d(dPlMaskOff + q.dr, pl.read_mask_off + q.r0, sizeof(uint64_t) * (nR + 1));
        rt::h2d(dPlCodes + q.dc, pl.codes + q.c0, sizeof(uint32_t) * (q.c1 - q.c0));
        rt::h2d(dPlMask + q.dm, pl.nmask + q.m0, sizeof(uint32_t) * (q.m1 - q.m0));
        if (!rt::streamWrite32(dChunksDone, c + 1)) rt::h2d(dChunksDone, ids + c + 1, sizeof(uint32_t));
        plBytes += 4 * (q.c1 - q.c0) + 4 * (q.m1 - q.m0) + 20ull * nR;
      }
    }
    streaming       = true;
    streamingPiles  = true;
  }

  void upload(const uint8_t* bases, const uint64_t* read_off, const uint32_t* locus_read_begin)
  {
    if (bases) dPlCodes = nullptr;
    streaming = streamingPiles = false;
    dBases  = bBases.as<uint8_t>(nBases + 64);
    dOff    = bReadOff.as<uint64_t>(nReadsTotal + 1);
    dBegin  = bLocusBegin.as<uint32_t>(nLoci + 1);
    dLoci   = bLoci.as<AsmLocusOut>(nLoci);
    dCont   = bContigs.as<AsmContigOut>(uint64_t(nLoci) * opt.max_assembly_count);
    dSeq    = bSeqArena.as<uint8_t>(devSeqCap);
    dBits   = bBitsArena.as<uint64_t>(devBitsCap);
    dCnt    = bCounters.as<uint64_t>(16);
    dWs     = bWs.as<uint8_t>(stride * grid);
    dGrowth = bGrowth.as<uint32_t>(2 * ctx->growthSize.size() + 2);
    dOrder  = bOrder.as<uint32_t>(nLoci);
    if (useFast) {
      dPunt = bPunt.as<uint32_t>(nLoci);
      if (!fastIds.empty()) rt::h2d(dOrder, fastIds.data(), sizeof(uint32_t) * fastIds.size());
      if (!bigIds.empty()) rt::h2d(dOrder + fastIds.size(), bigIds.data(), sizeof(uint32_t) * bigIds.size());  // (behind the small class' list)
      if (!genIds.empty()) rt::h2d(dPunt, genIds.data(), sizeof(uint32_t) * genIds.size());
      int maxGrid = 1, maxGridBig = 1;
      for (unsigned c = 0; c < manta_dev::LG_CLASSES; ++c) maxGrid = std::max(maxGrid, gridContig[c]);
      for (unsigned c = 0; c < manta_dev::LGL_CLASSES; ++c) maxGridBig = std::max(maxGridBig, gridContigBig[c]);
      (void)bLgArena.as<uint8_t>(lgArenaCap + 64);
      (void)bLgOff.as<uint64_t>(nLoci);
      (void)bLgClassIds.as<uint32_t>(uint64_t(manta_dev::LG_CLASSES) * std::max<size_t>(1, fastIds.size()));
      (void)bLgCnt.as<uint64_t>(16);
      (void)bCws.as<uint8_t>(cwsStride * uint64_t(maxGrid));
      if (!bigIds.empty()) {
        (void)bLgClassIdsBig.as<uint32_t>(uint64_t(manta_dev::LGL_CLASSES) * bigIds.size());
        (void)bCwsBig.as<uint8_t>(cwsStrideBig * uint64_t(maxGridBig));
        if (bigRounds) {
          (void)bLgIter.as<manta_dev::LgIter>(nLoci);
          (void)bLgPseudo.as<uint32_t>(pseudoArenaDw + 64);
          (void)bLgNext.as<uint32_t>(2 * bigIds.size());
          (void)bLgCyc.as<uint32_t>(bigIds.size());
          (void)bLgRounds.as<uint32_t>(8 * (manta_dev::LGL_MAX_ROUNDS + 1) + 32);
          (void)bRws.as<uint8_t>(rwsStride * uint64_t(gridRepeat));
          (void)bGws.as<uint8_t>(uint64_t(32) * manta_dev::LGL_POOL_OVF * uint64_t(gridBig));
        }
      }
    } else {
      rt::h2d(dOrder, order.data(), sizeof(uint32_t) * nLoci);
    }
    dMinWl = dMaxWl = nullptr;
    if (!locusMinWl.empty()) {
      dMinWl = bWl.as<uint32_t>(2ull * nLoci);
      dMaxWl = dMinWl + nLoci;
      rt::h2d(dMinWl, locusMinWl.data(), sizeof(uint32_t) * nLoci);
      rt::h2d(dMaxWl, locusMaxWl.data(), sizeof(uint32_t) * nLoci);
    }
    if (bases) {
      rt::h2d(dBases, bases, nBases);
      rt::h2d(dOff, read_off, sizeof(uint64_t) * (nReadsTotal + 1));
    }
    rt::h2d(dBegin, locus_read_begin, sizeof(uint32_t) * (nLoci + 1));
    rt::h2d(dGrowth, ctx->growthSize.data(), sizeof(uint32_t) * ctx->growthSize.size());
    rt::h2d(dGrowth + ctx->growthSize.size(), ctx->growthBuckets.data(), sizeof(uint32_t) * ctx->growthBuckets.size());
  }

  void launch()
  {
    rt::dzero(dCnt, sizeof(uint64_t) * 16);
    rt::dfill(dLoci, 0xff, sizeof(AsmLocusOut) * nLoci);
    AsmParams P;
    P.bases            = dBases;
    P.read_off         = dOff;
    P.locus_read_begin = dBegin;
    P.n_loci           = nLoci;
    P.opt = AsmOptsDev{opt.min_word_length, opt.max_word_length, opt.word_step_size, opt.min_coverage,
                       opt.min_conservative_coverage, opt.min_unused_reads, opt.min_support_reads, opt.max_assembly_count};
    P.counter        = reinterpret_cast<uint32_t*>(dCnt);
    P.ws             = dWs;
    P.ws_stride      = stride;
    P.cap_slots      = capSlots;
    P.cap_nodes      = capNodes;
    P.cap_words      = capWords;
    P.cap_reads      = capReads;
    P.max_contig_len = maxContigLen;
    P.w_max          = wMax;
    P.loci           = dLoci;
    P.contigs        = dCont;
    P.seq_arena      = dSeq;
    P.seq_cap        = devSeqCap;
    P.seq_used       = reinterpret_cast<unsigned long long*>(dCnt + 1);
    P.bits_arena     = dBits;
    P.bits_cap       = devBitsCap;
    P.bits_used      = reinterpret_cast<unsigned long long*>(dCnt + 2);
    P.phase_cycles   = reinterpret_cast<unsigned long long*>(dCnt + 4);
    P.growth_size    = dGrowth;
    P.growth_buckets = dGrowth + ctx->growthSize.size();
    P.n_growth       = uint32_t(ctx->growthSize.size());
    P.flags          = std::getenv("MANTA_AMD_SERIAL_WALK") ? ASM_FLAG_SERIAL_WALK : 0u;
    P.locus_min_wl   = dMinWl;
    P.locus_max_wl   = dMaxWl;
    P.locus_ids      = dOrder;
    P.pl_codes       = dPlCodes;
    P.pl_nmask       = dPlMask;
    P.pl_read_len    = dPlLen;
    P.pl_code_off    = dPlCodeOff;
    P.pl_mask_off    = dPlMaskOff;
    P.upload_chunks_done = streaming ? dChunksDone : nullptr;
    P.chunk_shift        = (streaming && !streamingPiles) ? dStream + 1 : nullptr;
    P.pl_chunk_shift     = streamingPiles ? dPlShift : nullptr;
    P.chunk_loci         = streaming ? chunkLoci : 0;
    P.reserved2          = 0;
    P.small_min_seed_reads = smallMinSeedReads;
    P.small_max_iterations = smallMaxIterations;
    P.punt_ids             = nullptr;
    P.punt_count           = nullptr;
    P.n_loci_dev           = nullptr;
    P.lds_bytes            = ASM_LDS_BYTES;
    P.stop_before          = 0;
    lastParams             = P;
    nRerun                 = 0;
    if (smallMode) {
      rt::launch(small_assemble_kernel, grid, 0, P);
      return;
    }
    // streamed upload: if the runtime moves a chunk with a shader copy instead of the DMA engine, that copy needs a free
    // workgroup slot while a persistent kernel runs -- the general kernel leaves one slot free on a quarter of the CUs (the fast
    // kernel's three workgroups per CU leave plenty)
    int g = grid;
    if (streaming && g >= ctx->cuCount * 16) g = rt::roundGrid(g - ctx->cuCount);
    P.lds_bytes = ASM_LDS_BYTES;
    if (useFast) {
      using namespace manta_dev;
      // dCnt: [0] work counter of graph_kernel's list, [14] length of the general kernel's own list (genIds + punts), [15] its
      // work counter.  bLgCnt (qwords): [0] bytes of the slab arena in use, [1..2] loci per size class, [3..4] the class launches'
      // work counters
      const uint32_t nGen = uint32_t(genIds.size());
      rt::h2d(reinterpret_cast<uint32_t*>(dCnt + 14), &nGen, sizeof(uint32_t));
      uint64_t* dLg = bLgCnt.as<uint64_t>(16);
      rt::dzero(dLg, sizeof(uint64_t) * 16);
      LgArgs A;
      A.P            = P;
      A.P.n_loci     = uint32_t(fastIds.size());
      A.P.punt_ids   = dPunt;
      A.P.punt_count = reinterpret_cast<uint32_t*>(dCnt + 14);
      A.G.arena        = bLgArena.as<uint8_t>(lgArenaCap + 64);
      A.G.arena_cap    = lgArenaCap;
      A.G.arena_used   = reinterpret_cast<unsigned long long*>(dLg);
      A.G.slab_off     = bLgOff.as<uint64_t>(nLoci);
      A.G.class_ids    = bLgClassIds.as<uint32_t>(uint64_t(LG_CLASSES) * std::max<size_t>(1, fastIds.size()));
      A.G.class_count  = reinterpret_cast<uint32_t*>(dLg + 1);
      A.G.class_stride = uint32_t(fastIds.size());
      for (unsigned c = 0; c < LG_CLASSES; ++c) A.G.class_bytes[c] = classBytes[c];
      A.G.cls        = 0;
      A.G.flags      = (std::getenv("MANTA_AMD_LG_NO_PROOF") ? LG_FLAG_NO_PROOF : 0u) | (std::getenv("MANTA_AMD_LG_NO_RESCUE") ? LG_FLAG_NO_RESCUE : 0u);
      A.G.stats      = reinterpret_cast<uint32_t*>(dLg + 8);
      int maxGrid = 1;
      for (unsigned c = 0; c < LG_CLASSES; ++c) maxGrid = std::max(maxGrid, gridContig[c]);
      A.G.cws        = bCws.as<uint8_t>(cwsStride * uint64_t(maxGrid));
      A.G.cws_stride = cwsStride;
      A.G.round = A.G.last_round = 0;
      A.G.iter        = nullptr;  // (the big class' rounds set these)
      A.G.parena      = nullptr;
      A.G.parena_cap  = 0;
      A.G.parena_used = nullptr;
      A.G.next_ids = A.G.next_count = A.G.cyc_ids = A.G.cyc_count = nullptr;
      A.G.rws        = nullptr;
      A.G.rws_stride = 0;
      A.G.gws        = nullptr;
      A.G.rprof      = nullptr;
      // streamed upload: a chunk the runtime moves with a shader copy needs a free workgroup slot (and, as far as this launch can
      // know, LDS): graph_kernel's two workgroups per CU own all 160 KB, so a quarter of the CUs keep one slot free -- without it
      // the copies never run and the persistent workgroups wait for their chunks forever (seen on hardware, round 4)
      int gf = gridFast;
      if (streaming && gf >= ctx->cuCount * 2) gf -= std::max(1, ctx->cuCount / 4);
      // (the instantiation by the longest first word length among the loci of this launch: keys of 2 / 4 / 8 dwords)
      if (!fastIds.empty()) {
        uint32_t firstWl = opt.min_word_length;
        if (!locusMinWl.empty()) {
          firstWl = 0;
          for (const uint32_t l : fastIds) firstWl = std::max(firstWl, locusMinWl[l]);
        }
        if (firstWl <= 32)
          rt::launchWG(graph_kernel<2>, gf, int(LG_WAVES), LG_BUDGET, A);
        else if (firstWl <= 64)
          rt::launchWG(graph_kernel<4>, gf, int(LG_WAVES), LG_BUDGET, A);
        else
          rt::launchWG(graph_kernel<8>, gf, int(LG_WAVES), LG_BUDGET, A);
      }
      // contig_kernel over the small class' lists.  With the big class' word-length rounds in the same launch sequence it goes right behind the
      // first round: the metric's tail does not wait behind rounds it has no part in, and afterFirstRound() -- the point after which the
      // records of every locus that is done at its first word length are final -- covers the small class too
      bool smallContigsLaunched = false;
      auto launchSmallContigs   = [&] {
        if (smallContigsLaunched) return;
        smallContigsLaunched = true;
        for (unsigned c = 0; c < LG_CLASSES && !fastIds.empty(); ++c) {
          if (!classBytes[c]) continue;
          A.G.cls       = c;
          A.P.counter   = reinterpret_cast<uint32_t*>(dLg + 3) + c;
          A.P.lds_bytes = classBytes[c];
          rt::launchSingle(contig_kernel, gridContig[c], classBytes[c], A);
        }
      };
      firstRoundHookRan = false;
      if (!bigIds.empty()) {
        // the big class: its own work list (behind the small class' in dOrder), class lists and counters; slabs and punts shared
        LgArgs B         = A;
        B.P.n_loci       = uint32_t(bigIds.size());
        B.P.locus_ids    = dOrder + fastIds.size();
        B.P.counter      = reinterpret_cast<uint32_t*>(dLg + 7);
        B.G.class_ids    = bLgClassIdsBig.as<uint32_t>(uint64_t(LGL_CLASSES) * bigIds.size());
        B.G.class_count  = reinterpret_cast<uint32_t*>(dLg + 5);
        B.G.class_stride = uint32_t(bigIds.size());
        for (unsigned c = 0; c < LG_CLASSES; ++c) B.G.class_bytes[c] = (c < LGL_CLASSES) ? classBytesBig[c] : 0u;
        B.G.stats        = reinterpret_cast<uint32_t*>(dLg + 11);
        int maxGridBig = 1;
        for (unsigned c = 0; c < LGL_CLASSES; ++c) maxGridBig = std::max(maxGridBig, gridContigBig[c]);
        B.G.cws        = bCwsBig.as<uint8_t>(cwsStrideBig * uint64_t(maxGridBig));
        B.G.cws_stride = cwsStrideBig;
        int gb = gridBig;
        if (streaming && gb >= ctx->cuCount) gb -= std::max(1, ctx->cuCount / 4);  // (as above: one workgroup owns a CU's whole LDS)
        // (the instantiation by the longest word length the kernel may meet: the first one without the rounds, any of them with)
        uint32_t firstWl = bigRounds ? opt.max_word_length : opt.min_word_length;
        if (!locusMinWl.empty()) {
          firstWl = 0;
          for (const uint32_t l : bigIds) firstWl = std::max(firstWl, bigRounds ? locusMaxWl[l] : locusMinWl[l]);
        }
        if (!bigRounds) {
          if (firstWl <= 80)
            rt::launchWG(graph_big_kernel<5>, gb, int(LGL_WAVES), LGL_BUDGET, B);
          else
            rt::launchWG(graph_big_kernel<8>, gb, int(LGL_WAVES), LGL_BUDGET, B);
          for (unsigned c = 0; c < LGL_CLASSES; ++c) {
            B.G.cls       = c;
            B.P.counter   = reinterpret_cast<uint32_t*>(dLg + 9) + c;
            B.P.lds_bytes = classBytesBig[c];
            rt::launchSingle(contig_big_kernel, gridContigBig[c], classBytesBig[c], B);
          }
        } else {
          // One round per word length.  Round r: graph_big_kernel over the round's list (round 0: the class' work list; later: what
          // contig_big_kernel of round r - 1 sent on -- a repeat hit, :872-910) -> repeat_big_kernel over the graphs without a proof of
          // acyclicity (peel, exact repeat search, LDS class) -> contig_big_kernel per LDS class.  Every round has its own counters
          // (bLgRounds, 8 dwords per round: [0] graph work counter, [1..2] loci per class, [3..4] the class launches' work counters,
          // [5] graphs without a proof, [6] repeat_big_kernel's work counter, [7] loci sent on); the rounds after the first are launched
          // blind -- their list lengths sit in device memory -- with small grids: a round without work costs four empty launches.
          uint32_t* rc = bLgRounds.as<uint32_t>(8 * (LGL_MAX_ROUNDS + 1) + 32);
          rt::dzero(rc, sizeof(uint32_t) * (8 * (LGL_MAX_ROUNDS + 1) + 32));
          uint32_t* nextBuf = bLgNext.as<uint32_t>(2 * bigIds.size());
          B.G.iter        = bLgIter.as<LgIter>(nLoci);
          B.G.parena      = bLgPseudo.as<uint32_t>(pseudoArenaDw + 64);
          B.G.parena_cap  = pseudoArenaDw;
          B.G.parena_used = reinterpret_cast<unsigned long long*>(rc + 8 * LGL_MAX_ROUNDS);
          B.G.cyc_ids     = bLgCyc.as<uint32_t>(bigIds.size());
          B.G.rws         = bRws.as<uint8_t>(rwsStride * uint64_t(gridRepeat));
          B.G.rws_stride  = rwsStride;
          B.G.gws         = bGws.as<uint8_t>(uint64_t(32) * LGL_POOL_OVF * uint64_t(gridBig));
          B.G.rprof       = std::getenv("MANTA_AMD_DEBUG") ? reinterpret_cast<unsigned long long*>(rc + 8 * (LGL_MAX_ROUNDS + 1)) : nullptr;
          B.G.last_round  = bigRounds - 1;
          const int later = int(std::max<size_t>(32, bigIds.size() / 4));
          const bool oneClassLater = !(std::getenv("MANTA_AMD_BIG_ONE_CLASS") && std::atoi(std::getenv("MANTA_AMD_BIG_ONE_CLASS")) == 0);
          for (uint32_t r = 0; r < bigRounds; ++r) {
            uint32_t* cr     = rc + 8 * r;
            LgArgs    R      = B;
            R.G.round        = r;
            R.G.class_count  = cr + 1;
            R.G.cyc_count    = cr + 5;
            R.G.next_ids     = nextBuf + size_t(r & 1u) * bigIds.size();
            R.G.next_count   = cr + 7;
            R.P.counter      = cr;
            if (r > 0) {
              R.P.locus_ids  = nextBuf + size_t((r - 1) & 1u) * bigIds.size();
              R.P.n_loci     = 0;
              R.P.n_loci_dev = rc + 8 * (r - 1) + 7;
            }
            // (the later rounds hold cyclic graphs, nearly all of which need the large LDS class anyway: one contig launch instead of two
            // saves a launch tail per word length)
            if (oneClassLater && r > 0)
              for (unsigned c = 0; c + 1 < LGL_CLASSES; ++c) R.G.class_bytes[c] = 0;
            const int gg = (r == 0) ? gb : std::min(gb, later);
            if (firstWl <= 80)
              rt::launchWG(graph_big_kernel<5>, gg, int(LGL_WAVES), LGL_BUDGET, R);
            else
              rt::launchWG(graph_big_kernel<8>, gg, int(LGL_WAVES), LGL_BUDGET, R);
            R.P.n_loci_dev = nullptr;
            R.P.counter    = cr + 6;
            R.P.lds_bytes  = RPB_LDS_BYTES;
            rt::launch(repeat_big_kernel, (r == 0) ? gridRepeat : std::min(gridRepeat, rt::roundGrid(2 * later)), RPB_LDS_BYTES, R);
            for (unsigned c = 0; c < LGL_CLASSES; ++c) {
              if (oneClassLater && r > 0 && c + 1 < LGL_CLASSES) continue;
              R.G.cls       = c;
              R.P.counter   = cr + 3 + c;
              R.P.lds_bytes = classBytesBig[c];
              rt::launchSingle(contig_big_kernel, (r == 0) ? gridContigBig[c] : std::min(gridContigBig[c], later), classBytesBig[c], R);
            }
            if (r == 0 && bigRounds > 1) {
              launchSmallContigs();
              if (afterFirstRound) {
                afterFirstRound();
                firstRoundHookRan = true;
              }
            }
          }
        }
      }
      launchSmallContigs();
      P.locus_ids  = dPunt;
      P.n_loci     = nLoci;
      P.n_loci_dev = reinterpret_cast<uint32_t*>(dCnt + 14);
      P.counter    = reinterpret_cast<uint32_t*>(dCnt + 15);
      // nothing for this launch unless the pipeline handed something back: a small grid then (its waves find the list
      // length in device memory); the full grid when the host already knows of loci outside the envelope
      // (the big class hands back the piles with a cyclic graph -- tandem repeats, one in ten of the config-4/5 shape -- and each of those is a
      // long dependent chain in this kernel: one wave per expected locus, not a queue)
      if (nGen == 0) g = std::min(g, rt::roundGrid(std::max<int>(ctx->cuCount * 4, int(std::min<uint64_t>(bigIds.size() / 8 + 1, uint64_t(g))))));
      rt::launch(assemble_kernel, g, ASM_LDS_BYTES, P);
    } else {
      rt::launch(assemble_kernel, g, ASM_LDS_BYTES, P);
    }
    // loci that did not fit the typical-case workspace are counted into dCnt[3] (see rerunCapacityFailures)
    // ... and so are the loci whose reads hold bytes outside {A,C,G,T,N} that cannot be masked exactly (ASM_E_ALPHABET): the same
    // pass runs those again on the byte-generic kernel
    CountStatusParams C;
    C.loci    = dLoci;
    C.n_loci  = nLoci;
    C.code    = ASM_E_TABLE_FULL;
    C.counter = reinterpret_cast<unsigned long long*>(dCnt + 3);
    rt::launch(count_status_kernel, rt::roundGrid(int(std::min<uint64_t>((nLoci + 63) / 64, 64))), 0, C);
    if (!dPlCodes) {
      C.code = ASM_E_ALPHABET;
      rt::launch(count_status_kernel, rt::roundGrid(int(std::min<uint64_t>((nLoci + 63) / 64, 64))), 0, C);
    }
  }

  /// Call once the assembler of launch() has finished, with the value of dCnt[3] (capacity_failures).  Loci whose pile did not
  /// fit the typical-case workspace (ASM_E_TABLE_FULL: many long contigs fed back as pseudo reads) run again, one wave each, on a
  /// workspace sized for the worst case; they write into the same records and arenas, so nothing downstream changes.
  void rerunCapacityFailures(const uint64_t failures)
  {
    if (failures == 0 || smallMode) return;
    std::vector<AsmLocusOut> st(nLoci);
    rt::d2h(st.data(), dLoci, sizeof(AsmLocusOut) * nLoci);
    std::vector<uint32_t> ids, alphaIds;
    for (uint32_t l = 0; l < nLoci; ++l) {
      if (st[l].status == ASM_E_TABLE_FULL) ids.push_back(l);
      if (st[l].status == ASM_E_ALPHABET) alphaIds.push_back(l);
    }
    nRerun = 0;
    const size_t wsBudget = workspaceBudget(size_t(32) << 30);
    if (!ids.empty()) {
      int g          = int(std::min<uint64_t>(ids.size(), std::max<uint64_t>(1, wsBudget / stride2)));
      g              = rt::roundGrid(std::min(g, std::max(1, ctx->cuCount * 4)));
      uint32_t* dIds = bFailIds.as<uint32_t>(ids.size());
      rt::h2d(dIds, ids.data(), sizeof(uint32_t) * ids.size());
      rt::dzero(dCnt + 12, sizeof(uint64_t) * 2);
      AsmParams P = lastParams;
      P.ws        = bWs2.as<uint8_t>(stride2 * uint64_t(g));
      P.ws_stride = stride2;
      P.cap_slots = capSlots2;
      P.cap_nodes = capNodes2;
      P.cap_words = capWords2;
      P.n_loci    = uint32_t(ids.size());
      P.locus_ids = dIds;
      P.counter   = reinterpret_cast<uint32_t*>(dCnt + 12);
      rt::launch(assemble_kernel, g, ASM_LDS_BYTES, P);
      rt::sync();
      nRerun = uint32_t(ids.size());
      // a locus that overflowed the typical-case workspace may only now reach the junk-byte tests (they run after the pack / after the
      // graph is built): what reports ASM_E_ALPHABET on the worst-case workspace joins the byte-generic run below
      rt::d2h(st.data(), dLoci, sizeof(AsmLocusOut) * nLoci);
      for (const uint32_t l : ids)
        if (st[l].status == ASM_E_ALPHABET) alphaIds.push_back(l);
      if (std::getenv("MANTA_AMD_DEBUG"))
        std::fprintf(stderr, "manta_amd: %zu of %u loci ran again on the worst-case workspace (%.1f MB per wave)\n", ids.size(), nLoci, double(stride2) / 1e6);
    }
    if (!alphaIds.empty()) {
      // byte-generic run (AssemblerT<8>: the reads' bytes as symbols, 4 per code dword) on a worst-case workspace
      const uint64_t    nCandMax = 2ull * opt.max_assembly_count;
      const uint32_t    capWordsG = uint32_t(std::min<uint64_t>(uint64_t(capWords2) * 4 + 64, 0x7fffffffull));
      const AsmWsLayout LG = asmWorkspaceLayout(capSlots2, capNodes2, capWordsG, capReads, maxContigLen, wMax, opt.max_assembly_count);
      const uint64_t    strideG = (LG.total + 255) & ~uint64_t(255);
      (void)nCandMax;
      int g          = int(std::min<uint64_t>(alphaIds.size(), std::max<uint64_t>(1, wsBudget / strideG)));
      g              = rt::roundGrid(std::min(g, std::max(1, ctx->cuCount * 4)));
      uint32_t* dIds = bFailIds.as<uint32_t>(alphaIds.size());
      rt::h2d(dIds, alphaIds.data(), sizeof(uint32_t) * alphaIds.size());
      rt::dzero(dCnt + 12, sizeof(uint64_t) * 2);
      AsmParams P = lastParams;
      P.ws        = bWs2.as<uint8_t>(strideG * uint64_t(g));
      P.ws_stride = strideG;
      P.cap_slots = capSlots2;
      P.cap_nodes = capNodes2;
      P.cap_words = capWordsG;
      P.n_loci    = uint32_t(alphaIds.size());
      P.locus_ids = dIds;
      P.counter   = reinterpret_cast<uint32_t*>(dCnt + 12);
#ifndef MANTA_DEV_NO_GENERIC
      rt::launch(assemble_generic_kernel, g, ASM_LDS_BYTES, P);
#else
      throw rt::Error("developer build without the byte-generic kernel");
#endif
      rt::sync();
      nRerun += uint32_t(alphaIds.size());
      if (std::getenv("MANTA_AMD_DEBUG"))
        std::fprintf(stderr, "manta_amd: %zu of %u loci ran again on the byte-generic kernel\n", alphaIds.size(), nLoci);
    }
  }

  /// Device -> pinned host staging of everything the assembler produced, with EXACT sizes: the fixed records and the
  /// arena counters first (one round trip), then exactly the used part of the text / bitset arenas.
  /// `moreCopies` lets a pipeline queue its own copies behind the second round so that one sync covers them.
  /// `firstCopies` / `moreCopies` let a pipeline queue its own copies in the first / second round trip; with
  /// `sparseContigs` false the (mostly empty) per-slot contig records stay on the device (the pipeline brings packed ones)
  /// The staging is split so that a pipeline can queue it right behind its last kernel: stageEnqueue() queues the counters,
  /// the locus records and -- speculatively -- the used part of the arenas as far as it is known or predicted (the sizes of
  /// the previous run of this stage, a quarter on top: consecutive blocks of a batch look alike); stageFinish(), after the
  /// stream has drained, fetches what the speculation missed (a second round trip only then).
  uint64_t seqCopied = 0, bitsCopied = 0, seqLast = 0, bitsLast = 0;
  bool     stageQueued = false;
  template <typename F0>
  void stageEnqueue(F0 firstCopies, bool sparseContigs = true)
  {
    hCnt  = pCnt.as<uint64_t>(16);
    hLoci = pLoci.as<AsmLocusOut>(nLoci);
    rt::d2hAsync(hCnt, dCnt, sizeof(uint64_t) * 16);
    rt::d2hAsync(hLoci, dLoci, sizeof(AsmLocusOut) * nLoci);
    if (sparseContigs) {
      hCont = pCont.as<AsmContigOut>(uint64_t(nLoci) * opt.max_assembly_count);
      rt::d2hAsync(hCont, dCont, sizeof(AsmContigOut) * uint64_t(nLoci) * opt.max_assembly_count);
    }
    seqCopied  = std::min<uint64_t>(devSeqCap, seqLast + seqLast / 4 + (seqLast ? 4096 : 0));
    bitsCopied = std::min<uint64_t>(devBitsCap, bitsLast + bitsLast / 4 + (bitsLast ? 512 : 0));
    hSeq       = pSeq.as<uint8_t>(seqCopied + 1);
    hBits      = pBits.as<uint64_t>(bitsCopied + 1);
    rt::d2hAsync(hSeq, dSeq, seqCopied);
    rt::d2hAsync(hBits, dBits, sizeof(uint64_t) * bitsCopied);
    firstCopies();
    stageQueued = true;
  }
  /// after the stream has drained.  moreCopies(queued&) queues what the pipeline still misses and sets `queued` if it did.
  template <typename F>
  void stageFinish(F moreCopies)
  {
    stageQueued = false;
    seqUsedDev  = std::min<uint64_t>(hCnt[1], devSeqCap);
    bitsUsedDev = std::min<uint64_t>(hCnt[2], devBitsCap);
    seqLast     = seqUsedDev;
    bitsLast    = bitsUsedDev;
    bool queued = false;
    if (seqUsedDev > seqCopied) {  // (the staging buffer keeps what it holds when it grows)
      hSeq = pSeq.as<uint8_t>(seqUsedDev + 1, true);
      rt::d2hAsync(hSeq + seqCopied, dSeq + seqCopied, seqUsedDev - seqCopied);
      queued = true;
    }
    if (bitsUsedDev > bitsCopied) {
      hBits = pBits.as<uint64_t>(bitsUsedDev + 1, true);
      rt::d2hAsync(hBits + bitsCopied, dBits + bitsCopied, sizeof(uint64_t) * (bitsUsedDev - bitsCopied));
      queued = true;
    }
    moreCopies(queued);
    if (queued) rt::sync();
    nContigsOut = pseudoBytesOut = pseudoCountOut = 0;
    for (uint32_t l = 0; l < nLoci; ++l) {
      const AsmLocusOut& h(hLoci[l]);
      if (h.status != ASM_OK) continue;
      nContigsOut += h.n_contigs;
      pseudoCountOut += h.n_pseudo;
      for (uint32_t q = 0; q < h.n_pseudo; ++q) pseudoBytesOut += hBits[h.pseudo_len_off + q];
    }
    staged = true;
    ldsFallbacks = useFast ? uint32_t(hCnt[14] & 0xffffffffu) - uint32_t(genIds.size()) : 0u;
    if (std::getenv("MANTA_AMD_DEBUG") && useFast) {
      uint32_t st[2] = {0, 0}, stBig[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, clsBig[2] = {0, 0};
      rt::d2h(st, bLgCnt.as<uint64_t>(16) + 8, sizeof(st));
      rt::d2h(stBig, bLgCnt.as<uint64_t>(16) + 11, sizeof(stBig));
      rt::d2h(clsBig, bLgCnt.as<uint64_t>(16) + 5, sizeof(clsBig));
      std::fprintf(stderr, "manta_amd: LDS assembler pipeline: %zu + %zu (big class) loci, %u handed to the general kernel (+ %zu outside its envelope); %u + %u graphs came "
                           "with a proof of acyclicity, %u + %u reads re-anchored; big class: %u / %u loci in its two contig LDS classes\n", fastIds.size(), bigIds.size(),
                   ldsFallbacks, genIds.size(), st[0], stBig[0], st[1], stBig[1], clsBig[0], clsBig[1]);
      if (!bigIds.empty())
        std::fprintf(stderr, "manta_amd: big class, %u word-length rounds; handed back: %u envelope, %u table / set pool, %u words / side tables / class, %u slab arena, %u by "
                             "repeat_big_kernel, %u by contig_big_kernel, %u pseudo arena, %u out of rounds\n", bigRounds, stBig[2], stBig[3], stBig[4], stBig[5], stBig[6],
                     stBig[7], stBig[8], stBig[9]);
      if (bigRounds) {
        unsigned long long rp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tot = 0;
        uint32_t           perRound[8 * manta_dev::LGL_MAX_ROUNDS];
        uint32_t*          rc = bLgRounds.as<uint32_t>(8 * (manta_dev::LGL_MAX_ROUNDS + 1) + 32);
        rt::d2h(rp, rc + 8 * (manta_dev::LGL_MAX_ROUNDS + 1), sizeof(rp));
        rt::d2h(perRound, rc, sizeof(perRound));
        for (int i = 0; i < 7; ++i) tot += rp[i];
        std::fprintf(stderr, "manta_amd: repeat_big_kernel clocks: renumbering %.1f%% component scan %.1f%% hash+insertion %.1f%% order1 %.1f%% order2 %.1f%% search in the reference's order %.1f%% rest %.1f%% (%.0f k clocks per graph without a proof, first three rounds)\n",
                     tot ? 100.0 * rp[0] / tot : 0, tot ? 100.0 * rp[1] / tot : 0, tot ? 100.0 * rp[2] / tot : 0, tot ? 100.0 * rp[3] / tot : 0, tot ? 100.0 * rp[4] / tot : 0,
                     tot ? 100.0 * rp[5] / tot : 0, tot ? 100.0 * rp[6] / tot : 0, perRound[5] ? double(tot) / 1e3 / double(std::max<uint32_t>(1, perRound[5] + perRound[8 + 5] + perRound[16 + 5])) : 0.0);
        std::fprintf(stderr, "manta_amd: rounds (graphs without a proof / loci sent on):");
        for (uint32_t r = 0; r < bigRounds; ++r) std::fprintf(stderr, " %u/%u", perRound[8 * r + 5], perRound[8 * r + 7]);
        std::fprintf(stderr, "\n");
      }
    }
    if (std::getenv("MANTA_AMD_PROFILE")) {
      static const char* namesGeneral[8] = {"pack", "table", "links", "cycle-check", "exact", "seed", "walk", "select+emit"};
#ifdef MANTA_LG_PROFILE_GRAPH
      static const char* namesLds[8]     = {"pack", "table", "links", "counts+radix", "ties+ids", "slab+sets+init", "preds+sibs", "spec+write"};
#else
      static const char* namesLds[8]     = {"pack", "table", "sort+records", "cycle-check", "slab write/read", "seed+replay", "walk", "select+emit"};
#endif
      const char* const* names = useFast ? namesLds : namesGeneral;
      uint64_t           tot = 0;
      for (int i = 0; i < 8; ++i) tot += hCnt[4 + i];
      std::fprintf(stderr, "manta_amd %s phase share (shader clocks summed over %u loci; graph_kernel: clocks of one wave of the workgroup):", useFast ? "graph_kernel + contig_kernel" : "assemble_kernel", nLoci);
      for (int i = 0; i < 8; ++i) std::fprintf(stderr, " %s=%.1f%%", names[i], tot ? 100.0 * double(hCnt[4 + i]) / double(tot) : 0.0);
      std::fprintf(stderr, " | avg clocks/locus=%.0f\n", double(tot) / nLoci);
    }
  }
  void stageOut()
  {
    stageEnqueue([] {});
    rt::sync();
    stageFinish([](bool&) {});
  }

  /// exact sizes compact() will write (valid after stageOut): contig records, text bytes, bitset qwords
  struct SparseContigs {
    const AsmStage* st;
    const AsmContigOut& operator()(uint32_t l, uint32_t c) const { return st->hCont[uint64_t(l) * st->opt.max_assembly_count + c]; }
  };
  void exactSizes(uint64_t& nContigs, uint64_t& seqBytes, uint64_t& bitsWords) const { exactSizes(SparseContigs{this}, nContigs, seqBytes, bitsWords); }
  template <typename ContigAt>
  void exactSizes(ContigAt contigAt, uint64_t& nContigs, uint64_t& seqBytes, uint64_t& bitsWords) const
  {
    nContigs = nContigsOut;
    seqBytes = pseudoBytesOut;
    bitsWords = pseudoCountOut;
    for (uint32_t l = 0; l < nLoci; ++l) {
      const AsmLocusOut& h(hLoci[l]);
      if (h.status != ASM_OK) continue;
      for (uint32_t c = 0; c < h.n_contigs; ++c) seqBytes += contigAt(l, c).seq_len;
      bitsWords += 2ull * h.n_words * h.n_contigs;
    }
  }

  /// what compact() writes for the loci [lBegin, lEnd)
  template <typename ContigAt>
  void rangeSizes(ContigAt contigAt, uint32_t lBegin, uint32_t lEnd, uint64_t& nContigs, uint64_t& seqBytes, uint64_t& bitsWords) const
  {
    nContigs = seqBytes = bitsWords = 0;
    for (uint32_t l = lBegin; l < lEnd; ++l) {
      const AsmLocusOut& h(hLoci[l]);
      if (h.status != ASM_OK) continue;
      nContigs += h.n_contigs;
      for (uint32_t c = 0; c < h.n_contigs; ++c) seqBytes += contigAt(l, c).seq_len;
      for (uint32_t q = 0; q < h.n_pseudo; ++q) seqBytes += hBits[h.pseudo_len_off + q];
      bitsWords += 2ull * h.n_words * h.n_contigs + h.n_pseudo;
    }
  }

  /// upper bounds of what compact() writes into the caller's arenas (cheap: counters only)
  void outputSizes(uint64_t& nContigs, uint64_t& seqBytes, uint64_t& bitsWords) const
  {
    uint64_t c[3];
    rt::d2h(c, dCnt, sizeof(c));
    nContigs  = uint64_t(nLoci) * opt.max_assembly_count + 1;
    seqBytes  = std::min<uint64_t>(c[1], devSeqCap) + 64;
    bitsWords = std::min<uint64_t>(c[2], devBitsCap) + 64;
  }

  /// staging -> the caller's records and arenas.  Offsets written into the records are relative to the arena pointers
  /// passed here plus `seqBase` / `bitsBase` / `contigBase` (a whole-batch call hands every block its own region of the
  /// caller's arenas).
  int compact(
      manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs, uint64_t contigs_cap, uint8_t* seq_arena,
      uint64_t seq_arena_cap, uint64_t* seq_arena_used, uint64_t* bits_arena, uint64_t bits_arena_cap, uint64_t* bits_arena_used,
      uint64_t contigBase = 0, uint64_t seqBase = 0, uint64_t bitsBase = 0)
  {
    return compact(SparseContigs{this}, loci, contigs, contigs_cap, seq_arena, seq_arena_cap, seq_arena_used, bits_arena, bits_arena_cap,
                   bits_arena_used, contigBase, seqBase, bitsBase);
  }
  template <typename ContigAt>
  int compact(
      ContigAt contigAt, manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs, uint64_t contigs_cap, uint8_t* seq_arena,
      uint64_t seq_arena_cap, uint64_t* seq_arena_used, uint64_t* bits_arena, uint64_t bits_arena_cap, uint64_t* bits_arena_used,
      uint64_t contigBase, uint64_t seqBase, uint64_t bitsBase, uint32_t lBegin = 0, uint32_t lEnd = ~0u)
  {
    // [lBegin, lEnd): the loci this call handles; the output pointers / bases are those of that range's first record
    uint64_t seqUsed = 0, bitsUsed = 0, nContigs = 0;
    int      worst = MANTA_OK;
    lEnd           = std::min(lEnd, nLoci);
    for (uint32_t l = lBegin; l < lEnd; ++l) {
      const AsmLocusOut&        h(hLoci[l]);
      manta_asm_locus_result_t& o(loci[l]);
      std::memset(&o, 0, sizeof(o));
      o.status       = asmStatusToAbi(h.status);
      o.first_contig = uint32_t(contigBase + nContigs);
      if (o.status != MANTA_OK) {
        worst = o.status;
        if (std::getenv("MANTA_AMD_DEBUG") || std::getenv("MANTA_AMD_DEBUG_STATUS"))
          std::fprintf(stderr, "manta_amd: locus %u device status %d (k=%u iter=%u)\n", l, h.status, h.final_word_length, h.n_iterations);
        continue;
      }
      o.n_contigs         = h.n_contigs;
      o.n_words           = h.n_words;
      o.n_pseudo          = h.n_pseudo;
      o.final_word_length = h.final_word_length;
      o.n_iterations      = h.n_iterations;
      o.cyclic_iterations = h.cyclic_iterations;
      if (nContigs + h.n_contigs > contigs_cap) return fail(ctx, MANTA_E_CAPACITY, "contig array too small");
      for (uint32_t c = 0; c < h.n_contigs; ++c) {
        const AsmContigOut& hc(contigAt(l, c));
        manta_asm_contig_t& oc(contigs[nContigs++]);
        if (seqUsed + hc.seq_len > seq_arena_cap || bitsUsed + 2ull * h.n_words > bits_arena_cap)
          return fail(ctx, MANTA_E_CAPACITY, "output arena too small");
        std::memcpy(seq_arena + seqUsed, hSeq + hc.seq_off, hc.seq_len);
        std::memcpy(bits_arena + bitsUsed, hBits + hc.bits_off, sizeof(uint64_t) * 2 * h.n_words);
        oc.seq_off            = seqBase + seqUsed;
        oc.seq_len            = hc.seq_len;
        oc.support_off        = bitsBase + bitsUsed;
        oc.reject_off         = bitsBase + bitsUsed + h.n_words;
        oc.seed_read_count    = smallMode ? hc.reserved : 0u;  // (runIterativeAssembler never writes it; runSmallAssembler does)
        oc.conservative_begin = hc.cons_begin;
        oc.conservative_end   = hc.cons_end;
        seqUsed += hc.seq_len;
        bitsUsed += 2ull * h.n_words;
      }
      uint64_t pBytes = 0;
      for (uint32_t q = 0; q < h.n_pseudo; ++q) pBytes += hBits[h.pseudo_len_off + q];
      if (seqUsed + pBytes > seq_arena_cap || bitsUsed + h.n_pseudo > bits_arena_cap)
        return fail(ctx, MANTA_E_CAPACITY, "output arena too small");
      std::memcpy(seq_arena + seqUsed, hSeq + h.pseudo_off, pBytes);
      std::memcpy(bits_arena + bitsUsed, hBits + h.pseudo_len_off, sizeof(uint64_t) * h.n_pseudo);
      o.pseudo_seq_off = seqBase + seqUsed;
      o.pseudo_len_off = bitsBase + bitsUsed;
      seqUsed += pBytes;
      bitsUsed += h.n_pseudo;
    }
    if (seq_arena_used) *seq_arena_used = seqUsed;
    if (bits_arena_used) *bits_arena_used = bitsUsed;
    if (worst != MANTA_OK) return fail(ctx, worst, "one or more loci failed; see per-locus status");
    return MANTA_OK;
  }

  int fetch(
      manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs, uint64_t contigs_cap, uint8_t* seq_arena,
      uint64_t seq_arena_cap, uint64_t* seq_arena_used, uint64_t* bits_arena, uint64_t bits_arena_cap, uint64_t* bits_arena_used)
  {
    stageOut();
    return compact(loci, contigs, contigs_cap, seq_arena, seq_arena_cap, seq_arena_used, bits_arena, bits_arena_cap, bits_arena_used);
  }
};

}  // namespace

struct manta_smallsv {
  manta_ctx_t*          ctx;
  AsmStage              asmStage;
  manta_asm_options_t   opt{};
  manta_align_scores_t  scores{};
  int32_t               largeIndel = 0;
  uint32_t              nLoci      = 0;
  uint64_t              refBytes = 0, maxRef = 0;
  bool                  uploaded = false, ran = false;
  DevBuf                dRefs, dRefOff, dCuts, dTasks, dInfo, dResults, dBucketIds, dBucketIds2, dSmall, dCigar, dTable, dPtrWs;
  rt::Event             evStart, evAsm, evSched, evAlign, refsReady;
  rt::Stream            main;  // everything of this pipeline except the aligner buckets
  rt::Stream            copy;  // streamed upload of the read bases (whole-batch calls)
  bool                  streamUploads = false, refsOnCopy = false;
  rt::Stream            side[3];
  rt::Event             sideDone[3];
  manta_smallsv_stats_t stats{};
  // device -> host staging (pinned)
  DevBuf                dFirst, dPacked, dCigPacked, dPackCnt;
  PinnedBuf             pFirst, pPacked, pCig, pPackCnt;
  uint32_t*             hFirst  = nullptr;
  PackedContigOut*      hPacked = nullptr;
  uint32_t*             hCig    = nullptr;
  uint32_t*             hPackCnt = nullptr;  // [0] packed contigs, [1] packed cigar words
  uint64_t              packLast[2] = {0, 0}, packedCopied = 0, cigCopied = 0;  // speculative staging (pipeStageEnqueue)
  PinnedBuf             pSmall;
  uint32_t              lastSmall[40] = {0};  // bucket counters of the previous run (grids of the next one)
  bool                  bucketHistory = false;
  bool                  stageBehindRun = false;  // whole-batch calls: the run queues the staging behind its last kernel
  bool                  staged = false;
  explicit manta_smallsv(manta_ctx_t* c) : ctx(c), asmStage(c) {}
};


struct manta_spanning {
  manta_ctx_t*          ctx;
  AsmStage              asmStage;
  manta_asm_options_t   opt{};
  manta_align_scores_t  scores{};
  int32_t               jumpScore = 0;
  uint32_t              nLoci     = 0;
  uint64_t              ref1Bytes = 0, ref2Bytes = 0;
  bool                  uploaded = false, ran = false;
  DevBuf                dRefs1, dRef1Off, dRefs2, dRef2Off, dCuts, dTasks, dTasks2, dInfo, dResults, dResults2, dBucketIds, dBucketIds2, dSmall,
      dCigar, dPtrWs;
  rt::Event             evStart, evAsm, evSched, evAlign, refsReady;
  rt::Stream            main;
  rt::Stream            copy;  // streamed upload of the read bases (whole-batch calls)
  bool                  streamUploads = false, refsOnCopy = false;
  rt::Stream            side[3];
  rt::Event             sideDone[3];
  manta_smallsv_stats_t stats{};
  DevBuf                dFirst, dPacked, dCigPacked, dPackCnt;
  PinnedBuf             pFirst, pPacked, pCig, pPackCnt;
  // the early alignment pass (spanningRunImpl): the contigs of the loci that are final after the first word length are aligned while the
  // word-length rounds of the tandem piles still run on `main`
  // (`early` is a CU-masked stream like its side streams: the runtime multiplexes ordinary streams onto a few hardware queues -- the first
  // hardware run had `early` on main's queue, behind every launch of the rounds; a masked stream owns its queue)
  std::unique_ptr<rt::Stream> early;
  std::unique_ptr<rt::Stream> sideEarly[3];  // its aligner buckets: CU-masked streams (a share of the CUs stays free for the rounds' kernels)
  int                   sideEarlyReserved = -1;
  rt::Event             evRound0;
  DevBuf                dPassMask;
  uint32_t              earlyLoci = 0;      // loci the last run aligned early
  std::vector<JumpCuts> hostCuts;  // the caller's cuts of the uploaded batch
  uint32_t*             hFirst  = nullptr;
  PackedContigOut*      hPacked = nullptr;
  uint32_t*             hCig    = nullptr;
  uint32_t*             hPackCnt = nullptr;
  uint64_t              packLast[2] = {0, 0}, packedCopied = 0, cigCopied = 0;
  bool                  stageBehindRun = false;
  bool                  staged = false;
  explicit manta_spanning(manta_ctx_t* c) : ctx(c), asmStage(c) {}
};

namespace {

/// the last kernel of a pipeline run: dense per-contig records + back-to-back CIGARs (pack_results_kernel)
template <typename Pipe>
void launchPack(Pipe* b, const AlignTaskDev* tasks, const AlignTaskDev* tasks2, const AlignResultDev* res, const AlignResultDev* res2,
                const SmallSvTaskInfo* infoSmall, const SpanTaskInfo* infoSpan, const uint32_t* cigar, uint64_t cigarCap)
{
  const uint32_t nLoci  = b->nLoci;
  const uint64_t nSlots = uint64_t(nLoci) * b->opt.max_assembly_count;
  PackParams     K;
  K.loci               = b->asmStage.dLoci;
  K.contigs            = b->asmStage.dCont;
  K.n_loci             = nLoci;
  K.max_assembly_count = b->opt.max_assembly_count;
  K.tasks              = tasks;
  K.tasks2             = tasks2;
  K.results            = res;
  K.results2           = res2;
  K.info_small         = infoSmall;
  K.info_span          = infoSpan;
  K.cigar              = cigar;
  K.first              = b->dFirst.template as<uint32_t>(nLoci);
  K.packed             = b->dPacked.template as<PackedContigOut>(nSlots);
  K.cigar_packed       = b->dCigPacked.template as<uint32_t>(cigarCap + 16);
  K.counters           = b->dPackCnt.template as<uint32_t>(4);
  rt::dzero(K.counters, 16);
  rt::launch(pack_results_kernel, rt::roundGrid(int(std::min<uint64_t>((nLoci + 63) / 64, uint64_t(std::max(1, b->ctx->cuCount * 8))))), 0, K);
}

}  // namespace

extern "C" {

uint32_t manta_abi_version(void) { return MANTA_ABI_VERSION; }
uint64_t manta_batch_stats_size(void) { return sizeof(manta_batch_stats_t); }

int manta_ctx_create(int device_id, manta_ctx_t** out)
{
  if (!out) {
    g_createError = "manta_ctx_create: null output pointer";
    return MANTA_E_INVALID_ARG;
  }
  *out = nullptr;
  try {
    rt::init(device_id);
    if (!stringHashMatchesLibstdcxx()) {
      g_createError = "this libstdc++'s std::hash<std::string> is not the Murmur variant the exact repeat search restates";
      return MANTA_E_UNSUPPORTED;
    }
    manta_ctx_t* ctx = new manta_ctx();
    ctx->deviceName  = rt::deviceName();
    ctx->cuCount     = rt::cuCount();
    ctx->deviceId    = rt::currentDevice();
    *out             = ctx;
    return MANTA_OK;
  } catch (const std::exception& e) {
    g_createError = e.what();
    return MANTA_E_NO_DEVICE;
  }
}

void manta_ctx_destroy(manta_ctx_t* ctx)
{
  delete ctx;
}

const char* manta_last_error(const manta_ctx_t* ctx)
{
  return ctx ? ctx->lastError.c_str() : g_createError.c_str();
}

const char* manta_ctx_device_name(const manta_ctx_t* ctx)
{
  return ctx ? ctx->deviceName.c_str() : "";
}

int manta_align_batch(
    manta_ctx_t* ctx, int kind, const manta_align_scores_t* scores, int32_t extra_score, uint32_t n_tasks,
    const manta_align_task_t* tasks, const uint8_t* seq_arena, uint64_t seq_arena_bytes, manta_align_result_t* results,
    uint32_t* cigar_arena, uint64_t cigar_arena_cap, uint64_t* cigar_arena_used)
{
  if (!ctx) return MANTA_E_INVALID_ARG;
  if (!scores || (n_tasks && (!tasks || !seq_arena || !results || !cigar_arena)))
    return fail(ctx, MANTA_E_INVALID_ARG, "manta_align_batch: null argument");
  if (kind < 0 || kind > 2) return fail(ctx, MANTA_E_INVALID_ARG, "manta_align_batch: unknown aligner kind");
  if (kind == MANTA_ALIGNER_JUMP && scores->is_allow_edge_insertion)
    return fail(ctx, MANTA_E_INVALID_ARG, "GlobalJumpAligner does not support isAllowEdgeInsertion");
  if (cigar_arena_used) *cigar_arena_used = 0;
  if (n_tasks == 0) return MANTA_OK;

  try {
    rt::setDevice(ctx->deviceId);  // the calling thread may have another device current (multi-GPU hosts)
    rt::ScopedStream onStream(ctx->stream);
    // ---- validate + bucket by columns-per-lane (E) ----
    std::vector<AlignTaskDev>          dev(n_tasks);
    std::vector<std::vector<uint32_t>> buckets(kNumESet);
    std::vector<uint64_t>              bucketMaxRef(kNumESet, 0);
    uint64_t                           cigarDevWords = 0;
    int                                worst         = MANTA_OK;
    for (uint32_t i = 0; i < n_tasks; ++i) {
      const manta_align_task_t& t(tasks[i]);
      manta_align_result_t&     r(results[i]);
      std::memset(&r, 0, sizeof(r));
      const bool jump = (kind == MANTA_ALIGNER_JUMP);
      auto outside = [&](uint64_t off, uint64_t len) { return off > seq_arena_bytes || len > seq_arena_bytes - off; };
      if (outside(t.query_off, t.query_len) || outside(t.ref1_off, t.ref1_len) || (jump && outside(t.ref2_off, t.ref2_len)))
        return fail(ctx, MANTA_E_INVALID_ARG, "manta_align_batch: task " + std::to_string(i) + " outside the sequence arena");
      if (t.query_len == 0 || t.ref1_len == 0 || (jump && t.ref2_len == 0)) {
        r.status = MANTA_E_EMPTY_SEQ;  // GlobalJumpAlignerImpl.hpp:50-58, GlobalAlignerImpl.hpp:44-49
        worst    = MANTA_E_EMPTY_SEQ;
        continue;
      }
      const int eIdx = pickE(t.query_len);
      AlignTaskDev& d(dev[i]);
      d.query     = reinterpret_cast<const uint8_t*>(uintptr_t(t.query_off));  // rebased onto the device arena below
      d.ref1      = reinterpret_cast<const uint8_t*>(uintptr_t(t.ref1_off));
      d.ref2      = reinterpret_cast<const uint8_t*>(uintptr_t(jump ? t.ref2_off : 0));
      d.query_len = t.query_len;
      d.ref1_len  = t.ref1_len;
      d.ref2_len  = jump ? t.ref2_len : 0;
      d.cigar_off = uint32_t(cigarDevWords);
      cigarDevWords += 4ull * t.query_len + 16;
      if (cigarDevWords > 0xffffffffull) return fail(ctx, MANTA_E_UNSUPPORTED, "manta_align_batch: batch too large (cigar workspace)");
      buckets[eIdx].push_back(i);
      bucketMaxRef[eIdx] = std::max<uint64_t>(bucketMaxRef[eIdx], alignSlabRefLen(kind, kESet[eIdx], d.query_len, uint64_t(d.ref1_len) + d.ref2_len));
    }

    // ---- stage ----
    uint8_t*        dSeq     = ctx->dSeq.as<uint8_t>(seq_arena_bytes);
    AlignTaskDev*   dTasks   = ctx->dTasks.as<AlignTaskDev>(n_tasks);
    AlignResultDev* dResults = ctx->dResults.as<AlignResultDev>(n_tasks);
    uint32_t*       dCigar   = ctx->dCigar.as<uint32_t>(cigarDevWords + 1);
    uint32_t*       dIds     = ctx->dTaskIds.as<uint32_t>(n_tasks);
    uint32_t*       dCounter = ctx->dCounter.as<uint32_t>(kNumESet);
    for (AlignTaskDev& d : dev) {
      d.query = dSeq + uintptr_t(d.query);
      d.ref1  = dSeq + uintptr_t(d.ref1);
      d.ref2  = dSeq + uintptr_t(d.ref2);
    }
    rt::h2d(dSeq, seq_arena, seq_arena_bytes);
    rt::h2d(dTasks, dev.data(), sizeof(AlignTaskDev) * n_tasks);
    rt::dzero(dCounter, sizeof(uint32_t) * kNumESet);
    rt::dzero(dResults, sizeof(AlignResultDev) * n_tasks);

    const int    maxWaves  = std::max(1, ctx->cuCount * alignWavesPerCu());
    const size_t wsBudget  = workspaceBudget(size_t(24) << 30);
    size_t       idsCursor = 0;
    for (int b = 0; b < kNumESet; ++b) {
      if (buckets[b].empty()) continue;
      const bool     pair   = alignUsesPairs(kind, b, scores->match, scores->mismatch, scores->open, scores->extend, scores->off_edge, extra_score,
                                             scores->is_allow_edge_insertion ? 1 : 0, bucketMaxRef[b]);
      const uint64_t stride = ((pair ? 2 : 1) * alignPtrSlabBytes(kind, kESet[b], bucketMaxRef[b]) + 255) & ~uint64_t(255);
      int            grid   = int(std::min<size_t>(pair ? (buckets[b].size() + 1) / 2 : buckets[b].size(), size_t(maxWaves)));
      grid                  = int(std::max<size_t>(1, std::min<size_t>(size_t(grid), wsBudget / stride)));
      grid                  = rt::roundGrid(grid);
      uint8_t* dWs          = ctx->dPtrWs.as<uint8_t>(stride * grid);
      rt::h2d(dIds + idsCursor, buckets[b].data(), sizeof(uint32_t) * buckets[b].size());
      AlignParams P;
      P.tasks          = dTasks;
      P.results        = dResults;
      P.cigar          = dCigar;
      P.task_ids       = dIds + idsCursor;
      P.n_tasks        = uint32_t(buckets[b].size());
      P.n_tasks_dev    = nullptr;
      P.counter        = dCounter + b;
      P.ptr_ws         = dWs;
      P.ptr_ws_stride  = stride;
      P.match          = scores->match;
      P.mismatch       = scores->mismatch;
      P.open           = scores->open;
      P.extend         = scores->extend;
      P.off_edge       = scores->off_edge;
      P.allow_edge_ins = scores->is_allow_edge_insertion ? 1 : 0;
      P.extra          = extra_score;
      rt::Event e0, e1;
      e0.record();
      launchAlignKind(kind, b, grid, P, pair);
      e1.record();
      rt::sync();  // dPtrWs may be re-sized by the next bucket
      if (std::getenv("MANTA_AMD_DEBUG"))
        std::fprintf(stderr, "manta_amd: align_kernel kind %d E=%d %zu tasks %.3f ms\n", kind, kESet[b], buckets[b].size(), rt::elapsedMs(e0, e1));
      idsCursor += buckets[b].size();
    }

    // ---- fetch + compact cigars into the caller's arena ----
    std::vector<AlignResultDev> hres(n_tasks);
    std::vector<uint32_t>       hcig(cigarDevWords + 1);
    rt::d2h(hres.data(), dResults, sizeof(AlignResultDev) * n_tasks);
    rt::d2h(hcig.data(), dCigar, sizeof(uint32_t) * cigarDevWords);
    uint64_t used = 0;
    for (uint32_t i = 0; i < n_tasks; ++i) {
      manta_align_result_t& r(results[i]);
      if (r.status != MANTA_OK) continue;
      const AlignResultDev& h(hres[i]);
      if (h.status != 0) {
        r.status = MANTA_E_DEVICE_FAULT;
        worst    = MANTA_E_DEVICE_FAULT;
        continue;
      }
      const uint64_t n = uint64_t(h.cigar1_len) + h.cigar2_len;
      if (used + n > cigar_arena_cap) return fail(ctx, MANTA_E_CAPACITY, "manta_align_batch: cigar arena too small");
      std::memcpy(cigar_arena + used, hcig.data() + dev[i].cigar_off, sizeof(uint32_t) * n);
      r.score            = h.score;
      r.is_jumped        = h.is_jumped;
      r.begin_pos1       = h.begin1;
      r.begin_pos2       = h.begin2;
      r.jump_insert_size = h.jump_insert_size;
      r.jump_range       = h.jump_range;
      r.cigar1_len       = h.cigar1_len;
      r.cigar2_len       = h.cigar2_len;
      r.cigar1_off       = used;
      r.cigar2_off       = used + h.cigar1_len;
      used += n;
    }
    if (cigar_arena_used) *cigar_arena_used = used;
    if (worst != MANTA_OK) return fail(ctx, worst, "manta_align_batch: one or more tasks failed; see per-task status");
    return MANTA_OK;
  } catch (const std::exception& e) {
    return fail(ctx, MANTA_E_HIP, e.what());
  }
}


int manta_assemble_batch(
    manta_ctx_t* ctx, const manta_asm_options_t* opt, uint32_t n_loci, const uint8_t* bases, const uint64_t* read_off,
    const uint32_t* locus_read_begin, manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs, uint64_t contigs_cap,
    uint8_t* seq_arena, uint64_t seq_arena_cap, uint64_t* seq_arena_used, uint64_t* bits_arena, uint64_t bits_arena_cap,
    uint64_t* bits_arena_used)
{
  if (!ctx) return MANTA_E_INVALID_ARG;
  if (!opt || (n_loci && (!bases || !read_off || !locus_read_begin || !loci || !contigs || !seq_arena || !bits_arena)))
    return fail(ctx, MANTA_E_INVALID_ARG, "manta_assemble_batch: null argument");
  if (seq_arena_used) *seq_arena_used = 0;
  if (bits_arena_used) *bits_arena_used = 0;
  if (n_loci == 0) return MANTA_OK;
  try {
    rt::setDevice(ctx->deviceId);  // the calling thread may have another device current (multi-GPU hosts)
    rt::ScopedStream onStream(ctx->stream);
    AsmStage st(ctx);
    int      rc = st.plan(*opt, n_loci, read_off, locus_read_begin);
    if (rc != MANTA_OK) {
      // refused as a whole (a locus outside the envelope sizes the shared workspace): every record says so, nothing is left unwritten
      for (uint32_t l = 0; l < n_loci; ++l) {
        std::memset(&loci[l], 0, sizeof(loci[l]));
        loci[l].status = rc;
      }
      return rc;
    }
    st.upload(bases, read_off, locus_read_begin);
    rt::Event e0, e1;
    e0.record();
    st.launch();
    e1.record();
    uint64_t capacityFailures = 0;
    rt::d2h(&capacityFailures, st.dCnt + 3, sizeof(uint64_t));  // (waits for the kernel)
    st.rerunCapacityFailures(capacityFailures);
    if (std::getenv("MANTA_AMD_DEBUG"))
      std::fprintf(stderr, "manta_amd: assemble_kernel %u loci %.3f ms\n", n_loci, rt::elapsedMs(e0, e1));
    return st.fetch(loci, contigs, contigs_cap, seq_arena, seq_arena_cap, seq_arena_used, bits_arena, bits_arena_cap, bits_arena_used);
  } catch (const std::exception& e) {
    return fail(ctx, MANTA_E_HIP, e.what());
  }
}

int manta_debug_repeat_words(
    manta_ctx_t* ctx, const manta_asm_options_t* opt, uint32_t n_reads, const uint8_t* bases, const uint64_t* read_off, char* out,
    uint64_t out_cap, uint32_t* n_words)
{
  if (!ctx) return MANTA_E_INVALID_ARG;
  if (!opt || !bases || !read_off || !out || !n_words) return fail(ctx, MANTA_E_INVALID_ARG, "manta_debug_repeat_words: null argument");
  *n_words = 0;
  try {
    rt::setDevice(ctx->deviceId);
    rt::ScopedStream onStream(ctx->stream);
    manta_asm_options_t o = *opt;
    o.max_word_length     = o.min_word_length;  // one word length: its graph is what the workspace holds afterwards
    const uint32_t begin[2] = {0, n_reads};
    AsmStage       st(ctx);
    int            rc = st.plan(o, 1, read_off, begin);
    if (rc != MANTA_OK) return rc;
    st.useFast = false;  // the general kernel: its workspace slab is read back below
    st.upload(bases, read_off, begin);
    rt::dzero(st.dWs, st.stride * uint64_t(st.grid));
    st.launch();
    AsmLocusOut lo;
    rt::d2h(&lo, st.dLoci, sizeof(lo));
    if (lo.status != ASM_OK) return fail(ctx, asmStatusToAbi(lo.status), "manta_debug_repeat_words: the locus did not assemble");
    const uint32_t nNodes = lo.reserved & 0x3ffffffu, slab = lo.reserved >> 26, k = lo.final_word_length;
    const AsmWsLayout L = asmWorkspaceLayout(st.capSlots, st.capNodes, st.capWords, st.capReads, st.maxContigLen, st.wMax, o.max_assembly_count);
    const uint8_t*    ws = st.dWs + st.stride * uint64_t(slab);
    std::vector<uint32_t> flag(nNodes), key(nNodes), codes(st.capWords + 2);
    rt::d2h(flag.data(), ws + L.node_flag, sizeof(uint32_t) * nNodes);
    rt::d2h(key.data(), ws + L.node_key, sizeof(uint32_t) * nNodes);
    rt::d2h(codes.data(), ws + L.codes, sizeof(uint32_t) * codes.size());
    std::vector<std::string> words;
    for (uint32_t nd = 0; nd < nNodes; ++nd) {
      if (!(flag[nd] & NF_REPEAT)) continue;
      std::string w(k, '?');
      for (uint32_t i = 0; i < k; ++i) {
        const uint32_t pb = key[nd] + i;
        w[i]              = "ACGT"[(codes[pb >> 4] >> (30 - 2 * (pb & 15))) & 3];
      }
      words.push_back(w);
    }
    std::sort(words.begin(), words.end());
    uint64_t used = 0;
    for (const std::string& w : words) {
      if (used + w.size() + 1 > out_cap) return fail(ctx, MANTA_E_CAPACITY, "manta_debug_repeat_words: output buffer too small");
      std::memcpy(out + used, w.data(), w.size());
      used += w.size();
      out[used++] = '\n';
    }
    if (used < out_cap) out[used] = 0;
    *n_words = uint32_t(words.size());
    return MANTA_OK;
  } catch (const std::exception& e) {
    return fail(ctx, MANTA_E_HIP, e.what());
  }
}

int manta_small_assemble_batch(
    manta_ctx_t* ctx, const manta_small_asm_options_t* opt, uint32_t n_loci, const uint8_t* bases, const uint64_t* read_off,
    const uint32_t* locus_read_begin, manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs, uint64_t contigs_cap,
    uint8_t* seq_arena, uint64_t seq_arena_cap, uint64_t* seq_arena_used, uint64_t* bits_arena, uint64_t bits_arena_cap,
    uint64_t* bits_arena_used)
{
  if (!ctx) return MANTA_E_INVALID_ARG;
  if (!opt || (n_loci && (!bases || !read_off || !locus_read_begin || !loci || !contigs || !seq_arena || !bits_arena)))
    return fail(ctx, MANTA_E_INVALID_ARG, "manta_small_assemble_batch: null argument");
  if (opt->word_step_size == 0 || opt->min_word_length < 2 || opt->max_assembly_iterations > 31)
    return fail(ctx, MANTA_E_INVALID_ARG, "manta_small_assemble_batch: wordStepSize 0, minWordLength < 2 or more than 31 iterations");
  if (seq_arena_used) *seq_arena_used = 0;
  if (bits_arena_used) *bits_arena_used = 0;
  if (n_loci == 0) return MANTA_OK;
  try {
    rt::setDevice(ctx->deviceId);  // the calling thread may have another device current (multi-GPU hosts)
    rt::ScopedStream onStream(ctx->stream);
    AsmStage st(ctx);
    // the slab and the record slots are those of the iterative assembler: one slot per iteration's contig + the isFiltered record
    manta_asm_options_t o{};
    o.min_word_length           = opt->min_word_length;
    o.max_word_length           = opt->max_word_length;
    o.word_step_size            = opt->word_step_size;
    o.min_contig_length         = opt->min_contig_length;
    o.min_coverage              = opt->min_coverage;
    o.min_conservative_coverage = opt->min_conservative_coverage;
    o.min_unused_reads          = 0;
    o.min_support_reads         = 0;
    o.max_assembly_count        = opt->max_assembly_iterations + 1;
    st.smallMode                = true;
    st.smallMinSeedReads        = opt->min_seed_reads;
    st.smallMaxIterations       = opt->max_assembly_iterations;
    int rc = st.plan(o, n_loci, read_off, locus_read_begin);
    if (rc != MANTA_OK) {
      for (uint32_t l = 0; l < n_loci; ++l) {
        std::memset(&loci[l], 0, sizeof(loci[l]));
        loci[l].status = rc;
      }
      return rc;
    }
    st.upload(bases, read_off, locus_read_begin);
    st.launch();
    rt::sync();
    return st.fetch(loci, contigs, contigs_cap, seq_arena, seq_arena_cap, seq_arena_used, bits_arena, bits_arena_cap, bits_arena_used);
  } catch (const std::exception& e) {
    return fail(ctx, MANTA_E_HIP, e.what());
  }
}

// ------------------------------------------------------------------------------------------------------
// fused small-SV pipeline
// ------------------------------------------------------------------------------------------------------
int manta_smallsv_create(
    manta_ctx_t* ctx, const manta_asm_options_t* opt, const manta_align_scores_t* scores, int32_t large_indel_score,
    manta_smallsv_t** out)
{
  if (!ctx) return MANTA_E_INVALID_ARG;
  if (!opt || !scores || !out) return fail(ctx, MANTA_E_INVALID_ARG, "manta_smallsv_create: null argument");
  try {
    rt::setDevice(ctx->deviceId);  // (the pipeline's streams and events belong to the context's device)
    manta_smallsv* b = new manta_smallsv(ctx);
    b->opt           = *opt;
    b->scores        = *scores;
    b->largeIndel    = large_indel_score;
    *out             = b;
  } catch (const std::exception& e) {
    return fail(ctx, MANTA_E_HIP, e.what());
  }
  return MANTA_OK;
}

void manta_smallsv_destroy(manta_smallsv_t* b)
{
  delete b;
}

int manta_smallsv_upload(
    manta_smallsv_t* b, uint32_t n_loci, const uint8_t* bases, const uint64_t* read_off, const uint32_t* locus_read_begin,
    const uint8_t* refs, const uint64_t* ref_off, const manta_ref_cuts_t* cuts)
{
  if (!b) return MANTA_E_INVALID_ARG;
  manta_ctx_t* ctx = b->ctx;
  if (n_loci == 0 || !bases || !read_off || !locus_read_begin || !refs || !ref_off || !cuts)
    return fail(ctx, MANTA_E_INVALID_ARG, "manta_smallsv_upload: null argument or empty batch");
  try {
    rt::setDevice(ctx->deviceId);  // the calling thread may have another device current (multi-GPU hosts)
    rt::ScopedStream onStream(b->main);
    b->uploaded = false;
    const double tP0 = nowMs();
    int rc      = b->asmStage.plan(b->opt, n_loci, read_off, locus_read_begin);
    if (rc != MANTA_OK) return rc;
    const double tP1 = nowMs();
    b->nLoci    = n_loci;
    b->refBytes = ref_off[n_loci];
    b->maxRef   = 0;
    for (uint32_t l = 0; l < n_loci; ++l) {
      if (ref_off[l + 1] < ref_off[l]) return fail(ctx, MANTA_E_INVALID_ARG, "manta_smallsv_upload: ref_off not monotone");
      b->maxRef = std::max<uint64_t>(b->maxRef, ref_off[l + 1] - ref_off[l]);
      if (cuts[l].leading_cut < 0 || cuts[l].trailing_cut < 0 || cuts[l].max_leading_cut < 0 || cuts[l].max_trailing_cut < 0)
        return fail(ctx, MANTA_E_INVALID_ARG, "manta_smallsv_upload: negative reference cut");
    }
    uint8_t*  dRefs   = b->dRefs.as<uint8_t>(b->refBytes + 16);
    uint64_t* dRefOff = b->dRefOff.as<uint64_t>(n_loci + 1);
    auto*     dCuts   = b->dCuts.as<SmallSvCuts>(n_loci);
    static_assert(sizeof(SmallSvCuts) == sizeof(manta_ref_cuts_t), "cuts layout");
    const bool streamed = b->streamUploads && !std::getenv("MANTA_AMD_NO_STREAM_UPLOAD");
    b->refsOnCopy       = streamed;
    if (streamed) {
      // the assembler does not read the reference windows: they travel on the copy stream BEHIND the read bases and the
      // schedule kernel waits for them (smallsvRunImpl); nothing but the small per-read arrays is waited for here
      b->asmStage.uploadStreamed(bases, read_off, locus_read_begin, b->copy);  // syncs the small copies, returns with the bases in flight
      rt::ScopedStream onCopy(b->copy);
      rt::h2d(dRefs, refs, b->refBytes);
      rt::h2d(dRefOff, ref_off, sizeof(uint64_t) * (n_loci + 1));
      rt::h2d(dCuts, cuts, sizeof(SmallSvCuts) * n_loci);
      b->refsReady.record();
    } else {
      rt::h2d(dRefs, refs, b->refBytes);
      rt::h2d(dRefOff, ref_off, sizeof(uint64_t) * (n_loci + 1));
      rt::h2d(dCuts, cuts, sizeof(SmallSvCuts) * n_loci);
      b->asmStage.upload(bases, read_off, locus_read_begin);
      rt::sync();
    }
    if (std::getenv("MANTA_AMD_DEBUG_TIMING"))
      std::fprintf(stderr, "manta_amd: smallsv_upload plan %.2f ms, upload (%s) %.2f ms\n", tP1 - tP0, b->asmStage.streaming ? "streamed" : "blocking", nowMs() - tP1);
    b->uploaded = true;
    return MANTA_OK;
  } catch (const std::exception& e) {
    return fail(ctx, MANTA_E_HIP, e.what());
  }
}

namespace {
int checkPiles(manta_ctx_t* ctx, const manta_packed_piles_t* pl, const char* who)
{
  if (!pl || !pl->codes || !pl->nmask || !pl->read_len || !pl->read_code_off || !pl->read_mask_off || !pl->locus_read_begin)
    return fail(ctx, MANTA_E_INVALID_ARG, std::string(who) + ": null pointer in the packed piles");
  return MANTA_OK;
}
}  // namespace

int manta_smallsv_upload_piles(
    manta_smallsv_t* b, uint32_t n_loci, const manta_packed_piles_t* piles, const uint8_t* refs, const uint64_t* ref_off,
    const manta_ref_cuts_t* cuts)
{
  if (!b) return MANTA_E_INVALID_ARG;
  manta_ctx_t* ctx = b->ctx;
  if (n_loci == 0 || !refs || !ref_off || !cuts) return fail(ctx, MANTA_E_INVALID_ARG, "manta_smallsv_upload_piles: null argument or empty batch");
  int rc = checkPiles(ctx, piles, "manta_smallsv_upload_piles");
  if (rc != MANTA_OK) return rc;
  try {
    rt::setDevice(ctx->deviceId);
    rt::ScopedStream onStream(b->main);
    b->uploaded = false;
    b->refsOnCopy = false;
    rc          = b->asmStage.plan(b->opt, n_loci, nullptr, piles->locus_read_begin, piles->read_len);
    if (rc != MANTA_OK) return rc;
    b->nLoci    = n_loci;
    b->refBytes = ref_off[n_loci];
    b->maxRef   = 0;
    for (uint32_t l = 0; l < n_loci; ++l) {
      if (ref_off[l + 1] < ref_off[l]) return fail(ctx, MANTA_E_INVALID_ARG, "manta_smallsv_upload_piles: ref_off not monotone");
      b->maxRef = std::max<uint64_t>(b->maxRef, ref_off[l + 1] - ref_off[l]);
      if (cuts[l].leading_cut < 0 || cuts[l].trailing_cut < 0 || cuts[l].max_leading_cut < 0 || cuts[l].max_trailing_cut < 0)
        return fail(ctx, MANTA_E_INVALID_ARG, "manta_smallsv_upload_piles: negative reference cut");
    }
    uint8_t*  dRefs   = b->dRefs.as<uint8_t>(b->refBytes + 16);
    uint64_t* dRefOff = b->dRefOff.as<uint64_t>(n_loci + 1);
    auto*     dCuts   = b->dCuts.as<SmallSvCuts>(n_loci);
    b->refsOnCopy     = b->streamUploads && !std::getenv("MANTA_AMD_NO_STREAM_UPLOAD");
    if (b->refsOnCopy) {  // whole-batch call: piles chunk by chunk behind the running assembler, references behind them (manta_smallsv_upload)
      b->asmStage.uploadPilesStreamed(*piles, b->copy);
      rt::ScopedStream onCopy(b->copy);
      rt::h2d(dRefs, refs, b->refBytes);
      rt::h2d(dRefOff, ref_off, sizeof(uint64_t) * (n_loci + 1));
      rt::h2d(dCuts, cuts, sizeof(SmallSvCuts) * n_loci);
      b->refsReady.record();
    } else {
      b->asmStage.uploadPiles(*piles);
      rt::h2d(dRefs, refs, b->refBytes);
      rt::h2d(dRefOff, ref_off, sizeof(uint64_t) * (n_loci + 1));
      rt::h2d(dCuts, cuts, sizeof(SmallSvCuts) * n_loci);
      rt::sync();
    }
    b->uploaded = true;
    return MANTA_OK;
  } catch (const std::exception& e) {
    return fail(ctx, MANTA_E_HIP, e.what());
  }
}

}  // extern "C"

/// Stage gates of a whole-batch call with several workers: at most one block assembles and at most one block
/// aligns at any time, so that block B's (memory-bound) assembler overlaps block A's (VALU-bound) aligners and transfers
/// instead of two persistent assemblers fighting for the same wave slots.
/// assembler waves per CU while another block's aligners share the device: 3 of the 4 wave slots per SIMD (128 VGPRs each)
static const int kPipelinedAsmWavesPerCu = std::getenv("MANTA_AMD_PIPELINED_ASM_WAVES") ? std::atoi(std::getenv("MANTA_AMD_PIPELINED_ASM_WAVES")) : 12;
struct StageGates {
  std::mutex asmMu, alignMu;
};
static std::mutex g_streamedAsmMuOfDevice[16];  // see smallsvRunImpl; one per device (id modulo 16)
static std::mutex& streamedAsmMu(const manta_ctx_t* ctx) { return g_streamedAsmMuOfDevice[unsigned(ctx->deviceId) % 16u]; }
struct GateLock {
  std::unique_lock<std::mutex> l;
  GateLock(StageGates* g, std::mutex StageGates::*m) { if (g) l = std::unique_lock<std::mutex>(g->*m); }
  void release() { if (l.owns_lock()) l.unlock(); }
};

/// failure exit of a pipeline run: let queued DMA reads of the caller's buffers (streamed upload) finish before the call returns
template <typename Pipe>
void drainCopyStream(Pipe* b) noexcept
{
  try {
    rt::ScopedStream onCopy(b->copy);
    rt::sync();
  } catch (...) {
  }
}

namespace {
template <typename Pipe>
void pipeStageEnqueue(Pipe* b);
}

int smallsvRunImpl(manta_smallsv_t* b, StageGates* gates)
{
  if (!b) return MANTA_E_INVALID_ARG;
  manta_ctx_t* ctx = b->ctx;
  if (!b->uploaded) return fail(ctx, MANTA_E_INVALID_ARG, "manta_smallsv_run: nothing uploaded");
  try {
    rt::setDevice(ctx->deviceId);  // the calling thread may have another device current (multi-GPU hosts)
    rt::ScopedStream onStream(b->main);
    const uint32_t nLoci   = b->nLoci;
    const uint32_t maxAsm  = b->opt.max_assembly_count;
    const uint64_t nSlots  = uint64_t(nLoci) * maxAsm;
    AsmStage&      as(b->asmStage);
    // ---- per-run device state ----
    AlignTaskDev*    dTasks   = b->dTasks.as<AlignTaskDev>(nSlots);
    SmallSvTaskInfo* dInfo    = b->dInfo.as<SmallSvTaskInfo>(nSlots);
    AlignResultDev*  dResults = b->dResults.as<AlignResultDev>(nSlots);
    uint32_t*        dBuckets = b->dBucketIds.as<uint32_t>(nSlots * kNumESet);
    uint32_t*        dSmall   = b->dSmall.as<uint32_t>(64);  // [0..11] counts, [16..27] maxref, [32] sched counter, [34..35] cigar_used, [40..51] align counters
    const uint32_t   tableCap = nextPow2(2ull * as.maxContigLen);
    const int        schedWaves = std::getenv("MANTA_AMD_SCHED_WAVES_PER_CU") ? std::max(1, std::atoi(std::getenv("MANTA_AMD_SCHED_WAVES_PER_CU"))) : 12;
    const uint32_t   schedChunk = std::getenv("MANTA_AMD_SCHED_CHUNK") ? uint32_t(std::max(1, std::atoi(std::getenv("MANTA_AMD_SCHED_CHUNK")))) : 2u;
    const int        schedGrid = rt::roundGrid(int(std::min<uint64_t>((nLoci + schedChunk - 1) / schedChunk, uint64_t(std::max(1, ctx->cuCount * schedWaves)))));
    uint32_t*        dTable   = b->dTable.as<uint32_t>(uint64_t(tableCap) * schedGrid);
    rt::dzero(dSmall, sizeof(uint32_t) * 64);
    rt::dzero(dResults, sizeof(AlignResultDev) * nSlots);

    const bool dbg = std::getenv("MANTA_AMD_DEBUG") != nullptr;
    auto       stage = [&](const char* what) {
      if (!dbg) return;
      rt::sync();
      std::fprintf(stderr, "manta_amd: smallsv_run %s\n", what);
      std::fflush(stderr);
    };
    stage("start");
    uint64_t asmCnt[4];  // [1] text bytes, [3] capacity failures
    {
      GateLock only(gates, &StageGates::asmMu);
      // A streamed-upload assembler polls for chunks whose copies may need a free workgroup slot (launch() leaves some);
      // a second persistent assembler would take exactly those slots and both would spin until the kernels' time-out.
      // Per device, process-wide: never two streamed assemblers on a device at once.
      std::unique_lock<std::mutex> streamedOnly(streamedAsmMu(ctx), std::defer_lock);
      if (as.streaming) streamedOnly.lock();
      as.stageQueued = false;  // (a run that failed behind its queued staging must not leave the flag to the next one)
      b->evStart.record();
      as.launch();
      b->evAsm.record();
      // loci that did not fit the typical-case workspace run again here (rare; the counter rides on a wait that is there anyway
      // in the whole-batch calls, the staged API pays one small read)
      rt::d2h(asmCnt, as.dCnt, sizeof(asmCnt));  // (also: the gate opens when the assembler has left the device)
      if (asmCnt[3]) {
        as.rerunCapacityFailures(asmCnt[3]);
        rt::d2h(asmCnt, as.dCnt, sizeof(asmCnt));  // the text arena grew
      }
    }
    stage("assembled");
    // CIGAR scratch, sized from what the assembler produced (as in spanningRunImpl): a task takes 4 * contig length + 16 words
    // (smallsv_schedule_kernel), every contig is aligned once, and the text arena counter bounds the summed contig lengths.  A
    // worst case per slot would be tens of GB at 65536-locus blocks and overflow the 32-bit offsets of the task records.
    const uint64_t cigarCap = 8ull * std::min<uint64_t>(asmCnt[1], as.devSeqCap) + 64;  // (8 words per text byte: smallsv_schedule_kernel)
    if (cigarCap + 16 > 0xffffffffull)
      return fail(ctx, MANTA_E_UNSUPPORTED, "manta_smallsv_run: alignment scratch of this block exceeds 2^32 words; use smaller blocks "
                                         "(manta_smallsv_batch splits a batch into blocks, manta_batch_plan_t::block_loci)");
    uint32_t* dCigar = b->dCigar.as<uint32_t>(cigarCap + 16);
    GateLock alignOnly(gates, &StageGates::alignMu);
    if (b->refsOnCopy) rt::curStreamWaits(b->refsReady);  // reference windows of a streamed upload (manta_smallsv_upload)

    ScheduleParams S;
    S.loci               = as.dLoci;
    S.contigs            = as.dCont;
    S.seq_arena          = as.dSeq;
    S.n_loci             = nLoci;
    S.max_assembly_count = maxAsm;
    S.refs               = static_cast<const uint8_t*>(b->dRefs.p);
    S.ref_off            = static_cast<const uint64_t*>(b->dRefOff.p);
    S.cuts               = static_cast<const SmallSvCuts*>(b->dCuts.p);
    S.tasks              = dTasks;
    S.info               = dInfo;
    S.bucket_ids         = dBuckets;
    S.bucket_count       = dSmall;
    S.bucket_maxref      = dSmall + 16;
    S.cigar_used         = reinterpret_cast<unsigned long long*>(dSmall + 34);
    S.cigar_cap          = cigarCap;
    S.counter            = dSmall + 32;
    S.table_ws           = dTable;
    S.table_cap          = tableCap;
    S.n_e                = kNumESet;
    S.chunk              = schedChunk;
    for (int i = 0; i < kNumESet; ++i) S.e_set[i] = uint32_t(kESet[i]);
    rt::launch(smallsv_schedule_kernel, schedGrid, SCHED_LDS_BYTES, S);
    // the pair-eligible buckets by descending reference length (bucket_sort_kernel): their align kernels read the sorted lists
    uint32_t  pairMask       = 0;
    uint32_t* dBucketsSorted = nullptr;
    for (int k = 0; k < kNumESet; ++k)
      if (alignUsesPairs(MANTA_ALIGNER_LARGE_INDEL, k, b->scores.match, b->scores.mismatch, b->scores.open, b->scores.extend, b->scores.off_edge,
                         b->largeIndel, b->scores.is_allow_edge_insertion ? 1 : 0, b->maxRef))
        pairMask |= 1u << k;
    if (pairMask) {
      dBucketsSorted = b->dBucketIds2.as<uint32_t>(nSlots * kNumESet);
      BucketSortParams BS;
      BS.tasks        = dTasks;
      BS.info         = dInfo;
      BS.ids_out      = dBucketsSorted;
      BS.bucket_count = dSmall;
      BS.total        = uint32_t(nSlots);
      BS.mask         = pairMask;
      rt::launchWG(bucket_sort_kernel, kNumESet, int(BS_WAVES), BS_LDS_BYTES, BS);
    }
    b->evSched.record();
    stage("scheduled");

    // Bucket sizes decide the alignment launches.  First run of a pipeline: one tiny D2H.  Later runs of a whole-batch call
    // do not wait for it: every bucket is launched with the task count read on the device (AlignParams::n_tasks_dev), its
    // grid sized from the previous run's counts (consecutive blocks look alike; the waves of a bucket pull tasks from a queue,
    // so a grid that is off costs time, never results) and its slabs from what the host knows (longest reference window,
    // longest possible contig).  The counts come back with the results.
    uint32_t hSmall[40];
    const int    maxWaves = std::max(1, ctx->cuCount * alignWavesPerCu());
    const size_t wsBudget = workspaceBudget(size_t(48) << 30);
    bool         fromHistory = b->bucketHistory && b->stageBehindRun && !std::getenv("MANTA_AMD_SYNC_BUCKETS");  // (no host wait)
    struct Launch {
      int      k, grid;
      uint64_t stride, slabOff;
      bool     pair;
    };
    std::vector<Launch> launches;
    uint64_t            slabBytes = 0;
    auto pairOf = [&](int k) {
      return alignUsesPairs(MANTA_ALIGNER_LARGE_INDEL, k, b->scores.match, b->scores.mismatch, b->scores.open, b->scores.extend, b->scores.off_edge,
                            b->largeIndel, b->scores.is_allow_edge_insertion ? 1 : 0, b->maxRef);
    };
    const bool batchCall = b->stageBehindRun && !std::getenv("MANTA_AMD_SYNC_BUCKETS");
    bool       haveCounts = false;
    if (batchCall && !fromHistory) {
      // first run of this pipeline: read the counts once, then launch exactly as the later runs will (same kernels, same slab
      // layout), so that the second run does not pay for a changed allocation or for kernels seen for the first time
      rt::d2h(hSmall, dSmall, sizeof(hSmall));
      std::memcpy(b->lastSmall, hSmall, sizeof(b->lastSmall));
      haveCounts  = true;
      fromHistory = true;
    }
    if (fromHistory) {
      for (int k = kNumESet - 1; k >= 0; --k) {
        const bool     pair   = pairOf(k);
        // (a bucket that was empty last time gets a token grid: tasks that do turn up are still aligned, just by few waves)
        const uint64_t tasks  = b->lastSmall[k] ? uint64_t(b->lastSmall[k]) + b->lastSmall[k] / 4 + 64 : 4;
        const uint64_t hint   = pair ? (tasks + 1) / 2 : tasks;  // work items: alignments, or pairs of them
        const uint64_t qBound = (kESet[k] == 32) ? std::max<uint64_t>(as.maxContigLen, 64ull * 32) : 64ull * uint64_t(kESet[k]);
        const uint64_t refLen = alignSlabRefLen(MANTA_ALIGNER_LARGE_INDEL, kESet[k], qBound, b->maxRef);
        const uint64_t stride = ((pair ? 2 : 1) * alignPtrSlabBytes(MANTA_ALIGNER_LARGE_INDEL, kESet[k], refLen) + 255) & ~uint64_t(255);
        const int      grid   = rt::roundGrid(int(std::min<uint64_t>(hint, uint64_t(maxWaves))));
        launches.push_back(Launch{k, grid, stride, slabBytes, pair});
        slabBytes += stride * uint64_t(grid);
      }
      if (slabBytes > wsBudget / 3) {  // too generous for this device right now: size from the real counts
        fromHistory = false;
        launches.clear();
        slabBytes = 0;
      }
    }
    if (!fromHistory) {
      if (!haveCounts) rt::d2h(hSmall, dSmall, sizeof(hSmall));
      for (int k = kNumESet - 1; k >= 0; --k) {  // widest (longest-running) buckets first
        const uint32_t cnt = hSmall[k];
        if (cnt == 0) continue;
        const bool     pair   = pairOf(k);
        const uint64_t stride = ((pair ? 2 : 1) * alignPtrSlabBytes(MANTA_ALIGNER_LARGE_INDEL, kESet[k], hSmall[16 + k]) + 255) & ~uint64_t(255);
        int            grid   = int(std::min<size_t>(pair ? (cnt + 1) / 2 : cnt, size_t(maxWaves)));
        grid                  = int(std::max<size_t>(1, std::min<size_t>(size_t(grid), (wsBudget / 3) / stride)));
        grid                  = rt::roundGrid(grid);
        launches.push_back(Launch{k, grid, stride, slabBytes, pair});
        slabBytes += stride * uint64_t(grid);
      }
    }
    b->stats.n_align_launches = 0;
    b->stats.n_alignments     = 0;
    b->stats.ptr_matrix_bytes = 0;
    // The E buckets are independent launches: they run on side streams so that the tail of one overlaps the others
    // (a launch's last alignments leave most of the device idle otherwise).  Each bucket gets its own slab region.
    {
      // The packed buckets share ONE launch and one queue (align_pair_multi_kernel: buckets with the fewest tasks first, so the
      // few long alignments of narrow buckets start at once and the bulk of the widest fills in behind them); every other
      // bucket is its own launch on a side stream.
      static const bool   noMerge = std::getenv("MANTA_AMD_NO_ALIGN_MERGE") != nullptr;  // A/B knob: one launch per packed bucket
      std::vector<Launch> merged;
      if (!noMerge) {
        std::vector<Launch> rest;
        for (const Launch& l : launches) (l.pair ? merged : rest).push_back(l);
        if (merged.size() < 2 || merged.size() > 6) {
          merged.clear();
        } else {
          launches.swap(rest);
          slabBytes = 0;
          for (Launch& l : launches) {
            l.slabOff = slabBytes;
            slabBytes += l.stride * uint64_t(l.grid);
          }
        }
      }
      uint64_t mergedStride = 0, mergedWaves = 0, mergedOff = slabBytes;
      const uint32_t* counts = fromHistory ? b->lastSmall : hSmall;
      if (!merged.empty()) {
        std::stable_sort(merged.begin(), merged.end(), [&](const Launch& x, const Launch& y) { return counts[x.k] < counts[y.k]; });
        for (const Launch& l : merged) {
          mergedStride = std::max(mergedStride, l.stride);
          mergedWaves += uint64_t(l.grid);
        }
        // (clamp first, then a whole number of workgroups -- rt::launch divides the grid by the workgroup's wave count -- and at least one)
        mergedWaves = std::min<uint64_t>(std::min<uint64_t>(mergedWaves, uint64_t(maxWaves)), (wsBudget / 3) / mergedStride);
        mergedWaves = std::max<uint64_t>(WV_WAVES_PER_WG, (mergedWaves / WV_WAVES_PER_WG) * WV_WAVES_PER_WG);
        slabBytes += mergedStride * mergedWaves;
      }
      uint8_t* dWsAll = b->dPtrWs.as<uint8_t>(slabBytes + 256);
      // (without the host read in between nothing else orders the side streams behind the schedule kernel)
      for (int i = 0; i < 3; ++i) rt::streamWaits(b->side[i], b->evSched);
      auto baseParams = [&]() {
        AlignParams P;
        P.tasks          = dTasks;
        P.results        = dResults;
        P.cigar          = dCigar;
        P.task_ids       = nullptr;
        P.n_tasks        = 0;
        P.n_tasks_dev    = nullptr;
        P.counter        = nullptr;
        P.ptr_ws         = nullptr;
        P.ptr_ws_stride  = 0;
        P.match          = b->scores.match;
        P.mismatch       = b->scores.mismatch;
        P.open           = b->scores.open;
        P.extend         = b->scores.extend;
        P.off_edge       = b->scores.off_edge;
        P.allow_edge_ins = b->scores.is_allow_edge_insertion ? 1 : 0;
        P.extra          = b->largeIndel;
        return P;
      };
      for (size_t i = 0; i < launches.size(); ++i) {
        const Launch& l(launches[i]);
        AlignParams   P  = baseParams();
        P.task_ids       = (l.pair ? dBucketsSorted : dBuckets) + uint64_t(l.k) * nSlots;
        P.n_tasks        = fromHistory ? uint32_t(nSlots) : hSmall[l.k];
        P.n_tasks_dev    = fromHistory ? dSmall + l.k : nullptr;
        P.counter        = dSmall + 40 + l.k;
        P.ptr_ws         = dWsAll + l.slabOff;
        P.ptr_ws_stride  = l.stride;
        {
          rt::ScopedStream onSide(b->side[i % 3]);  // (restores the pipeline's stream: nothing of this call runs on the null stream)
          launchAlignKind(MANTA_ALIGNER_LARGE_INDEL, l.k, l.grid, P, l.pair);
        }
        if (!fromHistory) {
          b->stats.n_align_launches++;
          b->stats.n_alignments += hSmall[l.k];
        }
      }
      // (the packed launch goes last: the buckets above are small or token grids that finish at once on an empty device; queued
      // behind a grid that fills every SIMD's registers they would sit in the dispatcher until its first waves retire)
      if (!merged.empty()) {
        PairMultiParams M;
        M.A               = baseParams();
        M.A.counter       = dSmall + 40 + merged[0].k;  // (the queue head of the first bucket serves the shared queue)
        M.A.ptr_ws        = dWsAll + mergedOff;
        M.A.ptr_ws_stride = mergedStride;
        M.bucket_ids      = dBucketsSorted;
        M.counts          = dSmall;
        M.n_slots         = uint32_t(nSlots);
        M.n_order         = uint32_t(merged.size());
        M.prio_work[0] = 2560u;  // (defaults from the config-2 measurement: a typical pair is E x (G + 64) ~ 5 x 384 = 1 920)
        M.prio_work[1] = 0u;
        M.prio_work[2] = 4096u;
        if (const char* e = std::getenv("MANTA_AMD_ALIGN_PRIO")) {  // experiments: "w1,w2,w3" (0 = level unused; "0,0,0" = no priorities)
          unsigned v[3] = {0, 0, 0};
          std::sscanf(e, "%u,%u,%u", &v[0], &v[1], &v[2]);
          for (int i = 0; i < 3; ++i) M.prio_work[i] = v[i];
        }
        for (size_t i = 0; i < 8; ++i) {
          M.order[i] = uint8_t(i < merged.size() ? merged[i].k : 0);
          M.e_of[i]  = uint8_t(i < merged.size() ? kESet[merged[i].k] : 0);
        }
        if (std::getenv("MANTA_AMD_DEBUG"))
          std::fprintf(stderr, "manta_amd: align_pair_multi_kernel: %llu waves over %zu packed buckets, slab stride %llu\n",
                       (unsigned long long)mergedWaves, merged.size(), (unsigned long long)mergedStride);
        rt::launch(align_pair_multi_kernel, int(mergedWaves), 0, M);  // on the pipeline's own stream
        if (!fromHistory) {
          b->stats.n_align_launches++;
          for (const Launch& l : merged) b->stats.n_alignments += hSmall[l.k];
        }
      }
      // the null stream (events, later copies) continues after every side stream has drained
      for (size_t i = 0; i < std::min<size_t>(launches.size(), 3); ++i) {
        b->sideDone[i].recordOn(b->side[i]);
        rt::curStreamWaits(b->sideDone[i]);
      }
    }
    launchPack(b, dTasks, nullptr, dResults, nullptr, dInfo, nullptr, dCigar, cigarCap);
    b->evAlign.record();
    uint32_t* hSmallPin = b->pSmall.as<uint32_t>(40);
    rt::d2hAsync(hSmallPin, dSmall, sizeof(uint32_t) * 40);
    if (b->stageBehindRun) pipeStageEnqueue(b);  // the results start for the host behind the last kernel: no wake-up in between
    rt::sync();
    std::memcpy(b->lastSmall, hSmallPin, sizeof(b->lastSmall));
    b->bucketHistory = true;
    if (fromHistory)
      for (int k = 0; k < kNumESet; ++k)
        if (b->lastSmall[k]) {
          b->stats.n_align_launches++;
          b->stats.n_alignments += b->lastSmall[k];
        }
    alignOnly.release();
    if (as.streaming) {  // all chunks were consumed by the kernel, so this returns at once; it closes the stream's error state
      rt::ScopedStream onCopy(b->copy);
      rt::sync();
    }
    stage("aligned");
    b->stats.assemble_ms = rt::elapsedMs(b->evStart, b->evAsm);
    b->stats.schedule_ms = rt::elapsedMs(b->evAsm, b->evSched);
    b->stats.align_ms    = rt::elapsedMs(b->evSched, b->evAlign);
    b->stats.total_ms    = rt::elapsedMs(b->evStart, b->evAlign);
    b->ran               = true;
    return MANTA_OK;
  } catch (const std::exception& e) {
    drainCopyStream(b);
    return fail(ctx, MANTA_E_HIP, e.what());
  }
}

extern "C" {

int manta_smallsv_run(manta_smallsv_t* b) { return smallsvRunImpl(b, nullptr); }

int manta_smallsv_stats(const manta_smallsv_t* b, manta_smallsv_stats_t* stats)
{
  if (!b || !stats) return MANTA_E_INVALID_ARG;
  *stats = b->stats;
  return MANTA_OK;
}

int manta_smallsv_output_sizes(const manta_smallsv_t* b, uint64_t* contigs, uint64_t* seq_bytes, uint64_t* bits_words, uint64_t* cigar_words)
{
  if (!b || !contigs || !seq_bytes || !bits_words || !cigar_words) return MANTA_E_INVALID_ARG;
  if (!b->ran) return fail(b->ctx, MANTA_E_INVALID_ARG, "manta_smallsv_output_sizes: run first");
  try {
    rt::setDevice(b->ctx->deviceId);  // the calling thread may have another device current (multi-GPU hosts)
    rt::ScopedStream onStream(const_cast<manta_smallsv_t*>(b)->main);
    b->asmStage.outputSizes(*contigs, *seq_bytes, *bits_words);
    uint32_t hSmall[40];
    rt::d2h(hSmall, b->dSmall.p, sizeof(hSmall));
    uint64_t cig = 0;
    std::memcpy(&cig, hSmall + 34, sizeof(uint64_t));
    *cigar_words = cig + 64;
    return MANTA_OK;
  } catch (const std::exception& e) {
    return fail(b->ctx, MANTA_E_HIP, e.what());
  }
}

}  // extern "C"

namespace {

/// device -> pinned staging of one finished pipeline run (both pipelines): counters + locus records + first-contig
/// index in the first round trip, then exactly the used part of every arena
template <typename Pipe>
void pipeStageEnqueue(Pipe* b)
{
  const uint32_t nLoci = b->nLoci;
  b->hPackCnt          = b->pPackCnt.template as<uint32_t>(4);
  b->hFirst            = b->pFirst.template as<uint32_t>(nLoci);
  b->asmStage.stageEnqueue(
      [&] {
        rt::d2hAsync(b->hPackCnt, b->dPackCnt.p, 16);
        rt::d2hAsync(b->hFirst, b->dFirst.p, sizeof(uint32_t) * nLoci);
        // packed contig records and CIGARs: as many as the previous run had, a quarter on top
        b->packedCopied = b->packLast[0] + b->packLast[0] / 4 + (b->packLast[0] ? 64 : 0);
        b->cigCopied    = b->packLast[1] + b->packLast[1] / 4 + (b->packLast[1] ? 1024 : 0);
        b->packedCopied = std::min<uint64_t>(b->packedCopied, b->dPacked.cap / sizeof(PackedContigOut));
        b->cigCopied    = std::min<uint64_t>(b->cigCopied, b->dCigPacked.cap / sizeof(uint32_t));
        b->hPacked      = b->pPacked.template as<PackedContigOut>(b->packedCopied + 1);
        b->hCig         = b->pCig.template as<uint32_t>(b->cigCopied + 1);
        rt::d2hAsync(b->hPacked, b->dPacked.p, sizeof(PackedContigOut) * b->packedCopied);
        rt::d2hAsync(b->hCig, b->dCigPacked.p, sizeof(uint32_t) * b->cigCopied);
      },
      false);
}
template <typename Pipe>
void pipeStageFinish(Pipe* b)
{
  b->asmStage.stageFinish([&](bool& queued) {
    const uint64_t nP = b->hPackCnt[0], nG = b->hPackCnt[1];
    b->packLast[0] = nP;
    b->packLast[1] = nG;
    if (nP > b->packedCopied) {
      b->hPacked = b->pPacked.template as<PackedContigOut>(nP + 1, true);
      rt::d2hAsync(b->hPacked + b->packedCopied, static_cast<const PackedContigOut*>(b->dPacked.p) + b->packedCopied,
                   sizeof(PackedContigOut) * (nP - b->packedCopied));
      queued = true;
    }
    if (nG > b->cigCopied) {
      b->hCig = b->pCig.template as<uint32_t>(nG + 1, true);
      rt::d2hAsync(b->hCig + b->cigCopied, static_cast<const uint32_t*>(b->dCigPacked.p) + b->cigCopied, sizeof(uint32_t) * (nG - b->cigCopied));
      queued = true;
    }
  });
  b->staged = true;
}
/// device -> pinned staging of one finished pipeline run (both pipelines).  A run that queued the staging behind its last
/// kernel (stageBehindRun) has done the first half already.
template <typename Pipe>
void pipeStage(Pipe* b)
{
  if (!b->asmStage.stageQueued) {
    pipeStageEnqueue(b);
    rt::sync();
  }
  pipeStageFinish(b);
}

template <typename Pipe>
struct PackedContigs {
  const Pipe* b;
  const AsmContigOut& operator()(uint32_t l, uint32_t c) const { return b->hPacked[b->hFirst[l] + c].contig; }
};

/// bytes a pipeStage moved over PCIe (for the batch statistics)
template <typename Pipe>
uint64_t pipeStagedBytes(const Pipe* b)
{
  return b->asmStage.seqUsedDev + 8 * b->asmStage.bitsUsedDev + (sizeof(AsmLocusOut) + 4) * uint64_t(b->nLoci) +
         sizeof(PackedContigOut) * uint64_t(b->hPackCnt[0]) + 4ull * b->hPackCnt[1] + 128 + 16;
}

/// staging -> caller records/arenas.  `loci` is this block's slice; `contigs` / `alignments` are the caller's whole arrays
/// and this block writes [contigBase, contigBase + contigs_cap); the three arenas are this block's regions, offsets in the
/// records are made relative to the caller's arena starts by adding the *Base values.
int smallsvCompact(
    manta_smallsv* b, manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs, manta_smallsv_alignment_t* alignments,
    uint64_t contigBase, uint64_t contigs_cap, uint8_t* seq_arena, uint64_t seq_arena_cap, uint64_t seqBase, uint64_t* seq_arena_used,
    uint64_t* bits_arena, uint64_t bits_arena_cap, uint64_t bitsBase, uint64_t* bits_arena_used, uint32_t* cigar_arena,
    uint64_t cigar_arena_cap, uint64_t cigarBase, uint64_t* cigar_arena_used, uint32_t lBegin = 0, uint32_t lEnd = ~0u,
    uint64_t* cellsOut = nullptr, uint64_t* ptrBytesOut = nullptr)
{
  // [lBegin, lEnd): the loci of this call (a whole-batch call compacts a block in a few ranges, one host thread each); every
  // output pointer / base is that of the range's first record
  manta_ctx_t* ctx = b->ctx;
  lEnd             = std::min(lEnd, b->nLoci);
  int rc = b->asmStage.compact(PackedContigs<manta_smallsv>{b}, loci, contigs + contigBase, contigs_cap, seq_arena, seq_arena_cap, seq_arena_used,
                               bits_arena, bits_arena_cap, bits_arena_used, contigBase, seqBase, bitsBase, lBegin, lEnd);
  if (rc != MANTA_OK && !perItemCode(rc)) return rc;
  uint64_t       used = 0, cells = 0, ptrBytes = 0;
  int            worst = rc;
  for (uint32_t l = lBegin; l < lEnd; ++l) {
    if (loci[l].status != MANTA_OK) continue;
    for (uint32_t c = 0; c < loci[l].n_contigs; ++c) {
      manta_smallsv_alignment_t& a(alignments[loci[l].first_contig + c]);
      std::memset(&a, 0, sizeof(a));
      const PackedContigOut& h(b->hPacked[b->hFirst[l] + c]);
      a.adjusted_leading_cut  = h.a;
      a.adjusted_trailing_cut = h.b;
      if (h.info_status != 0 || h.bucket < 0 || h.res_status != 0) {
        a.align.status = (h.info_status == 5) ? MANTA_E_DEVICE_FAULT : MANTA_E_UNSUPPORTED;
        worst          = a.align.status;
        if (std::getenv("MANTA_AMD_DEBUG") || std::getenv("MANTA_AMD_DEBUG_STATUS"))
          std::fprintf(stderr, "manta_amd: locus %u contig %u schedule status %d bucket %d align status %d\n", l, c, h.info_status, h.bucket, h.res_status);
        continue;
      }
      const uint64_t n = h.cigar1_len;
      if (used + n > cigar_arena_cap) return fail(ctx, MANTA_E_CAPACITY, "manta_smallsv_download: cigar arena too small");
      std::memcpy(cigar_arena + used, b->hCig + h.cigar_off, sizeof(uint32_t) * n);
      a.align.score      = h.score;
      a.align.is_jumped  = h.is_jumped;
      a.align.begin_pos1 = h.begin1 + h.a;  // SVCandidateAssemblyRefiner.cpp:2039
      a.align.cigar1_len = h.cigar1_len;
      a.align.cigar1_off = cigarBase + used;
      a.align.cigar2_off = cigarBase + used + n;
      used += n;
      cells += uint64_t(h.query_len) * h.ref_len;
      ptrBytes += 2ull * (uint64_t(h.query_len) + 1) * (uint64_t(h.ref_len) + 1);
    }
  }
  if (cellsOut) {  // ranged call: the caller adds the ranges up
    *cellsOut    = cells;
    *ptrBytesOut = ptrBytes;
  } else {
    b->stats.dp_cells         = cells;
    b->stats.ptr_matrix_bytes = ptrBytes;
  }
  if (cigar_arena_used) *cigar_arena_used = used;
  if (worst != MANTA_OK) return fail(ctx, worst, "manta_smallsv_download: one or more loci/contigs failed; see per-item status");
  return MANTA_OK;
}

/// CIGAR words smallsvCompact writes for the loci [lBegin, lEnd)
uint64_t smallsvCigarWords(const manta_smallsv* b, uint32_t lBegin, uint32_t lEnd)
{
  uint64_t n = 0;
  for (uint32_t l = lBegin; l < lEnd; ++l) {
    const AsmLocusOut& h(b->asmStage.hLoci[l]);
    if (h.status != ASM_OK) continue;
    for (uint32_t c = 0; c < h.n_contigs; ++c) {
      const PackedContigOut& pc(b->hPacked[b->hFirst[l] + c]);
      if (pc.info_status == 0 && pc.bucket >= 0 && pc.res_status == 0) n += pc.cigar1_len;
    }
  }
  return n;
}

}  // namespace

extern "C" {

int manta_smallsv_download(
    manta_smallsv_t* b, manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs, manta_smallsv_alignment_t* alignments,
    uint64_t contigs_cap, uint8_t* seq_arena, uint64_t seq_arena_cap, uint64_t* seq_arena_used, uint64_t* bits_arena,
    uint64_t bits_arena_cap, uint64_t* bits_arena_used, uint32_t* cigar_arena, uint64_t cigar_arena_cap,
    uint64_t* cigar_arena_used)
{
  if (!b) return MANTA_E_INVALID_ARG;
  manta_ctx_t* ctx = b->ctx;
  if (!b->ran) return fail(ctx, MANTA_E_INVALID_ARG, "manta_smallsv_download: run first");
  if (!loci || !contigs || !alignments || !seq_arena || !bits_arena || !cigar_arena)
    return fail(ctx, MANTA_E_INVALID_ARG, "manta_smallsv_download: null argument");
  try {
    rt::setDevice(ctx->deviceId);  // the calling thread may have another device current (multi-GPU hosts)
    rt::ScopedStream onStream(b->main);
    pipeStage(b);
    return smallsvCompact(b, loci, contigs, alignments, 0, contigs_cap, seq_arena, seq_arena_cap, 0, seq_arena_used, bits_arena,
                          bits_arena_cap, 0, bits_arena_used, cigar_arena, cigar_arena_cap, 0, cigar_arena_used);
  } catch (const std::exception& e) {
    return fail(ctx, MANTA_E_HIP, e.what());
  }
}

// ------------------------------------------------------------------------------------------------------
// fused spanning pipeline
// ------------------------------------------------------------------------------------------------------
int manta_spanning_create(
    manta_ctx_t* ctx, const manta_asm_options_t* opt, const manta_align_scores_t* scores, int32_t jump_score, manta_spanning_t** out)
{
  if (!ctx) return MANTA_E_INVALID_ARG;
  if (!opt || !scores || !out) return fail(ctx, MANTA_E_INVALID_ARG, "manta_spanning_create: null argument");
  if (scores->is_allow_edge_insertion)
    return fail(ctx, MANTA_E_INVALID_ARG, "GlobalJumpAligner does not support isAllowEdgeInsertion");
  try {
    rt::setDevice(ctx->deviceId);  // (the pipeline's streams and events belong to the context's device)
    manta_spanning* b = new manta_spanning(ctx);
    b->opt            = *opt;
    b->scores         = *scores;
    b->jumpScore      = jump_score;
    *out              = b;
  } catch (const std::exception& e) {
    return fail(ctx, MANTA_E_HIP, e.what());
  }
  return MANTA_OK;
}

void manta_spanning_destroy(manta_spanning_t* b)
{
  delete b;
}

int manta_spanning_upload(
    manta_spanning_t* b, uint32_t n_loci, const uint8_t* bases, const uint64_t* read_off, const uint32_t* locus_read_begin,
    const uint8_t* refs1, const uint64_t* ref1_off, const uint8_t* refs2, const uint64_t* ref2_off, const manta_jump_cuts_t* cuts)
{
  if (!b) return MANTA_E_INVALID_ARG;
  manta_ctx_t* ctx = b->ctx;
  if (n_loci == 0 || !bases || !read_off || !locus_read_begin || !refs1 || !ref1_off || !refs2 || !ref2_off || !cuts)
    return fail(ctx, MANTA_E_INVALID_ARG, "manta_spanning_upload: null argument or empty batch");
  try {
    rt::setDevice(ctx->deviceId);  // the calling thread may have another device current (multi-GPU hosts)
    rt::ScopedStream onStream(b->main);
    b->uploaded = false;
    int rc      = b->asmStage.plan(b->opt, n_loci, read_off, locus_read_begin);
    if (rc != MANTA_OK) return rc;
    b->nLoci     = n_loci;
    b->ref1Bytes = ref1_off[n_loci];
    b->ref2Bytes = ref2_off[n_loci];
    for (uint32_t l = 0; l < n_loci; ++l) {
      if (ref1_off[l + 1] < ref1_off[l] || ref2_off[l + 1] < ref2_off[l])
        return fail(ctx, MANTA_E_INVALID_ARG, "manta_spanning_upload: reference offsets not monotone");
      const manta_jump_cuts_t& c(cuts[l]);
      if (c.align1_leading_cut < 0 || c.align1_trailing_cut < 0 || c.align2_leading_cut < 0 || c.align2_trailing_cut < 0)
        return fail(ctx, MANTA_E_INVALID_ARG, "manta_spanning_upload: negative reference cut");
    }
    static_assert(sizeof(JumpCuts) == sizeof(manta_jump_cuts_t), "cuts layout");
    uint8_t*  dRefs1   = b->dRefs1.as<uint8_t>(b->ref1Bytes + 16);
    uint64_t* dRef1Off = b->dRef1Off.as<uint64_t>(n_loci + 1);
    uint8_t*  dRefs2   = b->dRefs2.as<uint8_t>(b->ref2Bytes + 16);
    uint64_t* dRef2Off = b->dRef2Off.as<uint64_t>(n_loci + 1);
    JumpCuts* dCuts    = b->dCuts.as<JumpCuts>(n_loci);
    auto      copyRefs = [&] {
      rt::h2d(dRefs1, refs1, b->ref1Bytes);
      rt::h2d(dRef1Off, ref1_off, sizeof(uint64_t) * (n_loci + 1));
      rt::h2d(dRefs2, refs2, b->ref2Bytes);
      rt::h2d(dRef2Off, ref2_off, sizeof(uint64_t) * (n_loci + 1));
      rt::h2d(dCuts, cuts, sizeof(JumpCuts) * n_loci);
    };
    b->hostCuts.assign(reinterpret_cast<const JumpCuts*>(cuts), reinterpret_cast<const JumpCuts*>(cuts) + n_loci);
    b->refsOnCopy = b->streamUploads && !std::getenv("MANTA_AMD_NO_STREAM_UPLOAD");
    if (b->refsOnCopy) {  // whole-batch call: as manta_smallsv_upload -- bases in chunks behind the running assembler, references behind them
      b->asmStage.uploadStreamed(bases, read_off, locus_read_begin, b->copy);
      rt::ScopedStream onCopy(b->copy);
      copyRefs();
      b->refsReady.record();
    } else {
      b->asmStage.upload(bases, read_off, locus_read_begin);
      copyRefs();
      rt::sync();
    }
    b->uploaded = true;
    return MANTA_OK;
  } catch (const std::exception& e) {
    return fail(ctx, MANTA_E_HIP, e.what());
  }
}

int manta_spanning_upload_piles(
    manta_spanning_t* b, uint32_t n_loci, const manta_packed_piles_t* piles, const uint8_t* refs1, const uint64_t* ref1_off,
    const uint8_t* refs2, const uint64_t* ref2_off, const manta_jump_cuts_t* cuts)
{
  if (!b) return MANTA_E_INVALID_ARG;
  manta_ctx_t* ctx = b->ctx;
  if (n_loci == 0 || !refs1 || !ref1_off || !refs2 || !ref2_off || !cuts)
    return fail(ctx, MANTA_E_INVALID_ARG, "manta_spanning_upload_piles: null argument or empty batch");
  int rc = checkPiles(ctx, piles, "manta_spanning_upload_piles");
  if (rc != MANTA_OK) return rc;
  try {
    rt::setDevice(ctx->deviceId);
    rt::ScopedStream onStream(b->main);
    b->uploaded = false;
    b->refsOnCopy = false;
    rc          = b->asmStage.plan(b->opt, n_loci, nullptr, piles->locus_read_begin, piles->read_len);
    if (rc != MANTA_OK) return rc;
    b->asmStage.uploadPiles(*piles);
    b->nLoci     = n_loci;
    b->ref1Bytes = ref1_off[n_loci];
    b->ref2Bytes = ref2_off[n_loci];
    for (uint32_t l = 0; l < n_loci; ++l) {
      if (ref1_off[l + 1] < ref1_off[l] || ref2_off[l + 1] < ref2_off[l])
        return fail(ctx, MANTA_E_INVALID_ARG, "manta_spanning_upload_piles: reference offsets not monotone");
      const manta_jump_cuts_t& c(cuts[l]);
      if (c.align1_leading_cut < 0 || c.align1_trailing_cut < 0 || c.align2_leading_cut < 0 || c.align2_trailing_cut < 0)
        return fail(ctx, MANTA_E_INVALID_ARG, "manta_spanning_upload_piles: negative reference cut");
    }
    rt::h2d(b->dRefs1.as<uint8_t>(b->ref1Bytes + 16), refs1, b->ref1Bytes);
    rt::h2d(b->dRef1Off.as<uint64_t>(n_loci + 1), ref1_off, sizeof(uint64_t) * (n_loci + 1));
    rt::h2d(b->dRefs2.as<uint8_t>(b->ref2Bytes + 16), refs2, b->ref2Bytes);
    rt::h2d(b->dRef2Off.as<uint64_t>(n_loci + 1), ref2_off, sizeof(uint64_t) * (n_loci + 1));
    rt::h2d(b->dCuts.as<JumpCuts>(n_loci), cuts, sizeof(JumpCuts) * n_loci);
    b->hostCuts.assign(reinterpret_cast<const JumpCuts*>(cuts), reinterpret_cast<const JumpCuts*>(cuts) + n_loci);
    rt::sync();
    b->uploaded = true;
    return MANTA_OK;
  } catch (const std::exception& e) {
    return fail(ctx, MANTA_E_HIP, e.what());
  }
}

}  // extern "C"

int spanningRunImpl(manta_spanning_t* b, StageGates* gates)
{
  if (!b) return MANTA_E_INVALID_ARG;
  manta_ctx_t* ctx = b->ctx;
  if (!b->uploaded) return fail(ctx, MANTA_E_INVALID_ARG, "manta_spanning_run: nothing uploaded");
  try {
    rt::setDevice(ctx->deviceId);  // the calling thread may have another device current (multi-GPU hosts)
    rt::ScopedStream onStream(b->main);
    const uint32_t nLoci  = b->nLoci;
    const uint32_t maxAsm = b->opt.max_assembly_count;
    const uint64_t nSlots = uint64_t(nLoci) * maxAsm;
    AsmStage&      as(b->asmStage);
    AlignTaskDev*   dTasks    = b->dTasks.as<AlignTaskDev>(nSlots);
    AlignTaskDev*   dTasks2   = b->dTasks2.as<AlignTaskDev>(nSlots);
    SpanTaskInfo*   dInfo     = b->dInfo.as<SpanTaskInfo>(nSlots);
    AlignResultDev* dResults  = b->dResults.as<AlignResultDev>(nSlots);
    AlignResultDev* dResults2 = b->dResults2.as<AlignResultDev>(nSlots);
    uint32_t*       dBuckets  = b->dBucketIds.as<uint32_t>(nSlots * kNumESet);
    uint32_t*       dBuckets2 = b->dBucketIds2.as<uint32_t>(nSlots * kNumESet);
    // [0..15] counts, [16..31] maxref, [32..47] counts2, [48..63] maxref2, [64..65] cigar_used, [72..87] / [88..103] align counters
    uint32_t*      dSmall   = b->dSmall.as<uint32_t>(128);
    rt::dzero(dSmall, sizeof(uint32_t) * 128);
    rt::dzero(dResults, sizeof(AlignResultDev) * nSlots);
    rt::dzero(dResults2, sizeof(AlignResultDev) * nSlots);

    const bool dbg   = std::getenv("MANTA_AMD_DEBUG") != nullptr;
    auto       stage = [&](const char* what) {
      if (!dbg) return;
      rt::sync();
      std::fprintf(stderr, "manta_amd: spanning_run %s\n", what);
      std::fflush(stderr);
    };
    // ---- the alignment stage as a function of (which loci, on which streams): it runs once over every locus, or -- the early pass --
    // first over the loci that are final after the assembler's first word length, concurrently with the later word lengths, and then
    // over the rest.
    SpanParams S;
    S.loci               = as.dLoci;
    S.contigs            = as.dCont;
    S.seq_arena          = as.dSeq;
    S.n_loci             = nLoci;
    S.max_assembly_count = maxAsm;
    S.refs1              = static_cast<const uint8_t*>(b->dRefs1.p);
    S.ref1_off           = static_cast<const uint64_t*>(b->dRef1Off.p);
    S.refs2              = static_cast<const uint8_t*>(b->dRefs2.p);
    S.ref2_off           = static_cast<const uint64_t*>(b->dRef2Off.p);
    S.cuts               = static_cast<const JumpCuts*>(b->dCuts.p);
    S.tasks              = dTasks;
    S.tasks2             = dTasks2;
    S.info               = dInfo;
    S.bucket_ids         = dBuckets;
    S.bucket_count       = dSmall;
    S.bucket_maxref      = dSmall + 16;
    S.bucket_ids2        = dBuckets2;
    S.bucket_count2      = dSmall + 32;
    S.bucket_maxref2     = dSmall + 48;
    S.cigar_used         = reinterpret_cast<unsigned long long*>(dSmall + 64);
    S.cigar_cap          = 0;
    S.results            = dResults;
    S.cigar              = nullptr;
    S.n_e                = kNumESet;
    for (int i = 0; i < kNumESet; ++i) S.e_set[i] = uint32_t(kESet[i]);
    S.pass_mask  = nullptr;
    S.pass_value = 0;
    const int glueGrid = rt::roundGrid(int(std::min<uint64_t>((nSlots + 63) / 64, uint64_t(std::max(1, ctx->cuCount * 8)))));
    b->stats.n_align_launches = 0;
    b->stats.n_alignments     = 0;
    const size_t wsBudget = workspaceBudget(size_t(48) << 30);
    uint32_t*    dCigar   = nullptr;
    // CIGAR scratch, sized from what the assembler produced: a task takes 4 * contig length + 16 words (spanFileTask), every
    // contig is aligned at most twice (second round: spanning_realign_kernel), and the text arena counter bounds the summed
    // contig lengths.  (A fixed worst case of max_contig_len per slot is ~40x the real need at 200 x 250 bp loci.)
    // `keepWords` > 0: the buffer already holds that many words of the early pass -- a larger one takes them over.
    auto cigarScratch = [&](const uint64_t seqUsed, const uint64_t keepWords) -> uint64_t {
      const uint64_t need = 2 * (4ull * std::min<uint64_t>(seqUsed, as.devSeqCap) + 16ull * nSlots) + 64;
      if (keepWords && (need + 16) * sizeof(uint32_t) > b->dCigar.cap) {
        DevBuf bigger;
        uint32_t* q = bigger.as<uint32_t>(need + 16);
        rt::d2d(q, b->dCigar.p, sizeof(uint32_t) * keepWords);
        rt::sync();
        std::swap(bigger.p, b->dCigar.p);
        std::swap(bigger.cap, b->dCigar.cap);
      }
      dCigar = b->dCigar.as<uint32_t>(need + 16);
      return need;
    };
    auto alignRound = [&](rt::Stream* const* sides, const int maxWaves, const uint32_t* hCounts, const uint32_t* hMaxref, const AlignTaskDev* tasks,
                          AlignResultDev* results, const uint32_t* bucketIds, uint32_t* counters) {
      // buckets on side streams, one slab region each (see manta_smallsv_run)
      struct Launch {
        int      k, grid;
        uint64_t stride, slabOff;
        bool     pair;
      };
      std::vector<Launch> launches;
      uint64_t            slabBytes = 0;
      // (sharing the wave slots among the buckets by work -- tasks x columns x steps -- so that they would end together was measured in
      // round 5 and lost: 170 ms against 86 per 16 384 config-5 loci.  A bucket's waves are long dependent chains; capping its grid only
      // lengthens them.)
      for (int k = kNumESet - 1; k >= 0; --k) {
        const uint32_t cnt = hCounts[k];
        if (cnt == 0) continue;
        // (two alignments per wave in packed 16-bit arithmetic where the scores allow it: align_jump_pair.hpp)
        const bool     pair   = alignUsesPairs(MANTA_ALIGNER_JUMP, k, b->scores.match, b->scores.mismatch, b->scores.open, b->scores.extend, b->scores.off_edge,
                                               b->jumpScore, 0, hMaxref[k]);
        const uint64_t stride = ((pair ? 2 : 1) * alignPtrSlabBytes(MANTA_ALIGNER_JUMP, kESet[k], hMaxref[k]) + 255) & ~uint64_t(255);
        int            grid   = int(std::min<size_t>(pair ? (cnt + 1) / 2 : cnt, size_t(maxWaves)));
        grid                  = int(std::max<size_t>(1, std::min<size_t>(size_t(grid), (wsBudget / 3) / stride)));
        grid                  = rt::roundGrid(grid);
        launches.push_back(Launch{k, grid, stride, slabBytes, pair});
        slabBytes += stride * uint64_t(grid);
      }
      if (launches.empty()) return;
      uint8_t* dWsAll = b->dPtrWs.as<uint8_t>(slabBytes + 256);
      for (size_t i = 0; i < launches.size(); ++i) {
        const Launch& l(launches[i]);
        AlignParams   P;
        P.tasks          = tasks;
        P.results        = results;
        P.cigar          = dCigar;
        P.task_ids       = bucketIds + uint64_t(l.k) * nSlots;
        P.n_tasks        = hCounts[l.k];
        P.n_tasks_dev    = nullptr;
        P.counter        = counters + l.k;
        P.ptr_ws         = dWsAll + l.slabOff;
        P.ptr_ws_stride  = l.stride;
        P.match          = b->scores.match;
        P.mismatch       = b->scores.mismatch;
        P.open           = b->scores.open;
        P.extend         = b->scores.extend;
        P.off_edge       = b->scores.off_edge;
        P.allow_edge_ins = 0;
        P.extra          = b->jumpScore;
        {
          rt::ScopedStream onSide(*sides[i % 3]);
          launchAlignKind(MANTA_ALIGNER_JUMP, l.k, l.grid, P, l.pair);
        }
        b->stats.n_align_launches++;
        b->stats.n_alignments += hCounts[l.k];
      }
      for (size_t i = 0; i < std::min<size_t>(launches.size(), 3); ++i) {
        b->sideDone[i].recordOn(*sides[i]);
        rt::curStreamWaits(b->sideDone[i]);
      }
      rt::sync();  // the next stage reads the results and may re-size the slab buffer
    };
    /// schedule -> round 1 -> re-align rule -> round 2 over the loci of S.pass_mask / S.pass_value, on the current stream + `sides`
    auto alignPass = [&](rt::Stream* const* sides, const int maxWaves, const char* what) {
      S.cigar = dCigar;
      rt::launch(spanning_schedule_kernel, glueGrid, 0, S);
      uint32_t hSmall[64];
      rt::d2h(hSmall, dSmall, sizeof(hSmall));
      alignRound(sides, maxWaves, hSmall, hSmall + 16, dTasks, dResults, dBuckets, dSmall + 72);
      stage(what);
      rt::launch(spanning_realign_kernel, rt::roundGrid(int(std::min<uint64_t>((nLoci + 63) / 64, uint64_t(std::max(1, ctx->cuCount * 8))))), 0, S);
      rt::d2h(hSmall, dSmall, sizeof(hSmall));
      alignRound(sides, maxWaves, hSmall + 32, hSmall + 48, dTasks2, dResults2, dBuckets2, dSmall + 88);
    };
    rt::Stream* const plainSides[3] = {&b->side[0], &b->side[1], &b->side[2]};

    // The early pass: after the first word length 94 % of the config-4/5 loci are final, while the tandem piles go through up to ten more
    // word lengths -- launches that are latency-bound on small lists and leave most of the device idle.  The jump aligner of the final loci
    // runs beside them: a mark kernel on the assembler's stream freezes "who is final" (span_mark_kernel), the pass itself runs on `early`
    // + CU-masked side streams (a persistent aligner grid on every CU would keep graph_big_kernel -- one workgroup owns a CU -- from ever
    // starting; MANTA_AMD_EARLY_RESERVE_CUS, default a quarter of the CUs, stay free for the rounds), the rest of the loci follow when the
    // assembler is done.  Not with stage gates (pipelined workers hold one stage at a time).  MANTA_AMD_EARLY_ALIGN=0 switches it off.
    static const bool earlyOff = std::getenv("MANTA_AMD_EARLY_ALIGN") && std::atoi(std::getenv("MANTA_AMD_EARLY_ALIGN")) == 0;
    const bool        early    = !earlyOff && !gates && as.useFast && as.bigRounds > 1 && !as.bigIds.empty();
    uint8_t*          dMask    = early ? b->dPassMask.as<uint8_t>(nLoci) : nullptr;
    uint64_t asmCnt[4];
    b->earlyLoci = 0;
    {
      GateLock only(gates, &StageGates::asmMu);
      std::unique_lock<std::mutex> streamedOnly(streamedAsmMu(ctx), std::defer_lock);  // see smallsvRunImpl
      if (as.streaming) streamedOnly.lock();
      as.stageQueued = false;  // (a run that failed behind its queued staging must not leave the flag to the next one)
      b->evStart.record();
      if (early) {
        as.afterFirstRound = [&] {
          SpanMarkParams K;
          K.loci    = as.dLoci;
          K.n_loci  = nLoci;
          K.mask    = dMask;
          K.counter = dSmall + 120;
          rt::launch(span_mark_kernel, rt::roundGrid(int(std::min<uint64_t>((nLoci + 63) / 64, uint64_t(std::max(1, ctx->cuCount * 4))))), 0, K);
          b->evRound0.record();
        };
      }
      struct HookReset {
        AsmStage& a;
        ~HookReset() { a.afterFirstRound = nullptr; }
      } hookReset{as};
      as.launch();
      b->evAsm.record();
      if (early && as.firstRoundHookRan) {
        // (everything below is queued behind evRound0 on `early`; `main` keeps running the word-length rounds)
        static const int reserveEnv = std::getenv("MANTA_AMD_EARLY_RESERVE_CUS") ? std::atoi(std::getenv("MANTA_AMD_EARLY_RESERVE_CUS")) : -1;
        const int reserve = std::max(8, std::min(ctx->cuCount - 8, reserveEnv >= 0 ? (reserveEnv / 8) * 8 : ((ctx->cuCount / 4) / 8) * 8));
        if (b->sideEarlyReserved != reserve) {
          b->early.reset(new rt::Stream(ctx->cuCount, reserve));
          for (int i = 0; i < 3; ++i) b->sideEarly[i].reset(new rt::Stream(ctx->cuCount, reserve));
          b->sideEarlyReserved = reserve;
          if (dbg) std::fprintf(stderr, "manta_amd: early alignment pass: %d of %d CUs reserved for the word-length rounds (CU mask %s)\n", reserve, ctx->cuCount,
                                b->sideEarly[0]->masked() ? "set" : "not available");
        }
        rt::ScopedStream onEarly(*b->early);
        rt::curStreamWaits(b->evRound0);
        if (b->refsOnCopy) rt::curStreamWaits(b->refsReady);
        rt::Stream* const earlySides[3] = {b->sideEarly[0].get(), b->sideEarly[1].get(), b->sideEarly[2].get()};
        uint64_t cnt0[4];
        uint32_t nEarly = 0;
        rt::d2h(cnt0, as.dCnt, sizeof(cnt0));  // (text arena in use: at least what the final loci wrote -- the rounds keep adding)
        rt::d2h(&nEarly, dSmall + 120, sizeof(nEarly));
        b->earlyLoci = nEarly;
        if (nEarly) {
          // (room for a quarter more text than is there now, so that the second pass seldom has to move the buffer)
          S.cigar_cap  = std::min<uint64_t>(cigarScratch(cnt0[1] + cnt0[1] / 4 + 65536, 0), 0xfffffff0ull);
          S.pass_mask  = dMask;
          S.pass_value = 1;
          static const int wavesEnv = std::getenv("MANTA_AMD_EARLY_WAVES_PER_CU") ? std::atoi(std::getenv("MANTA_AMD_EARLY_WAVES_PER_CU")) : 0;
          const int maxWavesEarly = std::max(1, (ctx->cuCount - (b->sideEarly[0]->masked() ? reserve : 0)) * (wavesEnv > 0 ? wavesEnv : alignWavesPerCu()));
          alignPass(earlySides, maxWavesEarly, "aligned round 1 (early pass)");
          stage("aligned round 2 (early pass)");
        }
      }
      rt::d2h(asmCnt, as.dCnt, sizeof(asmCnt));  // (waits for the assembler)
      if (asmCnt[3]) {  // loci that did not fit the typical-case workspace (see smallsvRunImpl)
        as.rerunCapacityFailures(asmCnt[3]);
        rt::d2h(asmCnt, as.dCnt, sizeof(asmCnt));  // the text arena grew
      }
    }
    stage("assembled");
    GateLock alignOnly(gates, &StageGates::alignMu);
    if (b->refsOnCopy) rt::curStreamWaits(b->refsReady);  // reference windows of a streamed upload (manta_spanning_upload)
    uint64_t cigarCap = 0;
    {
      uint64_t keepWords = 0;
      if (b->earlyLoci) {
        // second pass: fresh bucket lists (the allocator of the CIGAR scratch goes on where the early pass stopped)
        rt::d2h(&keepWords, dSmall + 64, sizeof(keepWords));
        keepWords = std::min<uint64_t>(keepWords, S.cigar_cap);
        rt::dzero(dSmall, sizeof(uint32_t) * 64);
        rt::dzero(dSmall + 72, sizeof(uint32_t) * 32);
        S.pass_mask  = dMask;
        S.pass_value = 0;
      }
      cigarCap = cigarScratch(asmCnt[1], keepWords);
      if (cigarCap + 16 > 0xffffffffull)
        return fail(ctx, MANTA_E_UNSUPPORTED, "manta_spanning_run: alignment scratch of this block exceeds 2^32 words; use smaller blocks "
                                           "(manta_spanning_batch splits a batch into blocks, manta_batch_plan_t::block_loci)");
      S.cigar_cap = cigarCap;
    }
    b->evSched.record();
    const int maxWaves = std::max(1, ctx->cuCount * alignWavesPerCu());
    alignPass(plainSides, maxWaves, "aligned round 1");
    launchPack(b, dTasks, dTasks2, dResults, dResults2, nullptr, dInfo, dCigar, cigarCap);
    b->evAlign.record();
    if (b->stageBehindRun) pipeStageEnqueue(b);
    rt::sync();
    alignOnly.release();
    if (as.streaming) {  // every chunk was consumed by the kernel, so this returns at once; it closes the copy stream's error state
      rt::ScopedStream onCopy(b->copy);
      rt::sync();
    }
    stage("aligned round 2");
    b->stats.assemble_ms = rt::elapsedMs(b->evStart, b->evAsm);
    b->stats.schedule_ms = rt::elapsedMs(b->evAsm, b->evSched);
    b->stats.align_ms    = rt::elapsedMs(b->evSched, b->evAlign);
    b->stats.total_ms    = rt::elapsedMs(b->evStart, b->evAlign);
    b->ran               = true;
    return MANTA_OK;
  } catch (const std::exception& e) {
    drainCopyStream(b);
    return fail(ctx, MANTA_E_HIP, e.what());
  }
}

extern "C" {

int manta_spanning_run(manta_spanning_t* b) { return spanningRunImpl(b, nullptr); }

int manta_spanning_stats(const manta_spanning_t* b, manta_smallsv_stats_t* stats)
{
  if (!b || !stats) return MANTA_E_INVALID_ARG;
  *stats = b->stats;
  return MANTA_OK;
}

int manta_spanning_output_sizes(const manta_spanning_t* b, uint64_t* contigs, uint64_t* seq_bytes, uint64_t* bits_words, uint64_t* cigar_words)
{
  if (!b || !contigs || !seq_bytes || !bits_words || !cigar_words) return MANTA_E_INVALID_ARG;
  if (!b->ran) return fail(b->ctx, MANTA_E_INVALID_ARG, "manta_spanning_output_sizes: run first");
  try {
    rt::setDevice(b->ctx->deviceId);  // the calling thread may have another device current (multi-GPU hosts)
    rt::ScopedStream onStream(const_cast<manta_spanning_t*>(b)->main);
    b->asmStage.outputSizes(*contigs, *seq_bytes, *bits_words);
    uint32_t hSmall[72];
    rt::d2h(hSmall, b->dSmall.p, sizeof(hSmall));
    uint64_t cig = 0;
    std::memcpy(&cig, hSmall + 64, sizeof(uint64_t));
    *cigar_words = cig + 64;
    return MANTA_OK;
  } catch (const std::exception& e) {
    return fail(b->ctx, MANTA_E_HIP, e.what());
  }
}

}  // extern "C"

namespace {

int spanningCompact(
    manta_spanning* b, manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs, manta_spanning_alignment_t* alignments,
    uint64_t contigBase, uint64_t contigs_cap, uint8_t* seq_arena, uint64_t seq_arena_cap, uint64_t seqBase, uint64_t* seq_arena_used,
    uint64_t* bits_arena, uint64_t bits_arena_cap, uint64_t bitsBase, uint64_t* bits_arena_used, uint32_t* cigar_arena,
    uint64_t cigar_arena_cap, uint64_t cigarBase, uint64_t* cigar_arena_used, uint32_t lBegin = 0, uint32_t lEnd = ~0u,
    uint64_t* cellsOut = nullptr, uint64_t* ptrBytesOut = nullptr)
{
  // [lBegin, lEnd) and the *Out totals: as smallsvCompact
  manta_ctx_t* ctx = b->ctx;
  lEnd             = std::min(lEnd, b->nLoci);
  int rc = b->asmStage.compact(PackedContigs<manta_spanning>{b}, loci, contigs + contigBase, contigs_cap, seq_arena, seq_arena_cap, seq_arena_used,
                               bits_arena, bits_arena_cap, bits_arena_used, contigBase, seqBase, bitsBase, lBegin, lEnd);
  if (rc != MANTA_OK && !perItemCode(rc)) return rc;
  uint64_t       used = 0, cells = 0, ptrBytes = 0;
  int            worst = rc;
  for (uint32_t l = lBegin; l < lEnd; ++l) {
    if (loci[l].status != MANTA_OK) continue;
    for (uint32_t c = 0; c < loci[l].n_contigs; ++c) {
      manta_spanning_alignment_t& a(alignments[loci[l].first_contig + c]);
      std::memset(&a, 0, sizeof(a));
      const PackedContigOut& h(b->hPacked[b->hFirst[l] + c]);
      const bool             uncut = h.a != 0;
      a.is_uncut                   = uncut ? 1 : 0;
      if (h.info_status != 0 || h.res_status != 0 || h.bucket < 0) {
        a.align.status = (h.info_status == 5) ? MANTA_E_DEVICE_FAULT : (h.info_status == 6) ? MANTA_E_EMPTY_SEQ : MANTA_E_UNSUPPORTED;
        worst          = a.align.status;
        continue;
      }
      const uint64_t n = uint64_t(h.cigar1_len) + h.cigar2_len;
      if (used + n > cigar_arena_cap) return fail(ctx, MANTA_E_CAPACITY, "manta_spanning_download: cigar arena too small");
      std::memcpy(cigar_arena + used, b->hCig + h.cigar_off, sizeof(uint32_t) * n);
      a.align.score            = h.score;
      a.align.is_jumped        = h.is_jumped;
      a.align.begin_pos1       = h.begin1 + (uncut ? 0 : b->hostCuts[l].a1Lead);  // SVCandidateAssemblyRefiner.cpp:1716-1717
      a.align.begin_pos2       = h.begin2 + (uncut ? 0 : b->hostCuts[l].a2Lead);
      a.align.jump_insert_size = h.jump_insert_size;
      a.align.jump_range       = h.jump_range;
      a.align.cigar1_len       = h.cigar1_len;
      a.align.cigar2_len       = h.cigar2_len;
      a.align.cigar1_off       = cigarBase + used;
      a.align.cigar2_off       = cigarBase + used + h.cigar1_len;
      used += n;
      cells += uint64_t(h.query_len) * h.ref_len;
      ptrBytes += (uint64_t(h.query_len) + 1) * (uint64_t(h.ref_len) + 2);
    }
  }
  if (cellsOut) {
    *cellsOut    = cells;
    *ptrBytesOut = ptrBytes;
  } else {
    b->stats.dp_cells         = cells;
    b->stats.ptr_matrix_bytes = ptrBytes;
  }
  if (cigar_arena_used) *cigar_arena_used = used;
  if (worst != MANTA_OK) return fail(ctx, worst, "manta_spanning_download: one or more loci/contigs failed; see per-item status");
  return MANTA_OK;
}

/// CIGAR words spanningCompact writes for the loci [lBegin, lEnd)
uint64_t spanningCigarWords(const manta_spanning* b, uint32_t lBegin, uint32_t lEnd)
{
  uint64_t n = 0;
  for (uint32_t l = lBegin; l < lEnd; ++l) {
    const AsmLocusOut& h(b->asmStage.hLoci[l]);
    if (h.status != ASM_OK) continue;
    for (uint32_t c = 0; c < h.n_contigs; ++c) {
      const PackedContigOut& pc(b->hPacked[b->hFirst[l] + c]);
      if (pc.info_status == 0 && pc.res_status == 0 && pc.bucket >= 0) n += uint64_t(pc.cigar1_len) + pc.cigar2_len;
    }
  }
  return n;
}

}  // namespace

extern "C" {

int manta_spanning_download(
    manta_spanning_t* b, manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs, manta_spanning_alignment_t* alignments,
    uint64_t contigs_cap, uint8_t* seq_arena, uint64_t seq_arena_cap, uint64_t* seq_arena_used, uint64_t* bits_arena,
    uint64_t bits_arena_cap, uint64_t* bits_arena_used, uint32_t* cigar_arena, uint64_t cigar_arena_cap,
    uint64_t* cigar_arena_used)
{
  if (!b) return MANTA_E_INVALID_ARG;
  manta_ctx_t* ctx = b->ctx;
  if (!b->ran) return fail(ctx, MANTA_E_INVALID_ARG, "manta_spanning_download: run first");
  if (!loci || !contigs || !alignments || !seq_arena || !bits_arena || !cigar_arena)
    return fail(ctx, MANTA_E_INVALID_ARG, "manta_spanning_download: null argument");
  try {
    rt::setDevice(ctx->deviceId);  // the calling thread may have another device current (multi-GPU hosts)
    rt::ScopedStream onStream(b->main);
    pipeStage(b);
    return spanningCompact(b, loci, contigs, alignments, 0, contigs_cap, seq_arena, seq_arena_cap, 0, seq_arena_used, bits_arena,
                           bits_arena_cap, 0, bits_arena_used, cigar_arena, cigar_arena_cap, 0, cigar_arena_used);
  } catch (const std::exception& e) {
    return fail(ctx, MANTA_E_HIP, e.what());
  }
}

}  // extern "C"



// ------------------------------------------------------------------------------------------------------
// per-locus word lengths, pinned host memory, whole-batch calls
// ------------------------------------------------------------------------------------------------------
manta_ctx::~manta_ctx()
{
  for (manta_smallsv* p : smallPool) delete p;
  for (manta_spanning* p : spanPool) delete p;
}

namespace {

int setWordLengths(manta_ctx_t* ctx, AsmStage& as, uint32_t n_loci, const uint32_t* minWl, const uint32_t* maxWl)
{
  as.locusMinWl.clear();
  as.locusMaxWl.clear();
  if (!minWl && !maxWl) return MANTA_OK;
  if (!minWl || !maxWl || n_loci == 0) return fail(ctx, MANTA_E_INVALID_ARG, "set_word_lengths: both arrays (or neither) must be given");
  as.locusMinWl.assign(minWl, minWl + n_loci);
  as.locusMaxWl.assign(maxWl, maxWl + n_loci);
  return MANTA_OK;
}

/// what the workers of one whole-batch call share: the block queue, the bump allocators over the caller's arenas, the
/// first fatal error and the statistics
struct BatchShared {
  std::vector<uint32_t> blockOrder;  // block indices, most expensive first
  std::atomic<uint32_t> next{0};
  std::atomic<uint64_t> contigsUsed{0}, seqUsed{0}, bitsUsed{0}, cigarUsed{0};
  std::mutex            mu;
  std::mutex            kernelMu;  // MANTA_BATCH_SERIAL_KERNELS
  bool                  serialKernels = false;
  std::vector<StageGates> gates;   // one per device; used whenever a device has more than one worker
  bool                  pipelineStages = false;
  uint32_t*             sharedQueue = nullptr;  // manta_batch_plan_t::shared_queue: the block counter of several processes
  std::vector<uint32_t> lociOfCtx;  // loci processed per context (device)
  /// next position of the block queue: this call's own counter, or the counter shared by the processes of the node
  uint32_t takeNext() { return sharedQueue ? __atomic_fetch_add(sharedQueue, 1u, __ATOMIC_RELAXED) : next.fetch_add(1); }
  int                   fatal = MANTA_OK, worst = MANTA_OK;
  std::string           msg;
  manta_batch_stats_t   st{};
  void                  error(int code, const std::string& m, bool isFatal)
  {
    std::lock_guard<std::mutex> g(mu);
    if (isFatal) {
      if (fatal == MANTA_OK) {
        fatal = code;
        msg   = m;
      }
    } else if (worst == MANTA_OK) {
      worst = code;
      msg   = m;
    }
  }
  bool stop()
  {
    std::lock_guard<std::mutex> g(mu);
    return fatal != MANTA_OK;
  }
};

/// default block size of a whole-batch call: one block per call is the measured optimum as long as the
/// device-side arenas stay moderate; beyond that, equal blocks of at most 65536 loci / 4 GiB of read bases
uint32_t autoBlockLoci(const uint32_t n_loci, const uint64_t totalBases)
{
  const uint64_t byLoci  = (uint64_t(n_loci) + 65535) / 65536;
  const uint64_t byBases = (totalBases + (uint64_t(4) << 30) - 1) / (uint64_t(4) << 30);
  const uint64_t nBlocks = std::max<uint64_t>(1, std::max(byLoci, byBases));
  return uint32_t((uint64_t(n_loci) + nBlocks - 1) / nBlocks);
}

/// several devices (or processes) on one queue: blocks small enough that the queue can balance uneven costs -- about eight
/// per puller would be ideal -- but not below what keeps a device efficient.  Measured on MI355X with the LDS assembler pipeline
/// (config-2 loci, one device): 10 000 loci as one block 971 k loci/s, as blocks of 5 000 778 k, of 4 096 633 k (fixed per-block
/// host work and kernel tails) -- hence a floor of 8 192.
uint32_t nodeBlockLoci(const uint32_t n_loci, const uint64_t totalBases)
{
  const uint32_t one = autoBlockLoci(n_loci, totalBases);
  return std::max<uint32_t>(1, std::min<uint32_t>(one, std::max<uint32_t>(8192, n_loci / 64)));
}

/// contiguous blocks of `blockLoci` loci, ordered by decreasing cost (reads x bases, the same estimate the kernels'
/// work queue uses): EdgeRetrieverBin.cpp:38-57 hands out contiguous edge ranges too, but statically
void planBlocks(BatchShared& sh, uint32_t n_loci, uint32_t blockLoci, const uint64_t* read_off, const uint32_t* locus_read_begin,
                const uint32_t* read_len = nullptr)
{
  const uint32_t        nBlocks = (n_loci + blockLoci - 1) / blockLoci;
  std::vector<uint64_t> cost(nBlocks, 0);
  for (uint32_t b = 0; b < nBlocks; ++b) {
    const uint32_t l0 = b * blockLoci, l1 = std::min(n_loci, l0 + blockLoci);
    for (uint32_t l = l0; l < l1; ++l) {
      const uint32_t rb = locus_read_begin[l], re = locus_read_begin[l + 1];
      uint64_t bases = 0;
      if (read_off)
        bases = read_off[re] - read_off[rb];
      else
        for (uint32_t r = rb; r < re; ++r) bases += read_len[r];
      cost[b] += bases * uint64_t(re - rb);
    }
  }
  sh.blockOrder.resize(nBlocks);
  for (uint32_t b = 0; b < nBlocks; ++b) sh.blockOrder[b] = b;
  std::stable_sort(sh.blockOrder.begin(), sh.blockOrder.end(), [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
}

template <typename T>
void rebase(std::vector<T>& out, const T* src, size_t first, size_t count)
{
  out.resize(count);
  const T base = src[first];
  for (size_t i = 0; i < count; ++i) out[i] = src[first + i] - base;
}

}  // namespace

extern "C" {

int manta_host_alloc(uint64_t bytes, void** out)
{
  if (!out) return MANTA_E_INVALID_ARG;
  try {
    *out = rt::hostAlloc(size_t(bytes));
    return MANTA_OK;
  } catch (const std::exception& e) {
    g_createError = e.what();
    *out          = nullptr;
    return MANTA_E_HIP;
  }
}

void manta_host_free(void* p)
{
  if (p) rt::hostFree(p);
}

int manta_smallsv_set_word_lengths(manta_smallsv_t* b, uint32_t n_loci, const uint32_t* min_word_length, const uint32_t* max_word_length)
{
  if (!b) return MANTA_E_INVALID_ARG;
  return setWordLengths(b->ctx, b->asmStage, n_loci, min_word_length, max_word_length);
}

int manta_spanning_set_word_lengths(manta_spanning_t* b, uint32_t n_loci, const uint32_t* min_word_length, const uint32_t* max_word_length)
{
  if (!b) return MANTA_E_INVALID_ARG;
  return setWordLengths(b->ctx, b->asmStage, n_loci, min_word_length, max_word_length);
}

namespace {
int smallsvBatchImpl(
    manta_ctx_t* const* ctxs, const uint32_t nCtx, uint32_t* lociPerDevice, const manta_asm_options_t* opt, const manta_align_scores_t* scores, int32_t large_indel_score, uint32_t n_loci,
    const uint8_t* bases, const uint64_t* read_off, const uint32_t* locus_read_begin, const manta_packed_piles_t* piles, const uint8_t* refs,
    const uint64_t* ref_off, const manta_ref_cuts_t* cuts, const uint32_t* locus_min_word_length, const uint32_t* locus_max_word_length,
    manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs, manta_smallsv_alignment_t* alignments, uint64_t contigs_cap,
    uint8_t* seq_arena, uint64_t seq_arena_cap, uint64_t* seq_arena_used, uint64_t* bits_arena, uint64_t bits_arena_cap,
    uint64_t* bits_arena_used, uint32_t* cigar_arena, uint64_t cigar_arena_cap, uint64_t* cigar_arena_used,
    const manta_batch_plan_t* plan, manta_batch_stats_t* stats)
{
  if (!ctxs || nCtx == 0 || !ctxs[0]) return MANTA_E_INVALID_ARG;
  manta_ctx_t* ctx = ctxs[0];  // (call-level errors are reported here)
  if (!opt || !scores || n_loci == 0 || (!piles && (!bases || !read_off)) || !locus_read_begin || !refs || !ref_off || !cuts || !loci ||
      !contigs || !alignments || !seq_arena || !bits_arena || !cigar_arena)
    return fail(ctx, MANTA_E_INVALID_ARG, "manta_smallsv_batch: null argument or empty batch");
  if ((locus_min_word_length == nullptr) != (locus_max_word_length == nullptr))
    return fail(ctx, MANTA_E_INVALID_ARG, "manta_smallsv_batch: per-locus word lengths need both arrays");
  for (uint32_t l = 0; l < n_loci; ++l)
    if (locus_read_begin[l + 1] < locus_read_begin[l] || ref_off[l + 1] < ref_off[l])
      return fail(ctx, MANTA_E_INVALID_ARG, "manta_smallsv_batch: offsets not monotone");
  const uint64_t totalBases = piles ? 0 : read_off[locus_read_begin[n_loci]] - read_off[locus_read_begin[0]];
  const bool     shared     = (plan && plan->shared_queue) || nCtx > 1;  // several devices pull from the queue
  const uint32_t blockLoci  = (plan && plan->block_loci) ? plan->block_loci : (shared ? nodeBlockLoci(n_loci, totalBases) : autoBlockLoci(n_loci, totalBases));
  BatchShared    sh;
  sh.serialKernels  = plan && (plan->flags & MANTA_BATCH_SERIAL_KERNELS);
  sh.sharedQueue    = plan ? plan->shared_queue : nullptr;
  sh.gates          = std::vector<StageGates>(nCtx);
  sh.lociOfCtx.assign(nCtx, 0);
  planBlocks(sh, n_loci, blockLoci, piles ? nullptr : read_off, locus_read_begin, piles ? piles->read_len : nullptr);
  const uint32_t nBlocks    = uint32_t(sh.blockOrder.size());
  const uint32_t perCtx     = std::max(1u, (plan && plan->n_workers) ? plan->n_workers : 1u);
  const uint32_t nWorkers   = std::max(1u, std::min(nBlocks, perCtx * nCtx));
  sh.pipelineStages         = perCtx > 1 && !std::getenv("MANTA_AMD_NO_STAGE_GATES");  // (experiments: concurrent workers without the stage gates)
  if (sh.sharedQueue)  // blocks another process of the node takes stay marked; the caller merges (bench.py: the final gather)
    for (uint32_t l = 0; l < n_loci; ++l) {
      std::memset(&loci[l], 0, sizeof(loci[l]));
      loci[l].status = MANTA_E_NOT_TAKEN;
    }
  try {
    for (uint32_t c = 0; c < nCtx; ++c) {
      rt::setDevice(ctxs[c]->deviceId);
      while (ctxs[c]->smallPool.size() < (nWorkers + nCtx - 1) / nCtx) ctxs[c]->smallPool.push_back(new manta_smallsv(ctxs[c]));
    }
  } catch (const std::exception& e) {
    return fail(ctx, MANTA_E_HIP, e.what());
  }
  const double tStart = nowMs();
  auto         worker = [&](const uint32_t w) {
    manta_ctx_t*   ctx = ctxs[w % nCtx];  // (shadows the call-level context: this worker's device)
    manta_smallsv* b   = ctx->smallPool[w / nCtx];
    try {
      rt::setDevice(ctx->deviceId);
      // a pooled pipeline that served other options or scores: its bucket counts say nothing about this call's contigs
      if (std::memcmp(&b->opt, opt, sizeof(*opt)) != 0 || std::memcmp(&b->scores, scores, sizeof(*scores)) != 0 || b->largeIndel != large_indel_score)
        b->bucketHistory = false;
      b->opt           = *opt;
      b->scores        = *scores;
      b->largeIndel    = large_indel_score;
      b->streamUploads = !(plan && (plan->flags & MANTA_BATCH_NO_STREAMED_UPLOAD));
      b->asmStage.wavesPerCuCap = sh.pipelineStages ? kPipelinedAsmWavesPerCu : 0;
      std::vector<uint64_t> rOff, fOff;
      std::vector<uint32_t> lBeg;
      while (!sh.stop()) {
        const uint32_t qi = sh.takeNext();
        if (qi >= nBlocks) break;
        const uint32_t blk = sh.blockOrder[qi];
        const uint32_t l0 = blk * blockLoci, l1 = std::min(n_loci, l0 + blockLoci), n = l1 - l0;
        const uint32_t r0 = locus_read_begin[l0], r1 = locus_read_begin[l1];
        if (!piles) rebase(rOff, read_off, r0, size_t(r1 - r0) + 1);
        rebase(lBeg, locus_read_begin, l0, size_t(n) + 1);
        rebase(fOff, ref_off, l0, size_t(n) + 1);
        setWordLengths(ctx, b->asmStage, n, locus_min_word_length ? locus_min_word_length + l0 : nullptr,
                       locus_max_word_length ? locus_max_word_length + l0 : nullptr);
        const double t0 = nowMs();
        int          rc;
        if (piles) {
          manta_packed_piles_t pl = *piles;
          pl.read_len            = piles->read_len + r0;
          pl.read_code_off       = piles->read_code_off + r0;
          pl.read_mask_off       = piles->read_mask_off + r0;
          pl.locus_read_begin    = lBeg.data();
          rc                     = manta_smallsv_upload_piles(b, n, &pl, refs + ref_off[l0], fOff.data(), cuts + l0);
        } else {
          rc = manta_smallsv_upload(b, n, bases + read_off[r0], rOff.data(), lBeg.data(), refs + ref_off[l0], fOff.data(), cuts + l0);
        }
        if (perItemCode(rc)) {  // a locus outside the supported envelope: this block's loci carry the code, the batch goes on
          for (uint32_t l = l0; l < l1; ++l) {
            std::memset(&loci[l], 0, sizeof(loci[l]));
            loci[l].status = rc;
          }
          sh.error(rc, lastErrorOf(ctx), false);
          continue;
        }
        if (rc != MANTA_OK) {
          sh.error(rc, lastErrorOf(ctx), true);
          break;
        }
        const double t1 = nowMs();
        {
          std::unique_lock<std::mutex> only(sh.kernelMu, std::defer_lock);
          if (sh.serialKernels) only.lock();
          b->stageBehindRun = true;
          rc = smallsvRunImpl(b, sh.pipelineStages ? &sh.gates[w % nCtx] : nullptr);
        }
        if (rc != MANTA_OK) {
          sh.error(rc, lastErrorOf(ctx), true);
          break;
        }
        const double t2 = nowMs();
        uint64_t     nC = 0, nS = 0, nB = 0, nG = 0;
        {
          rt::ScopedStream onStream(b->main);
          pipeStage(b);
        }
        const double tStage = nowMs();
        // compaction into the caller's arrays: a few contiguous locus ranges, one host thread each.  Pass 1 sizes the ranges,
        // the block then reserves its region of the caller's arenas, pass 2 writes every range at its own offset.
        struct Range {
          uint64_t nC = 0, nS = 0, nB = 0, nG = 0, cells = 0, ptrBytes = 0;
          uint64_t c0 = 0, s0 = 0, b0 = 0, g0 = 0;
          int      rc = MANTA_OK;
        };
        const unsigned     parts = hostParts(uint64_t(n) * 4);  // (a block of >= 1024 loci is worth the threads)
        std::vector<Range> rg(parts);
        hostParallel(n, parts, [&](unsigned t, uint64_t a, uint64_t z) {
          b->asmStage.rangeSizes(PackedContigs<manta_smallsv>{b}, uint32_t(a), uint32_t(z), rg[t].nC, rg[t].nS, rg[t].nB);
          rg[t].nG = smallsvCigarWords(b, uint32_t(a), uint32_t(z));
        });
        for (unsigned t = 0; t < parts; ++t) {
          rg[t].c0 = nC, rg[t].s0 = nS, rg[t].b0 = nB, rg[t].g0 = nG;
          nC += rg[t].nC, nS += rg[t].nS, nB += rg[t].nB, nG += rg[t].nG;
        }
        const double tSizes = nowMs();
        const uint64_t cBase = sh.contigsUsed.fetch_add(nC), sBase = sh.seqUsed.fetch_add(nS), bBase = sh.bitsUsed.fetch_add(nB),
                       gBase = sh.cigarUsed.fetch_add(nG);
        if (cBase + nC > contigs_cap || sBase + nS > seq_arena_cap || bBase + nB > bits_arena_cap || gBase + nG > cigar_arena_cap) {
          sh.error(MANTA_E_CAPACITY, "manta_smallsv_batch: caller arenas too small", true);
          break;
        }
        hostParallel(n, parts, [&](unsigned t, uint64_t a, uint64_t z) {
          const Range& r(rg[t]);
          rg[t].rc = smallsvCompact(b, loci + l0, contigs, alignments, cBase + r.c0, r.nC, seq_arena + sBase + r.s0, r.nS, sBase + r.s0, nullptr,
                                    bits_arena + bBase + r.b0, r.nB, bBase + r.b0, nullptr, cigar_arena + gBase + r.g0, r.nG, gBase + r.g0,
                                    nullptr, uint32_t(a), uint32_t(z), &rg[t].cells, &rg[t].ptrBytes);
        });
        rc = MANTA_OK;
        b->stats.dp_cells = b->stats.ptr_matrix_bytes = 0;
        for (const Range& r : rg) {
          if (r.rc != MANTA_OK && (rc == MANTA_OK || !perItemCode(r.rc))) rc = r.rc;
          b->stats.dp_cells += r.cells;
          b->stats.ptr_matrix_bytes += r.ptrBytes;
        }
        const double t3 = nowMs();
        if (std::getenv("MANTA_AMD_DEBUG_TIMING"))
          std::fprintf(stderr, "manta_amd: smallsv_batch block: upload %.2f ms, kernels %.2f, stage-out %.2f, sizes %.2f, compact %.2f\n", t1 - t0, t2 - t1,
                       tStage - t2, tSizes - tStage, t3 - tSizes);
        if (rc != MANTA_OK) {
          sh.error(rc, lastErrorOf(ctx), !perItemCode(rc));
          if (!perItemCode(rc)) break;
        }
        std::lock_guard<std::mutex> g(sh.mu);
        sh.lociOfCtx[w % nCtx] += n;
        sh.st.h2d_ms += t1 - t0;
        sh.st.kernel_ms += t2 - t1;
        sh.st.d2h_ms += t3 - t2;
        sh.st.assemble_ms += b->stats.assemble_ms;
        sh.st.schedule_ms += b->stats.schedule_ms;
        sh.st.align_ms += b->stats.align_ms;
        sh.st.n_alignments += b->stats.n_alignments;
        sh.st.n_align_launches += b->stats.n_align_launches;
        sh.st.dp_cells += b->stats.dp_cells;
        sh.st.ptr_matrix_bytes += b->stats.ptr_matrix_bytes;
        sh.st.n_loci_lds_small += b->asmStage.fastIds.size();
        sh.st.n_loci_lds_big += b->asmStage.bigIds.size();
        sh.st.n_loci_handed_back += b->asmStage.ldsFallbacks;
        sh.st.n_loci_general += b->asmStage.useFast ? b->asmStage.genIds.size() : size_t(n);
        sh.st.h2d_bytes += (piles ? b->asmStage.plBytes : (read_off[r1] - read_off[r0]) + 8ull * (r1 - r0 + 1)) + (ref_off[l1] - ref_off[l0]) +
                           12ull * (n + 1) + 16ull * n;
        sh.st.d2h_bytes += pipeStagedBytes(b);
      }
    } catch (const std::exception& e) {
      sh.error(MANTA_E_HIP, e.what(), true);
    }
  };
  std::vector<std::thread> threads;
  for (uint32_t w = 1; w < nWorkers; ++w) threads.emplace_back(worker, w);
  worker(0);
  for (std::thread& t : threads) t.join();
  sh.st.wall_ms   = nowMs() - tStart;
  sh.st.n_blocks  = nBlocks;
  sh.st.n_workers = nWorkers;
  if (stats) *stats = sh.st;
  if (lociPerDevice) for (uint32_t c = 0; c < nCtx; ++c) lociPerDevice[c] = sh.lociOfCtx[c];
  if (seq_arena_used) *seq_arena_used = sh.seqUsed.load();
  if (bits_arena_used) *bits_arena_used = sh.bitsUsed.load();
  if (cigar_arena_used) *cigar_arena_used = sh.cigarUsed.load();
  if (sh.fatal != MANTA_OK) return fail(ctx, sh.fatal, sh.msg);
  if (sh.worst != MANTA_OK) return fail(ctx, sh.worst, sh.msg);
  return MANTA_OK;
}
}  // namespace

int manta_smallsv_batch(
    manta_ctx_t* ctx, const manta_asm_options_t* opt, const manta_align_scores_t* scores, int32_t large_indel_score, uint32_t n_loci,
    const uint8_t* bases, const uint64_t* read_off, const uint32_t* locus_read_begin, const uint8_t* refs, const uint64_t* ref_off,
    const manta_ref_cuts_t* cuts, const uint32_t* locus_min_word_length, const uint32_t* locus_max_word_length,
    manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs, manta_smallsv_alignment_t* alignments, uint64_t contigs_cap,
    uint8_t* seq_arena, uint64_t seq_arena_cap, uint64_t* seq_arena_used, uint64_t* bits_arena, uint64_t bits_arena_cap,
    uint64_t* bits_arena_used, uint32_t* cigar_arena, uint64_t cigar_arena_cap, uint64_t* cigar_arena_used,
    const manta_batch_plan_t* plan, manta_batch_stats_t* stats)
{
  return smallsvBatchImpl(&ctx, 1, nullptr, opt, scores, large_indel_score, n_loci, bases, read_off, locus_read_begin, nullptr, refs, ref_off, cuts,
                          locus_min_word_length, locus_max_word_length, loci, contigs, alignments, contigs_cap, seq_arena, seq_arena_cap,
                          seq_arena_used, bits_arena, bits_arena_cap, bits_arena_used, cigar_arena, cigar_arena_cap, cigar_arena_used, plan, stats);
}

int manta_smallsv_batch_piles(
    manta_ctx_t* ctx, const manta_asm_options_t* opt, const manta_align_scores_t* scores, int32_t large_indel_score, uint32_t n_loci,
    const manta_packed_piles_t* piles, const uint8_t* refs, const uint64_t* ref_off, const manta_ref_cuts_t* cuts,
    const uint32_t* locus_min_word_length, const uint32_t* locus_max_word_length, manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs,
    manta_smallsv_alignment_t* alignments, uint64_t contigs_cap, uint8_t* seq_arena, uint64_t seq_arena_cap, uint64_t* seq_arena_used,
    uint64_t* bits_arena, uint64_t bits_arena_cap, uint64_t* bits_arena_used, uint32_t* cigar_arena, uint64_t cigar_arena_cap,
    uint64_t* cigar_arena_used, const manta_batch_plan_t* plan, manta_batch_stats_t* stats)
{
  if (!ctx) return MANTA_E_INVALID_ARG;
  const int rc = checkPiles(ctx, piles, "manta_smallsv_batch_piles");
  if (rc != MANTA_OK) return rc;
  return smallsvBatchImpl(&ctx, 1, nullptr, opt, scores, large_indel_score, n_loci, nullptr, nullptr, piles->locus_read_begin, piles, refs, ref_off, cuts,
                          locus_min_word_length, locus_max_word_length, loci, contigs, alignments, contigs_cap, seq_arena, seq_arena_cap,
                          seq_arena_used, bits_arena, bits_arena_cap, bits_arena_used, cigar_arena, cigar_arena_cap, cigar_arena_used, plan, stats);
}

namespace {
int spanningBatchImpl(
    manta_ctx_t* const* ctxs, const uint32_t nCtx, uint32_t* lociPerDevice, const manta_asm_options_t* opt, const manta_align_scores_t* scores, int32_t jump_score, uint32_t n_loci,
    const uint8_t* bases, const uint64_t* read_off, const uint32_t* locus_read_begin, const manta_packed_piles_t* piles, const uint8_t* refs1, const uint64_t* ref1_off,
    const uint8_t* refs2, const uint64_t* ref2_off, const manta_jump_cuts_t* cuts, const uint32_t* locus_min_word_length,
    const uint32_t* locus_max_word_length, manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs,
    manta_spanning_alignment_t* alignments, uint64_t contigs_cap, uint8_t* seq_arena, uint64_t seq_arena_cap, uint64_t* seq_arena_used,
    uint64_t* bits_arena, uint64_t bits_arena_cap, uint64_t* bits_arena_used, uint32_t* cigar_arena, uint64_t cigar_arena_cap,
    uint64_t* cigar_arena_used, const manta_batch_plan_t* plan, manta_batch_stats_t* stats)
{
  if (!ctxs || nCtx == 0 || !ctxs[0]) return MANTA_E_INVALID_ARG;
  manta_ctx_t* ctx = ctxs[0];  // (call-level errors are reported here)
  if (!opt || !scores || n_loci == 0 || (!piles && (!bases || !read_off)) || !locus_read_begin || !refs1 || !ref1_off || !refs2 || !ref2_off || !cuts ||
      !loci || !contigs || !alignments || !seq_arena || !bits_arena || !cigar_arena)
    return fail(ctx, MANTA_E_INVALID_ARG, "manta_spanning_batch: null argument or empty batch");
  if (scores->is_allow_edge_insertion) return fail(ctx, MANTA_E_INVALID_ARG, "GlobalJumpAligner does not support isAllowEdgeInsertion");
  if ((locus_min_word_length == nullptr) != (locus_max_word_length == nullptr))
    return fail(ctx, MANTA_E_INVALID_ARG, "manta_spanning_batch: per-locus word lengths need both arrays");
  for (uint32_t l = 0; l < n_loci; ++l)
    if (locus_read_begin[l + 1] < locus_read_begin[l] || ref1_off[l + 1] < ref1_off[l] || ref2_off[l + 1] < ref2_off[l])
      return fail(ctx, MANTA_E_INVALID_ARG, "manta_spanning_batch: offsets not monotone");
  const uint64_t totalBases = piles ? 0 : read_off[locus_read_begin[n_loci]] - read_off[locus_read_begin[0]];
  const bool     shared     = (plan && plan->shared_queue) || nCtx > 1;
  const uint32_t blockLoci  = (plan && plan->block_loci) ? plan->block_loci : (shared ? nodeBlockLoci(n_loci, totalBases) : autoBlockLoci(n_loci, totalBases));
  BatchShared    sh;
  sh.serialKernels  = plan && (plan->flags & MANTA_BATCH_SERIAL_KERNELS);
  sh.sharedQueue    = plan ? plan->shared_queue : nullptr;
  sh.gates          = std::vector<StageGates>(nCtx);
  sh.lociOfCtx.assign(nCtx, 0);
  planBlocks(sh, n_loci, blockLoci, piles ? nullptr : read_off, locus_read_begin, piles ? piles->read_len : nullptr);
  const uint32_t nBlocks  = uint32_t(sh.blockOrder.size());
  const uint32_t perCtx   = std::max(1u, (plan && plan->n_workers) ? plan->n_workers : 1u);
  const uint32_t nWorkers = std::max(1u, std::min(nBlocks, perCtx * nCtx));
  sh.pipelineStages       = perCtx > 1 && !std::getenv("MANTA_AMD_NO_STAGE_GATES");  // (experiments: concurrent workers without the stage gates)
  if (sh.sharedQueue)
    for (uint32_t l = 0; l < n_loci; ++l) {
      std::memset(&loci[l], 0, sizeof(loci[l]));
      loci[l].status = MANTA_E_NOT_TAKEN;
    }
  try {
    for (uint32_t c = 0; c < nCtx; ++c) {
      rt::setDevice(ctxs[c]->deviceId);
      while (ctxs[c]->spanPool.size() < (nWorkers + nCtx - 1) / nCtx) ctxs[c]->spanPool.push_back(new manta_spanning(ctxs[c]));
    }
  } catch (const std::exception& e) {
    return fail(ctx, MANTA_E_HIP, e.what());
  }
  const double tStart = nowMs();
  auto         worker = [&](const uint32_t w) {
    manta_ctx_t*    ctx = ctxs[w % nCtx];  // (shadows the call-level context: this worker's device)
    manta_spanning* b   = ctx->spanPool[w / nCtx];
    try {
      rt::setDevice(ctx->deviceId);
      b->opt       = *opt;
      b->scores    = *scores;
      b->jumpScore = jump_score;
      b->streamUploads = !(plan && (plan->flags & MANTA_BATCH_NO_STREAMED_UPLOAD));
      b->asmStage.wavesPerCuCap = sh.pipelineStages ? kPipelinedAsmWavesPerCu : 0;
      std::vector<uint64_t> rOff, f1Off, f2Off;
      std::vector<uint32_t> lBeg;
      while (!sh.stop()) {
        const uint32_t qi = sh.takeNext();
        if (qi >= nBlocks) break;
        const uint32_t blk = sh.blockOrder[qi];
        const uint32_t l0 = blk * blockLoci, l1 = std::min(n_loci, l0 + blockLoci), n = l1 - l0;
        const uint32_t r0 = locus_read_begin[l0], r1 = locus_read_begin[l1];
        if (!piles) rebase(rOff, read_off, r0, size_t(r1 - r0) + 1);
        rebase(lBeg, locus_read_begin, l0, size_t(n) + 1);
        rebase(f1Off, ref1_off, l0, size_t(n) + 1);
        rebase(f2Off, ref2_off, l0, size_t(n) + 1);
        setWordLengths(ctx, b->asmStage, n, locus_min_word_length ? locus_min_word_length + l0 : nullptr,
                       locus_max_word_length ? locus_max_word_length + l0 : nullptr);
        const double t0 = nowMs();
        int          rc;
        if (piles) {
          manta_packed_piles_t pl = *piles;
          pl.read_len            = piles->read_len + r0;
          pl.read_code_off       = piles->read_code_off + r0;
          pl.read_mask_off       = piles->read_mask_off + r0;
          pl.locus_read_begin    = lBeg.data();
          rc = manta_spanning_upload_piles(b, n, &pl, refs1 + ref1_off[l0], f1Off.data(), refs2 + ref2_off[l0], f2Off.data(), cuts + l0);
        } else {
          rc = manta_spanning_upload(b, n, bases + read_off[r0], rOff.data(), lBeg.data(), refs1 + ref1_off[l0], f1Off.data(), refs2 + ref2_off[l0],
                                     f2Off.data(), cuts + l0);
        }
        if (perItemCode(rc)) {  // a locus outside the supported envelope: this block's loci carry the code, the batch goes on
          for (uint32_t l = l0; l < l1; ++l) {
            std::memset(&loci[l], 0, sizeof(loci[l]));
            loci[l].status = rc;
          }
          sh.error(rc, lastErrorOf(ctx), false);
          continue;
        }
        if (rc != MANTA_OK) {
          sh.error(rc, lastErrorOf(ctx), true);
          break;
        }
        const double t1 = nowMs();
        {
          std::unique_lock<std::mutex> only(sh.kernelMu, std::defer_lock);
          if (sh.serialKernels) only.lock();
          b->stageBehindRun = true;
          rc = spanningRunImpl(b, sh.pipelineStages ? &sh.gates[w % nCtx] : nullptr);
        }
        if (rc != MANTA_OK) {
          sh.error(rc, lastErrorOf(ctx), true);
          break;
        }
        const double t2 = nowMs();
        uint64_t     nC = 0, nS = 0, nB = 0, nG = 0;
        {
          rt::ScopedStream onStream(b->main);
          pipeStage(b);
        }
        // compaction in a few locus ranges, one host thread each (as smallsvBatchImpl)
        struct Range {
          uint64_t nC = 0, nS = 0, nB = 0, nG = 0, cells = 0, ptrBytes = 0;
          uint64_t c0 = 0, s0 = 0, b0 = 0, g0 = 0;
          int      rc = MANTA_OK;
        };
        const unsigned     parts = hostParts(uint64_t(n) * 4);
        std::vector<Range> rg(parts);
        hostParallel(n, parts, [&](unsigned t, uint64_t a, uint64_t z) {
          b->asmStage.rangeSizes(PackedContigs<manta_spanning>{b}, uint32_t(a), uint32_t(z), rg[t].nC, rg[t].nS, rg[t].nB);
          rg[t].nG = spanningCigarWords(b, uint32_t(a), uint32_t(z));
        });
        for (unsigned t = 0; t < parts; ++t) {
          rg[t].c0 = nC, rg[t].s0 = nS, rg[t].b0 = nB, rg[t].g0 = nG;
          nC += rg[t].nC, nS += rg[t].nS, nB += rg[t].nB, nG += rg[t].nG;
        }
        const uint64_t cBase = sh.contigsUsed.fetch_add(nC), sBase = sh.seqUsed.fetch_add(nS), bBase = sh.bitsUsed.fetch_add(nB),
                       gBase = sh.cigarUsed.fetch_add(nG);
        if (cBase + nC > contigs_cap || sBase + nS > seq_arena_cap || bBase + nB > bits_arena_cap || gBase + nG > cigar_arena_cap) {
          sh.error(MANTA_E_CAPACITY, "manta_spanning_batch: caller arenas too small", true);
          break;
        }
        hostParallel(n, parts, [&](unsigned t, uint64_t a, uint64_t z) {
          const Range& r(rg[t]);
          rg[t].rc = spanningCompact(b, loci + l0, contigs, alignments, cBase + r.c0, r.nC, seq_arena + sBase + r.s0, r.nS, sBase + r.s0, nullptr,
                                     bits_arena + bBase + r.b0, r.nB, bBase + r.b0, nullptr, cigar_arena + gBase + r.g0, r.nG, gBase + r.g0,
                                     nullptr, uint32_t(a), uint32_t(z), &rg[t].cells, &rg[t].ptrBytes);
        });
        rc = MANTA_OK;
        b->stats.dp_cells = b->stats.ptr_matrix_bytes = 0;
        for (const Range& r : rg) {
          if (r.rc != MANTA_OK && (rc == MANTA_OK || !perItemCode(r.rc))) rc = r.rc;
          b->stats.dp_cells += r.cells;
          b->stats.ptr_matrix_bytes += r.ptrBytes;
        }
        const double t3 = nowMs();
        if (rc != MANTA_OK) {
          sh.error(rc, lastErrorOf(ctx), !perItemCode(rc));
          if (!perItemCode(rc)) break;
        }
        std::lock_guard<std::mutex> g(sh.mu);
        sh.lociOfCtx[w % nCtx] += n;
        sh.st.h2d_ms += t1 - t0;
        sh.st.kernel_ms += t2 - t1;
        sh.st.d2h_ms += t3 - t2;
        sh.st.assemble_ms += b->stats.assemble_ms;
        sh.st.schedule_ms += b->stats.schedule_ms;
        sh.st.align_ms += b->stats.align_ms;
        sh.st.n_alignments += b->stats.n_alignments;
        sh.st.n_align_launches += b->stats.n_align_launches;
        sh.st.dp_cells += b->stats.dp_cells;
        sh.st.ptr_matrix_bytes += b->stats.ptr_matrix_bytes;
        sh.st.n_loci_lds_small += b->asmStage.fastIds.size();
        sh.st.n_loci_lds_big += b->asmStage.bigIds.size();
        sh.st.n_loci_handed_back += b->asmStage.ldsFallbacks;
        sh.st.n_loci_general += b->asmStage.useFast ? b->asmStage.genIds.size() : size_t(n);
        sh.st.h2d_bytes += (piles ? b->asmStage.plBytes : (read_off[r1] - read_off[r0]) + 8ull * (r1 - r0 + 1)) + (ref1_off[l1] - ref1_off[l0]) + (ref2_off[l1] - ref2_off[l0]) +
                           20ull * (n + 1) + 16ull * n;
        sh.st.d2h_bytes += pipeStagedBytes(b);
      }
    } catch (const std::exception& e) {
      sh.error(MANTA_E_HIP, e.what(), true);
    }
  };
  std::vector<std::thread> threads;
  for (uint32_t w = 1; w < nWorkers; ++w) threads.emplace_back(worker, w);
  worker(0);
  for (std::thread& t : threads) t.join();
  sh.st.wall_ms   = nowMs() - tStart;
  sh.st.n_blocks  = nBlocks;
  sh.st.n_workers = nWorkers;
  if (stats) *stats = sh.st;
  if (lociPerDevice) for (uint32_t c = 0; c < nCtx; ++c) lociPerDevice[c] = sh.lociOfCtx[c];
  if (seq_arena_used) *seq_arena_used = sh.seqUsed.load();
  if (bits_arena_used) *bits_arena_used = sh.bitsUsed.load();
  if (cigar_arena_used) *cigar_arena_used = sh.cigarUsed.load();
  if (sh.fatal != MANTA_OK) return fail(ctx, sh.fatal, sh.msg);
  if (sh.worst != MANTA_OK) return fail(ctx, sh.worst, sh.msg);
  return MANTA_OK;
}
}  // namespace

int manta_spanning_batch(
    manta_ctx_t* ctx, const manta_asm_options_t* opt, const manta_align_scores_t* scores, int32_t jump_score, uint32_t n_loci,
    const uint8_t* bases, const uint64_t* read_off, const uint32_t* locus_read_begin, const uint8_t* refs1, const uint64_t* ref1_off,
    const uint8_t* refs2, const uint64_t* ref2_off, const manta_jump_cuts_t* cuts, const uint32_t* locus_min_word_length,
    const uint32_t* locus_max_word_length, manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs,
    manta_spanning_alignment_t* alignments, uint64_t contigs_cap, uint8_t* seq_arena, uint64_t seq_arena_cap, uint64_t* seq_arena_used,
    uint64_t* bits_arena, uint64_t bits_arena_cap, uint64_t* bits_arena_used, uint32_t* cigar_arena, uint64_t cigar_arena_cap,
    uint64_t* cigar_arena_used, const manta_batch_plan_t* plan, manta_batch_stats_t* stats)
{
  return spanningBatchImpl(&ctx, 1, nullptr, opt, scores, jump_score, n_loci, bases, read_off, locus_read_begin, nullptr, refs1, ref1_off, refs2, ref2_off,
                           cuts, locus_min_word_length, locus_max_word_length, loci, contigs, alignments, contigs_cap, seq_arena, seq_arena_cap,
                           seq_arena_used, bits_arena, bits_arena_cap, bits_arena_used, cigar_arena, cigar_arena_cap, cigar_arena_used, plan, stats);
}

/* the same with the read piles in packed form (what manta_read_piles_batch emits) */
int manta_spanning_batch_piles(
    manta_ctx_t* ctx, const manta_asm_options_t* opt, const manta_align_scores_t* scores, int32_t jump_score, uint32_t n_loci,
    const manta_packed_piles_t* piles, const uint8_t* refs1, const uint64_t* ref1_off, const uint8_t* refs2, const uint64_t* ref2_off,
    const manta_jump_cuts_t* cuts, const uint32_t* locus_min_word_length, const uint32_t* locus_max_word_length, manta_asm_locus_result_t* loci,
    manta_asm_contig_t* contigs, manta_spanning_alignment_t* alignments, uint64_t contigs_cap, uint8_t* seq_arena, uint64_t seq_arena_cap,
    uint64_t* seq_arena_used, uint64_t* bits_arena, uint64_t bits_arena_cap, uint64_t* bits_arena_used, uint32_t* cigar_arena,
    uint64_t cigar_arena_cap, uint64_t* cigar_arena_used, const manta_batch_plan_t* plan, manta_batch_stats_t* stats)
{
  if (!ctx) return MANTA_E_INVALID_ARG;
  const int rc = checkPiles(ctx, piles, "manta_spanning_batch_piles");
  if (rc != MANTA_OK) return rc;
  return spanningBatchImpl(&ctx, 1, nullptr, opt, scores, jump_score, n_loci, nullptr, nullptr, piles->locus_read_begin, piles, refs1, ref1_off, refs2,
                           ref2_off, cuts, locus_min_word_length, locus_max_word_length, loci, contigs, alignments, contigs_cap, seq_arena,
                           seq_arena_cap, seq_arena_used, bits_arena, bits_arena_cap, bits_arena_used, cigar_arena, cigar_arena_cap, cigar_arena_used,
                           plan, stats);
}

/* ------------------------------------------------------------------------------------------------------
 * manta_node_*: the GPUs of one node behind one block queue (include/manta_amd.h)
 * ---------------------------------------------------------------------------------------------------- */
struct manta_node {
  std::vector<manta_ctx_t*> ctxs;
  ~manta_node()
  {
    for (manta_ctx_t* c : ctxs) manta_ctx_destroy(c);
  }
};

int manta_node_create(const int32_t* device_ids, uint32_t n_devices, manta_node_t** out)
{
  if (!out || !device_ids || n_devices == 0) {
    g_createError = "manta_node_create: null argument or no device";
    return MANTA_E_INVALID_ARG;
  }
  *out = nullptr;
  std::unique_ptr<manta_node> node(new manta_node);
  for (uint32_t d = 0; d < n_devices; ++d) {
    manta_ctx_t* c  = nullptr;
    const int    rc = manta_ctx_create(device_ids[d], &c);
    if (rc != MANTA_OK) return rc;  // (the contexts created so far go with `node`)
    node->ctxs.push_back(c);
  }
  *out = node.release();
  return MANTA_OK;
}

void manta_node_destroy(manta_node_t* node) { delete node; }

uint32_t manta_node_device_count(const manta_node_t* node) { return node ? uint32_t(node->ctxs.size()) : 0u; }

const char* manta_node_last_error(const manta_node_t* node)
{
  return (node && !node->ctxs.empty()) ? manta_last_error(node->ctxs[0]) : g_createError.c_str();
}

int manta_node_smallsv_batch(
    manta_node_t* node, const manta_asm_options_t* opt, const manta_align_scores_t* scores, int32_t large_indel_score, uint32_t n_loci,
    const uint8_t* bases, const uint64_t* read_off, const uint32_t* locus_read_begin, const uint8_t* refs, const uint64_t* ref_off,
    const manta_ref_cuts_t* cuts, const uint32_t* locus_min_word_length, const uint32_t* locus_max_word_length,
    manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs, manta_smallsv_alignment_t* alignments, uint64_t contigs_cap,
    uint8_t* seq_arena, uint64_t seq_arena_cap, uint64_t* seq_arena_used, uint64_t* bits_arena, uint64_t bits_arena_cap,
    uint64_t* bits_arena_used, uint32_t* cigar_arena, uint64_t cigar_arena_cap, uint64_t* cigar_arena_used,
    const manta_batch_plan_t* plan, manta_batch_stats_t* stats, uint32_t* loci_per_device)
{
  if (!node) return MANTA_E_INVALID_ARG;
  return smallsvBatchImpl(node->ctxs.data(), uint32_t(node->ctxs.size()), loci_per_device, opt, scores, large_indel_score, n_loci, bases, read_off,
                          locus_read_begin, nullptr, refs, ref_off, cuts, locus_min_word_length, locus_max_word_length, loci, contigs, alignments,
                          contigs_cap, seq_arena, seq_arena_cap, seq_arena_used, bits_arena, bits_arena_cap, bits_arena_used, cigar_arena,
                          cigar_arena_cap, cigar_arena_used, plan, stats);
}

int manta_node_spanning_batch(
    manta_node_t* node, const manta_asm_options_t* opt, const manta_align_scores_t* scores, int32_t jump_score, uint32_t n_loci,
    const uint8_t* bases, const uint64_t* read_off, const uint32_t* locus_read_begin, const uint8_t* refs1, const uint64_t* ref1_off,
    const uint8_t* refs2, const uint64_t* ref2_off, const manta_jump_cuts_t* cuts, const uint32_t* locus_min_word_length,
    const uint32_t* locus_max_word_length, manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs,
    manta_spanning_alignment_t* alignments, uint64_t contigs_cap, uint8_t* seq_arena, uint64_t seq_arena_cap, uint64_t* seq_arena_used,
    uint64_t* bits_arena, uint64_t bits_arena_cap, uint64_t* bits_arena_used, uint32_t* cigar_arena, uint64_t cigar_arena_cap,
    uint64_t* cigar_arena_used, const manta_batch_plan_t* plan, manta_batch_stats_t* stats, uint32_t* loci_per_device)
{
  if (!node) return MANTA_E_INVALID_ARG;
  return spanningBatchImpl(node->ctxs.data(), uint32_t(node->ctxs.size()), loci_per_device, opt, scores, jump_score, n_loci, bases, read_off,
                           locus_read_begin, nullptr, refs1, ref1_off, refs2, ref2_off, cuts, locus_min_word_length, locus_max_word_length, loci, contigs,
                           alignments, contigs_cap, seq_arena, seq_arena_cap, seq_arena_used, bits_arena, bits_arena_cap, bits_arena_used,
                           cigar_arena, cigar_arena_cap, cigar_arena_used, plan, stats);
}

}  // extern "C"


// ------------------------------------------------------------------------------------------------------
// split-read scoring
// ------------------------------------------------------------------------------------------------------
extern "C" int manta_split_read_batch(
    manta_ctx_t* ctx, const double* ln_comp_error_prob, const double* ln_error_prob, uint32_t n_qscores, float ln_one_third,
    float ln_random_base, uint32_t n_tasks, const manta_split_task_t* tasks, const uint8_t* arena, uint64_t arena_bytes,
    manta_split_result_t* results)
{
  if (!ctx) return MANTA_E_INVALID_ARG;
  if (!ln_comp_error_prob || !ln_error_prob || n_qscores == 0 || (n_tasks && (!tasks || !arena || !results)))
    return fail(ctx, MANTA_E_INVALID_ARG, "manta_split_read_batch: null argument");
  if (n_tasks == 0) return MANTA_OK;
  try {
    rt::setDevice(ctx->deviceId);
    rt::ScopedStream onStream(ctx->stream);
    std::vector<SplitTaskDev> dev(n_tasks);
    uint8_t*                  dArena = ctx->dSeq.as<uint8_t>(arena_bytes + 16);
    auto outside = [&](uint64_t off, uint64_t len) { return off > arena_bytes || len > arena_bytes - off; };
    for (uint32_t i = 0; i < n_tasks; ++i) {
      const manta_split_task_t& t(tasks[i]);
      if (outside(t.query_off, t.query_len) || outside(t.qual_off, t.query_len) || outside(t.target_off, t.target_len))
        return fail(ctx, MANTA_E_INVALID_ARG, "manta_split_read_batch: task " + std::to_string(i) + " outside the arena");
      SplitTaskDev& d(dev[i]);
      d.query            = dArena + t.query_off;
      d.qual             = dArena + t.qual_off;
      d.target           = dArena + t.target_off;
      d.query_len        = t.query_len;
      d.target_len       = t.target_len;
      d.bp_begin         = t.bp_begin;
      d.bp_end           = t.bp_end;
      d.flank_score_size = t.flank_score_size;
      d.reserved         = 0;
    }
    SplitTaskDev*   dTasks = ctx->dSplitTasks.as<SplitTaskDev>(n_tasks);
    SplitResultDev* dRes   = ctx->dSplitResults.as<SplitResultDev>(n_tasks);
    double*         dTab   = ctx->dSplitTables.as<double>(2ull * n_qscores + 2);
    uint32_t*       dCount = ctx->dCounter.as<uint32_t>(kNumESet);
    rt::h2d(dArena, arena, arena_bytes);
    rt::h2d(dTasks, dev.data(), sizeof(SplitTaskDev) * n_tasks);
    rt::h2d(dTab, ln_comp_error_prob, sizeof(double) * n_qscores);
    rt::h2d(dTab + n_qscores, ln_error_prob, sizeof(double) * n_qscores);
    rt::dzero(dCount, sizeof(uint32_t));
    SplitParams P;
    P.tasks          = dTasks;
    P.results        = dRes;
    P.n_tasks        = n_tasks;
    P.n_q            = n_qscores;
    P.ln_comp_error  = dTab;
    P.ln_error       = dTab + n_qscores;
    P.ln_one_third   = ln_one_third;
    P.ln_random_base = ln_random_base;
    P.counter        = dCount;
    const int grid = rt::roundGrid(int(std::min<uint64_t>(n_tasks, uint64_t(std::max(1, ctx->cuCount * 32)))));
    rt::launch(split_read_kernel, grid, 0, P);
    std::vector<SplitResultDev> h(n_tasks);
    rt::d2h(h.data(), dRes, sizeof(SplitResultDev) * n_tasks);
    int worst = MANTA_OK;
    for (uint32_t i = 0; i < n_tasks; ++i) {
      manta_split_result_t& r(results[i]);
      std::memset(&r, 0, sizeof(r));
      const SplitResultDev& d(h[i]);
      r.status = (d.status == 0) ? MANTA_OK : (d.status == 3) ? MANTA_E_UNSUPPORTED : MANTA_E_INVALID_ARG;
      if (d.status == 1) r.status = MANTA_E_SPLIT_QUERY_NOT_SHORTER;
      if (d.status == 2) r.status = MANTA_E_SPLIT_EMPTY_SCAN;
      if (r.status != MANTA_OK) {
        worst = r.status;
        continue;
      }
      r.best_pos         = d.best_pos;
      r.best_ln_lhood    = d.best_ln_lhood;
      r.left_size        = d.left_size;
      r.hom_size         = d.hom_size;
      r.right_size       = d.right_size;
      r.left_mismatches  = d.left_mismatches;
      r.hom_mismatches   = d.hom_mismatches;
      r.right_mismatches = d.right_mismatches;
    }
    if (worst != MANTA_OK) return fail(ctx, worst, "manta_split_read_batch: one or more tasks failed; see per-task status");
    return MANTA_OK;
  } catch (const std::exception& e) {
    return fail(ctx, MANTA_E_HIP, e.what());
  }
}

extern "C" void manta_read_search_range(int32_t bp_begin, int32_t bp_end, int32_t* search_begin, int32_t* search_end)
{
  manta_read_scan_t sc;
  std::memset(&sc, 0, sizeof(sc));
  sc.bp_begin = bp_begin;
  sc.bp_end   = bp_end;
  int sb, se;
  ReadClass::searchRange(sc, sb, se);
  if (search_begin) *search_begin = sb;
  if (search_end) *search_end = se;
}

extern "C" int manta_read_piles_batch(
    manta_ctx_t* ctx, const manta_read_class_options_t* opt, uint32_t n_loci, const manta_read_locus_t* loci, uint32_t n_scans,
    const manta_read_scan_t* scans, uint32_t n_reads, const manta_bam_read_t* reads, const uint32_t* cigars, uint64_t n_cigar_words,
    const uint8_t* names, uint64_t names_bytes, const uint8_t* seqs, uint64_t seqs_bytes, const uint8_t* quals, uint64_t quals_bytes,
    const uint8_t* refs, uint64_t refs_bytes, uint8_t* decision, uint32_t* pile_index, manta_read_locus_result_t* results,
    uint32_t* codes, uint64_t codes_cap, uint64_t* codes_used, uint32_t* nmask, uint64_t mask_cap, uint64_t* mask_used,
    uint32_t* read_len, uint64_t* read_code_off, uint64_t* read_mask_off, uint32_t* pile_read, uint64_t reads_cap,
    uint64_t* reads_used, uint32_t* locus_read_begin)
{
  static const char* fn = "manta_read_piles_batch: ";
  if (!ctx) return MANTA_E_INVALID_ARG;
  // (the two offset arrays take their "one past the last read" entry whenever there is a candidate, also with no read at all)
  if (!opt || (n_loci && (!loci || !results || !locus_read_begin || !read_code_off || !read_mask_off)) || (n_scans && !scans) ||
      (n_reads && (!reads || !decision || !pile_index)) || (n_cigar_words && !cigars) || (names_bytes && !names) || (seqs_bytes && !seqs) ||
      (quals_bytes && !quals))
    return fail(ctx, MANTA_E_INVALID_ARG, std::string(fn) + "null argument");
  if (codes_used) *codes_used = 0;
  if (mask_used) *mask_used = 0;
  if (reads_used) *reads_used = 0;
  if (n_loci == 0) return MANTA_OK;
  // what the kernels index with must lie inside what was handed over
  uint64_t maxRange = 1, maxRecords = 1, codeBound = 0, maskBound = 0;
  std::vector<uint32_t> chunks;  // read_test_kernel's work list: 64 records of one query each (scan, first record, candidate)
  chunks.reserve(3 * (size_t(n_reads) / 64 + n_scans + 1));
  for (uint32_t l = 0; l < n_loci; ++l) {
    if (loci[l].scan_begin > loci[l].scan_end || loci[l].scan_end > n_scans)
      return fail(ctx, MANTA_E_INVALID_ARG, std::string(fn) + "candidate " + std::to_string(l) + ": scans outside the scan array");
    uint64_t recs = 0;
    for (uint32_t s = loci[l].scan_begin; s < loci[l].scan_end; ++s) {
      const manta_read_scan_t& sc(scans[s]);
      if (sc.read_begin > sc.read_end || sc.read_end > n_reads || sc.ref_off > refs_bytes || sc.ref_len > refs_bytes - sc.ref_off ||
          (sc.ref_len && !refs) || sc.bam_index >= (1u << 22))
        return fail(ctx, MANTA_E_INVALID_ARG, std::string(fn) + "scan " + std::to_string(s) + " outside the arrays");
      int sb, se;
      ReadClass::searchRange(sc, sb, se);
      maxRange = std::max<uint64_t>(maxRange, uint64_t(se > sb ? se - sb : 0) + 2);
      recs += sc.read_end - sc.read_begin;
      for (uint32_t base = sc.read_begin; base < sc.read_end; base += 64) {
        chunks.push_back(s);
        chunks.push_back(base);
        chunks.push_back(l);
      }
    }
    maxRecords = std::max(maxRecords, recs);
  }
  if (maxRange > (1ull << 26)) return fail(ctx, MANTA_E_UNSUPPORTED, std::string(fn) + "a breakend interval beyond 64 M bases");
  for (uint32_t i = 0; i < n_reads; ++i) {
    const manta_bam_read_t& r(reads[i]);
    const bool bad = r.cigar_off > n_cigar_words || r.n_cigar > n_cigar_words - r.cigar_off ||
                     ((r.tags & MANTA_READ_TAG_MC) && (r.mate_cigar_off > n_cigar_words || r.n_mate_cigar > n_cigar_words - r.mate_cigar_off)) ||
                     r.qname_off > names_bytes || r.qname_len > names_bytes - r.qname_off || r.seq_off > seqs_bytes ||
                     (uint64_t(r.read_len) + 1) / 2 > seqs_bytes - r.seq_off || r.qual_off > quals_bytes || r.read_len > quals_bytes - r.qual_off;
    if (bad) return fail(ctx, MANTA_E_INVALID_ARG, std::string(fn) + "record " + std::to_string(i) + " outside the arenas");
    codeBound += (uint64_t(r.read_len) + 15) / 16;
    maskBound += (uint64_t(r.read_len) + 31) / 32;
  }
  try {
    rt::setDevice(ctx->deviceId);
    rt::ScopedStream onStream(ctx->stream);
    uint64_t tableCap = 64;
    while (tableCap < 2 * maxRecords) tableCap *= 2;
    const uint64_t stride = 2 * maxRange + 1 + tableCap;
    // one workspace per wave, sized for the widest breakend interval of the batch: a multi-megabase interval shrinks the grid instead
    // of asking for (waves x that interval) bytes (the reference allocates searchRange.size() counters for the one candidate)
    const uint64_t wsFit  = std::max<uint64_t>(1, workspaceBudget(size_t(16) << 30) / (stride * 4));
    const int      grid   = rt::roundGrid(int(std::min<uint64_t>(std::min<uint64_t>(n_loci, wsFit), uint64_t(std::max(1, ctx->cuCount * 8)))));
    ReadClassParams P;
    std::memset(&P, 0, sizeof(P));
    P.opt    = *opt;
    P.n_loci = n_loci;
    auto up  = [&](DevBuf& b, const void* src, uint64_t bytes) {
      uint8_t* d = b.as<uint8_t>(bytes + 16);
      rt::h2d(d, src, bytes);
      return d;
    };
    P.loci   = reinterpret_cast<const manta_read_locus_t*>(up(ctx->dRc[0], loci, sizeof(manta_read_locus_t) * uint64_t(n_loci)));
    P.scans  = reinterpret_cast<const manta_read_scan_t*>(up(ctx->dRc[1], scans, sizeof(manta_read_scan_t) * uint64_t(n_scans)));
    P.reads  = reinterpret_cast<const manta_bam_read_t*>(up(ctx->dRc[2], reads, sizeof(manta_bam_read_t) * uint64_t(n_reads)));
    P.cigars = reinterpret_cast<const uint32_t*>(up(ctx->dRc[3], cigars, 4 * n_cigar_words));
    P.names  = up(ctx->dRc[4], names, names_bytes);
    P.seqs   = up(ctx->dRc[5], seqs, seqs_bytes);
    P.quals  = up(ctx->dRc[6], quals, quals_bytes);
    P.refs   = up(ctx->dRc[7], refs, refs_bytes);
    // outputs and workspace in one allocation each
    uint8_t* dOut = ctx->dRc[8].as<uint8_t>(uint64_t(n_reads) * 13 + uint64_t(n_loci) * (sizeof(manta_read_locus_result_t) + 16 + 32) + 256 +
                                            4 * (uint64_t(n_loci) + 1));
    P.pile_index   = reinterpret_cast<uint32_t*>(dOut);
    P.tmp          = P.pile_index + n_reads;
    P.results      = reinterpret_cast<manta_read_locus_result_t*>(P.tmp + n_reads);
    P.locus_counts = reinterpret_cast<uint32_t*>(P.results + n_loci);
    P.locus_base   = reinterpret_cast<unsigned long long*>(P.locus_counts + 4 * uint64_t(n_loci));
    P.locus_read_begin = reinterpret_cast<uint32_t*>(P.locus_base + 4 * (uint64_t(n_loci) + 1));
    P.counter      = P.locus_read_begin + n_loci + 1;  // four queue heads
    P.decision     = reinterpret_cast<uint8_t*>(P.counter + 4);
    P.pre          = P.decision + n_reads;
    P.chunks       = reinterpret_cast<const uint32_t*>(up(ctx->dRc[14], chunks.data(), 4 * chunks.size()));
    P.n_chunks     = uint32_t(chunks.size() / 3);
    P.ws           = ctx->dRc[9].as<uint32_t>(stride * uint64_t(grid));
    P.ws_stride    = stride;
    P.range_cap    = uint32_t(maxRange);
    P.table_cap    = uint32_t(tableCap);
    P.codes        = ctx->dRc[10].as<uint32_t>(codeBound + 4);
    P.nmask        = ctx->dRc[11].as<uint32_t>(maskBound + 4);
    P.read_len     = ctx->dRc[12].as<uint32_t>(2 * uint64_t(n_reads) + 4);
    P.pile_read    = P.read_len + n_reads + 1;
    P.read_code_off = ctx->dRc[13].as<unsigned long long>(2 * (uint64_t(n_reads) + 2));
    P.read_mask_off = P.read_code_off + n_reads + 1;
    rt::dzero(P.counter, 4 * sizeof(uint32_t));
    const int wide    = rt::roundGrid(std::max(1, ctx->cuCount * 16));
    auto      gridFor = [&](uint64_t items) { return std::min(wide, rt::roundGrid(int(std::max<uint64_t>(1, std::min<uint64_t>(items, 1u << 20))))); };
    rt::launch(read_test_kernel, gridFor(P.n_chunks), 0, P);
    rt::launch(read_class_kernel, grid, 0, P);
    rt::launch(read_pile_offsets_kernel, rt::roundGrid(1), 0, P);
    rt::launch(read_pile_pack_kernel, grid, 0, P);
    rt::launch(read_pile_bases_kernel, gridFor(uint64_t(n_reads) / 8 + 1), 0, P);
    unsigned long long totals[4] = {0, 0, 0, 0};
    rt::d2h(totals, P.locus_base + 4 * uint64_t(n_loci), sizeof(totals));  // (synchronizes)
    if (reads_used) *reads_used = totals[0];
    if (codes_used) *codes_used = totals[1];
    if (mask_used) *mask_used = totals[2];
    rt::d2hAsync(decision, P.decision, n_reads);
    rt::d2hAsync(pile_index, P.pile_index, 4 * uint64_t(n_reads));
    rt::d2hAsync(results, P.results, sizeof(manta_read_locus_result_t) * uint64_t(n_loci));
    rt::d2hAsync(locus_read_begin, P.locus_read_begin, 4 * (uint64_t(n_loci) + 1));
    const bool fits = totals[0] <= reads_cap && totals[1] <= codes_cap && totals[2] <= mask_cap &&
                      (totals[0] == 0 || (codes && nmask && read_len && pile_read));
    if (fits) {
      rt::d2hAsync(codes, P.codes, 4 * totals[1]);
      rt::d2hAsync(nmask, P.nmask, 4 * totals[2]);
      rt::d2hAsync(read_len, P.read_len, 4 * totals[0]);
      rt::d2hAsync(pile_read, P.pile_read, 4 * totals[0]);
      rt::d2hAsync(read_code_off, P.read_code_off, 8 * (totals[0] + 1));
      rt::d2hAsync(read_mask_off, P.read_mask_off, 8 * (totals[0] + 1));
    }
    rt::sync();
    if (!fits) return fail(ctx, MANTA_E_CAPACITY, std::string(fn) + "pile arrays too small (see *_used)");
    int worst = MANTA_OK;
    for (uint32_t l = 0; l < n_loci; ++l)
      if (results[l].status != MANTA_OK) worst = results[l].status;
    if (worst != MANTA_OK) return fail(ctx, worst, std::string(fn) + "one or more candidates failed; see per-candidate status");
    return MANTA_OK;
  } catch (const std::exception& e) {
    return fail(ctx, MANTA_E_HIP, e.what());
  }
}

#ifdef MANTA_WAVE_EMU
/// tests/emu only: speculation statistics of contig_kernel since the last call (loci done by it, walk rounds, walks,
/// accepted candidates, cache evictions)
extern "C" void manta_emu_fast_stats(unsigned long long* out)
{
  unsigned long long* v = manta_dev::fastStats();
  for (int i = 0; i < 8; ++i) {
    out[i] = v[i];
    v[i]   = 0;
  }
}
#endif
