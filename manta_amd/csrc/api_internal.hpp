// Internals shared by the host translation units of the C-ABI implementation (api.cpp: contexts, the single calls and the staged pipelines;
// api_batch.cpp: the whole-batch calls, the block queue and the node queue; api_reads.cpp: split-read scoring and read gathering).  Device
// buffers, the context, the assembler stage (AsmStage: plan / upload / launch / staging / compaction), the two pipeline objects and what a
// whole-batch worker needs of them.  Everything here is inline or a template: the product compiles the three sources separately
// (manta_amd/build.py, MANTA_TU_HOST), the wave-emulator build of tests/emu compiles them as one unit (api_unity.cpp).
#pragma once
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

namespace manta_host {

inline thread_local std::string g_createError;

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

inline double nowMs()
{
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

/// live contexts/pipelines of this process share the device's free memory (per-wave workspaces are sized from it)
inline std::atomic<int> g_liveWorkspaces{0};

}  // namespace manta_host
using namespace manta_host;

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

namespace manta_host {

inline int fail(manta_ctx_t* ctx, int code, const std::string& msg)
{
  if (ctx) {
    std::lock_guard<std::mutex> g(ctx->errMu);
    ctx->lastError = msg;
  }
  return code;
}

/// failures of single loci / alignments: reported in the per-item status, never fatal for the batch
inline bool perItemCode(int rc)
{
  return rc == MANTA_E_UNSUPPORTED || rc == MANTA_E_DEVICE_FAULT || rc == MANTA_E_EMPTY_SEQ;
}

inline std::string lastErrorOf(manta_ctx_t* ctx)
{
  std::lock_guard<std::mutex> g(ctx->errMu);
  return ctx->lastError;
}

const int kESet[]  = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 24, 32};
const int kNumESet = sizeof(kESet) / sizeof(kESet[0]);

inline int pickE(uint32_t qlen)
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
inline bool alignUsesPairs(int kind, int eIdx, int match, int mismatch, int open, int extend, int offEdge, int extra, int allowEdgeIns, uint64_t maxRef)
{
  static const bool off = std::getenv("MANTA_AMD_NO_ALIGN_PAIRS") != nullptr;  // A/B knob
  static const bool offJump = std::getenv("MANTA_AMD_NO_JUMP_PAIRS") != nullptr;
  if (off || maxRef > 0xfffeu) return false;
  if (kind == MANTA_ALIGNER_LARGE_INDEL) return pairEligible(kESet[eIdx], match, mismatch, open, extend, offEdge, extra, allowEdgeIns);
  // GlobalJumpAligner: align_jump_pair.hpp (maxRef: both references together -- the combined rows of a task)
  if (kind == MANTA_ALIGNER_JUMP) return !offJump && !allowEdgeIns && jumpPairEligible(kESet[eIdx], match, mismatch, open, extend, offEdge, extra);
  return false;
}

inline void launchJumpPair(int eIdx, int grid, const AlignParams& P)
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

inline void launchAlignPair(int eIdx, int grid, const AlignParams& P)
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
inline void launchAlignKind(int kind, int eIdx, int grid, const AlignParams& P, bool pair = false)
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

inline int alignWavesPerCu()
{
  return std::getenv("MANTA_AMD_ALIGN_WAVES_PER_CU") ? std::atoi(std::getenv("MANTA_AMD_ALIGN_WAVES_PER_CU")) : 16;
}

/// bucket_count() transitions of the libstdc++ this library is linked against, recorded from a live
/// std::unordered_map (the reference's repeat search iterates such maps: assembly/IterativeAssembler.cpp:630-641)
inline void recordGrowthSchedule(std::vector<uint32_t>& sizes, std::vector<uint32_t>& buckets, const uint32_t upTo)
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
inline bool stringHashMatchesLibstdcxx()
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
inline std::atomic<int> g_livePerDevice[kMaxDevices];
inline size_t workspaceBudget(const size_t capBytes)
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
/// plan()'s inner loops over one locus' reads: dwords of 2-bit codes (16 bases each) and the longest read (all ones: a negative step /
/// a read of 4 G bases).  The second form is the first compiled for AVX2 -- the host translation units are built for baseline x86-64,
/// where the loop stays scalar (~1.1 ns per read; the metric's batch has 800 k) -- and is taken where the CPU has it.
#define MANTA_SCAN_OFFSETS_BODY                                                      \
  uint64_t w = 0, hi = 0;                                                            \
  uint32_t m = 0;                                                                    \
  for (uint32_t i = 0; i < n; ++i) {                                                 \
    const uint64_t len = o[i + 1] - o[i];                                            \
    w += (len + 15) >> 4;                                                            \
    hi |= len >> 32;                                                                 \
    const uint32_t l32 = uint32_t(len);                                              \
    m                  = m > l32 ? m : l32;                                          \
  }                                                                                  \
  wOut       = w;                                                                    \
  longestOut = hi ? ~uint64_t(0) : uint64_t(m);
inline void scanOffsetsBase(const uint64_t* o, const uint32_t n, uint64_t& wOut, uint64_t& longestOut) { MANTA_SCAN_OFFSETS_BODY }
#if defined(__x86_64__) && !defined(MANTA_NO_AVX2_SCAN)
__attribute__((target("avx2"))) inline void scanOffsetsAvx2(const uint64_t* o, const uint32_t n, uint64_t& wOut, uint64_t& longestOut) { MANTA_SCAN_OFFSETS_BODY }
inline bool hostHasAvx2()
{
  static const bool has = __builtin_cpu_supports("avx2") != 0;
  return has;
}
#else
inline void scanOffsetsAvx2(const uint64_t* o, const uint32_t n, uint64_t& wOut, uint64_t& longestOut) { MANTA_SCAN_OFFSETS_BODY }
inline bool hostHasAvx2() { return false; }
#endif
#undef MANTA_SCAN_OFFSETS_BODY

static const unsigned kHostPartsMax = 8;
inline unsigned hostParts(const uint64_t n)
{
  static const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  if (const char* forced = std::getenv("MANTA_AMD_HOST_PARTS"))  // tests: take the multi-range paths on small batches too
    return unsigned(std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(n, kHostPartsMax), uint64_t(std::max(1, std::atoi(forced))))));
  if (n < 4096) return 1;
  return std::min(n < 262144 ? 4u : kHostPartsMax, hw);  // (eight for the passes over every read of a large batch)
}
/// seven helper threads per process, parked on a condition variable between jobs (starting std::threads per call costs more
/// than the loops they would share on a 256-core host)
class HostPool {
 public:
  static HostPool& get()
  {
    static HostPool pool;
    return pool;
  }
  /// fn(part, begin, end) for `parts` (<= kHostPartsMax) contiguous ranges of [0, n); part 0 runs on the caller's thread.  One job at a
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
    for (unsigned i = 0; i + 1 < kHostPartsMax; ++i) threads.emplace_back([this, i] { loop(i + 1); });
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

inline uint32_t nextPow2(uint64_t v)
{
  uint64_t p = 1;
  while (p < v) p <<= 1;
  return uint32_t(p);
}

inline int asmStatusToAbi(int st)
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
      const bool        avx2   = hostHasAvx2();
      hostParallel(n_loci, parts, [&](unsigned t, uint64_t l0, uint64_t l1) {
        // (the totals are kept in a local and stored once: the Part records of the threads share cache lines, and a store per locus into
        //  them made the threads take turns: 0.45 ms for the metric's 800 k reads on four threads, 0.07 ms now on eight)
        Part p;
        struct StoreAtExit {
          Part& to;
          Part& from;
          ~StoreAtExit() { to = from; }
        } storeAtExit{part[t], p};
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
            (avx2 ? scanOffsetsAvx2 : scanOffsetsBase)(o, re - rb, w, longest);
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
      order.resize(n_loci);
      bool uniform = true;  // every locus the same shape (the metric's batch): ties go in locus order, i.e. the identity -- no sort
      for (uint32_t l = 1; l < n_loci && uniform; ++l) uniform = cost[l] == cost[0];
      if (uniform) {
        for (uint32_t l = 0; l < n_loci; ++l) order[l] = l;
      } else {
        std::vector<std::pair<uint64_t, uint32_t>> keyed(n_loci);  // (inverted cost, locus): ascending = most expensive first, ties in locus order
        for (uint32_t l = 0; l < n_loci; ++l) keyed[l] = std::make_pair(~(cost[l] & ~(uint64_t(3) << 62)), l);
        std::sort(keyed.begin(), keyed.end());
        for (uint32_t l = 0; l < n_loci; ++l) order[l] = keyed[l].second;
      }
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
      // (worst-case slabs: ~176 KB per big-class locus against ~130 KB in use on config-5 piles; a bound from the pile alone does not get
      // below it -- 200 x 250 bases hold more word instances than the class admits words -- and an arena that runs out only hands loci
      // back, so a caller short of memory lowers MANTA_AMD_WS_BUDGET_GB)
      if (std::getenv("MANTA_AMD_DEBUG"))
        std::fprintf(stderr, "manta_amd: slab arena of the LDS pipeline: %.1f MB for %zu + %zu loci (cap %.1f MB)\n", double(lgArenaCap) / 1e6, fastIds.size(), bigIds.size(),
                     double(wsBudget / 2) / 1e6);
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
  static const uint32_t kStreamChunks = 32;  ///< at most (array sizes)
  /// chunks a streamed upload is cut into: the kernel cannot start on a chunk before all of it has landed, so the last chunk's loci are the
  /// tail behind the DMA (1 / chunks of the kernel's work); every chunk costs a copy command and a counter write
  /// Workgroup slots a streamed launch of the LDS pipeline leaves free for the runtime's copy kernels and stream writes (launch()): two
  /// per XCD.  Workgroups are dealt to the eight XCDs round-robin and stay there, so what matters is a free slot in EVERY XCD: with 4 free
  /// slots (one in each of four XCDs) a 16 384-locus config-5 block starved until the kernel's time-out, with 16 it runs -- and the quarter
  /// of the CUs that rounds 4-5 left free cost graph_kernel 12 % and graph_big_kernel 25 % of their workgroups for the whole launch
  /// (metric step 8.83 -> 8.60 ms, 16 384 config-5 loci 268 -> 257 ms).  MANTA_AMD_STREAM_FREE_WGS overrides (rounded up to whole eights).
  static int streamFreeSlots(const int cuCount)
  {
    static const int forced = std::getenv("MANTA_AMD_STREAM_FREE_WGS") ? std::max(1, std::atoi(std::getenv("MANTA_AMD_STREAM_FREE_WGS"))) : 0;
    const int        want   = forced ? forced : 16;
    return std::max(1, std::min(((want + 7) / 8) * 8, cuCount / 4));  // (never more than the quarter of the CUs of rounds 4-5: small devices, the emulator)
  }
  static uint32_t streamChunks()
  {
    static const uint32_t n = std::getenv("MANTA_AMD_STREAM_CHUNKS") ? uint32_t(std::max(1, std::min(int(kStreamChunks), std::atoi(std::getenv("MANTA_AMD_STREAM_CHUNKS"))))) : 16u;
    return n;
  }
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
    const bool chunksQueued = preStreamed && preLoci == nLoci;  // startStream() ran for this batch: the chunks are in flight already
    if (preStreamed && !chunksQueued) {  // (cannot happen: startStream() and plan() are given the same batch)
      rt::ScopedStream onCopy(copyStream);
      rt::sync();
    }
    preStreamed = false;
    StreamLayout L;
    if (chunksQueued)
      L = preLayout;
    else
      L = streamLayout(nLoci, read_off, locus_read_begin);
    chunkLoci                 = L.chunkLoci;
    const uint64_t savedBases = nBases;
    nBases                    = L.cursor + 64;  // device arena incl. the per-chunk padding
    upload(nullptr, nullptr, chunksQueued ? nullptr : locus_read_begin);  // allocations + the small arrays (order, word lengths, growth schedule, locus begins)
    nBases  = savedBases;
    dPlCodes = nullptr;
    if (!chunksQueued) rt::h2d(dOff, read_off, sizeof(uint64_t) * (nReadsTotal + 1));
    dStream = bStream.as<uint32_t>(1 + kStreamChunks);
    rt::h2d(dStream, L.shift, sizeof(uint32_t) * (1 + kStreamChunks));
    if (!chunksQueued) resetChunkCounter(copyStream);
    if (chunksQueued) rt::curStreamWaits(evPreSmall);  // (read offsets and locus table: ahead of the chunks on the copy stream)
    rt::sync();  // the counter is zero and the small arrays are in place before the first chunk can land / the kernel starts
    if (!chunksQueued) queueChunks(L, bases, copyStream);
    streaming = true;
  }

  /// where the chunks of a streamed upload lie on the host and on the device (chunks of whole loci, in locus order)
  struct StreamLayout {
    uint32_t chunkLoci = 0, nChunks = 0;
    uint64_t cursor = 0;  ///< bytes of the device arena the chunks take (per-chunk padding included)
    uint64_t hostBegin[kStreamChunks + 1] = {0}, devBegin[kStreamChunks + 1] = {0};
    uint32_t shift[1 + kStreamChunks] = {0};  ///< AsmParams::chunk_shift: device - host offset of chunk c at [1 + c]
    bool     monotone = true;
  };
  static StreamLayout streamLayout(const uint32_t n, const uint64_t* read_off, const uint32_t* locus_read_begin)
  {
    StreamLayout L;
    L.chunkLoci = std::max<uint32_t>(1, (n + streamChunks() - 1) / streamChunks());
    L.nChunks   = (n + L.chunkLoci - 1) / L.chunkLoci;
    for (uint32_t c = 0; c < L.nChunks; ++c) {
      const uint32_t l0 = c * L.chunkLoci, l1 = std::min(n, l0 + L.chunkLoci);
      L.hostBegin[c]    = read_off[locus_read_begin[l0]];
      const uint64_t end = read_off[locus_read_begin[l1]];
      if (end < L.hostBegin[c]) L.monotone = false;
      const uint64_t len = end - L.hostBegin[c];
      L.devBegin[c]      = L.cursor;
      // (modulo 2^32: the kernel adds it in 32-bit arithmetic to a 64-bit offset; device offsets only grow by the padding, so the shifts stay small)
      L.shift[1 + c]     = uint32_t(L.devBegin[c] - L.hostBegin[c]);
      L.cursor           = (L.cursor + len + 64 + 255) & ~uint64_t(255);
    }
    L.hostBegin[L.nChunks] = read_off[locus_read_begin[n]];
    return L;
  }
  void resetChunkCounter(rt::Stream& copyStream)
  {
    if (!dChunksDone) dChunksDone = static_cast<uint32_t*>(rt::dmallocFine(64));
    uint32_t* ids = pChunkIds.as<uint32_t>(kStreamChunks + 1);
    for (uint32_t c = 0; c <= kStreamChunks; ++c) ids[c] = c;
    {  // a call that failed half way may have left counter bumps queued on the copy stream: none may land after the reset
      rt::ScopedStream onCopy(copyStream);
      rt::sync();
    }
    rt::h2d(dChunksDone, ids, sizeof(uint32_t));  // = 0
  }
  void queueChunks(const StreamLayout& L, const uint8_t* bases, rt::Stream& copyStream)
  {
    // one copy per chunk, each followed by a stream-ordered 32-bit write of the counter (command processor; a 4-byte copy
    // if the runtime refuses): the counter says c+1 only after chunk c is in HBM.  Nothing here needs a workgroup slot --
    // the persistent assembler, or another process' kernels, may own every one of them.
    const uint32_t*   ids = pChunkIds.as<uint32_t>(kStreamChunks + 1);
    rt::ScopedStream onCopy(copyStream);
    for (uint32_t c = 0; c < L.nChunks; ++c) {
      rt::h2d(dBases + L.devBegin[c], bases + L.hostBegin[c], L.hostBegin[c + 1] - L.hostBegin[c]);
      if (!rt::streamWrite32(dChunksDone, c + 1)) rt::h2d(dChunksDone, ids + c + 1, sizeof(uint32_t));
    }
  }
  /// The first thing a streamed upload does, BEFORE plan(): the read bases start for the device while the host still sizes the batch
  /// (plan()'s pass over every read offset: ~0.5 ms for the metric's 800 k reads, which the kernel used to spend waiting for chunks
  /// later).  Needs nothing of the plan: chunks are whole loci in locus order.  False (nothing queued): offsets that are not monotone
  /// at the chunk boundaries -- plan() names the error.  The caller drains copyStream if it fails before uploadStreamed().
  bool                  preStreamed = false;
  rt::Event             evPreSmall;
  uint32_t              preLoci     = 0;
  StreamLayout          preLayout;
  bool startStream(const uint32_t n_loci, const uint8_t* bases, const uint64_t* read_off, const uint32_t* locus_read_begin, rt::Stream& copyStream)
  {
    preStreamed = false;
    for (uint32_t l = 0; l < n_loci; ++l)
      if (locus_read_begin[l + 1] < locus_read_begin[l]) return false;
    preLayout = streamLayout(n_loci, read_off, locus_read_begin);
    if (!preLayout.monotone) return false;
    dBases = bBases.as<uint8_t>(preLayout.cursor + 64 + 64);  // (upload() asks for the same sizes again)
    const uint32_t nReads = locus_read_begin[n_loci];
    dOff   = bReadOff.as<uint64_t>(size_t(nReads) + 1);
    dBegin = bLocusBegin.as<uint32_t>(size_t(n_loci) + 1);
    resetChunkCounter(copyStream);
    rt::sync();
    {
      // the per-read offsets (6.4 MB for the metric's batch) and the locus table go first, on the chunks' own stream: on another stream,
      // issued later, they share the link with the chunks -- or queue behind all of them on the same DMA engine (measured: the host then
      // waits 2.4 ms for its small arrays).  uploadStreamed() makes the pipeline's stream wait for them.
      rt::ScopedStream onCopy(copyStream);
      rt::h2d(dOff, read_off, sizeof(uint64_t) * (size_t(nReads) + 1));
      rt::h2d(dBegin, locus_read_begin, sizeof(uint32_t) * (size_t(n_loci) + 1));
      evPreSmall.record();
    }
    queueChunks(preLayout, bases, copyStream);
    preStreamed = true;
    preLoci     = n_loci;
    return true;
  }

  /// uploadStreamed() for packed piles: per chunk the slices of the five pile arrays (codes, N masks, read lengths and the two
  /// per-read offset arrays) are copied to line-aligned device positions, then the counter is bumped; the kernel finds a
  /// locus' slices through three per-chunk shifts (AsmParams::pl_chunk_shift).  Only the locus table goes first.
  void uploadPilesStreamed(const manta_packed_piles_t& pl, rt::Stream& copyStream)
  {
    chunkLoci = std::max<uint32_t>(1, (nLoci + streamChunks() - 1) / streamChunks());
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
          (void)bGws.as<uint8_t>(uint64_t(manta_dev::LGL_GWS_BYTES) * uint64_t(gridBig));
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
    if (locus_read_begin) rt::h2d(dBegin, locus_read_begin, sizeof(uint32_t) * (nLoci + 1));  // (nullptr: startStream() has sent it)
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
      // streamed upload: a chunk the runtime moves with a shader copy -- and every hipStreamWriteValue32, a one-workgroup kernel --
      // needs a free workgroup slot: graph_kernel's two workgroups per CU own all 160 KB of LDS and every VGPR, so a few slots stay
      // free (streamFreeSlots) -- without them the copies never run and the persistent workgroups wait for their chunks forever
      // (seen on hardware, round 4)
      int gf = gridFast;
      if (streaming && gf >= ctx->cuCount * 2) gf -= streamFreeSlots(ctx->cuCount);
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
        if (streaming && gb >= ctx->cuCount) gb -= streamFreeSlots(ctx->cuCount);  // (as above: one workgroup owns a CU's whole LDS)
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
          B.G.gws         = bGws.as<uint8_t>(uint64_t(LGL_GWS_BYTES) * uint64_t(gridBig));
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
    bool queued = false;
    if (!earlyStaged) {
      seqUsedDev  = std::min<uint64_t>(hCnt[1], devSeqCap);
      bitsUsedDev = std::min<uint64_t>(hCnt[2], devBitsCap);
      seqLast     = seqUsedDev;
      bitsLast    = bitsUsedDev;
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
    }
    moreCopies(queued);
    if (queued) rt::sync();
    if (!earlyStaged) stagedTotals();  // (an early staging has done this: finishEarly)
    earlyStaged = false;
  }
  /// The assembler's outputs for the host WHILE THE ALIGNERS RUN (whole-batch small-SV calls, smallsvRunImpl): queued on the copy
  /// stream as soon as the run has read the assembler's counters `cnt` -- the assembler has left the device, the arenas' used sizes are
  /// known, nothing is speculative -- so that the caller can compact them into its arrays before the alignments arrive.  Brings the
  /// per-slot contig records (the packed ones are written after the aligners).
  bool earlyStaged = false;
  void stageEarly(rt::Stream& copyStream, const uint64_t* cnt, rt::Event& done)
  {
    rt::ScopedStream onCopy(copyStream);
    hCnt  = pCnt.as<uint64_t>(16);
    hLoci = pLoci.as<AsmLocusOut>(nLoci);
    hCont = pCont.as<AsmContigOut>(uint64_t(nLoci) * opt.max_assembly_count);
    seqCopied  = std::min<uint64_t>(cnt[1], devSeqCap);
    bitsCopied = std::min<uint64_t>(cnt[2], devBitsCap);
    hSeq       = pSeq.as<uint8_t>(seqCopied + 1);
    hBits      = pBits.as<uint64_t>(bitsCopied + 1);
    rt::d2hAsync(hCnt, dCnt, sizeof(uint64_t) * 16);
    rt::d2hAsync(hLoci, dLoci, sizeof(AsmLocusOut) * nLoci);
    rt::d2hAsync(hCont, dCont, sizeof(AsmContigOut) * uint64_t(nLoci) * opt.max_assembly_count);
    rt::d2hAsync(hSeq, dSeq, seqCopied);
    rt::d2hAsync(hBits, dBits, sizeof(uint64_t) * bitsCopied);
    done.record();
    earlyStaged = true;
  }
  /// after the host has waited for stageEarly()'s event
  void finishEarly()
  {
    seqUsedDev  = seqCopied;
    bitsUsedDev = bitsCopied;
    seqLast     = seqUsedDev;
    bitsLast    = bitsUsedDev;
    stagedTotals();
  }
  /// the pipeline's own copies behind an early staging (pipeStageEnqueue)
  template <typename F0>
  void stageEnqueueRest(F0 firstCopies)
  {
    firstCopies();
    stageQueued = true;
  }
  void stagedTotals()
  {
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

}  // namespace manta_host
using namespace manta_host;

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
  // whole-batch calls: the host's work on the assembler's outputs (sizes, compaction into the caller's arrays) while the aligners run.
  // The run stages those outputs on the copy stream as soon as the assembler has left the device (AsmStage::stageEarly) and calls the
  // hook after its last launch, before it waits for the device.
  std::function<void()> whileAligning;
  rt::Event             evEarlyStaged;
  PinnedBuf             hostOff[3], hostBegin;  // whole-batch worker: a block's rebased offset arrays (api_batch.cpp: rebase), kept across calls
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
  std::function<void()> whileAligning;  // as manta_smallsv::whileAligning (called while the last alignment pass runs)
  rt::Event             evEarlyStaged;
  PinnedBuf             hostOff[3], hostBegin;  // whole-batch worker: a block's rebased offset arrays, kept across calls
  explicit manta_spanning(manta_ctx_t* c) : ctx(c), asmStage(c) {}
};

namespace manta_host {

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

}  // namespace manta_host
using namespace manta_host;

namespace manta_host {
inline int checkPiles(manta_ctx_t* ctx, const manta_packed_piles_t* pl, const char* who)
{
  if (!pl || !pl->codes || !pl->nmask || !pl->read_len || !pl->read_code_off || !pl->read_mask_off || !pl->locus_read_begin)
    return fail(ctx, MANTA_E_INVALID_ARG, std::string(who) + ": null pointer in the packed piles");
  return MANTA_OK;
}
}  // namespace manta_host
using namespace manta_host;

/// Stage gates of a whole-batch call with several workers: at most one block assembles and at most one block
/// aligns at any time, so that block B's (memory-bound) assembler overlaps block A's (VALU-bound) aligners and transfers
/// instead of two persistent assemblers fighting for the same wave slots.
/// assembler waves per CU while another block's aligners share the device: 3 of the 4 wave slots per SIMD (128 VGPRs each)
static const int kPipelinedAsmWavesPerCu = std::getenv("MANTA_AMD_PIPELINED_ASM_WAVES") ? std::atoi(std::getenv("MANTA_AMD_PIPELINED_ASM_WAVES")) : 12;
struct StageGates {
  std::mutex asmMu, alignMu;
};
inline std::mutex g_streamedAsmMuOfDevice[16];  // see smallsvRunImpl; one per device (id modulo 16)
inline std::mutex& streamedAsmMu(const manta_ctx_t* ctx) { return g_streamedAsmMuOfDevice[unsigned(ctx->deviceId) % 16u]; }
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

/// one run of a fused pipeline (api.cpp); `gates` = the stage gates of a whole-batch call with several workers, else nullptr
int smallsvRunImpl(manta_smallsv_t* b, StageGates* gates);
int spanningRunImpl(manta_spanning_t* b, StageGates* gates);

namespace manta_host {

/// device -> pinned staging of one finished pipeline run (both pipelines): counters + locus records + first-contig
/// index in the first round trip, then exactly the used part of every arena
template <typename Pipe>
void pipeStageEnqueue(Pipe* b)
{
  const uint32_t nLoci = b->nLoci;
  b->hPackCnt          = b->pPackCnt.template as<uint32_t>(4);
  b->hFirst            = b->pFirst.template as<uint32_t>(nLoci);
  auto packCopies = [&] {
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
      };
  if (b->asmStage.earlyStaged)  // the assembler's outputs are on the host already (AsmStage::stageEarly)
    b->asmStage.stageEnqueueRest(packCopies);
  else
    b->asmStage.stageEnqueue(packCopies, false);
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
inline int smallsvCompactAlign(
    manta_smallsv* b, const manta_asm_locus_result_t* loci, manta_smallsv_alignment_t* alignments, uint32_t* cigar_arena, uint64_t cigar_arena_cap,
    uint64_t cigarBase, uint64_t* cigar_arena_used, uint32_t lBegin, uint32_t lEnd, uint64_t* cellsOut, uint64_t* ptrBytesOut, int worst);
inline int smallsvCompact(
    manta_smallsv* b, manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs, manta_smallsv_alignment_t* alignments,
    uint64_t contigBase, uint64_t contigs_cap, uint8_t* seq_arena, uint64_t seq_arena_cap, uint64_t seqBase, uint64_t* seq_arena_used,
    uint64_t* bits_arena, uint64_t bits_arena_cap, uint64_t bitsBase, uint64_t* bits_arena_used, uint32_t* cigar_arena,
    uint64_t cigar_arena_cap, uint64_t cigarBase, uint64_t* cigar_arena_used, uint32_t lBegin = 0, uint32_t lEnd = ~0u,
    uint64_t* cellsOut = nullptr, uint64_t* ptrBytesOut = nullptr)
{
  // [lBegin, lEnd): the loci of this call (a whole-batch call compacts a block in a few ranges, one host thread each); every
  // output pointer / base is that of the range's first record
  lEnd   = std::min(lEnd, b->nLoci);
  int rc = b->asmStage.compact(PackedContigs<manta_smallsv>{b}, loci, contigs + contigBase, contigs_cap, seq_arena, seq_arena_cap, seq_arena_used,
                               bits_arena, bits_arena_cap, bits_arena_used, contigBase, seqBase, bitsBase, lBegin, lEnd);
  if (rc != MANTA_OK && !perItemCode(rc)) return rc;
  return smallsvCompactAlign(b, loci, alignments, cigar_arena, cigar_arena_cap, cigarBase, cigar_arena_used, lBegin, lEnd, cellsOut, ptrBytesOut, rc);
}
/// the second half of smallsvCompact: the alignment records and CIGARs of the loci [lBegin, lEnd), whose locus records (first_contig,
/// n_contigs) AsmStage::compact has written; `worst`: what that call returned (a per-item code or MANTA_OK)
inline int smallsvCompactAlign(
    manta_smallsv* b, const manta_asm_locus_result_t* loci, manta_smallsv_alignment_t* alignments, uint32_t* cigar_arena, uint64_t cigar_arena_cap,
    uint64_t cigarBase, uint64_t* cigar_arena_used, uint32_t lBegin, uint32_t lEnd, uint64_t* cellsOut, uint64_t* ptrBytesOut, int worst)
{
  manta_ctx_t* ctx = b->ctx;
  lEnd             = std::min(lEnd, b->nLoci);
  uint64_t used = 0, cells = 0, ptrBytes = 0;
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
inline uint64_t smallsvCigarWords(const manta_smallsv* b, uint32_t lBegin, uint32_t lEnd)
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

}  // namespace manta_host
using namespace manta_host;

namespace manta_host {

inline int spanningCompactAlign(
    manta_spanning* b, const manta_asm_locus_result_t* loci, manta_spanning_alignment_t* alignments, uint32_t* cigar_arena, uint64_t cigar_arena_cap,
    uint64_t cigarBase, uint64_t* cigar_arena_used, uint32_t lBegin, uint32_t lEnd, uint64_t* cellsOut, uint64_t* ptrBytesOut, int worst);
inline int spanningCompact(
    manta_spanning* b, manta_asm_locus_result_t* loci, manta_asm_contig_t* contigs, manta_spanning_alignment_t* alignments,
    uint64_t contigBase, uint64_t contigs_cap, uint8_t* seq_arena, uint64_t seq_arena_cap, uint64_t seqBase, uint64_t* seq_arena_used,
    uint64_t* bits_arena, uint64_t bits_arena_cap, uint64_t bitsBase, uint64_t* bits_arena_used, uint32_t* cigar_arena,
    uint64_t cigar_arena_cap, uint64_t cigarBase, uint64_t* cigar_arena_used, uint32_t lBegin = 0, uint32_t lEnd = ~0u,
    uint64_t* cellsOut = nullptr, uint64_t* ptrBytesOut = nullptr)
{
  // [lBegin, lEnd) and the *Out totals: as smallsvCompact
  lEnd   = std::min(lEnd, b->nLoci);
  int rc = b->asmStage.compact(PackedContigs<manta_spanning>{b}, loci, contigs + contigBase, contigs_cap, seq_arena, seq_arena_cap, seq_arena_used,
                               bits_arena, bits_arena_cap, bits_arena_used, contigBase, seqBase, bitsBase, lBegin, lEnd);
  if (rc != MANTA_OK && !perItemCode(rc)) return rc;
  return spanningCompactAlign(b, loci, alignments, cigar_arena, cigar_arena_cap, cigarBase, cigar_arena_used, lBegin, lEnd, cellsOut, ptrBytesOut, rc);
}
/// the second half of spanningCompact (as smallsvCompactAlign)
inline int spanningCompactAlign(
    manta_spanning* b, const manta_asm_locus_result_t* loci, manta_spanning_alignment_t* alignments, uint32_t* cigar_arena, uint64_t cigar_arena_cap,
    uint64_t cigarBase, uint64_t* cigar_arena_used, uint32_t lBegin, uint32_t lEnd, uint64_t* cellsOut, uint64_t* ptrBytesOut, int worst)
{
  manta_ctx_t* ctx = b->ctx;
  lEnd             = std::min(lEnd, b->nLoci);
  uint64_t used = 0, cells = 0, ptrBytes = 0;
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
inline uint64_t spanningCigarWords(const manta_spanning* b, uint32_t lBegin, uint32_t lEnd)
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

}  // namespace manta_host
using namespace manta_host;

using namespace manta_host;
