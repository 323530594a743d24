// C-ABI implementation (include/manta_amd.h).  Compiled by hipcc for gfx950 into manta_amd/libmanta_amd.so.
// Host code here only stages buffers, buckets work and launches the kernels; all arithmetic of the hot path
// runs in the HIP kernels of align_kernels.hpp / assemble_kernels.hpp.
#include "../../include/manta_amd.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "align_kernels.hpp"
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

}  // namespace

struct manta_ctx {
  std::string lastError;
  std::string deviceName;
  int         cuCount = 0;
  // align scratch
  DevBuf dSeq, dTasks, dResults, dCigar, dTaskIds, dCounter, dPtrWs;
};

namespace {

int fail(manta_ctx_t* ctx, int code, const std::string& msg)
{
  if (ctx) ctx->lastError = msg;
  return code;
}

const int kESet[]  = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 24, 32};
const int kNumESet = sizeof(kESet) / sizeof(kESet[0]);

int pickE(uint32_t qlen)
{
  const uint32_t need = (qlen + 63) / 64;
  for (int i = 0; i < kNumESet; ++i)
    if (uint32_t(kESet[i]) >= need) return i;
  return -1;
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

void launchAlignKind(int kind, int eIdx, int grid, const AlignParams& P)
{
  if (kind == MANTA_ALIGNER_GLOBAL)
    launchAlignE<0>(eIdx, grid, P);
  else if (kind == MANTA_ALIGNER_LARGE_INDEL)
    launchAlignE<1>(eIdx, grid, P);
  else
    launchAlignE<2>(eIdx, grid, P);
}

}  // namespace

extern "C" {

int manta_ctx_create(int device_id, manta_ctx_t** out)
{
  if (!out) {
    g_createError = "manta_ctx_create: null output pointer";
    return MANTA_E_INVALID_ARG;
  }
  *out = nullptr;
  try {
    rt::init(device_id);
    manta_ctx_t* ctx = new manta_ctx();
    ctx->deviceName  = rt::deviceName();
    ctx->cuCount     = rt::cuCount();
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
      if (t.query_off + t.query_len > seq_arena_bytes || t.ref1_off + t.ref1_len > seq_arena_bytes ||
          (jump && t.ref2_off + t.ref2_len > seq_arena_bytes))
        return fail(ctx, MANTA_E_INVALID_ARG, "manta_align_batch: task " + std::to_string(i) + " outside the sequence arena");
      if (t.query_len == 0 || t.ref1_len == 0 || (jump && t.ref2_len == 0)) {
        r.status = MANTA_E_EMPTY_SEQ;  // GlobalJumpAlignerImpl.hpp:50-58, GlobalAlignerImpl.hpp:44-49
        worst    = MANTA_E_EMPTY_SEQ;
        continue;
      }
      const int eIdx = pickE(t.query_len);
      if (eIdx < 0) {
        r.status = MANTA_E_UNSUPPORTED;
        worst    = MANTA_E_UNSUPPORTED;
        continue;
      }
      AlignTaskDev& d(dev[i]);
      d.query_off = t.query_off;
      d.ref1_off  = t.ref1_off;
      d.ref2_off  = jump ? t.ref2_off : 0;
      d.query_len = t.query_len;
      d.ref1_len  = t.ref1_len;
      d.ref2_len  = jump ? t.ref2_len : 0;
      d.cigar_off = uint32_t(cigarDevWords);
      cigarDevWords += 4ull * t.query_len + 16;
      if (cigarDevWords > 0xffffffffull) return fail(ctx, MANTA_E_UNSUPPORTED, "manta_align_batch: batch too large (cigar workspace)");
      buckets[eIdx].push_back(i);
      bucketMaxRef[eIdx] = std::max<uint64_t>(bucketMaxRef[eIdx], uint64_t(d.ref1_len) + d.ref2_len);
    }

    // ---- stage ----
    uint8_t*        dSeq     = ctx->dSeq.as<uint8_t>(seq_arena_bytes);
    AlignTaskDev*   dTasks   = ctx->dTasks.as<AlignTaskDev>(n_tasks);
    AlignResultDev* dResults = ctx->dResults.as<AlignResultDev>(n_tasks);
    uint32_t*       dCigar   = ctx->dCigar.as<uint32_t>(cigarDevWords + 1);
    uint32_t*       dIds     = ctx->dTaskIds.as<uint32_t>(n_tasks);
    uint32_t*       dCounter = ctx->dCounter.as<uint32_t>(kNumESet);
    rt::h2d(dSeq, seq_arena, seq_arena_bytes);
    rt::h2d(dTasks, dev.data(), sizeof(AlignTaskDev) * n_tasks);
    rt::dzero(dCounter, sizeof(uint32_t) * kNumESet);
    rt::dzero(dResults, sizeof(AlignResultDev) * n_tasks);

    const int    maxWaves  = std::max(1, ctx->cuCount * 8);
    const size_t wsBudget  = std::min<size_t>(rt::freeBytes() / 2, size_t(24) << 30);
    size_t       idsCursor = 0;
    for (int b = 0; b < kNumESet; ++b) {
      if (buckets[b].empty()) continue;
      const uint64_t stride = (alignPtrSlabBytes(kind, kESet[b], bucketMaxRef[b]) + 255) & ~uint64_t(255);
      int            grid   = int(std::min<size_t>(buckets[b].size(), size_t(maxWaves)));
      grid                  = int(std::max<size_t>(1, std::min<size_t>(size_t(grid), wsBudget / stride)));
      uint8_t* dWs          = ctx->dPtrWs.as<uint8_t>(stride * grid);
      rt::h2d(dIds + idsCursor, buckets[b].data(), sizeof(uint32_t) * buckets[b].size());
      AlignParams P;
      P.seq            = dSeq;
      P.tasks          = dTasks;
      P.results        = dResults;
      P.cigar          = dCigar;
      P.task_ids       = dIds + idsCursor;
      P.n_tasks        = uint32_t(buckets[b].size());
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
      launchAlignKind(kind, b, grid, P);
      rt::sync();  // dPtrWs may be re-sized by the next bucket
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

}  // extern "C"
