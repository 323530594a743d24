// TEST INFRASTRUCTURE: lock-step 64-lane wavefront emulator.
//
// Lets the CPU-only test tier execute the REAL kernel bodies of manta_amd/csrc (compiled with -DMANTA_WAVE_EMU
// into tests/emu/libmanta_amd_emu.so).  Each lane is a cooperative fiber (ucontext) on one OS thread; every
// cross-lane primitive is a rendezvous: all live lanes run up to it, exchange through a ping-pong buffer, and
// continue.  Between rendezvous points exactly one lane runs, so "atomics" are plain read-modify-writes.
// This is NOT a product code path: manta_amd/ never builds, loads or falls back to it.
#pragma once
#include <ucontext.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define WV_DEV inline
#define WV_KERNEL
#define WV_KERNEL_OCC(w)
#define WV_KERNEL_SINGLE
#define WV_KERNEL_WG(n)
#define WV_WAVES_PER_SIMD(n)
#define WV_WAVES_PER_WG 1
#define WV_HD inline
#define WV_DEV_COLD inline
#define WV_DEV_CALL inline

namespace wv_emu {

struct Wave {
  static const int      N = 64;
  ucontext_t            sched;
  ucontext_t            lanes[N];
  std::vector<char>     stacks;
  bool                  done[N];
  int                   cur = 0;
  uint64_t              xbuf[2][N];
  unsigned              opSeq[N];    // number of rendezvous each lane has executed (convergence check)
  const char*           opTag[N];
  std::vector<char>     ldsMem;   // single-wave launches: the wave's own LDS
  char*                 ldsBase = nullptr;  // the (16-byte aligned) LDS this wave sees (shared by the waves of a workgroup)
  int                   block = 0, nblocks = 1;
  int                   waveInWg = 0, wgWaves = 1;
  bool                  atBarrier = false;
  std::function<void()> body;
};

inline Wave*& W()
{
  static thread_local Wave* w = nullptr;
  return w;
}

inline void trampoline()
{
  Wave* w = W();
  w->body();
  w->done[w->cur] = true;
  swapcontext(&w->lanes[w->cur], &w->sched);
}

inline void yieldLane(const char* tag)
{
  Wave* w          = W();
  w->opSeq[w->cur] += 1;
  w->opTag[w->cur] = tag;
  swapcontext(&w->lanes[w->cur], &w->sched);
}

/// MANTA_EMU_LANE_ORDER=reverse steps the lanes of a wave from 63 down to 0 between rendezvous.  A kernel whose result depends on
/// the order in which lanes run between two rendezvous is missing a wv::sync(): the tests run the sensitive suites both ways.
inline bool laneOrderReversed()
{
  static const bool rev = [] {
    const char* e = std::getenv("MANTA_EMU_LANE_ORDER");
    return e && std::strcmp(e, "reverse") == 0;
  }();
  return rev;
}

/// run `grid` workgroups (one wave each) sequentially
inline void launch(int grid, size_t ldsBytes, const std::function<void()>& body)
{
  static const size_t STACK = 256 * 1024;
  Wave                w;
  w.stacks.resize(STACK * Wave::N);
  w.ldsMem.assign(ldsBytes + 64, 0);
  w.body    = body;
  w.nblocks = grid;
  Wave* saved = W();
  W()         = &w;
  {
    char* p   = w.ldsMem.data();
    w.ldsBase = p + ((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15);
  }
  for (int b = 0; b < grid; ++b) {
    w.block = b;
    std::fill(w.ldsMem.begin(), w.ldsMem.end(), char(0xcd));  // LDS is NOT zero-initialised on hardware
    for (int l = 0; l < Wave::N; ++l) {
      w.done[l]  = false;
      w.opSeq[l] = 0;
      w.opTag[l] = "start";
      getcontext(&w.lanes[l]);
      w.lanes[l].uc_stack.ss_sp   = w.stacks.data() + STACK * l;
      w.lanes[l].uc_stack.ss_size = STACK;
      w.lanes[l].uc_link          = &w.sched;
      makecontext(&w.lanes[l], (void (*)())trampoline, 0);
    }
    while (true) {
      bool anyLive = false;
      for (int li = 0; li < Wave::N; ++li) {
        const int l = laneOrderReversed() ? Wave::N - 1 - li : li;
        if (w.done[l]) continue;
        anyLive = true;
        w.cur   = l;
        swapcontext(&w.sched, &w.lanes[l]);
      }
      if (!anyLive) break;
      // convergence check: all live lanes must sit at the same rendezvous
      unsigned    seq = 0;
      const char* tag = nullptr;
      for (int l = 0; l < Wave::N; ++l) {
        if (w.done[l]) continue;
        if (!tag) {
          seq = w.opSeq[l];
          tag = w.opTag[l];
        } else if (w.opSeq[l] != seq || w.opTag[l] != tag) {
          std::fprintf(stderr, "wave_emu: divergent rendezvous: lane %d at '%s'#%u vs '%s'#%u (block %d)\n", l, w.opTag[l],
                       w.opSeq[l], tag, seq, b);
          std::abort();
        }
      }
    }
  }
  W() = saved;
}

/// run `grid` workgroups of `nWaves` cooperating wavefronts (shared LDS, workgroup barrier) one after the other.  The waves of a
/// workgroup are stepped round-robin, one rendezvous at a time; a wave that sits at the workgroup barrier is held until every
/// wave that has not finished sits there too (finished waves leave the barrier's count, as on the hardware).
inline void launchWG(int grid, int nWaves, size_t ldsBytes, const std::function<void()>& body)
{
  static const size_t STACK = 256 * 1024;
  std::vector<Wave>   waves(nWaves);
  std::vector<char>   lds(ldsBytes + 64, 0);
  char*               ldsAligned = lds.data() + ((16 - (reinterpret_cast<uintptr_t>(lds.data()) & 15)) & 15);
  for (int v = 0; v < nWaves; ++v) {
    Wave& w(waves[v]);
    w.stacks.resize(STACK * Wave::N);
    w.body     = body;
    w.nblocks  = grid;
    w.waveInWg = v;
    w.wgWaves  = nWaves;
    w.ldsBase  = ldsAligned;
  }
  Wave* saved = W();
  for (int b = 0; b < grid; ++b) {
    std::fill(lds.begin(), lds.end(), char(0xcd));
    for (Wave& w : waves) {
      w.block     = b;
      w.atBarrier = false;
      W()         = &w;
      for (int l = 0; l < Wave::N; ++l) {
        w.done[l]  = false;
        w.opSeq[l] = 0;
        w.opTag[l] = "start";
        getcontext(&w.lanes[l]);
        w.lanes[l].uc_stack.ss_sp   = w.stacks.data() + STACK * l;
        w.lanes[l].uc_stack.ss_size = STACK;
        w.lanes[l].uc_link          = &w.sched;
        makecontext(&w.lanes[l], (void (*)())trampoline, 0);
      }
    }
    while (true) {
      bool anyLive = false, progressed = false;
      for (Wave& w : waves) {
        bool live = false;
        for (int l = 0; l < Wave::N; ++l) live = live || !w.done[l];
        if (!live) continue;
        anyLive = true;
        if (w.atBarrier) continue;
        W() = &w;
        for (int li = 0; li < Wave::N; ++li) {
          const int l = laneOrderReversed() ? Wave::N - 1 - li : li;
          if (w.done[l]) continue;
          w.cur = l;
          swapcontext(&w.sched, &w.lanes[l]);
        }
        progressed      = true;
        unsigned    seq = 0;
        const char* tag = nullptr;
        for (int l = 0; l < Wave::N; ++l) {
          if (w.done[l]) continue;
          if (!tag) {
            seq = w.opSeq[l];
            tag = w.opTag[l];
          } else if (w.opSeq[l] != seq || w.opTag[l] != tag) {
            std::fprintf(stderr, "wave_emu: divergent rendezvous: wave %d lane %d at '%s'#%u vs '%s'#%u (workgroup %d)\n", w.waveInWg, l,
                         w.opTag[l], w.opSeq[l], tag, seq, b);
            std::abort();
          }
        }
        if (tag && std::strcmp(tag, "wgbarrier") == 0) w.atBarrier = true;
      }
      if (!anyLive) break;
      bool allThere = true;
      for (Wave& w : waves) {
        bool live = false;
        for (int l = 0; l < Wave::N; ++l) live = live || !w.done[l];
        if (live && !w.atBarrier) allThere = false;
      }
      if (allThere) {
        for (Wave& w : waves) w.atBarrier = false;
      } else if (!progressed) {
        std::fprintf(stderr, "wave_emu: workgroup %d is stuck: some waves wait at the barrier, the others have no lane to run\n", b);
        std::abort();
      }
    }
  }
  W() = saved;
}

template <typename T>
inline T exchange(T v, int src, const char* tag)
{
  Wave*     w = W();
  const int p = w->opSeq[w->cur] & 1;
  uint64_t  raw = 0;
  std::memcpy(&raw, &v, sizeof(T));
  w->xbuf[p][w->cur] = raw;
  yieldLane(tag);
  T r;
  std::memcpy(&r, &w->xbuf[p][src & 63], sizeof(T));
  return r;
}

}  // namespace wv_emu

namespace wv {

inline int lane() { return wv_emu::W()->cur; }
inline int block() { return wv_emu::W()->block; }
inline int nblocks() { return wv_emu::W()->nblocks; }
inline char* lds(const unsigned) { return wv_emu::W()->ldsBase; }

inline int   block_single() { return block(); }
inline char* lds_single() { return lds(0); }
/// cooperative workgroups (launchWG): this wave's index in its workgroup, the workgroup's wave count, the workgroup barrier, and
/// a polling step that lets the other waves of the workgroup run
inline int  wave_in_wg() { return wv_emu::W()->waveInWg; }
inline int  wg_waves() { return wv_emu::W()->wgWaves; }
inline void wg_barrier() { wv_emu::yieldLane("wgbarrier"); }
inline void spin() { wv_emu::yieldLane("spin"); }
inline void fence_wg() {}
inline void atomic_store(unsigned* p, unsigned v) { *p = v; }

inline int shr1(int v, int fill)
{
  const int l = lane();
  const int r = wv_emu::exchange(v, l - 1, "shr1");
  return (l == 0) ? fill : r;
}
inline unsigned shr1(unsigned v, unsigned fill) { return unsigned(shr1(int(v), int(fill))); }

inline int shfl(int v, int src) { return wv_emu::exchange(v, src, "shfl"); }
inline unsigned shfl(unsigned v, int src) { return wv_emu::exchange(v, src, "shfl"); }
inline uint64_t shfl(uint64_t v, int src) { return wv_emu::exchange(v, src, "shfl64"); }

inline int readlane(int v, int src) { return wv_emu::exchange(v, src, "readlane"); }
inline unsigned readlane(unsigned v, int src) { return wv_emu::exchange(v, src, "readlane"); }
inline uint64_t readlane(uint64_t v, int src) { return wv_emu::exchange(v, src, "readlane64"); }

inline uint64_t ballot(bool p)
{
  wv_emu::Wave* w   = wv_emu::W();
  const int     par = w->opSeq[w->cur] & 1;
  w->xbuf[par][w->cur] = p ? 1 : 0;
  wv_emu::yieldLane("ballot");
  uint64_t m = 0;
  for (int l = 0; l < 64; ++l)
    if (!w->done[l] && w->xbuf[par][l]) m |= (uint64_t(1) << l);
  return m;
}
inline bool any(bool p) { return ballot(p) != 0; }

inline int first(int v)
{
  wv_emu::Wave* w = wv_emu::W();
  int           f = 0;
  while (w->done[f]) ++f;
  return wv_emu::exchange(v, f, "first");
}
inline unsigned first(unsigned v) { return unsigned(first(int(v))); }

inline void sync() { wv_emu::yieldLane("sync"); }

inline unsigned atomic_add(unsigned* p, unsigned v) { const unsigned o = *p; *p = o + v; return o; }
inline unsigned atomic_sub(unsigned* p, unsigned v) { const unsigned o = *p; *p = o - v; return o; }
inline unsigned atomic_and(unsigned* p, unsigned v) { const unsigned o = *p; *p = o & v; return o; }
inline unsigned atomic_exch(unsigned* p, unsigned v) { const unsigned o = *p; *p = v; return o; }
inline unsigned atomic_min(unsigned* p, unsigned v) { const unsigned o = *p; if (v < o) *p = v; return o; }
inline unsigned atomic_max(unsigned* p, unsigned v) { const unsigned o = *p; if (v > o) *p = v; return o; }
inline unsigned atomic_cas(unsigned* p, unsigned cmp, unsigned v) { const unsigned o = *p; if (o == cmp) *p = v; return o; }
inline unsigned atomic_or(unsigned* p, unsigned v) { const unsigned o = *p; *p = o | v; return o; }
inline unsigned long long atomic_or(unsigned long long* p, unsigned long long v) { const unsigned long long o = *p; *p = o | v; return o; }
inline unsigned long long atomic_add(unsigned long long* p, unsigned long long v) { const unsigned long long o = *p; *p = o + v; return o; }
inline unsigned atomic_load(const unsigned* p) { return *p; }
inline unsigned long long atomic_load(const unsigned long long* p) { return *p; }
inline unsigned atomic_load_system(const unsigned* p) { return *p; }
inline void sleep() {}
inline void setprio(unsigned) {}
inline void fence_acquire() {}

// packed 16-bit arithmetic (two lanes of 16 bits per 32-bit value), as the hardware's v_pk_* instructions compute it
inline uint32_t pk_add_sat_i16(uint32_t a, uint32_t b)
{
  uint32_t r = 0;
  for (int h = 0; h < 2; ++h) {
    int v = int(int16_t(a >> (16 * h))) + int(int16_t(b >> (16 * h)));
    v     = (v > 32767) ? 32767 : ((v < -32768) ? -32768 : v);
    r |= (uint32_t(v) & 0xffffu) << (16 * h);
  }
  return r;
}
inline uint32_t pk_max_i16(uint32_t a, uint32_t b)
{
  uint32_t r = 0;
  for (int h = 0; h < 2; ++h) {
    const int x = int16_t(a >> (16 * h)), y = int16_t(b >> (16 * h));
    r |= (uint32_t(x > y ? x : y) & 0xffffu) << (16 * h);
  }
  return r;
}
inline uint32_t pk_min_u16(uint32_t a, uint32_t b)
{
  uint32_t r = 0;
  for (int h = 0; h < 2; ++h) {
    const uint32_t x = (a >> (16 * h)) & 0xffffu, y = (b >> (16 * h)) & 0xffffu;
    r |= (x < y ? x : y) << (16 * h);
  }
  return r;
}
inline uint32_t pk_mad_u16(uint32_t a, uint32_t b, uint32_t c)
{
  uint32_t r = 0;
  for (int h = 0; h < 2; ++h) {
    const uint32_t x = (a >> (16 * h)) & 0xffffu, y = (b >> (16 * h)) & 0xffffu, z = (c >> (16 * h)) & 0xffffu;
    r |= ((x * y + z) & 0xffffu) << (16 * h);
  }
  return r;
}

inline uint64_t clock() { return 0; }
inline int popc(unsigned v) { return __builtin_popcount(v); }
inline int popc(uint64_t v) { return __builtin_popcountll(v); }
inline int ctz(uint64_t v) { return __builtin_ctzll(v); }
inline int clz(uint64_t v) { return __builtin_clzll(v); }

}  // namespace wv
