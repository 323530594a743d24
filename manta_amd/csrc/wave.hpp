// Wave-level primitives for the gfx950 kernels (one workgroup == one 64-lane wavefront everywhere in this
// library).  Kernels use ONLY these wrappers for lane ids, cross-lane traffic, LDS and atomics, so that the
// identical kernel source can also be run by the lock-step wave emulator in tests/emu/ (test infrastructure:
// it exists so the CPU-only test tier can execute the real kernel bodies; the product library is never built
// that way and has no CPU path).
#pragma once
#include <cstdint>

// Translation units of the product library.  manta_amd/build.py compiles csrc/kernels_tu.cpp once per kernel family (-DMANTA_TU=<id>: that
// family's kernels are DEFINED, device code and host stubs) and the host sources with MANTA_TU_HOST (every kernel only DECLARED: a kernel's
// handle is an ordinary external symbol), all in parallel; editing host code no longer recompiles ~70 kernel instantiations.  A TU built
// without MANTA_TU (the wave-emulator build of tests/emu, the profile variants) defines everything, as the single-TU build did.
#define MANTA_TU_ALL 0
#define MANTA_TU_HOST 1
#define MANTA_TU_ASM 2          // assemble_kernel, small_assemble_kernel
#define MANTA_TU_ASM_GENERIC 3  // assemble_generic_kernel
#define MANTA_TU_GRAPH 4        // graph_kernel<2 / 4 / 8>
#define MANTA_TU_GRAPH_BIG 5    // graph_big_kernel<5 / 8>
#define MANTA_TU_CONTIG 6       // contig_kernel, contig_big_kernel
#define MANTA_TU_REPEAT 7       // repeat_big_kernel
#define MANTA_TU_ALIGN0 8       // align_kernel<0, E>
#define MANTA_TU_ALIGN1 9       // align_kernel<1, E>
#define MANTA_TU_ALIGN2 10      // align_kernel<2, E>
#define MANTA_TU_ALIGN_PAIR 11  // align_pair_kernel<E>, align_pair_multi_kernel
#define MANTA_TU_JUMP_PAIR 12   // align_jump_pair_kernel<E>
#define MANTA_TU_GLUE 13        // pipeline_kernels.hpp, split_kernels.hpp, read_class_kernels.hpp
#define MANTA_TU_COUNT 14
#ifndef MANTA_TU
#define MANTA_TU MANTA_TU_ALL
#endif
#define MANTA_TU_DEFINES(tu) (MANTA_TU == MANTA_TU_ALL || MANTA_TU == (tu))
// (a template kernel is explicitly instantiated in its own TU and `extern template` -- no implicit instantiation -- in every other one:
// the lists behind the kernels' definitions)

#ifdef MANTA_WAVE_EMU
#include "wave_emu.hpp"  // tests/emu/
#else

#include <hip/hip_runtime.h>

#define WV_DEV __device__ __forceinline__
// Workgroups are 4 INDEPENDENT wavefronts (256 threads): the hardware admits only ~8 workgroups per CU, so
// single-wave workgroups would cap residency at 8 waves/CU.  The waves of a workgroup never synchronise with each
// other; every "block" below is one wavefront.
#define WV_WAVES_PER_WG 4
#define WV_KERNEL __global__ __launch_bounds__(64 * WV_WAVES_PER_WG)
// same, additionally asking the register allocator for at least `w` resident waves per SIMD
#define WV_KERNEL_OCC(w) __global__ __launch_bounds__(64 * WV_WAVES_PER_WG, w)
// single-wave workgroups (the LDS-resident assembler: one locus per workgroup, the workgroup's LDS is the wave's)
#define WV_KERNEL_SINGLE __global__ __launch_bounds__(64)
// cooperative workgroups of `n` wavefronts working on ONE item (shared LDS, workgroup barrier)
#define WV_KERNEL_WG(n) __global__ __launch_bounds__(64 * (n))
// register budget: the kernel must fit `n` of its wavefronts on a SIMD (512 / n VGPRs)
#define WV_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n, n)))
#define WV_HD __host__ __device__ inline
// cold paths (measured: real out-of-line calls cost more than they save on gfx950, so this is still inline)
#define WV_DEV_COLD __device__ __forceinline__
// a real call (once per work item, never per step): keeps the caller's register budget at the callee's own
#define WV_DEV_CALL __device__ __attribute__((noinline))

extern __shared__ __attribute__((aligned(16))) char wv_dyn_lds[];

namespace wv {

WV_DEV int lane() { return int(threadIdx.x & 63u); }
/// index of this wavefront among all wavefronts of the launch (its workspace slot)
WV_DEV int block() { return int(blockIdx.x) * WV_WAVES_PER_WG + __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)); }
WV_DEV int nblocks() { return int(gridDim.x) * WV_WAVES_PER_WG; }
/// this wavefront's share of the dynamic LDS (ldsBytes passed to rt::launch is PER WAVE)
WV_DEV char* lds(const unsigned bytesPerWave) { return wv_dyn_lds + size_t(__builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6))) * bytesPerWave; }

/// single-wave workgroups: workgroup index and the whole dynamic LDS
WV_DEV int   block_single() { return int(blockIdx.x); }
WV_DEV char* lds_single() { return wv_dyn_lds; }
/// cooperative workgroups: this wave's index in its workgroup, the wave count, the workgroup barrier (s_barrier behind a full
/// wait for this wave's memory operations: LDS writes of all waves are visible to all afterwards), a polling back-off
WV_DEV int  wave_in_wg() { return __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)); }
WV_DEV int  wg_waves() { return int(blockDim.x >> 6); }
WV_DEV void wg_barrier() { __syncthreads(); }
WV_DEV void spin() { __builtin_amdgcn_s_sleep(2); }
/// orders this lane's earlier LDS / memory writes before its later ones as the other waves of the workgroup see them
WV_DEV void fence_wg() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); }
WV_DEV void atomic_store(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

/// lane l receives lane (l-1)'s value; lane 0 receives `fill`   (v_mov_b32_dpp wave_shr:1)
WV_DEV int shr1(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false); }
WV_DEV unsigned shr1(unsigned v, unsigned fill) { return unsigned(shr1(int(v), int(fill))); }

/// arbitrary gather (ds_bpermute_b32)
WV_DEV int shfl(int v, int src) { return __builtin_amdgcn_ds_bpermute(src << 2, v); }
WV_DEV unsigned shfl(unsigned v, int src) { return unsigned(shfl(int(v), src)); }
WV_DEV uint64_t shfl(uint64_t v, int src)
{
  const unsigned lo = shfl(unsigned(v), src), hi = shfl(unsigned(v >> 32), src);
  return (uint64_t(hi) << 32) | lo;
}

/// wave-uniform source lane (v_readlane_b32)
WV_DEV int readlane(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
WV_DEV unsigned readlane(unsigned v, int src) { return unsigned(__builtin_amdgcn_readlane(int(v), src)); }
WV_DEV uint64_t readlane(uint64_t v, int src)
{
  return (uint64_t(readlane(unsigned(v >> 32), src)) << 32) | readlane(unsigned(v), src);
}
WV_DEV int first(int v) { return __builtin_amdgcn_readfirstlane(v); }
WV_DEV unsigned first(unsigned v) { return unsigned(__builtin_amdgcn_readfirstlane(int(v))); }

WV_DEV uint64_t ballot(bool p) { return __ballot(p); }
WV_DEV bool any(bool p) { return __ballot(p) != 0; }

/// makes this wave's earlier LDS/global writes visible to its other lanes.  A wavefront executes in lock step, so no
/// s_barrier is needed (and none may be used: sibling wavefronts of the workgroup run unrelated work items);
/// the fence drains this wave's outstanding memory operations.
WV_DEV void sync()
{
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

WV_DEV unsigned atomic_add(unsigned* p, unsigned v) { return atomicAdd(p, v); }
WV_DEV unsigned atomic_cas(unsigned* p, unsigned cmp, unsigned v) { return atomicCAS(p, cmp, v); }
WV_DEV unsigned atomic_or(unsigned* p, unsigned v) { return atomicOr(p, v); }
WV_DEV unsigned atomic_sub(unsigned* p, unsigned v) { return atomicSub(p, v); }
WV_DEV unsigned atomic_and(unsigned* p, unsigned v) { return atomicAnd(p, v); }
WV_DEV unsigned atomic_exch(unsigned* p, unsigned v) { return atomicExch(p, v); }
WV_DEV unsigned atomic_min(unsigned* p, unsigned v) { return atomicMin(p, v); }
WV_DEV unsigned atomic_max(unsigned* p, unsigned v) { return atomicMax(p, v); }
WV_DEV unsigned long long atomic_or(unsigned long long* p, unsigned long long v) { return atomicOr(p, v); }
WV_DEV unsigned long long atomic_add(unsigned long long* p, unsigned long long v) { return atomicAdd(p, v); }
/// L1-bypassing load of a word other lanes update with atomics
WV_DEV unsigned atomic_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
WV_DEV unsigned long long atomic_load(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
/// system-scope load of a word in fine-grained memory that the copy engine updates while the kernel runs
WV_DEV unsigned atomic_load_system(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
/// back-off inside a polling loop
WV_DEV void sleep() { __builtin_amdgcn_s_sleep(32); }
/// issue priority of this wave among the waves of its SIMD (s_setprio 0..3; wave-uniform argument)
WV_DEV void setprio(const unsigned level)
{
  if (level >= 3)
    __builtin_amdgcn_s_setprio(3);
  else if (level == 2)
    __builtin_amdgcn_s_setprio(2);
  else if (level == 1)
    __builtin_amdgcn_s_setprio(1);
  else
    __builtin_amdgcn_s_setprio(0);
}
/// drop this CU's (possibly stale) L1 lines: needed before plain re-reads of memory that was updated by L2 atomics
WV_DEV void fence_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }

/// shader-clock timestamp (s_memtime) for the optional per-phase profile
WV_DEV uint64_t clock() { return uint64_t(__builtin_readcyclecounter()); }

// packed 16-bit arithmetic: two int16 / uint16 per register (v_pk_add_i16 clamp, v_pk_max_i16, v_pk_min_u16, v_pk_mad_u16)
typedef short          pk_s2 __attribute__((ext_vector_type(2)));
typedef unsigned short pk_u2 __attribute__((ext_vector_type(2)));
WV_DEV uint32_t pk_add_sat_i16(uint32_t a, uint32_t b)
{
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(pk_s2, a), __builtin_bit_cast(pk_s2, b)));
}
WV_DEV uint32_t pk_max_i16(uint32_t a, uint32_t b)
{
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(pk_s2, a), __builtin_bit_cast(pk_s2, b)));
}
// (these two as the instruction itself: from the vector builtins the compiler rewrites min(x, 1) * c + d into compares and selects
// per half -- six instructions where two do)
WV_DEV uint32_t pk_min_u16(uint32_t a, uint32_t b)
{
  uint32_t r;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
WV_DEV uint32_t pk_mad_u16(uint32_t a, uint32_t b, uint32_t c)
{
  uint32_t r;
  asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

WV_DEV int popc(unsigned v) { return __popc(v); }
WV_DEV int popc(uint64_t v) { return __popcll(v); }
WV_DEV int ctz(uint64_t v) { return __builtin_ctzll(v); }  // v != 0
WV_DEV int clz(uint64_t v) { return __builtin_clzll(v); }  // v != 0

}  // namespace wv
#endif
