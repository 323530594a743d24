// What would folding the k-mer table slot and the node record into ONE 64-byte entry buy?
// This models the table pass of assemble_kernel: one wave per pile, a private slab per wave (far past L2 / MALL in total),
// every lane inserts k-mer instances with the duplicate structure of a read pile (coverage ~30 plus singleton error words):
//   layout A (today): 8-byte slot {first occurrence, node id} -> key compare against the codes of the first occurrence ->
//                     atomic OR into the node's 64-byte record (support set)                    = three arrays, three sectors
//   layout B:         one 64-byte entry {first occurrence, id, count, links, support}: probe + atomic OR in the same sector
// Reports time per instance for both.  Developer tool, not part of the product or the tests.
//   hipcc --offload-arch=gfx950 -O3 -o table_layout table_layout.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
      std::exit(1);                                                                \
    }                                                                              \
  } while (0)

__device__ inline uint32_t mix(uint32_t h)
{
  h ^= h >> 16;
  h *= 0x7feb352du;
  h ^= h >> 15;
  h *= 0x846ca68bu;
  h ^= h >> 16;
  return h;
}

struct Params {
  uint8_t* ws;
  uint64_t stride;      // bytes per wave
  uint32_t slots;       // power of two
  uint32_t instances;   // per pile
  uint32_t distinctHot; // "true" words (each seen ~coverage times)
  uint32_t piles;       // piles per wave
};

// word id of instance i: 55 % of the instances hit one of the hot words, the rest are singletons
__device__ inline uint32_t wordOf(uint32_t pile, uint32_t i, const Params& P)
{
  const uint32_t r = mix(pile * 0x9e3779b9u + i);
  if ((r & 127u) < 70u) return mix(pile ^ ((r >> 8) % P.distinctHot)) | 1u;
  return mix(r ^ 0x5bd1e995u) | 1u;
}

template <int LAYOUT>
__global__ __launch_bounds__(256) void tablePass(const Params P, uint32_t* sink)
{
  const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  uint8_t*       slab = P.ws + uint64_t(wave) * P.stride;
  const uint32_t mask = P.slots - 1;
  // layout A: slots (8 B) | codes (4 B per instance) | records (64 B per slot);  layout B: entries (64 B per slot)
  unsigned long long* slotsA = reinterpret_cast<unsigned long long*>(slab);
  uint32_t*           codesA = reinterpret_cast<uint32_t*>(slab + 8ull * P.slots);
  uint8_t*            recA   = slab + 8ull * P.slots + 4ull * P.instances + 64;
  uint8_t*            entB   = slab;
  uint32_t            acc    = 0;
  for (uint32_t pile = 0; pile < P.piles; ++pile) {
    // reset (the real pass memsets its table per word length)
    if (LAYOUT == 0) {
      for (uint32_t s = lane; s < P.slots; s += 64) slotsA[s] = ~0ull;
      for (uint32_t i = lane; i < P.instances; i += 64) codesA[i] = wordOf(pile + wave * 977u, i, P);
    } else {
      for (uint32_t s = lane; s < P.slots * 4; s += 64) reinterpret_cast<uint4*>(entB)[s] = make_uint4(~0u, ~0u, 0u, 0u);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i0 = 0; i0 < P.instances; i0 += 64) {
      const uint32_t i = i0 + lane;
      if (i >= P.instances) continue;
      const uint32_t w = wordOf(pile + wave * 977u, i, P);
      uint32_t       s = mix(w) & mask;
      for (uint32_t probe = 0; probe <= mask; ++probe) {
        if (LAYOUT == 0) {
          const unsigned long long pr = __hip_atomic_load(&slotsA[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          uint32_t cur = uint32_t(pr);
          if (cur == ~0u) {
            cur = atomicCAS(reinterpret_cast<unsigned int*>(&slotsA[s]), ~0u, i);
            if (cur == ~0u) {
              for (int q = 0; q < 4; ++q) reinterpret_cast<unsigned long long*>(recA + 64ull * s + 32)[q] = 0;
              cur = i;
            }
          }
          if (codesA[cur] == w) break;  // key compare through the first occurrence
        } else {
          uint32_t* e   = reinterpret_cast<uint32_t*>(entB + 64ull * s);
          uint32_t  cur = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (cur == ~0u) {
            cur = atomicCAS(e, ~0u, w);  // the entry carries its own key word (here: the id; the kernel: the first occurrence + a tag)
            if (cur == ~0u) cur = w;
          }
          if (cur == w) break;
        }
        s = (s + 1) & mask;
      }
      // support OR: read index = i / 128
      unsigned long long* sup = reinterpret_cast<unsigned long long*>((LAYOUT == 0 ? recA : entB) + 64ull * s + 32) + ((i >> 13) & 3);
      acc += uint32_t(atomicOr(sup, 1ull << ((i >> 7) & 63)));
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (acc == 0x12345u) sink[0] = acc;
}

int main()
{
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  Params P;
  P.instances   = 9600;   // 80 reads x (150 - 31 + 1)
  P.distinctHot = 1500;
  P.slots       = 8192;
  P.piles       = 3;
  const int waves = prop.multiProcessorCount * 16;
  P.stride        = (8ull * P.slots + 4ull * P.instances + 64 + 64ull * P.slots + 255) & ~255ull;
  CK(hipMalloc(&P.ws, P.stride * waves));
  uint32_t* sink;
  CK(hipMalloc(&sink, 64));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int layout = 0; layout < 2; ++layout) {
    auto run = [&] {
      if (layout == 0)
        hipLaunchKernelGGL((tablePass<0>), dim3(waves / 4), dim3(256), 0, 0, P, sink);
      else
        hipLaunchKernelGGL((tablePass<1>), dim3(waves / 4), dim3(256), 0, 0, P, sink);
    };
    run();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    run();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    const double inst = double(waves) * P.piles * P.instances;
    std::printf("layout %c: %.3f ms for %d waves x %u piles x %u instances = %.2f G instances/s (%.1f ns per wave-step of 64)\n", layout ? 'B' : 'A', ms,
                waves, P.piles, P.instances, inst / (ms * 1e-3) * 1e-9, ms * 1e6 / (double(P.piles) * (P.instances / 64.0)));
  }
  return 0;
}
