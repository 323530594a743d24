// LDS ceilings for the two assembler kernels of the metric's path, measured the way valu_ceiling.hip measured the
// aligner's VALU ceiling.  Developer tool; not part of the product or the tests.
//
//   gather   what graph_kernel's table pass and link phase do: every lane reads 16 bytes (ds_read_b128) at a RANDOM 16-byte slot of an
//            80 KB region of LDS (the workgroup's share: two 8-wave workgroups per CU = 4 waves per SIMD, as graph_kernel runs), 4 such
//            reads in flight per lane, nothing else in the loop but the address hash.  Result: wave-level 16-byte gathers per second
//            per CU -> the ceiling for "probes per locus / time".  Also with sequential (conflict-free) addresses: what the random
//            addresses cost.
//   chain    what contig_kernel's walk step does: one lane-private DEPENDENT chain of LDS reads (the next address comes out of the data:
//            8-byte records, ds_read_b64), one wave per workgroup, 8 / 3 / 1 workgroups per CU (contig_kernel's 20 KB / 53 KB LDS classes
//            and a whole-CU class).  Result: ns per dependent read per wave -> the floor of a walk step that needs R dependent reads.
//
//   hipcc --offload-arch=gfx950 -O3 -o lds_ceiling lds_ceiling.hip ; ./lds_ceiling [only: gather|chain]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
      std::exit(1);                                                                \
    }                                                                              \
  } while (0)

extern __shared__ __attribute__((aligned(16))) char dyn_lds[];

__device__ inline uint32_t mix(uint32_t h)
{
  h ^= h >> 16;
  h *= 0x7feb352du;
  h ^= h >> 15;
  h *= 0x846ca68bu;
  h ^= h >> 16;
  return h;
}

// RANDOM: 1 = hashed slot per lane, 0 = lane l reads slot (base + l): consecutive 16-byte slots, no bank conflict beyond the b128 pattern
template <int RANDOM>
__global__ __launch_bounds__(1024) void lds_gather(const int iters, const unsigned slots, uint32_t* sink)
{
  uint4*         tbl = reinterpret_cast<uint4*>(dyn_lds);
  const unsigned tid = threadIdx.x;
  for (unsigned i = tid; i < slots; i += blockDim.x) tbl[i] = make_uint4(i, i * 3u, i * 5u, i * 7u);
  __syncthreads();
  uint32_t h = mix(blockIdx.x * 977u + tid + 1u), acc = 0;
  for (int i = 0; i < iters; ++i) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned s = RANDOM ? (mix(h + uint32_t(u) * 0x9e3779b9u) % slots) : ((h + uint32_t(u) * 64u + (tid & 63u)) % slots);
      v[u]             = tbl[s];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    h = RANDOM ? mix(h + 0x1234567u) : (h + 256u);
  }
  if (acc == 0x12345u) sink[0] = acc;
}

__global__ __launch_bounds__(64) void lds_chain(const int iters, const unsigned recs, uint32_t* sink)
{
  uint64_t*      tbl  = reinterpret_cast<uint64_t*>(dyn_lds);
  const unsigned lane = threadIdx.x;
  for (unsigned i = lane; i < recs; i += 64) tbl[i] = (uint64_t(mix(i * 2654435761u + 17u) % recs)) | (uint64_t(i) << 32);
  __syncthreads();
  unsigned at  = mix(blockIdx.x * 64u + lane + 1u) % recs;
  uint32_t acc = 0;
  for (int i = 0; i < iters; ++i) {
    const uint64_t r = tbl[at];  // ds_read_b64, the next index is in the low half
    at               = unsigned(r & 0xffffffffu);
    acc += unsigned(r >> 32);
  }
  if (acc == 0x12345u) sink[0] = acc + at;
}

int main(int argc, char** argv)
{
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int   cus  = prop.multiProcessorCount;
  const char* only = argc > 1 ? argv[1] : "";
  uint32_t*   sink = nullptr;
  CK(hipMalloc(&sink, 64));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(lds_gather<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(lds_gather<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(lds_chain), hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
  const double clk = double(prop.clockRate) * 1e3;  // Hz
  std::printf("# %s, %d CUs, %.2f GHz\n", prop.gcnArchName, cus, clk * 1e-9);
  if (!*only || !std::strcmp(only, "gather")) {
    std::printf("# gather: 16-byte LDS reads per lane, 4 in flight; workgroups of W waves with B bytes of LDS each, G workgroups per CU\n");
    std::printf("# pattern waves_per_wg lds_bytes_per_wg wgs_per_cu | G wave-gathers/s (device)  M wave-gathers/s per CU  clocks per wave-gather per CU  GB/s of LDS per CU\n");
    struct Cfg {
      int waves, ldsBytes, perCu;
    };
    const Cfg cfgs[] = {{8, 81920, 2}, {8, 81920, 1}, {16, 163840, 1}, {4, 40960, 4}};
    for (int random = 1; random >= 0; --random)
      for (const Cfg& c : cfgs) {
        const int      grid  = cus * c.perCu;
        const int      iters = 4000;
        const unsigned slots = unsigned(c.ldsBytes / 16);
        auto           run   = [&]() {
          if (random)
            hipLaunchKernelGGL(lds_gather<1>, dim3(grid), dim3(64 * c.waves), c.ldsBytes, 0, iters, slots, sink);
          else
            hipLaunchKernelGGL(lds_gather<0>, dim3(grid), dim3(64 * c.waves), c.ldsBytes, 0, iters, slots, sink);
        };
        run();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        run();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        const double gathers = double(grid) * c.waves * iters * 4;  // wave-level
        const double rate    = gathers / (ms * 1e-3);
        std::printf("%-10s %2d %6d %d | %8.2f  %8.1f  %6.2f  %8.1f\n", random ? "random" : "sequential", c.waves, c.ldsBytes, c.perCu, rate * 1e-9, rate / cus * 1e-6,
                    clk / (rate / cus), rate / cus * 64 * 16 * 1e-9);
      }
  }
  if (!*only || !std::strcmp(only, "chain")) {
    std::printf("# chain: one dependent ds_read_b64 chain per lane, single-wave workgroups with B bytes of LDS, G per CU\n");
    std::printf("# lds_bytes_per_wg wgs_per_cu | ns per dependent read (per wave)  clocks per read  M wave-reads/s per CU\n");
    const int cfg[][2] = {{20480, 8}, {54272, 3}, {81920, 2}, {163840, 1}};
    for (const auto& c : cfg) {
      const int      grid  = cus * c[1];
      const int      iters = 200000;
      const unsigned recs  = unsigned(c[0] / 8);
      auto           run   = [&]() { hipLaunchKernelGGL(lds_chain, dim3(grid), dim3(64), c[0], 0, iters, recs, sink); };
      run();
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(a));
      run();
      CK(hipEventRecord(b));
      CK(hipEventSynchronize(b));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, a, b));
      const double ns = double(ms) * 1e6 / iters;
      std::printf("%6d %d | %7.2f  %6.1f  %8.1f\n", c[0], c[1], ns, ns * clk * 1e-9, double(c[1]) / ns * 1e3);
    }
  }
  return 0;
}
