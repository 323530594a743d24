// Ceiling for the access pattern of assemble_kernel: wave-level SCATTERED loads into a private per-wave
// slab (2.3 MB x 4096 resident waves, far past L2 / MALL).  Measures wave-loads per second for
//   mode 0  independent loads (8 in flight per wave)    -> the vector-memory issue / L1 tag ceiling
//   mode 1  dependent chain (next address from the data) -> what latency alone allows at this occupancy
// with 64 or 4 active lanes and 4 / 16 bytes per lane.  Developer tool; not part of the product or the tests.
//   hipcc --offload-arch=gfx950 -O3 -o gather_ceiling gather_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                     \
      std::exit(1);                                                                    \
    }                                                                                  \
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

template <int BYTES, int MODE>
__global__ __launch_bounds__(64) void gather(const uint32_t* base, uint64_t slabDwords, int iters, int activeLanes, uint32_t* sink)
{
  const uint32_t* slab = base + uint64_t(blockIdx.x) * slabDwords;
  const unsigned  lane = threadIdx.x;
  uint32_t        acc  = 0;
  if (int(lane) >= activeLanes) return;
  const uint32_t mask = uint32_t(slabDwords / 4 - 1);  // 16-byte granules, slabDwords/4 is a power of two
  uint32_t       h    = mix(blockIdx.x * 64u + lane + 1u);
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t g = mix(h + uint32_t(u) * 0x9e3779b9u) & mask;
        if (BYTES == 4) {
          acc += slab[uint64_t(g) * 4];
        } else {
          const uint4 v = *reinterpret_cast<const uint4*>(slab + uint64_t(g) * 4);
          acc += v.x ^ v.y ^ v.z ^ v.w;
        }
      }
      h = mix(h + 0x1234567u);
    } else {
      const uint32_t g = h & mask;
      uint32_t       got;
      if (BYTES == 4) {
        got = slab[uint64_t(g) * 4];
      } else {
        const uint4 v = *reinterpret_cast<const uint4*>(slab + uint64_t(g) * 4);
        got           = v.x ^ v.y ^ v.z ^ v.w;
      }
      acc += got;
      h = mix(h + got + 1u);
    }
  }
  if (acc == 0x12345u) sink[0] = acc;
}

int main(int argc, char** argv)
{
  int cus = 256;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  cus = prop.multiProcessorCount;
  const uint64_t slabDwords = (argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 2ull << 20) / 4;  // bytes -> dwords (2 MiB default)
  const int      maxWaves   = cus * 16;
  uint32_t *     d = nullptr, *sink = nullptr;
  CK(hipMalloc(&d, uint64_t(maxWaves) * slabDwords * 4));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(d, 1, uint64_t(maxWaves) * slabDwords * 4));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  std::printf("# %d CUs, slab %.2f MiB per wave\n# mode bytes lanes waves_per_cu  G_wave_loads_per_s  ns_per_load_per_wave  cycles_per_load_per_CU(2.4GHz)\n", cus,
              double(slabDwords) * 4 / 1048576.0);
  // optional: argv[2..5] = mode bytes lanes waves_per_cu -> just that configuration (for rocprofv3 --pmc passes)
  const bool one = argc > 5;
  for (int mode = 0; mode < 2; ++mode)
    for (int bytes : {4, 16})
      for (int lanes : {64, 16, 4})
        for (int wpc : {4, 8, 16}) {
          if (one && (mode != std::atoi(argv[2]) || bytes != std::atoi(argv[3]) || lanes != std::atoi(argv[4]) || wpc != std::atoi(argv[5]))) continue;
          const int grid  = cus * wpc;
          const int iters = mode == 0 ? 2000 : 4000;
          auto      run   = [&]() {
            if (mode == 0 && bytes == 4) hipLaunchKernelGGL((gather<4, 0>), dim3(grid), dim3(64), 0, 0, d, slabDwords, iters, lanes, sink);
            if (mode == 0 && bytes == 16) hipLaunchKernelGGL((gather<16, 0>), dim3(grid), dim3(64), 0, 0, d, slabDwords, iters, lanes, sink);
            if (mode == 1 && bytes == 4) hipLaunchKernelGGL((gather<4, 1>), dim3(grid), dim3(64), 0, 0, d, slabDwords, iters, lanes, sink);
            if (mode == 1 && bytes == 16) hipLaunchKernelGGL((gather<16, 1>), dim3(grid), dim3(64), 0, 0, d, slabDwords, iters, lanes, sink);
          };
          run();
          CK(hipDeviceSynchronize());
          CK(hipEventRecord(a));
          run();
          CK(hipEventRecord(b));
          CK(hipEventSynchronize(b));
          float ms = 0;
          CK(hipEventElapsedTime(&ms, a, b));
          const double loads = double(grid) * iters * (mode == 0 ? 8 : 1);
          const double rate  = loads / (ms * 1e-3);
          std::printf("%d %2d %2d %2d  %8.2f  %8.1f  %8.1f\n", mode, bytes, lanes, wpc, rate * 1e-9, double(ms) * 1e6 / (double(iters) * (mode == 0 ? 8 : 1)),
                      2.4e9 / (rate / cus));
        }
  return 0;
}
