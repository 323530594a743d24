// issue rate of packed 16-bit integer VALU ops (v_pk_add_i16 clamp, v_pk_max_i16) against plain 32-bit ops (v_add_u32, v_max_i32)
// on gfx950: the packed pair aligner (align_pair.hpp) assumes they issue at the same rate.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short pk_s2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned seed, int iters)
{
  unsigned a = threadIdx.x * 2654435761u + seed, b = a ^ 0x5bd1e995u, c = a + 77, d = b + 99, e = a * 3, f = b * 5, g = c * 7, h = d * 9;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (MODE == 0) {  // 8 independent chains of 32-bit add / max
        a = a + b; c = max(int(c), int(d)); e = e + f; g = max(int(g), int(h));
        b = b + a; d = max(int(d), int(c)); f = f + e; h = max(int(h), int(g));
      } else {  // same with packed saturating add / packed max
        pk_s2 A = __builtin_bit_cast(pk_s2, a), B = __builtin_bit_cast(pk_s2, b), C = __builtin_bit_cast(pk_s2, c), D = __builtin_bit_cast(pk_s2, d);
        pk_s2 E = __builtin_bit_cast(pk_s2, e), F = __builtin_bit_cast(pk_s2, f), G = __builtin_bit_cast(pk_s2, g), H = __builtin_bit_cast(pk_s2, h);
        A = __builtin_elementwise_add_sat(A, B); C = __builtin_elementwise_max(C, D); E = __builtin_elementwise_add_sat(E, F); G = __builtin_elementwise_max(G, H);
        B = __builtin_elementwise_add_sat(B, A); D = __builtin_elementwise_max(D, C); F = __builtin_elementwise_add_sat(F, E); H = __builtin_elementwise_max(H, G);
        a = __builtin_bit_cast(unsigned, A); b = __builtin_bit_cast(unsigned, B); c = __builtin_bit_cast(unsigned, C); d = __builtin_bit_cast(unsigned, D);
        e = __builtin_bit_cast(unsigned, E); f = __builtin_bit_cast(unsigned, F); g = __builtin_bit_cast(unsigned, G); h = __builtin_bit_cast(unsigned, H);
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
}
int main()
{
  unsigned* d;
  hipMalloc(&d, 4096 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000, grid = 256 * 4;  // 16 waves per CU
  for (int mode = 0; mode < 2; ++mode)
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, d, 1u, iters);
      else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, d, 1u, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double instr = double(grid) * 4 * iters * 16 * 8;  // wave-instructions
      printf("%s: %.3f ms, %.1f G wave-instr/s (%.2f clk per instr per SIMD at 2.4 GHz)\n", mode ? "v_pk_add_i16 clamp / v_pk_max_i16" : "v_add_u32 / v_max_i32", ms,
             instr / ms / 1e6, 1024.0 * 2.4e9 / (instr / (ms * 1e-3)));
    }
  return 0;
}
