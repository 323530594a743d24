// VALU issue ceiling of gfx950 for the instructions the aligners are made of.
//
// Every instruction is an `asm volatile` statement, so the compiler can neither fuse, reorder away nor drop anything (the ISA of
// the loops is dumped next to the numbers: `hipcc -save-temps`).  Swept: the instruction (32-bit integer add / max, packed 16-bit
// add / max, fp32 fma, packed fp32 fma), the number of INDEPENDENT dependency chains per wave (1 = one dependent chain), and the
// number of waves per SIMD (1, 2, 4, 8).  Reported: wave-instructions per second over the chip and clocks per wave64 instruction
// per SIMD at the clock the run sustained (s_memtime delta over s_memrealtime delta, 100 MHz reference).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

enum {
  OP_ADD_U32 = 0, OP_MAX_I32, OP_PK_ADD_I16, OP_PK_MAX_I16, OP_FMA_F32, OP_PK_FMA_F32,
  // the rest of what the aligners' inner loops are made of, and candidates to replace the half-rate ones
  OP_MAX3_I32, OP_MAX_F32, OP_MAX3_F32, OP_ADD_F32, OP_MOV_DPP, OP_AND_B32, OP_AND_OR_B32, OP_LSHL_OR_B32, OP_CNDMASK, OP_MAX_U32,
  OP_MAX_I16, OP_ADD3_U32, OP_PK_ADD_U16, OP_BFE_U32, OP_PERM_B32, OP_MAX_U16, OP_XOR_B32, OP_SUB_U32, OP_LSHLREV_B32, OP_MAD_U32_U24,
  N_OPS
};
static const char* kOpName[N_OPS] = {"v_add_u32", "v_max_i32", "v_pk_add_i16 clamp", "v_pk_max_i16", "v_fma_f32", "v_pk_fma_f32",
                                     "v_max3_i32", "v_max_f32", "v_max3_f32", "v_add_f32", "v_mov_b32_dpp wave_shr:1", "v_and_b32", "v_and_or_b32", "v_lshl_or_b32",
                                     "v_cndmask_b32", "v_max_u32", "v_max_i16", "v_add3_u32", "v_pk_add_u16", "v_bfe_u32", "v_perm_b32", "v_max_u16",
                                     "v_xor_b32", "v_sub_u32", "v_lshlrev_b32", "v_mad_u32_u24"};

// one asm statement = 8 instructions over CHAINS accumulators (acc[i % CHAINS]): nothing the compiler could put in between
#define VC_BLOCK8(INS, TAIL)                                                                                                   \
  asm volatile(INS " %0, %0, %8" TAIL "\n\t" INS " %1, %1, %8" TAIL "\n\t" INS " %2, %2, %8" TAIL "\n\t" INS " %3, %3, %8" TAIL "\n\t" \
               INS " %4, %4, %8" TAIL "\n\t" INS " %5, %5, %8" TAIL "\n\t" INS " %6, %6, %8" TAIL "\n\t" INS " %7, %7, %8" TAIL         \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                                  \
               : "v"(b))
#define VC_BLOCK4(INS, TAIL)                                                                                                   \
  asm volatile(INS " %0, %0, %4" TAIL "\n\t" INS " %1, %1, %4" TAIL "\n\t" INS " %2, %2, %4" TAIL "\n\t" INS " %3, %3, %4" TAIL "\n\t" \
               INS " %0, %0, %4" TAIL "\n\t" INS " %1, %1, %4" TAIL "\n\t" INS " %2, %2, %4" TAIL "\n\t" INS " %3, %3, %4" TAIL         \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)                                                                          \
               : "v"(b))
#define VC_BLOCK2(INS, TAIL)                                                                                                   \
  asm volatile(INS " %0, %0, %2" TAIL "\n\t" INS " %1, %1, %2" TAIL "\n\t" INS " %0, %0, %2" TAIL "\n\t" INS " %1, %1, %2" TAIL "\n\t" \
               INS " %0, %0, %2" TAIL "\n\t" INS " %1, %1, %2" TAIL "\n\t" INS " %0, %0, %2" TAIL "\n\t" INS " %1, %1, %2" TAIL         \
               : "+v"(a0), "+v"(a1)                                                                                              \
               : "v"(b))
#define VC_BLOCK1(INS, TAIL)                                                                                                   \
  asm volatile(INS " %0, %0, %1" TAIL "\n\t" INS " %0, %0, %1" TAIL "\n\t" INS " %0, %0, %1" TAIL "\n\t" INS " %0, %0, %1" TAIL "\n\t" \
               INS " %0, %0, %1" TAIL "\n\t" INS " %0, %0, %1" TAIL "\n\t" INS " %0, %0, %1" TAIL "\n\t" INS " %0, %0, %1" TAIL         \
               : "+v"(a0)                                                                                                        \
               : "v"(b))
#define VC_DISPATCH(INS, TAIL)                   \
  do {                                           \
    if (CHAINS == 8) VC_BLOCK8(INS, TAIL);       \
    else if (CHAINS == 4) VC_BLOCK4(INS, TAIL);  \
    else if (CHAINS == 2) VC_BLOCK2(INS, TAIL);  \
    else VC_BLOCK1(INS, TAIL);                   \
  } while (0)

// CHAINS independent accumulators, 64 instructions per loop trip
template <int OP, int CHAINS, typename T>
__device__ __forceinline__ T body(const T seed, const T b, const int iters)
{
  T a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (OP == OP_ADD_U32) VC_DISPATCH("v_add_u32", "");
      if (OP == OP_MAX_I32) VC_DISPATCH("v_max_i32", "");
      if (OP == OP_PK_ADD_I16) VC_DISPATCH("v_pk_add_i16", " clamp");
      if (OP == OP_PK_MAX_I16) VC_DISPATCH("v_pk_max_i16", "");
      if (OP == OP_FMA_F32) VC_DISPATCH("v_fma_f32", ", 1.0");
      if (OP == OP_PK_FMA_F32) VC_DISPATCH("v_pk_fma_f32", ", 1.0");
      if (OP == OP_MAX3_I32) VC_DISPATCH("v_max3_i32", ", 3");
      if (OP == OP_MAX_F32) VC_DISPATCH("v_max_f32", "");
      if (OP == OP_MAX3_F32) VC_DISPATCH("v_max3_f32", ", 1.0");
      if (OP == OP_ADD_F32) VC_DISPATCH("v_add_f32", "");
      if (OP == OP_AND_B32) VC_DISPATCH("v_and_b32", "");
      if (OP == OP_AND_OR_B32) VC_DISPATCH("v_and_or_b32", ", 7");
      if (OP == OP_LSHL_OR_B32) VC_DISPATCH("v_lshl_or_b32", ", 7");
      if (OP == OP_CNDMASK) VC_DISPATCH("v_cndmask_b32", ", vcc");
      if (OP == OP_MAX_U32) VC_DISPATCH("v_max_u32", "");
      if (OP == OP_MAX_I16) VC_DISPATCH("v_max_i16", "");
      if (OP == OP_ADD3_U32) VC_DISPATCH("v_add3_u32", ", 3");
      if (OP == OP_PK_ADD_U16) VC_DISPATCH("v_pk_add_u16", "");
      if (OP == OP_BFE_U32) VC_DISPATCH("v_bfe_u32", ", 3");
      if (OP == OP_PERM_B32) VC_DISPATCH("v_perm_b32", ", 3");
      if (OP == OP_MAX_U16) VC_DISPATCH("v_max_u16", "");
      if (OP == OP_XOR_B32) VC_DISPATCH("v_xor_b32", "");
      if (OP == OP_SUB_U32) VC_DISPATCH("v_sub_u32", "");
      if (OP == OP_LSHLREV_B32) VC_DISPATCH("v_lshlrev_b32", "");
      if (OP == OP_MAD_U32_U24) VC_DISPATCH("v_mad_u32_u24", ", 3");
      if (OP == OP_MOV_DPP) {
        // (two operands only: %0 <- dpp(%0); 8 independent moves, or one chain)
        if (CHAINS == 1)
          asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                       "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                       "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                       "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf"
                       : "+v"(a0));
        else
          asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                       "v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                       "v_mov_b32_dpp %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %5, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                       "v_mov_b32_dpp %6, %7 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %7, %0 wave_shr:1 row_mask:0xf bank_mask:0xf"
                       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      }
    }
  }
  return a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

template <int OP, int CHAINS>
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned long long* clk, unsigned seed, int iters)
{
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  if (OP == OP_PK_FMA_F32) {
    const unsigned long long x = body<OP, CHAINS, unsigned long long>((unsigned long long)(threadIdx.x) * 0x3f8000003f800000ull + seed, 0x3f8000013f800001ull, iters);
    out[blockIdx.x * blockDim.x + threadIdx.x] = unsigned(x) ^ unsigned(x >> 32);
  } else {
    out[blockIdx.x * blockDim.x + threadIdx.x] = body<OP, CHAINS, unsigned>(threadIdx.x * 2654435761u + seed, 0x00010001u, iters);
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    clk[0] = __builtin_readcyclecounter() - t0;
    clk[1] = wall_clock64() - r0;
  }
}

typedef void (*kern_t)(unsigned*, unsigned long long*, unsigned, int);
template <int OP>
static kern_t pick(const int chains)
{
  switch (chains) {
    case 1: return k<OP, 1>;
    case 2: return k<OP, 2>;
    case 4: return k<OP, 4>;
    default: return k<OP, 8>;
  }
}
template <int OP>
static kern_t pickFrom(const int op, const int chains)
{
  if (op == OP) return pick<OP>(chains);
  if constexpr (OP + 1 < N_OPS) return pickFrom<OP + 1>(op, chains);
  return nullptr;
}
static kern_t pickOp(const int op, const int chains) { return pickFrom<0>(op, chains); }

int main(int argc, char** argv)
{
  const bool quick = argc > 1 && atoi(argv[1]) == 1;  // the counter passes: one configuration per instruction
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  unsigned*           d;
  unsigned long long* dclk;
  hipMalloc(&d, size_t(cus) * 8 * 256 * 4);
  hipMalloc(&dclk, 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  printf("# %s, %d CUs, nominal %d MHz\n", prop.name, cus, prop.clockRate / 1000);
  printf("# op | chains | waves/SIMD | ms | G wave-instr/s (chip) | shader MHz (s_memtime/s_memrealtime) | clocks per wave64 instr per SIMD\n");
  const int iters = 4000;
  for (int op = 0; op < N_OPS; ++op)
    for (int chains = 1; chains <= 8; chains *= 2)
      for (int wps = 1; wps <= 8; wps *= 2) {
        if (quick && !(chains == 8 && wps == 4)) continue;
        if (op >= OP_MAX3_I32 && !((chains == 1 || chains == 8) && wps != 2)) continue;
        const int    grid = cus * wps;  // 256-thread workgroups: one wave per SIMD each
        const kern_t fn   = pickOp(op, chains);
        float        best = 1e30f;
        unsigned long long hclk[2] = {0, 0};
        for (int rep = 0; rep < 3; ++rep) {
          hipEventRecord(e0);
          hipLaunchKernelGGL(fn, dim3(grid), dim3(256), 0, 0, d, dclk, 1u + rep, iters);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          if (ms < best) {
            best = ms;
            hipMemcpy(hclk, dclk, 16, hipMemcpyDeviceToHost);
          }
        }
        const double instr = double(grid) * 4 * double(iters) * 64;  // wave-instructions of the timed loop
        const double mhz   = hclk[1] ? double(hclk[0]) / double(hclk[1]) * 100.0 : 0.0;
        const double perS  = instr / (best * 1e-3);
        const double clkPer = (mhz > 0 ? mhz * 1e6 : 2.4e9) * double(cus) * 4 / perS;
        printf("%-20s | %d | %d | %.3f | %.1f | %.0f | %.2f\n", kOpName[op], chains, wps, best, perS / 1e9, mhz, clkPer);
      }
  return 0;
}
