// Micro-benchmark of vector-instruction issue on gfx950: how many wave-instructions per cycle one SIMD
// issues for the instruction kinds the BVH node test is made of, at 1, 2, 4, 6 and 8 waves per SIMD.
//
// Every wave runs `iters` iterations of 64 instructions of one kind on eight independent register
// chains (inline asm, so the instruction is the one named), stamps the shader clock (s_memtime) before
// and after, and the host reports
//   cycles per wave-instruction per SIMD = launch time x in-kernel clock / (waves per SIMD x instructions per wave)
// (in-kernel clock = s_memtime ticks per s_memrealtime tick x 100 MHz), the median wave's own cycles per
// instruction, and G wave-instructions/s for the chip.  The `node_test` rows time
// the library's own node test (node_test_words of csrc/intersect.h) on register-resident node words; its
// VALU instruction count per test comes from a counter pass (SQ_INSTS_VALU) over this program or from
// its disassembly.
//
// Waves per SIMD are set by construction, not by hoping for the dispatcher: a block of 256 k threads
// puts k waves on each of a CU's four SIMDs, and its dynamic LDS request (96 KB, or 64 KB for two
// blocks) lets exactly one (or two) blocks onto a CU; the grid is CUs x blocks per CU, so all are resident.
//
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/micro/valu_issue tools/micro/valu_issue.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include "../../path_tracer_amd/csrc/intersect.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum Op { FMA32, FMA64, CVT_UBYTE, MINMAX32, PK_FMA32, ADD64, MUL64, MAX3, CNDMASK, CMP32, MOV, AND_SHIFT, FMA_MIX, PERM, CNDMASK_SGPR, FMA32_AND, CVT_FMA, PERM_MIX, MOV_DPP, MAX_I32, MAX3_I32, CMP_I32, BFI, BFE, LSHL_ADD, ADD_F32, MUL_F32, OR_SDWA, CMP_CND_VCC, CVT_AND_FMA, MAXF_AND_FMA, CND_AND_FMA, MAXI_AND_FMA, MED3, CND_E64_VCC, CMP_CND_PAIRS, CND_VCC_SET, ADD_U32, N_OPS };
static const char *kOpName[N_OPS] = {"v_fma_f32", "v_fma_f64", "v_cvt_f32_ubyte0", "v_max_f32 / v_min_f32", "v_pk_fma_f32", "v_add_f64", "v_mul_f64",
                                     "v_max3_f32", "v_cndmask_b32 (vcc)", "v_cmp_le_f32", "v_mov_b32", "v_and_b32 / v_lshrrev_b32",
                                     "v_fma_mix_f32 (f16 subnormal source)", "v_perm_b32", "v_cndmask_b32 (sgpr pair)", "v_fma_f32 / v_and_b32 alternating",
                                     "4 v_cvt_f32_ubyteK + 4 v_fma_f32 (per 8)", "2 v_perm_b32 + 4 v_fma_mix_f32 (per 6, counted as 8)", "v_mov_b32_dpp quad_perm",
                                     "v_max_i32 / v_min_i32", "v_max3_i32", "v_cmp_le_i32", "v_bfi_b32", "v_bfe_u32", "v_lshl_add_u32", "v_add_f32", "v_mul_f32",
                                     "v_or_b32_sdwa (byte source)", "v_cmp_le_f32 vcc + 7 v_cndmask_b32 vcc (per 8)", "v_cvt_f32_ubyte0 / v_fma_f32 alternating",
                                     "v_max_f32 / v_fma_f32 alternating", "v_cndmask_b32 (sgpr pair) / v_fma_f32 alternating", "v_max_i32 / v_fma_f32 alternating", "v_med3_f32",
                                     "v_cndmask_b32_e64 (vcc as the sgpr pair)", "v_cmp_lt_f32 vcc / v_cndmask_b32_e32 vcc alternating",
                                     "v_cndmask_b32_e32 vcc, vcc set once by s_mov before the loop", "v_add_u32"};

struct WaveStamp {
  uint64_t cycles;    // s_memtime difference around the loop
  uint64_t realtime;  // s_memrealtime difference (100 MHz)
  uint64_t real0;     // s_memrealtime at the start of the loop (one clock for the whole chip)
  uint32_t hw_id, xcc_id;
};

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

// LANES: which lanes of every wave run the loop (the others wait at the end): 0 all, 1 the low 32, 2 the low 16, 3 the even ones,
// 4 lanes 0-15 and 32-47
template <int OP, int LANES = 0>
__global__ void __launch_bounds__(1024, 8) stream(int iters, WaveStamp *stamps, float *sink) {
  extern __shared__ char lds_pad[];  // only sizes the block's LDS request (blocks per CU)
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  float a[8];
  double d[8];
  float2 p[8];
  uint32_t u[8], h[8];
  const uint64_t sel = 0x5555aaaa3333ccccull ^ blockIdx.x;
  for (int k = 0; k < 8; k++) {
    h[k] = ((tid + k) & 0xffu) | (((tid * 7u + k) & 0xffu) << 16);
    a[k] = 1.0f + 1e-3f * (float)((tid + k) & 15);
    d[k] = 1.0 + 1e-3 * (double)((tid + k) & 15);
    p[k] = float2{a[k], a[k] + 0.5f};
    u[k] = tid * 2654435761u + k;
  }
  const float b = 1.0f - 1e-7f * (float)(tid & 3), c = 1e-6f;
  const double bd = 1.0 - 1e-12 * (double)(tid & 3), cd = 1e-9;
  const float2 bp{b, b}, cp{c, c};
  __syncthreads();
  const uint64_t t0 = __builtin_readcyclecounter();
  const uint64_t r0 = wall_clock64();
  const uint32_t ln = threadIdx.x & 63u;
  const bool runs = LANES == 0 ? true : LANES == 1 ? ln < 32u : LANES == 2 ? ln < 16u : LANES == 3 ? (ln & 1u) == 0u : (ln & 16u) == 0u;
  if (runs)
  for (int i = 0; i < iters; i++) {
    if (OP == FMA32) {
#define X(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
      REP64(X)
#undef X
    } else if (OP == FMA64) {
#define X(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[k]) : "v"(bd), "v"(cd));
      REP64(X)
#undef X
    } else if (OP == CVT_UBYTE) {
#define X(k) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "+v"(a[k]) : "v"(u[k]));
      REP64(X)
#undef X
    } else if (OP == MINMAX32) {
#define X(k) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c)); asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[(k + 4) & 7]) : "v"(b));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if (OP == PK_FMA32) {
#define X(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(bp), "v"(cp));
      REP64(X)
#undef X
    } else if (OP == ADD64) {
#define X(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[k]) : "v"(cd));
      REP64(X)
#undef X
    } else if (OP == MUL64) {
#define X(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[k]) : "v"(bd));
      REP64(X)
#undef X
    } else if (OP == MAX3) {
#define X(k) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
      REP64(X)
#undef X
    } else if (OP == CNDMASK) {
#define X(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b) : );
      REP64(X)
#undef X
    } else if (OP == CMP32) {
      // one block, so that the compiler puts no hazard s_nop between the compares
      asm volatile(".rept 8\n v_cmp_le_f32 vcc, %0, %8\n v_cmp_le_f32 vcc, %1, %8\n v_cmp_le_f32 vcc, %2, %8\n v_cmp_le_f32 vcc, %3, %8\n"
                   " v_cmp_le_f32 vcc, %4, %8\n v_cmp_le_f32 vcc, %5, %8\n v_cmp_le_f32 vcc, %6, %8\n v_cmp_le_f32 vcc, %7, %8\n .endr"
                   :
                   : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(b)
                   : "vcc");
    } else if (OP == MOV) {
#define X(k) asm volatile("v_mov_b32 %0, %1" : "+v"(a[k]) : "v"(u[k]));
      REP64(X)
#undef X
    } else if (OP == FMA_MIX) {
      // the f16 halves of u[] hold small integers: subnormal f16 values n * 2^-24
#define X(k) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(a[k]) : "v"(h[k]), "v"(b));
      REP64(X)
#undef X
    } else if (OP == PERM) {
#define X(k) asm volatile("v_perm_b32 %0, %0, %0, %1" : "+v"(u[k]) : "s"(0x0c010c00u));
      REP64(X)
#undef X
    } else if (OP == CNDMASK_SGPR) {
#define X(k) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "s"(sel));
      REP64(X)
#undef X
    } else if (OP == FMA32_AND) {
#define X(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c)); asm volatile("v_and_b32 %0, 0xff, %0" : "+v"(u[k]));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if (OP == CVT_FMA) {
      // what four planes of the node test cost now: a conversion and an fma each
#define X(k) asm volatile("v_cvt_f32_ubyte0 %0, %4\n v_cvt_f32_ubyte1 %1, %4\n v_cvt_f32_ubyte2 %2, %4\n v_cvt_f32_ubyte3 %3, %4\n" \
                          "v_fma_f32 %0, %0, %5, %6\n v_fma_f32 %1, %1, %5, %6\n v_fma_f32 %2, %2, %5, %6\n v_fma_f32 %3, %3, %5, %6" \
                          : "+v"(a[k & 3]), "+v"(a[4 + (k & 3)]), "+v"(p[k].x), "+v"(p[k].y) : "v"(u[k]), "v"(b), "v"(c));
      REP8(X)
#undef X
    } else if (OP == PERM_MIX) {
      // the same four planes through two byte->half expansions and four mixed-precision fmas
#define X(k) asm volatile("v_perm_b32 %4, %6, %6, %9\n v_perm_b32 %5, %6, %6, %10\n" \
                          "v_fma_mix_f32 %0, %4, %7, %8 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %4, %7, %8 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n" \
                          "v_fma_mix_f32 %2, %5, %7, %8 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %5, %7, %8 op_sel:[1,0,0] op_sel_hi:[1,0,0]" \
                          : "+v"(a[k & 3]), "+v"(a[4 + (k & 3)]), "+v"(p[k].x), "+v"(p[k].y), "=&v"(h[0]), "=&v"(h[1]) : "v"(u[k]), "v"(b), "v"(c), "s"(0x0c010c00u), "s"(0x0c030c02u));
      REP8(X)
#undef X
    } else if (OP == MOV_DPP) {
#define X(k) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(a[k]) : "v"(a[(k + 1) & 7]));
      REP64(X)
#undef X
    } else if (OP == MAX_I32) {
#define X(k) asm volatile("v_max_i32 %0, %0, %1" : "+v"(u[k]) : "v"(h[0])); asm volatile("v_min_i32 %0, %0, %1" : "+v"(u[(k + 4) & 7]) : "v"(h[1]));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if (OP == MAX3_I32) {
#define X(k) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(u[k]) : "v"(h[0]), "v"(h[1]));
      REP64(X)
#undef X
    } else if (OP == CMP_I32) {
      asm volatile(".rept 8\n v_cmp_le_i32 vcc, %0, %8\n v_cmp_le_i32 vcc, %1, %8\n v_cmp_le_i32 vcc, %2, %8\n v_cmp_le_i32 vcc, %3, %8\n"
                   " v_cmp_le_i32 vcc, %4, %8\n v_cmp_le_i32 vcc, %5, %8\n v_cmp_le_i32 vcc, %6, %8\n v_cmp_le_i32 vcc, %7, %8\n .endr"
                   :
                   : "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]), "v"(u[4]), "v"(u[5]), "v"(u[6]), "v"(u[7]), "v"(h[0])
                   : "vcc");
    } else if (OP == BFI) {
#define X(k) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(u[k]) : "v"(h[0]), "v"(h[1]));
      REP64(X)
#undef X
    } else if (OP == BFE) {
#define X(k) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(u[k]));
      REP64(X)
#undef X
    } else if (OP == LSHL_ADD) {
#define X(k) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[k]) : "v"(h[0]));
      REP64(X)
#undef X
    } else if (OP == ADD_F32) {
#define X(k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
      REP64(X)
#undef X
    } else if (OP == MUL_F32) {
#define X(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
      REP64(X)
#undef X
    } else if (OP == OR_SDWA) {
#define X(k) asm volatile("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(a[k]) : "v"(u[k]), "v"(h[2]));
      REP64(X)
#undef X
    } else if (OP == CMP_CND_VCC) {
#define X(k) asm volatile("v_cmp_le_f32 vcc, %8, %9\n v_cndmask_b32 %0, %0, %9, vcc\n v_cndmask_b32 %1, %1, %9, vcc\n v_cndmask_b32 %2, %2, %9, vcc\n v_cndmask_b32 %3, %3, %9, vcc\n" \
                          "v_cndmask_b32 %4, %4, %9, vcc\n v_cndmask_b32 %5, %5, %9, vcc\n v_cndmask_b32 %6, %6, %9, vcc" \
                          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(p[k].x), "v"(b) : "vcc");
      REP8(X)
#undef X
    } else if (OP == CVT_AND_FMA) {
#define X(k) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "+v"(p[k].x) : "v"(u[k])); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if (OP == MAXF_AND_FMA) {
#define X(k) asm volatile("v_max_f32 %0, %0, %1" : "+v"(p[k].x) : "v"(c)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if (OP == CND_AND_FMA) {
#define X(k) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(p[k].x) : "v"(b), "s"(sel)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if (OP == MAXI_AND_FMA) {
#define X(k) asm volatile("v_max_i32 %0, %0, %1" : "+v"(u[k]) : "v"(h[0])); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if (OP == MED3) {
#define X(k) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
      REP64(X)
#undef X
    } else if (OP == CND_E64_VCC) {
#define X(k) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b) : );
      REP64(X)
#undef X
    } else if (OP == CMP_CND_PAIRS) {
#define X(k) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[k]) : "v"(p[k].x), "v"(b) : "vcc");
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if (OP == CND_VCC_SET) {
      if (i == 0) asm volatile("s_mov_b64 vcc, %0" : : "s"(sel) : "vcc");
#define X(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b) : );
      REP64(X)
#undef X
    } else if (OP == ADD_U32) {
#define X(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[k]) : "v"(h[0]));
      REP64(X)
#undef X
    } else if (OP == AND_SHIFT) {
#define X(k) asm volatile("v_and_b32 %0, 0xff, %0" : "+v"(u[k])); asm volatile("v_lshrrev_b32 %0, 8, %0" : "+v"(u[(k + 4) & 7]));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  const uint64_t r1 = wall_clock64();
  float s = 0.f;
  for (int k = 0; k < 8; k++) s += a[k] + (float)d[k] + p[k].x + p[k].y + (float)u[k] + (float)h[k];
  sink[tid] = s;
  if ((threadIdx.x & 63) == 0) {
    WaveStamp w;
    w.cycles = t1 - t0;
    w.realtime = r1 - r0;
    w.real0 = r0;
    w.hw_id = __builtin_amdgcn_s_getreg(4 | (31 << 11));    // HW_REG_HW_ID
    w.xcc_id = __builtin_amdgcn_s_getreg(20 | (31 << 11));  // HW_REG_XCC_ID
    stamps[tid / 64] = w;
  }
}

// More one-instruction streams, from a table: X(id, name, asm on chain register %0 with operands %1 (a VGPR) and %2 (a VGPR), type of the chain)
// type 0: 32-bit integer chain, 1: fp32 chain, 2: fp64 chain
#define MORE_OPS(X)                                                    \
  X(0, "v_or_b32", "v_or_b32 %0, %0, %1", 0)                           \
  X(1, "v_xor_b32", "v_xor_b32 %0, %0, %1", 0)                         \
  X(2, "v_lshlrev_b32", "v_lshlrev_b32 %0, 1, %0", 0)                  \
  X(3, "v_sub_u32", "v_sub_u32 %0, %0, %1", 0)                         \
  X(4, "v_subrev_f32", "v_subrev_f32 %0, %1, %0", 1)                   \
  X(5, "v_fmac_f32", "v_fmac_f32 %0, %1, %2", 1)                       \
  X(6, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %1", 0)                   \
  X(7, "v_mad_u64_u32", "v_mad_u64_u32 %0, vcc, %1, %2, %0", 2)        \
  X(8, "v_add_co_u32", "v_add_co_u32 %0, vcc, %0, %1", 0)              \
  X(9, "v_cvt_f32_f64", "v_cvt_f32_f64 %1, %0", 2)                     \
  X(10, "v_rcp_f64", "v_rcp_f64 %0, %0", 2)                            \
  X(11, "v_sqrt_f64", "v_sqrt_f64 %0, %0", 2)                          \
  X(12, "v_rcp_f32", "v_rcp_f32 %0, %0", 1)                            \
  X(13, "v_ashrrev_i32", "v_ashrrev_i32 %0, 1, %0", 0)                 \
  X(14, "v_mul_u32_u24", "v_mul_u32_u24 %0, %0, %1", 0)                \
  X(15, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %1, %2", 0)            \
  X(16, "v_cvt_f32_u32", "v_cvt_f32_u32 %0, %0", 0)                    \
  X(17, "v_min_f64", "v_min_f64 %0, %0, %3", 2)                        \
  X(18, "v_lshl_or_b32", "v_lshl_or_b32 %0, %0, 1, %1", 0)             \
  X(19, "v_and_or_b32", "v_and_or_b32 %0, %0, %1, %2", 0)              \
  X(20, "v_add3_u32", "v_add3_u32 %0, %0, %1, %2", 0)                  \
  X(21, "v_cmp_lt_f64", "v_cmp_lt_f64 vcc, %0, %3", 2)                 \
  X(22, "v_bfrev_b32", "v_bfrev_b32 %0, %0", 0)                        \
  X(23, "v_min_u32", "v_min_u32 %0, %0, %1", 0)
constexpr int kMoreOps = 24;

template <int ID>
__global__ void __launch_bounds__(1024, 8) stream_more(int iters, WaveStamp *stamps, float *sink) {
  extern __shared__ char lds_pad[];
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t u[8];
  float a[8];
  double d[8];
  for (int k = 0; k < 8; k++) {
    u[k] = tid * 2654435761u + k;
    a[k] = 1.0f + 1e-3f * (float)((tid + k) & 15);
    d[k] = 1.0 + 1e-3 * (double)((tid + k) & 15);
  }
  const uint32_t x1 = tid | 1u, x2 = tid ^ 0x55u;
  const float f1 = 1.0f - 1e-7f * (float)(tid & 3), f2 = 1e-6f;
  const double d3 = 1.0 + 1e-9 * (double)(tid & 7);
  __syncthreads();
  const uint64_t t0 = __builtin_readcyclecounter();
  const uint64_t r0 = wall_clock64();
  for (int i = 0; i < iters; i++) {
#define X(id, name, text, type)                                                                                                   \
    if (ID == id) {                                                                                                                \
      if (type == 0) { for (int r = 0; r < 8; r++) { _Pragma("unroll") for (int k = 0; k < 8; k++) asm volatile(text : "+v"(u[k]) : "v"(x1), "v"(x2), "v"(d3) : "vcc"); } } \
      else if (type == 1) { for (int r = 0; r < 8; r++) { _Pragma("unroll") for (int k = 0; k < 8; k++) asm volatile(text : "+v"(a[k]) : "v"(f1), "v"(f2), "v"(d3) : "vcc"); } } \
      else { for (int r = 0; r < 8; r++) { _Pragma("unroll") for (int k = 0; k < 8; k++) asm volatile(text : "+v"(d[k]) : "v"(x1), "v"(x2), "v"(d3) : "vcc"); } } \
    }
    MORE_OPS(X)
#undef X
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  const uint64_t r1 = wall_clock64();
  float sacc = 0.f;
  for (int k = 0; k < 8; k++) sacc += a[k] + (float)d[k] + (float)u[k];
  sink[tid] = sacc;
  if ((threadIdx.x & 63) == 0) {
    WaveStamp w;
    w.cycles = t1 - t0;
    w.realtime = r1 - r0;
    w.real0 = r0;
    w.hw_id = __builtin_amdgcn_s_getreg(4 | (31 << 11));
    w.xcc_id = __builtin_amdgcn_s_getreg(20 | (31 << 11));
    stamps[tid / 64] = w;
  }
}

// The library's node test on node words held in registers: `tests` tests per wave, the words perturbed by every
// result so that no test can be hoisted.  SORTED: the closest-hit form (entry distances and the sorting network).
template <bool SORTED>
__global__ void __launch_bounds__(1024, 8) node_test_loop(int iters, WaveStamp *stamps, float *sink) {
  extern __shared__ char lds_pad[];
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  mipt::Trav T;
  T.org = mipt::d3{0.1 * (tid & 7), 0.2, 0.3};
  T.dir = mipt::normalize(mipt::d3{0.3 + 0.01 * (tid & 31), 0.5, -0.8});
  T.c = mipt::make_cull(T.org, T.dir, 100.0);
  T.tlimit = 1000.0f;
  uint4 q0{__float_as_uint(-1.0f), __float_as_uint(-2.0f), __float_as_uint(-3.0f), 0x007a7a7au | (4u << 24)};
  uint4 q1{0x30201000u + tid, 0x31211101u, 0x32221202u, 0xf0e0d0c0u};
  uint4 q2{0xf1e1d1c1u, 0xf2e2d2c2u, 0u, 0u};
  int4 q3{1, 2, 3, 4};
  uint32_t acc = 0;
  __syncthreads();
  const uint64_t t0 = __builtin_readcyclecounter();
  const uint64_t r0 = wall_clock64();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      int ch[4];
      float t[4];
      bool hit[4];
      mipt::node_test_words<SORTED>(T, q0, q1, q2, q3, ch, t, hit);
      if (SORTED) {
        MIPT_CSWAP(0, 1) MIPT_CSWAP(2, 3) MIPT_CSWAP(0, 2) MIPT_CSWAP(1, 3) MIPT_CSWAP(1, 2)
        acc += (uint32_t)ch[0] + (t[1] < 2.0e38f ? 1u : 0u) + (t[3] < 2.0e38f ? 2u : 0u);
      } else {
        acc += (hit[0] ? 1u : 0u) + (hit[1] ? 2u : 0u) + (hit[2] ? 4u : 0u) + (hit[3] ? 8u : 0u);
      }
      q1.x ^= acc;
      q2.y += acc & 3u;
      q0.x ^= acc & 1u;   // the frame origin and the step exponents change too: nothing of the test is loop-invariant
      q0.w ^= (acc & 1u) << 1;
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  const uint64_t r1 = wall_clock64();
  sink[tid] = (float)acc;
  if ((threadIdx.x & 63) == 0) {
    WaveStamp w;
    w.cycles = t1 - t0;
    w.realtime = r1 - r0;
    w.real0 = r0;
    w.hw_id = __builtin_amdgcn_s_getreg(4 | (31 << 11));
    w.xcc_id = __builtin_amdgcn_s_getreg(20 | (31 << 11));
    stamps[tid / 64] = w;
  }
}

struct Shape { int waves_per_simd, block, blocks_per_cu; size_t lds; };
static const Shape kShapes[] = {{1, 256, 1, 96 << 10}, {2, 512, 1, 96 << 10}, {4, 1024, 1, 96 << 10}, {6, 768, 2, 60 << 10}, {8, 1024, 2, 60 << 10}};

typedef void (*Kern)(int, WaveStamp *, float *);

struct Result { double cyc_per_inst_simd, cyc_per_inst_wave, ginst_chip, clock_ghz, resident; int waves_min, waves_max; };

static Result run(Kern k, const Shape &sh, int cus, int iters, double inst_per_iter, WaveStamp *d_st, float *d_sink) {
  CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh.lds));
  const int grid = cus * sh.blocks_per_cu;
  const int n_waves = grid * sh.block / 64;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(grid), dim3(sh.block), sh.lds, 0, iters / 8 + 1, d_st, d_sink);  // warm-up
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k, dim3(grid), dim3(sh.block), sh.lds, 0, iters, d_st, d_sink);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  CHECK(hipGetLastError());
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<WaveStamp> st(n_waves);
  CHECK(hipMemcpy(st.data(), d_st, n_waves * sizeof(WaveStamp), hipMemcpyDeviceToHost));
  std::vector<uint64_t> cyc(n_waves);
  std::map<uint64_t, int> per_simd;
  double clock = 0, busy = 0;
  uint64_t first = ~0ull, last = 0;
  for (int i = 0; i < n_waves; i++) {
    busy += (double)st[i].realtime;
    first = std::min(first, st[i].real0);
    last = std::max(last, st[i].real0 + st[i].realtime);
    cyc[i] = st[i].cycles;
    per_simd[((uint64_t)st[i].xcc_id << 32) | (st[i].hw_id & 0xfff0u)]++;
    clock += (double)st[i].cycles / (double)std::max<uint64_t>(1, st[i].realtime) * 0.1;  // GHz
  }
  std::nth_element(cyc.begin(), cyc.begin() + n_waves / 2, cyc.end());
  Result r;
  const double inst_per_wave = inst_per_iter * iters;
  // a wave's own stamps span only the time it was resident and running beside the others (the issue arbiter favours
  // the oldest wave, so waves end one after the other): the per-SIMD figure comes from the launch's wall time
  r.cyc_per_inst_wave = (double)cyc[n_waves / 2] / inst_per_wave;
  r.ginst_chip = inst_per_wave * n_waves / (ms * 1e-3) / 1e9;
  r.clock_ghz = clock / n_waves;
  r.cyc_per_inst_simd = r.clock_ghz * 1e9 * (ms * 1e-3) / (inst_per_wave * sh.waves_per_simd);
  r.resident = busy / (double)(last - first) / (cus * 4.0);  // waves per SIMD in flight, averaged over the launch
  r.waves_min = 1 << 30;
  r.waves_max = 0;
  for (auto &kv : per_simd) { r.waves_min = std::min(r.waves_min, kv.second); r.waves_max = std::max(r.waves_max, kv.second); }
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
  return r;
}

// Does v_fma_mix_f32 read a subnormal fp16 source exactly (no flush)?  200 * 2^-24 * (1.5 * 2^24) + 0.25 = 300.25.
__global__ void probe(float *out) {
  const uint32_t h = 0x00ff00c8u;
  const float a = 1.5f * 16777216.0f, b = 0.25f;
  float lo, hi;
  asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(h), "v"(a), "v"(b));
  asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(hi) : "v"(h), "v"(a), "v"(b));
  if (threadIdx.x == 0) { out[0] = lo; out[1] = hi; }
}

int main(int argc, char **argv) {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  int iters = argc > 1 ? atoi(argv[1]) : 20000;
  WaveStamp *d_st;
  float *d_sink;
  CHECK(hipMalloc(&d_st, (size_t)cus * 2 * 16 * sizeof(WaveStamp)));
  CHECK(hipMalloc(&d_sink, (size_t)cus * 2 * 1024 * sizeof(float)));
  float h_probe[2];
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_sink);
  CHECK(hipMemcpy(h_probe, d_sink, sizeof(h_probe), hipMemcpyDeviceToHost));
  printf("{\"device\": \"%s\", \"cus\": %d, \"simds\": %d, \"nominal_mhz\": %.0f, \"iters\": %d,\n"
         " \"fma_mix_subnormal_probe\": {\"low_half\": %.4f, \"high_half\": %.4f, \"expected\": [300.25, 382.75]},\n"
         " \"rows\": [\n",
         prop.name, cus, cus * 4, prop.clockRate / 1000.0, iters, h_probe[0], h_probe[1]);
  static const Kern kStreams[N_OPS] = {stream<FMA32>, stream<FMA64>, stream<CVT_UBYTE>, stream<MINMAX32>, stream<PK_FMA32>, stream<ADD64>, stream<MUL64>,
                                       stream<MAX3>, stream<CNDMASK>, stream<CMP32>, stream<MOV>, stream<AND_SHIFT>, stream<FMA_MIX>, stream<PERM>, stream<CNDMASK_SGPR>,
                                       stream<FMA32_AND>, stream<CVT_FMA>, stream<PERM_MIX>, stream<MOV_DPP>, stream<MAX_I32>, stream<MAX3_I32>, stream<CMP_I32>, stream<BFI>,
                                       stream<BFE>, stream<LSHL_ADD>, stream<ADD_F32>, stream<MUL_F32>, stream<OR_SDWA>, stream<CMP_CND_VCC>, stream<CVT_AND_FMA>,
                                       stream<MAXF_AND_FMA>, stream<CND_AND_FMA>, stream<MAXI_AND_FMA>, stream<MED3>, stream<CND_E64_VCC>, stream<CMP_CND_PAIRS>,
                                       stream<CND_VCC_SET>, stream<ADD_U32>};
  bool first = true;
  auto emit = [&](const char *name, const Shape &sh, const Result &r, double inst_per_iter, const char *unit) {
    printf("%s  {\"stream\": \"%s\", \"waves_per_simd\": %d, \"waves_in_flight_per_simd\": %.2f, \"waves_per_simd_seen\": [%d, %d], \"cycles_per_%s_per_simd\": %.3f, "
           "\"cycles_per_%s_median_wave\": %.2f, \"G_%s_per_s_chip\": %.1f, \"in_kernel_clock_ghz\": %.3f}",
           first ? "" : ",\n", name, sh.waves_per_simd, r.resident, r.waves_min, r.waves_max, unit, r.cyc_per_inst_simd, unit, r.cyc_per_inst_wave, unit, r.ginst_chip, r.clock_ghz);
    first = false;
    (void)inst_per_iter;
  };
  const char *only = argc > 2 ? argv[2] : nullptr;  // run only the streams whose name contains this
  for (int op = 0; op < N_OPS; op++)
    if (!only || strstr(kOpName[op], only))
    for (const Shape &sh : kShapes) emit(kOpName[op], sh, run(kStreams[op], sh, cus, iters, 64.0, d_st, d_sink), 64.0, "instruction");
  {
#define X(id, name, text, type) {name, stream_more<id>},
    struct { const char *name; Kern k; } more[] = {MORE_OPS(X)};
#undef X
    for (auto &mk : more)
      if (!only || strstr(mk.name, only))
        for (const Shape &sh : kShapes)
          if (sh.waves_per_simd == 1 || sh.waves_per_simd == 4 || sh.waves_per_simd == 8) emit(mk.name, sh, run(mk.k, sh, cus, iters, 64.0, d_st, d_sink), 64.0, "instruction");
  }
  // does an instruction whose upper half (or three quarters) of the wave is switched off issue faster?
  {
    struct { const char *name; Kern k; } masked[] = {
        {"v_fma_f32, low 32 lanes only", stream<FMA32, 1>}, {"v_fma_f32, low 16 lanes only", stream<FMA32, 2>}, {"v_fma_f32, even lanes only", stream<FMA32, 3>},
        {"v_fma_f32, lanes 0-15 and 32-47 only", stream<FMA32, 4>},
        {"v_fma_f64, low 32 lanes only", stream<FMA64, 1>}, {"v_fma_f64, low 16 lanes only", stream<FMA64, 2>}, {"v_fma_f64, even lanes only", stream<FMA64, 3>},
        {"v_fma_f64, lanes 0-15 and 32-47 only", stream<FMA64, 4>},
        {"v_cvt_f32_ubyte0, low 32 lanes only", stream<CVT_UBYTE, 1>}, {"v_cvt_f32_ubyte0, low 16 lanes only", stream<CVT_UBYTE, 2>}};
    for (auto &mk : masked)
      if (!only || strstr(mk.name, only))
        for (const Shape &sh : kShapes) emit(mk.name, sh, run(mk.k, sh, cus, iters, 64.0, d_st, d_sink), 64.0, "instruction");
  }
  for (const Shape &sh : kShapes) emit("node_test any-hit (4 children, unsorted)", sh, run(node_test_loop<false>, sh, cus, iters / 8, 4.0, d_st, d_sink), 4.0, "node_test");
  for (const Shape &sh : kShapes) emit("node_test closest-hit (entry distances + sort)", sh, run(node_test_loop<true>, sh, cus, iters / 8, 4.0, d_st, d_sink), 4.0, "node_test");
  printf("\n]}\n");
  return 0;
}
