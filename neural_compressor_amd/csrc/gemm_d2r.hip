// gemm_d2r.hip -- K4a, third generation: fused INT4 -> bf16 dequant-GEMM with the weights DIRECT TO REGISTERS.
//
// Replaces INCWeightOnlyLinear.forward (reference modules.py:594-610 = recover() once + F.linear) for M > 64 rows, like the
// producer / consumer kernel in gemm.hip, with the same 256 x 256 x 64 tile and bit-identical outputs (every accumulator
// receives the same MFMAs in the same order).  What changes is who touches the weights:
//
//   * a packed word of the optimum layout (8 consecutive k of ONE output column, modules.py:254-260) IS one lane's A operand
//     of v_mfma_f32_32x32x16_bf16 once dequantised.  The workgroup has FOUR waves, one per SIMD, and wave w owns ALL 256 rows
//     of the tile x the 64 columns [64 w, 64 w + 64): every weight of the tile is needed by exactly one wave, so that wave
//     loads its packed words straight from global memory (two 128-byte segments per request), dequantises them in registers
//     (the fp8-decoder arithmetic of dequant8: bit-identical to inc_woq_dequant) and multiplies -- the dequantised tile never
//     exists in LDS: no producer waves, no 32 KiB of ds_write_b128 per K-step, no W fragment reads, no redundant arithmetic;
//   * one wave per SIMD owns the SIMD's whole 512-entry register file: 2 x 8 accumulator tiles of 32 x 32 (256 registers),
//     two x-fragment sets and two W-fragment sets (software pipeline one k16 group deep), three sets of packed words (loaded
//     two K-steps ahead);
//   * LDS holds only x: NS stages of 32 KiB filled by LDS-DMA (global_load_lds_dwordx4, 8 pieces of 1 KiB per wave and step,
//     XOR-swizzled chunks, NS - 1 steps ahead), read as 8 conflict-free ds_read_b128 per k16 group and wave.
//
// Per K-step and wave: 64 MFMAs, 32 fragment reads, 8 packed words (152 VALU) + 2 group parameters, 12 small loads, 8 DMA pieces,
// one barrier.  With a single wave per SIMD nothing else hides an instruction, so the step is written as 16 sub-regions of
// 4 MFMAs, each carrying its share of the reads / arithmetic / requests, pinned with sched_barrier fences.
#include <type_traits>

#include "gemm_common.hpp"

namespace {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
constexpr int D2R_THREADS = 256;
constexpr int D2R_CPITCH = TN * 2 + 16;  // epilogue image of the bf16 tile in LDS: 528-byte rows

template <int NS>
constexpr int d2r_smem_bytes() {
  return NS * T_ASTAGE > TM * D2R_CPITCH ? NS * T_ASTAGE : TM * D2R_CPITCH;
}


// compile-time loop: f(std::integral_constant<int, B>) ... f(std::integral_constant<int, E - 1>)
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}
// one accumulator register (AGPR R) of the direct-to-register kernel
template <int R>
__device__ __forceinline__ float acc_read() {
  float v;
  asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(v) : "i"(R));
  return v;
}
template <int R>
__device__ __forceinline__ void acc_zero() {
  asm volatile("v_accvgpr_write_b32 a%c0, 0" : : "i"(R));
}
// a0 .. a255 as an asm clobber list: reserves the accumulator half of the register file in the kernel descriptor
#define INC_D2R_AGPR_CLOBBERS \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", \
  "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", \
  "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", \
  "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", \
  "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", \
  "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", \
  "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", \
  "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", \
  "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", \
  "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", \
  "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", \
  "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", \
  "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", \
  "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", \
  "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", \
  "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"
#define INC_D2R_ZERO_ACC()                                 \
  asm volatile("" : : : INC_D2R_AGPR_CLOBBERS);            \
  static_for<0, 256>([&](auto R) { acc_zero<R.value>(); })
#define INC_SB() __builtin_amdgcn_sched_barrier(0)

// ABL (harness build only, timing-only, WRONG results): bit 0 no dequant arithmetic, 2 no x LDS-DMA, 3 no W loads,
// 4 no fragment reads, 5 no MFMA, 6 no per-step barrier, 7 no epilogue stores
template <bool IS_BF16, int NS, int ABL>
__global__ __launch_bounds__(D2R_THREADS) void woq_gemm_w4_d2r_kernel(
    const uint16_t* __restrict__ x, const uint32_t* __restrict__ qweight, const uint16_t* __restrict__ scales,
    const uint32_t* __restrict__ qzeros, const uint16_t* __restrict__ bias, uint16_t* __restrict__ y, int64_t M, int64_t N,
    int64_t K, int64_t NW, int g_shift, int y_vec_ok, float* __restrict__ partial, int steps_per_split) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int D = NS - 1;                   // the x DMA runs D steps ahead
  constexpr int VM_STEADY = D >= 3 ? 28 : 20;  // see "counted waits" below
  const int tiles_n = (int)((N + TN - 1) / TN);
  const int tiles_m = (int)((M + TM - 1) / TM);
  const int nwg = tiles_m * tiles_n;
  int wg = blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = wg % 8, idx = wg / 8;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;  // bijective XCD remap
  }
  const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
  const int64_t m0 = (int64_t)tm * TM, n0 = (int64_t)tn * TN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk_all = (int)(K / TK);
  const int kbase = blockIdx.y * steps_per_split;
  const int nk = min(steps_per_split, nk_all - kbase);
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const bool lds_epilogue = !partial && (y_vec_ok & 2) && m0 + TM <= M && n0 + TN <= N && (ABL & 128) == 0;
  const float inv_u = fp8_unit_inverse();

  // ---- x tile by LDS-DMA: piece i of this wave = LDS rows (wave*8+i)*8 .. +7, 16-byte chunk XOR-ed by (row >> 1) & 7 ----
  // Running state (scalar registers, advanced once per step in a fenced slot far from the requests that read them): `xptr` = the
  // tile the NEXT pieces fetch (clamped at the last tile: the tail requests re-read valid memory and are never used), `dma_off` = the
  // LDS stage they fill.
  uint32_t avoff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int R = (wave * 8 + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((R >> 1) & 7);
    int64_t row = m0 + R;
    if (row > M - 1) row = M - 1;  // rows past M are computed from a valid row and never stored
    avoff[i] = (uint32_t)(((row - m0) * K + 8 * c) * 2);
  }
  const uint16_t* xptr = x + m0 * K + (int64_t)kbase * TK;
  int xt = 0;
  uint32_t dma_off = 0;
  const uint32_t dma_lds0 = __builtin_amdgcn_readfirstlane(lds0 + wave * 8192);
#define INC_D2R_DMA(I)                                                                                                        \
  if constexpr ((ABL & 4) == 0) {                                                                                             \
    /* M0 = LDS base of the piece; s_nop 4 covers M0 -> LDS-DMA and a freshly written scalar base */                          \
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(avoff[I]), "s"(xptr), "s"(dma_lds0 + dma_off), "i"((I)*1024) : "memory", "scc"); \
  }
  const uint16_t* const xbase = xptr;
  auto advance_x = [&]() {  // scalar arithmetic only (no branch): the tile index saturates at the last tile
    asm volatile("" : "+s"(xt), "+s"(dma_off));  // inputs pinned too: the arithmetic starts HERE, behind the slot's MFMA
    xt = min(xt + 1, nk - 1);
    xptr = xbase + (uint32_t)(xt * TK);
    dma_off = dma_off + T_ASTAGE == NS * T_ASTAGE ? 0u : dma_off + T_ASTAGE;
    asm volatile("" : "+s"(xptr), "+s"(dma_off), "+s"(xt));  // computed HERE, in this slot (the compiler would sink it to its first use)
  };

  // ---- packed words: this lane's A operands.  Fragment (nf, kk): column n0 + 64 wave + 32 nf + (lane & 31), k-octet 2 kk + (lane >> 5)
  uint32_t wvoff[8], svoff[2], zvoff[2];  // wvoff[2 kk + nf]
  int zshift[2];
#pragma unroll
  for (int nf = 0; nf < 2; ++nf) {
    int64_t ncol = n0 + wave * 64 + nf * 32 + (lane & 31);
    if (ncol > N - 1) ncol = N - 1;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) wvoff[2 * kk + nf] = (uint32_t)((((int64_t)(2 * kk + (lane >> 5))) * N + ncol) * 4);
    svoff[nf] = (uint32_t)(ncol * 2);
    zvoff[nf] = (uint32_t)((ncol / 8) * 4);
    zshift[nf] = 4 * (int)(ncol % 8);
  }
  // running pointers of the NEXT register loads (tile `wt`, clamped at the last tile) and of its group's parameters
  const uint32_t* wptr = qweight + (int64_t)kbase * (TK / 8) * N;
  const int64_t g0 = g_shift >= 0 ? (((int64_t)kbase * TK) >> g_shift) : 0;
  const uint16_t* sptr = scales + g0 * N;
  const uint32_t* zptr = qzeros + g0 * NW;
  const uint16_t* const sbase = sptr;
  const uint32_t* const zbase = zptr;
  const int g0base = (int)g0;
  int wt = 0;
  const uint32_t* const wbase = wptr;
  const int64_t wstride = (int64_t)(TK / 8) * N;
  const int gsh = g_shift >= 6 ? g_shift - 6 : -1;  // tiles per group = 1 << gsh; -1: a single group
  const uint32_t wstride32 = (uint32_t)wstride, n32 = (uint32_t)N, nw32 = (uint32_t)NW;  // every offset below is < 2^31 elements (inc_woq_gemm checks)
  auto advance_w = [&]() {  // scalar arithmetic only (no branch)
    asm volatile("" : "+s"(wt));
    wt = min(wt + 1, nk - 1);
    wptr = wbase + (uint32_t)wt * wstride32;
    const uint32_t gi = gsh >= 0 ? (uint32_t)((kbase + wt) >> gsh) - (uint32_t)g0base : 0u;
    sptr = sbase + gi * n32;
    zptr = zbase + gi * nw32;
    asm volatile("" : "+s"(wptr), "+s"(sptr), "+s"(zptr), "+s"(wt));  // see advance_x
  };
  uint32_t W[3][8], SC[3][2], ZW[3][2];  // [tile % 3][2 kk + nf]
  // the step's 12 register loads as three requests of four: part 0 = words of kk 0, 1; part 1 = kk 2, 3; part 2 = scales + zero words
#define INC_D2R_LOADW(SET, PART)                                                                                                   \
  if constexpr ((ABL & 8) != 0) {                                                                                                  \
    asm volatile("" : "=v"(W[SET][4 * (PART)]), "=v"(W[SET][4 * (PART) + 1]), "=v"(W[SET][4 * (PART) + 2]), "=v"(W[SET][4 * (PART) + 3])); \
  } else {                                                                                                                         \
    asm volatile(                                                                                                                  \
        "s_nop 4\n\t"                                                                                                              \
        "global_load_dword %0, %4, %8\n\t"                                                                                         \
        "global_load_dword %1, %5, %8\n\t"                                                                                         \
        "global_load_dword %2, %6, %8\n\t"                                                                                         \
        "global_load_dword %3, %7, %8"                                                                                             \
        : "=&v"(W[SET][4 * (PART)]), "=&v"(W[SET][4 * (PART) + 1]), "=&v"(W[SET][4 * (PART) + 2]), "=&v"(W[SET][4 * (PART) + 3])   \
        : "v"(wvoff[4 * (PART)]), "v"(wvoff[4 * (PART) + 1]), "v"(wvoff[4 * (PART) + 2]), "v"(wvoff[4 * (PART) + 3]), "s"(wptr)    \
        : "memory");                                                                                                               \
  }
#define INC_D2R_LOADP(SET)                                                                                                         \
  if constexpr ((ABL & 8) != 0) {                                                                                                  \
    asm volatile("" : "=v"(SC[SET][0]), "=v"(SC[SET][1]), "=v"(ZW[SET][0]), "=v"(ZW[SET][1]));                                     \
  } else {                                                                                                                         \
    asm volatile(                                                                                                                  \
        "s_nop 4\n\t"                                                                                                              \
        "global_load_ushort %0, %4, %8\n\t"                                                                                        \
        "global_load_ushort %1, %5, %8\n\t"                                                                                        \
        "global_load_dword %2, %6, %9\n\t"                                                                                         \
        "global_load_dword %3, %7, %9"                                                                                             \
        : "=&v"(SC[SET][0]), "=&v"(SC[SET][1]), "=&v"(ZW[SET][0]), "=&v"(ZW[SET][1])                                               \
        : "v"(svoff[0]), "v"(svoff[1]), "v"(zvoff[0]), "v"(zvoff[1]), "s"(sptr), "s"(zptr)                                         \
        : "memory");                                                                                                               \
  }
  // counted waits.  Program order of a step's 20 requests: 12 register loads (tile t+2), then 8 DMA pieces (tile t+D).  At the
  // wait of step t the words of tile t+1 (issued in step t-1) and the x pieces of tile t+1 (issued in step t+1-D) must be
  // back: D = 2 -> everything up to step t-1, this step's 20 stay in flight; D >= 3 -> step t-1's 8 pieces stay in flight too.
#define INC_D2R_WAIT(SET, NN)                                                                                                       \
  if constexpr ((ABL & 12) == 12)                                                                                                   \
    asm volatile("" : "+v"(W[SET][0]), "+v"(W[SET][1]), "+v"(W[SET][2]), "+v"(W[SET][3]), "+v"(W[SET][4]), "+v"(W[SET][5]), "+v"(W[SET][6]), \
                 "+v"(W[SET][7]), "+v"(SC[SET][0]), "+v"(SC[SET][1]), "+v"(ZW[SET][0]), "+v"(ZW[SET][1]) : : "memory");             \
  else                                                                                                                              \
    asm volatile("s_waitcnt vmcnt(%12)"                                                                                             \
                 : "+v"(W[SET][0]), "+v"(W[SET][1]), "+v"(W[SET][2]), "+v"(W[SET][3]), "+v"(W[SET][4]), "+v"(W[SET][5]), "+v"(W[SET][6]), \
                   "+v"(W[SET][7]), "+v"(SC[SET][0]), "+v"(SC[SET][1]), "+v"(ZW[SET][0]), "+v"(ZW[SET][1])                          \
                 : "i"(NN)                                                                                                          \
                 : "memory");

  // ---- dequantisation of one packed word in two halves of three instruction slots each (dequant8's FORM 0 arithmetic, bit-identical):
  // a slot is what rides behind ONE MFMA.  Temporaries are named so that the slots can be separated by scheduling fences.
  float sc[2], nzs[2];
  uint32_t mlo, mhi;
  f32x2 cq, dq;
  float f0, f1, f2, f3;
  // sched_barrier fences only bind the machine scheduler; instruction selection and the IR passes before it (SLP vectoriser, sinking)
  // move pure arithmetic freely.  Every slot therefore ends by passing what it produced through an empty volatile asm: volatile
  // statements (the MFMAs among them) keep their order, so a slot's arithmetic is bracketed between the MFMA in front of it (its
  // inputs were pinned by the previous slot) and the one behind it.
#define INC_PIN1(a) asm volatile("" : "+v"(a))
#define INC_PIN2(a, b) asm volatile("" : "+v"(a), "+v"(b))
#define INC_PIN3(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c))
#define INC_PIN4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
  auto dqa = [&](int slot, uint32_t w, int nf, uint4& o) {  // half A: k 0..3 of the word -> o.x, o.y
    if constexpr ((ABL & 1) != 0) {
      if (slot == 0) { mlo = w; mhi = w ^ __float_as_uint(sc[nf]); INC_PIN2(mlo, mhi); }
      if (slot == 2) { o.x = mlo; o.y = mhi; INC_PIN2(o.x, o.y); }
      return;
    }
    if (slot == 0) { INC_PIN1(w); mlo = w & 0x0F0F0F0Fu; mhi = (w >> 4) & 0x0F0F0F0Fu; cq = __builtin_amdgcn_cvt_pk_f32_fp8((int)mlo, false); INC_PIN3(mlo, mhi, cq); }
    if (slot == 1) { dq = __builtin_amdgcn_cvt_pk_f32_fp8((int)mhi, false); f0 = fma_single(cq[0], sc[nf], nzs[nf]); f2 = fma_single(cq[1], sc[nf], nzs[nf]); INC_PIN3(dq, f0, f2); }
    if (slot == 2) { f1 = fma_single(dq[0], sc[nf], nzs[nf]); f3 = fma_single(dq[1], sc[nf], nzs[nf]); o.x = cvt_pair<IS_BF16>(f0, f1); o.y = cvt_pair<IS_BF16>(f2, f3); INC_PIN2(o.x, o.y); }
  };
  auto dqb = [&](int slot, int nf, uint4& o) {  // half B: k 4..7 -> o.z, o.w (masks left by half A)
    if constexpr ((ABL & 1) != 0) {
      if (slot == 2) { o.z = mlo ^ __float_as_uint(nzs[nf]); o.w = mhi; INC_PIN2(o.z, o.w); }
      return;
    }
    if (slot == 0) { cq = __builtin_amdgcn_cvt_pk_f32_fp8((int)mlo, true); dq = __builtin_amdgcn_cvt_pk_f32_fp8((int)mhi, true); f0 = fma_single(cq[0], sc[nf], nzs[nf]); INC_PIN3(cq, dq, f0); }
    if (slot == 1) { f1 = fma_single(dq[0], sc[nf], nzs[nf]); f2 = fma_single(cq[1], sc[nf], nzs[nf]); f3 = fma_single(dq[1], sc[nf], nzs[nf]); INC_PIN3(f1, f2, f3); }
    if (slot == 2) { o.z = cvt_pair<IS_BF16>(f0, f1); o.w = cvt_pair<IS_BF16>(f2, f3); INC_PIN2(o.z, o.w); }
  };
  // group parameters of column nf from register set `set`, in two slots
  float gp_s0;
  uint32_t gp_z;
  auto gpar = [&](int slot, int set, int nf) {
    if (slot == 0) {
      INC_PIN2(SC[set][nf], ZW[set][nf]);
      gp_s0 = f16_bits_to_f32((uint16_t)SC[set][nf]);
      gp_z = ((ZW[set][nf] >> zshift[nf]) & 15u) + 1u;  // modules.py:407-410 (stored zp - 1; wraps above 15)
      INC_PIN2(gp_s0, gp_z);
    } else {
      gp_z = gp_z > 15u ? 0u : gp_z;
      nzs[nf] = -(float)gp_z * gp_s0;
      sc[nf] = gp_s0 * inv_u;
      INC_PIN2(nzs[nf], sc[nf]);
    }
  };

  // ---- x fragments: B operand (mf): rows 32 mf + (lane & 31), 16-byte chunk (2 kk + (lane >> 5)) ^ ((row >> 1) & 7) --------------
  const int a_sw = ((lane & 31) >> 1) & 7, a_hi = lane >> 5;
  const char* const xrow = smem + (lane & 31) * 128;
  uint32_t xchunk[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) xchunk[kk] = (uint32_t)(((2 * kk + a_hi) ^ a_sw) << 4);
  uint32_t rd_off = 0, rd_nxt = 0;  // LDS stage of the tile being multiplied / of the next one
  auto read_x = [&](uint32_t stage_off, int kk, int mf, uint4& dst) {
    if constexpr ((ABL & 16) != 0) { asm volatile("" : "=v"(dst.x), "=v"(dst.y), "=v"(dst.z), "=v"(dst.w)); return; }
    dst = *reinterpret_cast<const uint4*>(xrow + stage_off + mf * 4096 + xchunk[kk]);
  };

  // The 256 accumulators are the AGPR half of the SIMD's register file, addressed LITERALLY (a[16 j : 16 j + 15] for accumulator
  // j = 8 nf + mf) by asm statements only: the compiler never sees them as values, so it cannot shuffle them between the two halves
  // of the file (it did: 1400 spilled registers with `f32x16 acc[2][8]`), and the volatile asm MFMAs keep their order against the
  // requests, reads and waits around them.  The first statement clobbers a0..a255, which makes the kernel descriptor allocate
  // them; audit after every edit: no compiler-generated v_accvgpr_* and no scratch in the ISA (tools/audit_d2r.py).
  uint4 X[2][8], Wf[2][2];
  INC_D2R_ZERO_ACC();
#define INC_D2R_MMA(CUR, NF, MF)                                                                                                   \
  {                                                                                                                                \
    const u32x4 av_ = {Wf[CUR][NF].x, Wf[CUR][NF].y, Wf[CUR][NF].z, Wf[CUR][NF].w};                                               \
    const u32x4 bv_ = {X[CUR][MF].x, X[CUR][MF].y, X[CUR][MF].z, X[CUR][MF].w};                                                   \
    if constexpr ((ABL & 32) != 0) asm volatile("" : : "v"(av_), "v"(bv_));                                                       \
    else if constexpr (IS_BF16)                                                                                                    \
      asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" : : "v"(av_), "v"(bv_), "i"(16 * (8 * (NF) + (MF))), "i"(16 * (8 * (NF) + (MF)) + 15)); \
    else                                                                                                                           \
      asm volatile("v_mfma_f32_32x32x16_f16 a[%c2:%c3], %0, %1, a[%c2:%c3]" : : "v"(av_), "v"(bv_), "i"(16 * (8 * (NF) + (MF))), "i"(16 * (8 * (NF) + (MF)) + 15)); \
  }

  // ---- prologue: words of tiles 0, 1 and x tiles 0 .. D-1 requested; tile 0 complete ---------------------------------------
  INC_D2R_LOADW(0, 0) INC_D2R_LOADW(0, 1) INC_D2R_LOADP(0)
  INC_D2R_DMA(0) INC_D2R_DMA(1) INC_D2R_DMA(2) INC_D2R_DMA(3) INC_D2R_DMA(4) INC_D2R_DMA(5) INC_D2R_DMA(6) INC_D2R_DMA(7)
  advance_w();
  advance_x();
  INC_SB();
  INC_D2R_LOADW(1, 0) INC_D2R_LOADW(1, 1) INC_D2R_LOADP(1)
  advance_w();
#pragma unroll
  for (int d = 1; d < D; ++d) {
    INC_SB();
    INC_D2R_DMA(0) INC_D2R_DMA(1) INC_D2R_DMA(2) INC_D2R_DMA(3) INC_D2R_DMA(4) INC_D2R_DMA(5) INC_D2R_DMA(6) INC_D2R_DMA(7)
    advance_x();
  }
  INC_SB();
  INC_D2R_WAIT(0, VM_STEADY)  // outstanding allowed: tile 1's 12 words + (D-1) x 8 pieces = 20 (D = 2) or 28 (D = 3)
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int nf = 0; nf < 2; ++nf) {
    gpar(0, 0, nf); gpar(1, 0, nf);
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) dqa(sl, W[0][nf], nf, Wf[0][nf]);
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) dqb(sl, nf, Wf[0][nf]);
  }
#pragma unroll
  for (int mf = 0; mf < 8; ++mf) read_x(0u, 0, mf, X[0][mf]);
  INC_SB();

  // One sub-region = 4 MFMAs (accumulators (nf, mb .. mb+3) with nf = Q >> 1, mb = 4 (Q & 1)) of fragment set CUR, each followed by
  // its slot: S0..S3 are statements (fragment reads of the other set, dequantisation slots, parameter slots), fenced so that every
  // slot stays behind its MFMA.  PRE is the sub-region's request (register loads, a DMA piece, the counted wait) or empty; the slot
  // in front of a request (S3) is kept empty so that the request is all that sits between two MFMAs.
#define INC_D2R_SUB(CUR, Q, PRE, S0, S1, S2, S3)                       \
  PRE                                                                  \
  INC_SB();                                                            \
  INC_D2R_MMA(CUR, (Q) >> 1, 4 * ((Q) & 1) + 0) S0; INC_SB();          \
  INC_D2R_MMA(CUR, (Q) >> 1, 4 * ((Q) & 1) + 1) S1; INC_SB();          \
  INC_D2R_MMA(CUR, (Q) >> 1, 4 * ((Q) & 1) + 2) S2; INC_SB();          \
  INC_D2R_MMA(CUR, (Q) >> 1, 4 * ((Q) & 1) + 3) S3; INC_SB();
#define INC_NONE (void)0
  // a group that prepares fragment set NXT (k16 group KK of stage offset ST, words 2 KK + nf of set WS) while it multiplies set CUR
#define INC_D2R_GROUP(CUR, NXT, ST, KK, WS, P0, P1, P2, P3, E3)                                                                                    \
  INC_D2R_SUB(CUR, 0, P0, (read_x(ST, KK, 0, X[NXT][0]), dqa(0, W[WS][2 * (KK)], 0, Wf[NXT][0])), (read_x(ST, KK, 1, X[NXT][1]), dqa(1, 0u, 0, Wf[NXT][0])), \
              dqa(2, 0u, 0, Wf[NXT][0]), INC_NONE)                                                                                                \
  INC_D2R_SUB(CUR, 1, P1, (read_x(ST, KK, 2, X[NXT][2]), dqb(0, 0, Wf[NXT][0])), (read_x(ST, KK, 3, X[NXT][3]), dqb(1, 0, Wf[NXT][0])),           \
              dqb(2, 0, Wf[NXT][0]), INC_NONE)                                                                                                    \
  INC_D2R_SUB(CUR, 2, P2, (read_x(ST, KK, 4, X[NXT][4]), dqa(0, W[WS][2 * (KK) + 1], 1, Wf[NXT][1])), (read_x(ST, KK, 5, X[NXT][5]), dqa(1, 0u, 1, Wf[NXT][1])), \
              dqa(2, 0u, 1, Wf[NXT][1]), INC_NONE)                                                                                                \
  INC_D2R_SUB(CUR, 3, P3, (read_x(ST, KK, 6, X[NXT][6]), dqb(0, 1, Wf[NXT][1])), (read_x(ST, KK, 7, X[NXT][7]), dqb(1, 1, Wf[NXT][1])),           \
              dqb(2, 1, Wf[NXT][1]), E3)
  // One K-step.  `SET` = t % 3 (compile time: register sets); the LDS stages are run-time offsets.
  //   groups 0..2 (kk = g of tile t): multiply set g & 1, prepare kk = g + 1 of the same stage / register set; their requests:
  //            group 0: the 12 register loads of tile t+2 and DMA piece 0 of tile t+D; group 1: pieces 1..4; group 2: pieces 5..7 and
  //            the counted wait for tile t+1 (words + this wave's x pieces); parameters of column 0 behind it
  //   group 3: parameters of column 1, words kk = 0 of tile t+1; the step's barrier after its first sub-region, then the x fragments
  //            kk = 0 of the next stage; the running pointers advance in its last slot
#define INC_D2R_STEP(SET)                                                                                                       \
  {                                                                                                                             \
    constexpr int s1_ = ((SET) + 1) % 3, s2_ = ((SET) + 2) % 3;                                                                  \
    rd_nxt = rd_off + T_ASTAGE == NS * T_ASTAGE ? 0u : rd_off + T_ASTAGE;                                                       \
    INC_D2R_GROUP(0, 1, rd_off, 1, SET, INC_D2R_LOADW(s2_, 0), INC_D2R_LOADW(s2_, 1), INC_D2R_LOADP(s2_), INC_D2R_DMA(0), INC_NONE) \
    INC_D2R_GROUP(1, 0, rd_off, 2, SET, INC_D2R_DMA(1), INC_D2R_DMA(2), INC_D2R_DMA(3), INC_D2R_DMA(4), INC_NONE)                \
    INC_D2R_GROUP(0, 1, rd_off, 3, SET, INC_D2R_DMA(5), INC_D2R_DMA(6), INC_D2R_DMA(7), INC_D2R_WAIT(s1_, VM_STEADY), gpar(0, s1_, 0)) \
    INC_D2R_SUB(1, 0, INC_NONE;, (gpar(1, s1_, 0), dqa(0, W[s1_][0], 0, Wf[0][0])), (dqa(1, 0u, 0, Wf[0][0]), gpar(0, s1_, 1)),  \
                dqa(2, 0u, 0, Wf[0][0]), advance_x())                                                                           \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                          \
    if constexpr ((ABL & 64) == 0) __builtin_amdgcn_s_barrier();                                                                \
    INC_D2R_SUB(1, 1, INC_NONE;, (read_x(rd_nxt, 0, 0, X[0][0]), dqb(0, 0, Wf[0][0])), (read_x(rd_nxt, 0, 1, X[0][1]), dqb(1, 0, Wf[0][0])), \
                (read_x(rd_nxt, 0, 2, X[0][2]), dqb(2, 0, Wf[0][0])), gpar(1, s1_, 1))                                           \
    INC_D2R_SUB(1, 2, INC_NONE;, (read_x(rd_nxt, 0, 3, X[0][3]), dqa(0, W[s1_][1], 1, Wf[0][1])), (read_x(rd_nxt, 0, 4, X[0][4]), dqa(1, 0u, 1, Wf[0][1])), \
                (read_x(rd_nxt, 0, 5, X[0][5]), dqa(2, 0u, 1, Wf[0][1])), advance_w())                                           \
    INC_D2R_SUB(1, 3, INC_NONE;, (read_x(rd_nxt, 0, 6, X[0][6]), dqb(0, 1, Wf[0][1])), (read_x(rd_nxt, 0, 7, X[0][7]), dqb(1, 1, Wf[0][1])), \
                dqb(2, 1, Wf[0][1]), rd_off = rd_nxt)                                                                           \
  }
  for (int t0 = 0; t0 < nk; t0 += 3) {
    INC_D2R_STEP(0)
    if (t0 + 1 >= nk) break;
    INC_D2R_STEP(1)
    if (t0 + 2 >= nk) break;
    INC_D2R_STEP(2)
  }
#undef INC_D2R_STEP
#undef INC_D2R_GROUP
#undef INC_D2R_SUB
#undef INC_D2R_MMA
#undef INC_D2R_WAIT
#undef INC_D2R_LOADP
#undef INC_D2R_LOADW
#undef INC_D2R_DMA
#undef INC_NONE
#undef INC_PIN1
#undef INC_PIN2
#undef INC_PIN3
#undef INC_PIN4
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped tail requests
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs' results: nothing the compiler emits may read them early
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped tail requests
  if constexpr ((ABL & 128) != 0) return;

  // ---- epilogue: accumulator j = 8 nf + mf, register r of it: D row (n) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column (m) = lane & 31 ----
  if (lds_epilogue) {
    // full tile, 16-byte aligned y: through LDS (the stages are dead), then whole 512-byte rows per store instruction
    __builtin_amdgcn_s_barrier();  // every wave has drained its DMA (vmcnt(0) above) and finished its fragment reads
    static_for<0, 8>([&](auto NFRQ) {
      constexpr int nf = NFRQ.value >> 2, rq = NFRQ.value & 3;
      const int nl = wave * 64 + nf * 32 + 8 * rq + 4 * (lane >> 5);
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      if (bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = cvt16<IS_BF16>(bias[n0 + nl + e]);
      }
      static_for<0, 8>([&](auto MF) {
        constexpr int mf = MF.value, r0 = 16 * (8 * nf + mf) + 4 * rq;
        const int ml = mf * 32 + (lane & 31);
        *reinterpret_cast<uint2*>(smem + ml * D2R_CPITCH + nl * 2) =
            make_uint2(cvt_pair<IS_BF16>(acc_read<r0>() + bv[0], acc_read<r0 + 1>() + bv[1]), cvt_pair<IS_BF16>(acc_read<r0 + 2>() + bv[2], acc_read<r0 + 3>() + bv[3]));
      });
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    uint16_t* const ytile = y + m0 * N + n0;
#pragma unroll 4
    for (int i = tid; i < TM * (TN / 8); i += D2R_THREADS) {  // 16-byte chunk i: row i / 32, columns 8 (i % 32) .. +7
      const int row = i >> 5, c = i & 31;
      const uint4 v = *reinterpret_cast<const uint4*>(smem + row * D2R_CPITCH + c * 16);
      *reinterpret_cast<uint4*>(ytile + (int64_t)row * N + c * 8) = v;
    }
    return;
  }
  float* const slab = partial ? partial + (int64_t)blockIdx.y * M * N : nullptr;  // split-K: raw fp32 tile into this split's slab
  static_for<0, 8>([&](auto NFRQ) {
    constexpr int nf = NFRQ.value >> 2, rq = NFRQ.value & 3;
    const int64_t nb = n0 + wave * 64 + nf * 32 + 8 * rq + 4 * (lane >> 5);
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias && !slab) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (nb + e < N) bv[e] = cvt16<IS_BF16>(bias[nb + e]);
    }
    static_for<0, 8>([&](auto MF) {
      constexpr int mf = MF.value, r0 = 16 * (8 * nf + mf) + 4 * rq;
      const float vv[4] = {acc_read<r0>() + bv[0], acc_read<r0 + 1>() + bv[1], acc_read<r0 + 2>() + bv[2], acc_read<r0 + 3>() + bv[3]};
      const int64_t m = m0 + mf * 32 + (lane & 31);
      if (m < M) {
        if (slab) {  // (bias and conversion happen in the finalize kernel)
          float* dst = slab + m * N + nb;
          if (nb + 4 <= N && (N % 4) == 0) {
            *reinterpret_cast<float4*>(dst) = make_float4(vv[0], vv[1], vv[2], vv[3]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (nb + e < N) dst[e] = vv[e];
          }
        } else {
          uint16_t* dst = y + m * N + nb;
          if ((y_vec_ok & 1) && nb + 4 <= N) {
            *reinterpret_cast<uint2*>(dst) = make_uint2(cvt_pair<IS_BF16>(vv[0], vv[1]), cvt_pair<IS_BF16>(vv[2], vv[3]));
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (nb + e < N) dst[e] = IS_BF16 ? f32_to_bf16_bits(vv[e]) : f32_to_f16_bits(vv[e]);
          }
        }
      }
    });
  });
}
#undef INC_SB

}  // namespace

// Launcher used by inc_woq_gemm (gemm.hip).  `abl` selects a timing-only ablation in the harness build (0 in the product).
int inc_launch_woq_gemm_d2r(const uint16_t* x, const uint32_t* qw, const uint16_t* scales, const uint32_t* qz, const uint16_t* bias,
                            uint16_t* y, int64_t M, int64_t N, int64_t K, int64_t NW, int g_shift, int y_vec_ok, float* part, int steps,
                            int splits, bool bf, int ns, int abl, hipStream_t s) {
  const unsigned grid = (unsigned)(ceil_div64(M, TM) * ceil_div64(N, TN));
  dim3 g2(grid, (unsigned)splits);
#define INC_D2R(B, NS_, A)                                                                                                             \
  {                                                                                                                                    \
    constexpr int smem = d2r_smem_bytes<NS_>();                                                                                        \
    static std::atomic<uint64_t> attr_set{0};                                                                                          \
    if (inc_attr_needed(attr_set)) {                                                                                                   \
      (void)hipFuncSetAttribute((const void*)woq_gemm_w4_d2r_kernel<B, NS_, A>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);    \
      inc_attr_done(attr_set);                                                                                                         \
    }                                                                                                                                  \
    woq_gemm_w4_d2r_kernel<B, NS_, A><<<g2, D2R_THREADS, smem, s>>>(x, qw, scales, qz, bias, y, M, N, K, NW, g_shift, y_vec_ok, part, steps); \
  }
  if (!bf) {
    INC_D2R(false, 4, 0)
  }
#ifdef INC_KBENCH
  else if (ns == 3) INC_D2R(true, 3, 0)
  else if (abl == 1) INC_D2R(true, 4, 1)
  else if (abl == 4) INC_D2R(true, 4, 4)
  else if (abl == 8) INC_D2R(true, 4, 8)
  else if (abl == 12) INC_D2R(true, 4, 12)
  else if (abl == 16) INC_D2R(true, 4, 16)
  else if (abl == 32) INC_D2R(true, 4, 32)
  else if (abl == 61) INC_D2R(true, 4, 61)   /* MFMA + barrier only */
  else if (abl == 125) INC_D2R(true, 4, 125) /* MFMA only */
  else if (abl == 128) INC_D2R(true, 4, 128)
#endif
  else INC_D2R(true, 4, 0)
#undef INC_D2R
  return 0;
}
