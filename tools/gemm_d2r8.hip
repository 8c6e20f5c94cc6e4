// gemm_d2r8.hip -- K4a, the direct-to-register dequant-GEMM with TWO waves per SIMD (eight per workgroup).
//
// gemm_d2r.hip gives every SIMD one wave that owns 256 x 64 of the tile and all 512 registers; nothing hides that wave's own VMEM
// issue, LDS waits or the step's rendezvous (tools/kbench d2r: 1280 TFLOP/s against 1720 with global traffic and barrier removed).
// Here the 256 x 256 x 64 tile is worked by eight waves as 2 (m) x 4 (n): wave w owns rows 128 (w >> 2) .. + 127 of the columns
// 64 (w & 3) .. + 63 -- 8 accumulator tiles of 32 x 32 in a[0:127], at most 128 arch registers -- so each SIMD holds the two waves
// w and w + 4, which need the SAME packed words (the second request hits the CU's L1) and dequantise them redundantly: twice the
// VALU work per MFMA, none of it in LDS, and whenever one wave waits the other multiplies.  x as before: four 32 KiB stages by
// LDS-DMA (four 1 KiB pieces per wave and step), XOR-swizzled, one barrier per step.  Per output element the MFMAs and their order
// are those of the four-wave kernel: bit-identical results.
//
// MEASURED (tools/kbench d2r, profiles/r3h_kbench_d2r8.log): bit-identical on every case and 6 % SLOWER than the four-wave kernel
// (4096^3: 1203 vs 1284 TFLOP/s; 1163 / 1260 / 1191 vs 1239 / 1355 / 1267 on the other shapes).  Hiding one wave's stalls behind
// another is not what the four-wave kernel lacks: its limit is what the SIMD can issue and the CU can be fed per step, and this
// form doubles the dequantisation issue.  Kept as the A/B partner in the harness library only (tools/Makefile, harness flag 97);
// it is not part of libinc_mi355x.so.
#include <type_traits>

#include "gemm_common.hpp"

namespace {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
constexpr int D8_THREADS = 512;
constexpr int D8_NS = 4;
constexpr int D8_CPITCH = TN * 2 + 16;

template <int B, int E, class F>
__device__ __forceinline__ void static_for8(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for8<B + 1, E>(f);
  }
}
template <int R>
__device__ __forceinline__ float acc_read8() {
  float v;
  asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(v) : "i"(R));
  return v;
}
template <int R>
__device__ __forceinline__ void acc_zero8() {
  asm volatile("v_accvgpr_write_b32 a%c0, 0" : : "i"(R));
}
#define INC_D8_AGPR_CLOBBERS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"
#define INC_SB() __builtin_amdgcn_sched_barrier(0)

template <bool IS_BF16>
__global__ __launch_bounds__(D8_THREADS) void woq_gemm_w4_d2r8_kernel(
    const uint16_t* __restrict__ x, const uint32_t* __restrict__ qweight, const uint16_t* __restrict__ scales,
    const uint32_t* __restrict__ qzeros, const uint16_t* __restrict__ bias, uint16_t* __restrict__ y, int64_t M, int64_t N,
    int64_t K, int64_t NW, int g_shift, int y_vec_ok, float* __restrict__ partial, int steps_per_split) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NS = D8_NS, D = NS - 1;
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
  const int wn = wave & 3, wm = wave >> 2;
  const int nk_all = (int)(K / TK);
  const int kbase = blockIdx.y * steps_per_split;
  const int nk = min(steps_per_split, nk_all - kbase);
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const bool lds_epilogue = !partial && (y_vec_ok & 2) && m0 + TM <= M && n0 + TN <= N;
  const float inv_u = fp8_unit_inverse();

  // ---- x tile by LDS-DMA: piece i of this wave = LDS rows (wave * 4 + i) * 8 .. + 7, 16-byte chunk XOR-ed by (row >> 1) & 7 ----
  uint32_t avoff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int R = (wave * 4 + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((R >> 1) & 7);
    int64_t row = m0 + R;
    if (row > M - 1) row = M - 1;  // rows past M are computed from a valid row and never stored
    avoff[i] = (uint32_t)(((row - m0) * K + 8 * c) * 2);
  }
  const uint16_t* xptr = x + m0 * K + (int64_t)kbase * TK;
  int xt = 0;
  uint32_t dma_off = 0;
  const uint32_t dma_lds0 = __builtin_amdgcn_readfirstlane(lds0 + wave * 4096);
#define INC_D8_DMA(I) \
  asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(avoff[I]), "s"(xptr), "s"(dma_lds0 + dma_off), "i"((I)*1024) : "memory", "scc");
  const uint16_t* const xbase = xptr;
  auto advance_x = [&]() {
    asm volatile("" : "+s"(xt), "+s"(dma_off));
    xt = min(xt + 1, nk - 1);
    xptr = xbase + (uint32_t)(xt * TK);
    dma_off = dma_off + T_ASTAGE == NS * T_ASTAGE ? 0u : dma_off + T_ASTAGE;
    asm volatile("" : "+s"(xptr), "+s"(dma_off), "+s"(xt));
  };

  // ---- packed words: accumulator row r = lane & 31 of fragment nf is column n0 + 64 wn + 2 r + nf (see gemm_d2r.hip) ----
  uint32_t wvoff[4], svoff, zvoff;
  int zshift[2];
  {
    int64_t ncol = n0 + wn * 64 + 2 * (lane & 31);
    if (ncol > N - 2) ncol = N - 2;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) wvoff[kk] = (uint32_t)((((int64_t)(2 * kk + (lane >> 5))) * N + ncol) * 4);
    svoff = (uint32_t)(ncol * 2);
    zvoff = (uint32_t)((ncol / 8) * 4);
    zshift[0] = 4 * (int)(ncol % 8);
    zshift[1] = zshift[0] + 4;
  }
  const uint32_t* wptr = qweight + (int64_t)kbase * (TK / 8) * N;
  const int64_t g0 = g_shift >= 0 ? (((int64_t)kbase * TK) >> g_shift) : 0;
  const uint16_t* sptr = scales + g0 * N;
  const uint32_t* zptr = qzeros + g0 * NW;
  const uint16_t* const sbase = sptr;
  const uint32_t* const zbase = zptr;
  const int g0base = (int)g0;
  int wt = 0;
  const uint32_t* const wbase = wptr;
  const int gsh = g_shift >= 6 ? g_shift - 6 : -1;
  uint32_t gi_ = 0;
  const uint32_t wstride32 = (uint32_t)((TK / 8) * N), n32 = (uint32_t)N, nw32 = (uint32_t)NW;
  auto advance_w = [&]() {
    asm volatile("" : "+s"(wt));
    wt = min(wt + 1, nk - 1);
    wptr = wbase + (uint32_t)wt * wstride32;
    gi_ = gsh >= 0 ? (uint32_t)((kbase + wt) >> gsh) - (uint32_t)g0base : 0u;
    sptr = sbase + gi_ * n32;
    zptr = zbase + gi_ * nw32;
    asm volatile("" : "+s"(wptr), "+s"(wt), "+s"(sptr), "+s"(zptr));
  };
  u32x2 W[3][4];          // [tile % 3][kk]: .x = word of fragment 0, .y = fragment 1
  uint32_t SC[3], ZW[3];
#define INC_D8_LOADW(SET, PART)                                                                \
  asm volatile("global_load_dwordx2 %0, %2, %4\n\tglobal_load_dwordx2 %1, %3, %4"              \
               : "=&v"(W[SET][2 * (PART)]), "=&v"(W[SET][2 * (PART) + 1])                       \
               : "v"(wvoff[2 * (PART)]), "v"(wvoff[2 * (PART) + 1]), "s"(wptr)                  \
               : "memory");
#define INC_D8_LOADP(SET)                                                                      \
  asm volatile("global_load_dword %0, %2, %4\n\tglobal_load_dword %1, %3, %5"                  \
               : "=&v"(SC[SET]), "=&v"(ZW[SET])                                                 \
               : "v"(svoff), "v"(zvoff), "s"(sptr), "s"(zptr)                                   \
               : "memory");
  // counted wait.  A step issues 6 register loads (tile t+2) and then 4 DMA pieces (tile t+D).  At the wait of step t the words of
  // tile t+1 (step t-1) and this wave's x pieces of tile t+1 (step t+1-D = t-2) must be back: this step's 10 requests and step t-1's
  // 4 pieces stay in flight.
#define INC_D8_WAIT(SET) \
  asm volatile("s_waitcnt vmcnt(14)" : "+v"(W[SET][0]), "+v"(W[SET][1]), "+v"(W[SET][2]), "+v"(W[SET][3]), "+v"(SC[SET]), "+v"(ZW[SET]) : : "memory");

  // ---- x fragments (B operands): rows 128 wm + 32 mf + (lane & 31), 16-byte chunk (2 kk + (lane >> 5)) ^ ((row >> 1) & 7) ----
  const int a_sw = ((lane & 31) >> 1) & 7, a_hi = lane >> 5;
  uint32_t xaddr[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) xaddr[kk] = lds0 + (uint32_t)(wm * 16384 + (lane & 31) * 128) + (uint32_t)(((2 * kk + a_hi) ^ a_sw) << 4);
  uint32_t rd_off = 0, rd_nxt = 0;
  u32x4 X[2][4];
  uint32_t xa_;
#define INC_PIN1(a) asm volatile("" : "+v"(a))
#define INC_D8_XADDR(ST, KK) xa_ = xaddr[KK] + (ST); INC_PIN1(xa_)
#define INC_LGKM(NN) asm volatile("s_waitcnt lgkmcnt(%0)" : : "i"(NN) : "memory")

  uint4 Wf[2][2];
  float sc[2], nzs[2];
  uint32_t mlo, mhi;
  float gp_s0;
  uint32_t gp_z;
  asm volatile("" : : : INC_D8_AGPR_CLOBBERS);
  static_for8<0, 128>([&](auto R) { acc_zero8<R.value>(); });

  // literal temporaries v118..v127 (clobbered): f16 conversion v118, v119; f0..f3 = v120..v123; cq = v[124:125]; dq = v[126:127]
#define D8_MFMA(MN, I) MN " a[%c[aj]+" #I "*16:%c[aj]+" #I "*16+15], %[wf], %[x" #I "], a[%c[aj]+" #I "*16:%c[aj]+" #I "*16+15]\n\t"
#define D8_RD(I) "ds_read_b128 %[r" #I "], %[xa] offset:%c[o" #I "]\n\t"
#define D8_DQA0 "v_and_b32 %[mlo], 0xf0f0f0f, %[w]\n\tv_lshrrev_b32 %[mhi], 4, %[w]\n\tv_cvt_pk_f32_fp8 v[124:125], %[mlo]\n\tv_and_b32 %[mhi], 0xf0f0f0f, %[mhi]\n\t"
#define D8_DQA1 "v_cvt_pk_f32_fp8 v[126:127], %[mhi]\n\tv_fma_f32 v120, v124, %[sc], %[nz]\n\tv_fma_f32 v122, v125, %[sc], %[nz]\n\t"
#define D8_DQA2(CVTP) "v_fma_f32 v121, v126, %[sc], %[nz]\n\tv_fma_f32 v123, v127, %[sc], %[nz]\n\t" CVTP("%[oa]", "v120", "v121") CVTP("%[ob]", "v122", "v123")
#define D8_DQB0 "v_cvt_pk_f32_fp8_sdwa v[124:125], %[mlo] src0_sel:WORD_1\n\tv_cvt_pk_f32_fp8_sdwa v[126:127], %[mhi] src0_sel:WORD_1\n\t"
#define D8_DQB1 "v_fma_f32 v120, v124, %[sc], %[nz]\n\tv_fma_f32 v121, v126, %[sc], %[nz]\n\tv_fma_f32 v122, v125, %[sc], %[nz]\n\t"
#define D8_DQB2(CVTP) "v_fma_f32 v123, v127, %[sc], %[nz]\n\t" CVTP("%[oc]", "v120", "v121") CVTP("%[od]", "v122", "v123")
#define D8_GP0_LO "v_cvt_f32_f16 %[gs], %[scw]\n\tv_bfe_u32 %[gz], %[zw], %[zsh], 4\n\tv_add_u32 %[gz], 1, %[gz]\n\t"
#define D8_GP0_HI "v_cvt_f32_f16_sdwa %[gs], %[scw] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\tv_bfe_u32 %[gz], %[zw], %[zsh], 4\n\tv_add_u32 %[gz], 1, %[gz]\n\t"
#define D8_GP1 "v_cmp_gt_u32 vcc, 16, %[gz]\n\tv_cndmask_b32 %[gz], 0, %[gz], vcc\n\tv_cvt_f32_u32 %[gz], %[gz]\n\tv_mul_f32_e64 %[nz], %[gs], -%[gz]\n\tv_mul_f32 %[sc], %[iu], %[gs]\n\t"
#define D8_CVTP_BF16(Dd, A, Bb) "v_cvt_pk_bf16_f32 " Dd ", " A ", " Bb "\n\t"
#define D8_CVTP_F16(Dd, A, Bb) "v_cvt_f16_f32 v118, " A "\n\tv_cvt_f16_f32 v119, " Bb "\n\tv_pack_b32_f16 " Dd ", v118, v119\n\t"
#define D8_TMPS "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127"
  // inputs of every sub-region: the W fragment of fragment NF (A operand), the four x fragments (B operands), accumulator base 4 NF
#define D8_IN(CUR, NF) \
  [wf] "v"(wf_), [x0] "v"(X[CUR][0]), [x1] "v"(X[CUR][1]), [x2] "v"(X[CUR][2]), [x3] "v"(X[CUR][3]), [aj] "i"(16 * 4 * (NF))
#define D8_WF(CUR, NF) const u32x4 wf_ = {Wf[CUR][NF].x, Wf[CUR][NF].y, Wf[CUR][NF].z, Wf[CUR][NF].w};
#define D8_MN (IS_BF16 ? 0 : 1)

  // kind R (fragment 0's sub-region of groups 0..2): the four x fragments of set NXT and the whole of WORD -> Wf[NXT][0]
#define D8_ASM_R(MN, CVTP, CUR, NXT)                                                                                              \
  asm volatile(D8_MFMA(MN, 0) D8_RD(0) D8_RD(1) D8_DQA0 D8_MFMA(MN, 1) D8_RD(2) D8_RD(3) D8_DQA1 D8_MFMA(MN, 2) D8_DQA2(CVTP) D8_DQB0 \
                   D8_MFMA(MN, 3) D8_DQB1 D8_DQB2(CVTP)                                                                           \
               : [r0] "=&v"(X[NXT][0]), [r1] "=&v"(X[NXT][1]), [r2] "=&v"(X[NXT][2]), [r3] "=&v"(X[NXT][3]), [oa] "=&v"(Wf[NXT][0].x), \
                 [ob] "=&v"(Wf[NXT][0].y), [oc] "=&v"(Wf[NXT][0].z), [od] "=&v"(Wf[NXT][0].w), [mlo] "=&v"(mlo), [mhi] "=&v"(mhi)    \
               : D8_IN(CUR, 0), [xa] "v"(xa_), [w] "v"(w_), [sc] "v"(sc[0]), [nz] "v"(nzs[0]), [o0] "i"(0), [o1] "i"(4096),         \
                 [o2] "i"(8192), [o3] "i"(12288)                                                                                   \
               : D8_TMPS)
#define INC_D8_SUB_R(CUR, NXT, WORD)                                              \
  {                                                                               \
    D8_WF(CUR, 0)                                                                 \
    const uint32_t w_ = (WORD);                                                   \
    if constexpr (IS_BF16) D8_ASM_R("v_mfma_f32_32x32x16_bf16", D8_CVTP_BF16, CUR, NXT); \
    else D8_ASM_R("v_mfma_f32_32x32x16_f16", D8_CVTP_F16, CUR, NXT);              \
  }
  // kind W (fragment 1's sub-region of groups 0..2): the whole of WORD -> Wf[NXT][1]
#define D8_ASM_W(MN, CVTP, CUR, NXT)                                                                                              \
  asm volatile(D8_MFMA(MN, 0) D8_DQA0 D8_MFMA(MN, 1) D8_DQA1 D8_DQA2(CVTP) D8_MFMA(MN, 2) D8_DQB0 D8_DQB1 D8_MFMA(MN, 3) D8_DQB2(CVTP) \
               : [oa] "=&v"(Wf[NXT][1].x), [ob] "=&v"(Wf[NXT][1].y), [oc] "=&v"(Wf[NXT][1].z), [od] "=&v"(Wf[NXT][1].w),             \
                 [mlo] "=&v"(mlo), [mhi] "=&v"(mhi)                                                                                \
               : D8_IN(CUR, 1), [w] "v"(w_), [sc] "v"(sc[1]), [nz] "v"(nzs[1])                                                      \
               : D8_TMPS)
#define INC_D8_SUB_W(CUR, NXT, WORD)                                              \
  {                                                                               \
    D8_WF(CUR, 1)                                                                 \
    const uint32_t w_ = (WORD);                                                   \
    if constexpr (IS_BF16) D8_ASM_W("v_mfma_f32_32x32x16_bf16", D8_CVTP_BF16, CUR, NXT); \
    else D8_ASM_W("v_mfma_f32_32x32x16_f16", D8_CVTP_F16, CUR, NXT);              \
  }
  // kind W + the first half of fragment 0's NEXT group parameters (group 2's second sub-region)
#define D8_ASM_WG(MN, CVTP, CUR, NXT, SET)                                                                                        \
  asm volatile(D8_MFMA(MN, 0) D8_DQA0 D8_MFMA(MN, 1) D8_DQA1 D8_DQA2(CVTP) D8_MFMA(MN, 2) D8_DQB0 D8_DQB1 D8_MFMA(MN, 3) D8_DQB2(CVTP) \
                   D8_GP0_LO                                                                                                      \
               : [oa] "=&v"(Wf[NXT][1].x), [ob] "=&v"(Wf[NXT][1].y), [oc] "=&v"(Wf[NXT][1].z), [od] "=&v"(Wf[NXT][1].w),             \
                 [mlo] "=&v"(mlo), [mhi] "=&v"(mhi), [gs] "=&v"(gp_s0), [gz] "=&v"(gp_z)                                           \
               : D8_IN(CUR, 1), [w] "v"(w_), [sc] "v"(sc[1]), [nz] "v"(nzs[1]), [scw] "v"(SC[SET]), [zw] "v"(ZW[SET]),              \
                 [zsh] "v"(zshift[0])                                                                                              \
               : D8_TMPS)
#define INC_D8_SUB_WG(CUR, NXT, WORD, SET)                                        \
  {                                                                               \
    D8_WF(CUR, 1)                                                                 \
    const uint32_t w_ = (WORD);                                                   \
    if constexpr (IS_BF16) D8_ASM_WG("v_mfma_f32_32x32x16_bf16", D8_CVTP_BF16, CUR, NXT, SET); \
    else D8_ASM_WG("v_mfma_f32_32x32x16_f16", D8_CVTP_F16, CUR, NXT, SET);        \
  }
  // group 3, fragment 0 (before the barrier: no reads): fragment 0's new parameters, the first word of tile t+1 -> Wf[0][0], then the
  // first half of fragment 1's parameters
#define D8_ASM_G0(MN, CVTP, SET)                                                                                                  \
  asm volatile(D8_MFMA(MN, 0) D8_GP1 D8_DQA0 D8_MFMA(MN, 1) D8_DQA1 D8_DQA2(CVTP) D8_MFMA(MN, 2) D8_DQB0 D8_DQB1 D8_MFMA(MN, 3)      \
                   D8_DQB2(CVTP)                                                                                                  \
               : [oa] "=&v"(Wf[0][0].x), [ob] "=&v"(Wf[0][0].y), [oc] "=&v"(Wf[0][0].z), [od] "=&v"(Wf[0][0].w), [mlo] "=&v"(mlo),   \
                 [mhi] "=&v"(mhi), [sc] "=&v"(sc[0]), [nz] "=&v"(nzs[0]), [gz] "+v"(gp_z)                                           \
               : D8_IN(1, 0), [w] "v"(w_), [gs] "v"(gp_s0), [iu] "v"(inv_u)                                                         \
               : D8_TMPS, "vcc")
#define INC_D8_SUB_G0(SET)                                                        \
  {                                                                               \
    D8_WF(1, 0)                                                                   \
    const uint32_t w_ = W[SET][0].x;                                              \
    if constexpr (IS_BF16) D8_ASM_G0("v_mfma_f32_32x32x16_bf16", D8_CVTP_BF16, SET); \
    else D8_ASM_G0("v_mfma_f32_32x32x16_f16", D8_CVTP_F16, SET);                  \
  }
  // group 3, fragment 1 (behind the barrier): the x fragments of the next stage, fragment 1's new parameters, its first word -> Wf[0][1]
#define D8_ASM_G1(MN, CVTP, SET)                                                                                                  \
  asm volatile(D8_MFMA(MN, 0) D8_RD(0) D8_RD(1) D8_GP0_HI D8_MFMA(MN, 1) D8_RD(2) D8_RD(3) D8_GP1 D8_DQA0 D8_MFMA(MN, 2) D8_DQA1     \
                   D8_DQA2(CVTP) D8_MFMA(MN, 3) D8_DQB0 D8_DQB1 D8_DQB2(CVTP)                                                      \
               : [r0] "=&v"(X[0][0]), [r1] "=&v"(X[0][1]), [r2] "=&v"(X[0][2]), [r3] "=&v"(X[0][3]), [oa] "=&v"(Wf[0][1].x),         \
                 [ob] "=&v"(Wf[0][1].y), [oc] "=&v"(Wf[0][1].z), [od] "=&v"(Wf[0][1].w), [mlo] "=&v"(mlo), [mhi] "=&v"(mhi),         \
                 [sc] "=&v"(sc[1]), [nz] "=&v"(nzs[1]), [gs] "=&v"(gp_s0), [gz] "=&v"(gp_z)                                         \
               : D8_IN(1, 1), [xa] "v"(xa_), [w] "v"(w_), [iu] "v"(inv_u), [scw] "v"(SC[SET]), [zw] "v"(ZW[SET]),                   \
                 [zsh] "v"(zshift[1]), [o0] "i"(0), [o1] "i"(4096), [o2] "i"(8192), [o3] "i"(12288)                                \
               : D8_TMPS, "vcc")
#define INC_D8_SUB_G1(SET)                                                        \
  {                                                                               \
    D8_WF(1, 1)                                                                   \
    const uint32_t w_ = W[SET][0].y;                                              \
    if constexpr (IS_BF16) D8_ASM_G1("v_mfma_f32_32x32x16_bf16", D8_CVTP_BF16, SET); \
    else D8_ASM_G1("v_mfma_f32_32x32x16_f16", D8_CVTP_F16, SET);                  \
  }

  // ---- prologue: words of tiles 0, 1 and x tiles 0 .. D-1 requested; tile 0 complete ----
  INC_D8_LOADW(0, 0) INC_D8_LOADW(0, 1) INC_D8_LOADP(0)
  INC_D8_DMA(0) INC_D8_DMA(1) INC_D8_DMA(2) INC_D8_DMA(3)
  advance_w();
  advance_x();
  INC_SB();
  INC_D8_LOADW(1, 0) INC_D8_LOADW(1, 1) INC_D8_LOADP(1)
  advance_w();
#pragma unroll
  for (int d = 1; d < D; ++d) {
    INC_SB();
    INC_D8_DMA(0) INC_D8_DMA(1) INC_D8_DMA(2) INC_D8_DMA(3)
    advance_x();
  }
  INC_SB();
  INC_D8_WAIT(0)  // outstanding allowed: tile 1's 6 requests + (D - 1) x 4 pieces = 14
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int nf = 0; nf < 2; ++nf) {
    const float s0 = f16_bits_to_f32((uint16_t)(nf ? SC[0] >> 16 : SC[0]));
    uint32_t zz = ((ZW[0] >> zshift[nf]) & 15u) + 1u;
    zz = zz > 15u ? 0u : zz;
    nzs[nf] = -(float)zz * s0;
    sc[nf] = s0 * inv_u;
    Wf[0][nf] = dequant8<IS_BF16>(nf ? W[0][0].y : W[0][0].x, sc[nf], nzs[nf]);
  }
  INC_D8_XADDR(0u, 0);
  asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:4096\n\tds_read_b128 %2, %4 offset:8192\n\tds_read_b128 %3, %4 offset:12288"
               : "=&v"(X[0][0]), "=&v"(X[0][1]), "=&v"(X[0][2]), "=&v"(X[0][3])
               : "v"(xa_)
               : "memory");
  INC_SB();

  // One K-step.  Groups 0..2 (kk = g of tile t) multiply fragment set g & 1 and prepare kk = g + 1; group 3 multiplies kk = 3, takes the
  // next tile's parameters and first word, and holds the step's barrier between its two sub-regions.  The step's requests sit between the
  // sub-regions: 6 register loads of tile t+2 (groups 0, 1), the wave's 4 DMA pieces of tile t+D (groups 1, 2), the counted wait.
#define INC_D8_GROUP(CUR, NXT, KK, SET, P0, P1, LASTW)                                                  \
  INC_D8_XADDR(rd_off, KK);                                                                             \
  P0 INC_LGKM(0); INC_SB();                                                                             \
  INC_D8_SUB_R(CUR, NXT, W[SET][KK].x)                                                                  \
  INC_SB(); P1 INC_SB();                                                                                \
  LASTW                                                                                                 \
  INC_SB();
#define INC_D8_STEP(SET)                                                                                \
  {                                                                                                     \
    constexpr int s1_ = ((SET) + 1) % 3, s2_ = ((SET) + 2) % 3;                                          \
    rd_nxt = rd_off + T_ASTAGE == NS * T_ASTAGE ? 0u : rd_off + T_ASTAGE;                               \
    INC_D8_GROUP(0, 1, 1, SET, INC_D8_LOADW(s2_, 0), INC_D8_LOADW(s2_, 1), INC_D8_SUB_W(0, 1, W[SET][1].y)) \
    INC_D8_GROUP(1, 0, 2, SET, INC_D8_LOADP(s2_) INC_D8_DMA(0), INC_D8_DMA(1), INC_D8_SUB_W(1, 0, W[SET][2].y)) \
    INC_D8_GROUP(0, 1, 3, SET, INC_D8_DMA(2), INC_D8_DMA(3) INC_D8_WAIT(s1_), INC_D8_SUB_WG(0, 1, W[SET][3].y, s1_)) \
    INC_LGKM(0); INC_SB();                                                                              \
    INC_D8_SUB_G0(s1_)                                                                                  \
    INC_SB();                                                                                           \
    advance_x();                                                                                        \
    __builtin_amdgcn_s_barrier();                                                                       \
    INC_D8_XADDR(rd_nxt, 0);                                                                            \
    INC_SB();                                                                                           \
    INC_D8_SUB_G1(s1_)                                                                                  \
    INC_SB();                                                                                           \
    advance_w();                                                                                        \
    rd_off = rd_nxt;                                                                                    \
    INC_SB();                                                                                           \
  }
  for (int t0 = 0; t0 < nk; t0 += 3) {
    INC_D8_STEP(0)
    if (t0 + 1 >= nk) break;
    INC_D8_STEP(1)
    if (t0 + 2 >= nk) break;
    INC_D8_STEP(2)
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

  // ---- epilogue.  Accumulator j = 4 nf + mf, register r = 4 rq + e of it: D row i = e + 8 rq + 4 (lane >> 5) = output column
  // 64 wn + 2 i + nf, D column = x row 128 wm + 32 mf + (lane & 31): for a fixed (rq, lane >> 5) the two fragments hold the eight
  // consecutive columns c0 = 64 wn + 16 rq + 8 (lane >> 5) .. + 7 of one output row.
  if (lds_epilogue) {
    __builtin_amdgcn_s_barrier();
    static_for8<0, 4>([&](auto RQ) {
      constexpr int rq = RQ.value;
      const int c0 = wn * 64 + 16 * rq + 8 * (lane >> 5);
      float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (bias) {
#pragma unroll
        for (int j = 0; j < 8; ++j) bv[j] = cvt16<IS_BF16>(bias[n0 + c0 + j]);
      }
      static_for8<0, 4>([&](auto MF) {
        constexpr int mf = MF.value, r0 = 16 * mf + 4 * rq, r1 = 16 * (4 + mf) + 4 * rq;
        const int ml = wm * 128 + mf * 32 + (lane & 31);
        *reinterpret_cast<uint4*>(smem + ml * D8_CPITCH + c0 * 2) =
            make_uint4(cvt_pair<IS_BF16>(acc_read8<r0>() + bv[0], acc_read8<r1>() + bv[1]), cvt_pair<IS_BF16>(acc_read8<r0 + 1>() + bv[2], acc_read8<r1 + 1>() + bv[3]),
                       cvt_pair<IS_BF16>(acc_read8<r0 + 2>() + bv[4], acc_read8<r1 + 2>() + bv[5]), cvt_pair<IS_BF16>(acc_read8<r0 + 3>() + bv[6], acc_read8<r1 + 3>() + bv[7]));
      });
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    uint16_t* const ytile = y + m0 * N + n0;
#pragma unroll 4
    for (int i = tid; i < TM * (TN / 8); i += D8_THREADS) {
      const int row = i >> 5, c = i & 31;
      const uint4 v = *reinterpret_cast<const uint4*>(smem + row * D8_CPITCH + c * 16);
      const u32x4 vv = {v.x, v.y, v.z, v.w};
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(ytile + (int64_t)row * N + c * 8), "v"(vv) : "memory");
    }
    return;
  }
  float* const slab = partial ? partial + (int64_t)blockIdx.y * M * N : nullptr;
  static_for8<0, 4>([&](auto RQ) {
    constexpr int rq = RQ.value;
    const int64_t nb = n0 + wn * 64 + 16 * rq + 8 * (lane >> 5);
    float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (bias && !slab) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (nb + j < N) bv[j] = cvt16<IS_BF16>(bias[nb + j]);
    }
    static_for8<0, 4>([&](auto MF) {
      constexpr int mf = MF.value, r0 = 16 * mf + 4 * rq, r1 = 16 * (4 + mf) + 4 * rq;
      const float vv_[8] = {acc_read8<r0>() + bv[0],     acc_read8<r1>() + bv[1],     acc_read8<r0 + 1>() + bv[2], acc_read8<r1 + 1>() + bv[3],
                           acc_read8<r0 + 2>() + bv[4], acc_read8<r1 + 2>() + bv[5], acc_read8<r0 + 3>() + bv[6], acc_read8<r1 + 3>() + bv[7]};
      const int64_t m = m0 + wm * 128 + mf * 32 + (lane & 31);
      if (m < M) {
        if (slab) {
          float* dst = slab + m * N + nb;
          if (nb + 8 <= N && (N % 4) == 0) {
            *reinterpret_cast<float4*>(dst) = make_float4(vv_[0], vv_[1], vv_[2], vv_[3]);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(vv_[4], vv_[5], vv_[6], vv_[7]);
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (nb + j < N) dst[j] = vv_[j];
          }
        } else {
          uint16_t* dst = y + m * N + nb;
          if ((y_vec_ok & 1) && nb + 8 <= N) {
            *reinterpret_cast<uint2*>(dst) = make_uint2(cvt_pair<IS_BF16>(vv_[0], vv_[1]), cvt_pair<IS_BF16>(vv_[2], vv_[3]));
            *reinterpret_cast<uint2*>(dst + 4) = make_uint2(cvt_pair<IS_BF16>(vv_[4], vv_[5]), cvt_pair<IS_BF16>(vv_[6], vv_[7]));
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (nb + j < N) dst[j] = IS_BF16 ? f32_to_bf16_bits(vv_[j]) : f32_to_f16_bits(vv_[j]);
          }
        }
      }
    });
  });
}
#undef INC_SB

}  // namespace

int inc_launch_woq_gemm_d2r8(const uint16_t* x, const uint32_t* qw, const uint16_t* scales, const uint32_t* qz, const uint16_t* bias,
                             uint16_t* y, int64_t M, int64_t N, int64_t K, int64_t NW, int g_shift, int y_vec_ok, float* part, int steps,
                             int splits, bool bf, hipStream_t s) {
  const unsigned grid = (unsigned)(ceil_div64(M, TM) * ceil_div64(N, TN));
  dim3 g2(grid, (unsigned)splits);
  constexpr int smem = D8_NS * T_ASTAGE > TM * D8_CPITCH ? D8_NS * T_ASTAGE : TM * D8_CPITCH;
  static std::atomic<uint64_t> attr_set{0};
  if (inc_attr_needed(attr_set)) {
    (void)hipFuncSetAttribute((const void*)woq_gemm_w4_d2r8_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute((const void*)woq_gemm_w4_d2r8_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    inc_attr_done(attr_set);
  }
  if (bf) woq_gemm_w4_d2r8_kernel<true><<<g2, D8_THREADS, smem, s>>>(x, qw, scales, qz, bias, y, M, N, K, NW, g_shift, y_vec_ok, part, steps);
  else woq_gemm_w4_d2r8_kernel<false><<<g2, D8_THREADS, smem, s>>>(x, qw, scales, qz, bias, y, M, N, K, NW, g_shift, y_vec_ok, part, steps);
  return 0;
}
