// REJECTED EXPERIMENT (round 6, profiles/NOTES.md): correct (all GEMM parity tests green) and SLOWER than the split-K streaming kernel --
// q+k+v 12.3 vs 11.4 us, gate+up 17.9 vs 17.6, 11008 x 4096 11.7 vs 10.5 us cold.  Kept out of the build; to try it again: put it back into
// csrc/, add it to the Makefiles and restore the two routing hooks in gemm.hip (git log).
//
// gemv_fullk.hip -- K4b', decode (M <= 4) of packed 4-bit modules without split-K (round 6).
//
// The streaming kernel (gemm.hip, woq_gemv_w4_kernel) cuts K into slices over workgroups to put 512+ of them on the chip and pays for
// it after the last MFMA: write-through partials, a drain, a ticket, the last arriver's reload.  Timing-only ablations of its body on
// the cold q+k+v launch (tools/gemv_lab, profiles/NOTES.md round 6) showed four ADDITIVE phases: 3.4 us launch + wave ramp, 3.6 us of HBM
// streaming, 2.05 us of dequantise + MFMA, 1.65 us of hand-off = 10.8 us -- nothing overlaps, because every wave issues all of its
// loads, waits for all of them, computes, and hands over.  Here:
//   * a workgroup of EIGHT waves owns 64 columns over the WHOLE of K: the waves take an eighth of the K-steps each and meet in LDS --
//     one hop inside the workgroup, no partials, no ticket, y written once;
//   * a wave requests its K-steps in chunks of four (one group of 128 k), FOUR chunks ahead -- at K = 4096 its whole share before the
//     first MFMA -- and multiplies chunk c as soon as chunk c has landed (the queue returns in order: counted waits), so the VALU /
//     MFMA work runs underneath the rest of the HBM stream;
//   * x (M x K, a few KiB, L2-resident) is staged per wave into LDS once -- 16-byte fragment reads per step instead of 16 registers per
//     step -- which keeps the kernel at <= 128 registers: two workgroups (16 waves) per CU.
// Same arithmetic per weight as every other path (fp8-decoder trick, one rounding to the 16-bit type: bit-identical to inc_woq_dequant),
// fp32 accumulation in a fixed order (a wave's steps ascending, then waves 0..7): deterministic.
// Reference semantics: INCWeightOnlyLinear.forward (modules.py:594-610) = F.linear(x, recover()).
#include "gemm_common.hpp"

namespace {

typedef __attribute__((ext_vector_type(2))) uint32_t fk_u32x2;
constexpr int FK_WAVES = 8;
constexpr int FK_CH = 4;     // K-steps (32 k each) per chunk: one group of 128 k
constexpr int FK_RING = 4;   // chunks requested ahead: at K = 4096 a wave's whole share (16 steps, 16 KiB) is in flight before its first MFMA

template <bool IS_BF16, int MR, int S>  // MR = rows of x (1..4); S = K-steps per wave (16: K = 4096, 32: K = 8192) -- compile-time, so
                                       // that the whole request / multiply sequence is straight-line code with COUNTED waits
__global__ __launch_bounds__(64 * FK_WAVES, 2) void woq_gemv_fullk_kernel(GemvBatch args, const uint16_t* __restrict__ x, int M, int64_t K, int g_shift) {
  __shared__ __attribute__((aligned(16))) uint16_t xs[FK_WAVES][MR][FK_CH * 32 * 8];  // a wave's x slice: up to 32 steps of 32 k per row
  __shared__ float red[FK_WAVES][MR][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int jn = lane & 15, oct = lane >> 4;
  const float inv_u = fp8_unit_inverse();
  const int b = (int)blockIdx.x;
  int p = 0;
#pragma unroll
  for (int i = 1; i < GEMV_MAX_BATCH; ++i)
    if (i < args.n && b >= args.first[i]) p = i;
  p = __builtin_amdgcn_readfirstlane(p);
  const int64_t N = args.N[p], NW = (N + 7) / 8;
  const uint32_t* __restrict__ qweight = args.qweight[p];
  const uint16_t* __restrict__ scales = args.scales[p];
  const uint32_t* __restrict__ qzeros = args.qzeros[p];
  const int64_t n0 = (int64_t)(b - args.first[p]) * 64;
  int64_t ncol = n0 + 4 * jn;
  if (ncol > N - 4) ncol = N - 4;  // clamped lanes recompute valid columns; their results are not stored
  const int zsh = 4 * (int)(ncol & 7);
  const int step0 = wave * S;
  constexpr int nch = S / FK_CH;

  struct Chunk {
    uint4 w[FK_CH];
    fk_u32x2 s;
    uint32_t z;
  };
  auto issue = [&](Chunk& c, int ch) {
    const int st0 = step0 + ch * FK_CH;
    const int64_t g = g_shift >= 0 ? (((int64_t)st0 * 32) >> g_shift) : 0;
#pragma unroll
    for (int s = 0; s < FK_CH; ++s) c.w[s] = *reinterpret_cast<const uint4*>(qweight + ((int64_t)(st0 + s) * 4 + oct) * N + ncol);
    c.s = *reinterpret_cast<const fk_u32x2*>(scales + g * N + ncol);
    c.z = qzeros[g * NW + (ncol >> 3)];
  };
  // x slice of this wave -> LDS by LDS-DMA, FIRST in the wave's in-order queue (no registers; x is L2-resident): row m, k in
  // [step0 * 32, (step0 + S) * 32) = S / 16 pieces of 1 KiB.  Every later wait for a chunk of weights (the compiler counts ITS loads,
  // which are all younger) implies that these pieces have landed.
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    const int mm = m < M ? m : M - 1;
#pragma unroll
    for (int piece = 0; piece < S / 16; ++piece) {
      const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)&xs[wave][m][piece * 512]);
      lds_dma_1k(x + (int64_t)mm * K + (int64_t)step0 * 32 + piece * 512, dst, (uint32_t)lane * 16);
    }
  }
  Chunk ring[FK_RING];
#pragma unroll
  for (int r = 0; r < FK_RING; ++r)
    if (r < nch) issue(ring[r], r);
  __builtin_amdgcn_sched_barrier(0);  // (hipcc sinks plain loads to their first use: the requests above stay above)
  f32x4 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](const Chunk& c, int ch) {
    float scu[4], nzs[4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const float sc = f16_bits_to_f32((uint16_t)(c.s[cc >> 1] >> (16 * (cc & 1))));
      uint32_t zz = ((c.z >> (zsh + 4 * cc)) & 15u) + 1u;  // modules.py:407-410 (stored zp - 1; wraps above 15)
      zz = zz > 15u ? 0u : zz;
      scu[cc] = sc * inv_u;
      nzs[cc] = -(float)zz * sc;
    }
#pragma unroll
    for (int s = 0; s < FK_CH; ++s) {
      // A fragment: lane (jn, oct) = row jn of x, k-octet oct of this step; rows >= M are zero
      uint4 a = make_uint4(0u, 0u, 0u, 0u);
      if (jn < MR && jn < M) a = *reinterpret_cast<const uint4*>(&xs[wave][jn][(ch * FK_CH + s) * 32 + 8 * oct]);
      const uint32_t ww[4] = {c.w[s].x, c.w[s].y, c.w[s].z, c.w[s].w};
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) acc[cc] = mfma16<IS_BF16>(a, dequant8<IS_BF16>(ww[cc], scu[cc], nzs[cc]), acc[cc]);
    }
  };
  // chunk ch lives in ring[ch % FK_RING]; when it has been multiplied its buffer takes chunk ch + FK_RING (K = 8192 only)
#pragma unroll
  for (int ch = 0; ch < nch; ++ch) {
    compute(ring[ch % FK_RING], ch);
    if (ch + FK_RING < nch) {
      issue(ring[ch % FK_RING], ch + FK_RING);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- the eight waves meet in LDS: D of an MFMA = column 4 jn + c, row 4 oct + r; rows < M <= 4 live in the oct == 0 lanes ---------
  if (oct == 0) {
#pragma unroll
    for (int r = 0; r < MR; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) red[wave][r][4 * jn + c] = acc[c][r];
  }
  __syncthreads();
  const uint16_t* __restrict__ bias = args.bias[p];
  uint16_t* __restrict__ y = args.y[p];
  for (int idx = tid; idx < MR * 64; idx += 64 * FK_WAVES) {
    const int m = idx >> 6, cidx = idx & 63;
    const int64_t n = n0 + cidx;
    if (m < M && n < N) {
      float v = red[0][m][cidx];
#pragma unroll
      for (int wv = 1; wv < FK_WAVES; ++wv) v += red[wv][m][cidx];  // fixed order
      if (bias) v += cvt16<IS_BF16>(bias[n]);
      y[(int64_t)m * N + n] = IS_BF16 ? f32_to_bf16_bits(v) : f32_to_f16_bits(v);
    }
  }
}

}  // namespace

// M <= 4, groups of 128 k or more (a chunk of four K-steps lies inside one group) or a single group, a whole number of chunks per
// wave and 16 or 32 steps per wave (K = 4096 or 8192: whole 1 KiB pieces of x per wave), and enough strips that most CUs get a workgroup
bool inc_woq_gemv_fullk_ok(int64_t M, int64_t K, int g_shift, int64_t strips) {
  return M >= 1 && M <= 4 && (g_shift == -1 || g_shift >= 7) && K > 0 && (K == 4096 || K == 8192) && strips >= 128;
}

int inc_launch_woq_gemv_fullk(const GemvBatch& args, const uint16_t* x, int64_t M, int64_t K, int g_shift, int64_t strips, bool bf, hipStream_t s) {
#define INC_FK(F, R) { if (K == 4096) woq_gemv_fullk_kernel<F, R, 16><<<(unsigned)strips, 64 * FK_WAVES, 0, s>>>(args, x, (int)M, K, g_shift); \
                       else woq_gemv_fullk_kernel<F, R, 32><<<(unsigned)strips, 64 * FK_WAVES, 0, s>>>(args, x, (int)M, K, g_shift); }
  if (bf) { if (M == 1) INC_FK(true, 1) else if (M == 2) INC_FK(true, 2) else INC_FK(true, 4) }
  else { if (M == 1) INC_FK(false, 1) else if (M == 2) INC_FK(false, 2) else INC_FK(false, 4) }
#undef INC_FK
  INC_LAUNCH_RETURN();
}
