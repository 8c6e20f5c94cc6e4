// chol.hip -- K6': the diagonal-block kernel of the blocked "inverse Cholesky factor" used by GPTQ.
//
// Reference (neural_compressor/torch/algorithms/weight_only/gptq.py:1228-1230):
//     H = torch.linalg.cholesky(H); H = torch.cholesky_inverse(H); H = torch.linalg.cholesky(H, upper=True)
// i.e. the upper Cholesky factor U of H^-1 (H^-1 = U^T U).  LAPACK/rocSOLVER run that as three O(K^3)
// factorisations whose unblocked diagonal kernels dominate on the GPU (rocSOLVER potf2: 316 us per
// block, 12 % of the whole GPTQ run in profiles/r1c).  The host side (gptq.py: inverse_cholesky_upper)
// uses the identity
//     J H J = Lr Lr^T (lower Cholesky of the index-reversed matrix)   =>   U = J Lr^-1 J
// so one blocked Cholesky plus one blocked triangular inverse (both GEMM-dominated) replace the trio.
// This file holds the only non-GEMM piece: for one 128x128 diagonal block, factor it AND invert the
// factor, entirely inside one workgroup's registers + LDS.
//
//   phase 1  right-looking Cholesky; the 128x128 block lives in registers, one 8x8 sub-block per
//            thread (16x16 threads); per column k: the owner of (k,k) publishes sqrt(a_kk), the owners of
//            column k scale it (true division) and publish l_ik through LDS, every thread applies the
//            rank-1 update to its sub-block.  Two barriers per column.
//   phase 2  X = L^-1 by forward substitution, one thread per column of X, L broadcast out of LDS.
#include <math.h>

#include "common.hpp"

namespace {

constexpr int CB = 128;         // block edge
constexpr int CP = CB + 1;      // LDS pitch (floats)

__global__ __launch_bounds__(256) void chol_diag_block_kernel(float* __restrict__ A, int64_t lda, int n,
                                                              float* __restrict__ Linv, int64_t ldi,
                                                              int32_t* __restrict__ info, int tag) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* Ls = reinterpret_cast<float*>(smem_raw);  // [CB][CP]: L (phase 2 reads it)
  float* Xs = Ls + CB * CP;                         // [CB][CP]: X^T staging: Xs[k][c] = X[k][c]
  float* colbuf = Xs + CB * CP;                     // [CB] current column of L, [CB] = pivot
  const int tid = threadIdx.x;
  const int ti = tid >> 4, tj = tid & 15;           // sub-block row / column (8x8 each)

  // load the block (rows/cols >= n: identity, so the padded part factors to identity)
  float a[8][8];
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int i = ti * 8 + r, j = tj * 8 + c;
      float v = (i == j) ? 1.f : 0.f;
      if (i < n && j < n) v = (j <= i) ? A[(int64_t)i * lda + j] : A[(int64_t)j * lda + i];  // lower triangle is the source
      a[r][c] = v;
    }
  bool bad = false;
  for (int kb = 0; kb < CB / 8; ++kb) {
#pragma unroll
    for (int kr = 0; kr < 8; ++kr) {  // kr is a literal after unrolling: every register index below is static
      const int k = kb * 8 + kr;
      // pivot
      if (ti == kb && tj == kb) {
        const float p = a[kr][kr];
        if (!(p > 0.f)) bad = true;
        colbuf[CB] = sqrtf(p);
      }
      __syncthreads();
      const float d = colbuf[CB];
      // column k: l_ik = a_ik / d for i > k, l_kk = d (owners: tj == kb)
      if (tj == kb) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int i = ti * 8 + r;
          const float v = (i == k) ? d : (i > k ? a[r][kr] / d : 0.f);
          a[r][kr] = v;
          colbuf[i] = v;
        }
      }
      __syncthreads();
      // rank-1 update of the trailing lower part: a_ij -= l_ik * l_jk for j > k.  No barrier after it: the next
      // column's pivot lives in the registers of the thread that just updated it, and colbuf[0..127] is only
      // rewritten after the next barrier, which every thread reaches after finishing these reads.
      if (ti >= tj && ti * 8 + 7 > k) {
        float li[8], lj[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) li[r] = colbuf[ti * 8 + r];
#pragma unroll
        for (int c = 0; c < 8; ++c) lj[c] = colbuf[tj * 8 + c];
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
          for (int c = 0; c < 8; ++c)
            if (tj * 8 + c > k) a[r][c] = fmaf(-li[r], lj[c], a[r][c]);
      }
    }
  }
  if (bad && info) atomicMax(info, tag);
  // write L: registers -> LDS (zero above the diagonal) and -> global
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int i = ti * 8 + r, j = tj * 8 + c;
      const float v = j <= i ? a[r][c] : 0.f;
      Ls[i * CP + j] = v;
      if (i < n && j < n) A[(int64_t)i * lda + j] = v;
    }
  __syncthreads();
  // phase 2: X = L^-1 by recursive doubling inside LDS.
  //   base: the sixteen 8x8 diagonal blocks by forward substitution (one thread each);
  //   level b = 8, 16, 32, 64: every pair of inverted b-blocks [[A,0],[C,B]] gets X21 = -B^-1 (C A^-1): first
  //   T = C A^-1 is parked TRANSPOSED in the (unused) upper-triangle mirror of the X21 block, then X21 is formed from it.
  for (int idx = tid; idx < CB * CP; idx += 256) Xs[idx] = 0.f;
  __syncthreads();
  if (tid < CB / 8) {
    const int o = tid * 8;
    for (int c = 0; c < 8; ++c) {
      for (int i = c; i < 8; ++i) {
        float sum = (i == c) ? 1.f : 0.f;
        for (int k = c; k < i; ++k) sum = fmaf(-Ls[(o + i) * CP + o + k], Xs[(o + k) * CP + o + c], sum);
        Xs[(o + i) * CP + o + c] = sum / Ls[(o + i) * CP + o + i];
      }
    }
  }
  __syncthreads();
  for (int lb = 3; lb < 7; ++lb) {
    const int bsz = 1 << lb, per_pair = bsz * bsz, total = (CB / (2 * bsz)) * per_pair;
    // T[r][c] = sum_k C[r][k] * A^-1[k][c]  (A^-1 lower: k >= c)  -> Xs[s1 + c][s2 + r]
    for (int idx = tid; idx < total; idx += 256) {
      const int pr = idx >> (2 * lb), rem = idx & (per_pair - 1), r = rem >> lb, c = rem & (bsz - 1);
      const int s1 = pr * 2 * bsz, s2 = s1 + bsz;
      float t = 0.f;
      for (int k = c; k < bsz; ++k) t = fmaf(Ls[(s2 + r) * CP + s1 + k], Xs[(s1 + k) * CP + s1 + c], t);
      Xs[(s1 + c) * CP + s2 + r] = t;
    }
    __syncthreads();
    // X21[r][c] = - sum_k B^-1[r][k] * T[k][c]  (B^-1 lower: k <= r)
    for (int idx = tid; idx < total; idx += 256) {
      const int pr = idx >> (2 * lb), rem = idx & (per_pair - 1), r = rem >> lb, c = rem & (bsz - 1);
      const int s1 = pr * 2 * bsz, s2 = s1 + bsz;
      float x = 0.f;
      for (int k = 0; k <= r; ++k) x = fmaf(Xs[(s2 + r) * CP + s2 + k], Xs[(s1 + c) * CP + s2 + k], x);
      Xs[(s2 + r) * CP + s1 + c] = -x;
    }
    __syncthreads();
  }
  for (int idx = tid; idx < CB * CB; idx += 256) {
    const int i = idx >> 7, j = idx & 127;
    if (i < n && j < n) Linv[(int64_t)i * ldi + j] = j <= i ? Xs[i * CP + j] : 0.f;
  }
}

}  // namespace

extern "C" {

int inc_chol_diag_block(float* A, int64_t lda, int n, float* Linv, int64_t ldi, int32_t* info, int tag,
                        inc_stream_t stream) {
  INC_CHECK_ARG(A && Linv && n > 0 && n <= CB && lda >= n && ldi >= n);
  const size_t smem = (size_t)(2 * CB * CP + CB + 4) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)chol_diag_block_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  chol_diag_block_kernel<<<1, 256, smem, inc_s(stream)>>>(A, lda, n, Linv, ldi, info, tag);
  INC_LAUNCH_RETURN();
}

}  // extern "C"
