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
//   phase 1  right-looking Cholesky in 8-column panels; the 128x128 block lives in registers, one 8x8 tile per thread
//            (16x16 threads).  Per panel: (a) the owner of the diagonal tile factors it and inverts the 8x8 factor in
//            registers, (b) the tiles below it become A_tile * inv8^T, (c) every trailing lower tile gets
//            A_tile -= L_i L_j^T with both 8x8 panel pieces read from LDS.  Two barriers per PANEL (32 in total; a
//            column-at-a-time version needed 256 and ran 4x longer).
//   phase 2  X = L^-1 by recursive doubling in LDS, seeded with the 8x8 inverses of phase 1.
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
  float* colbuf = Xs + CB * CP;                     // panel of L [CB][8] + inverse of its diagonal factor [8][8]
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
  // Xs must be zero outside what the phases below write (padding, strict upper part)
  for (int idx = tid; idx < CB * CP; idx += 256) Xs[idx] = 0.f;
  __syncthreads();
  bool bad = false;
  float* pan = colbuf;               // [CB][8] current 8-column panel of L (row-major, 32 B per row)
  float* inv8 = colbuf + CB * 8;     // [8][8] inverse of the panel's 8x8 diagonal factor
  for (int kb = 0; kb < CB / 8; ++kb) {
    // (a) the owner of diagonal tile kb factors it (8x8 Cholesky in registers) and inverts the factor
    if (ti == kb && tj == kb) {
      // one division per pivot (its reciprocal); the 28 column scalings and the 36 rows of the inverse multiply by it -- this
      // thread is the serial part of every panel step, and an fp32 division is a ~10-instruction dependent sequence
      float rd[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float p = a[k][k];
        if (!(p > 0.f)) bad = true;
        const float d = sqrtf(p);
        a[k][k] = d;
        rd[k] = 1.0f / d;
#pragma unroll
        for (int i = k + 1; i < 8; ++i) a[i][k] = a[i][k] * rd[k];
#pragma unroll
        for (int j = k + 1; j < 8; ++j)
#pragma unroll
          for (int i = j; i < 8; ++i) a[i][j] = fmaf(-a[i][k], a[j][k], a[i][j]);
      }
      float iv[8][8];
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (i < c) { iv[i][c] = 0.f; continue; }
          float sum = (i == c) ? 1.f : 0.f;
#pragma unroll
          for (int k = c; k < i; ++k) sum = fmaf(-a[i][k], iv[k][c], sum);
          iv[i][c] = sum * rd[i];
        }
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const float v = c <= i ? a[i][c] : 0.f;
          a[i][c] = v;                                   // strict upper part of the factor is zero
          pan[(kb * 8 + i) * 8 + c] = v;
          inv8[i * 8 + c] = iv[i][c];
          Xs[(kb * 8 + i) * CP + kb * 8 + c] = iv[i][c];  // base level of phase 2
        }
    }
    __syncthreads();
    // (b) tiles below it in the panel: L_tile = A_tile * inv8^T  (row r: new[c] = sum_{m<=c} a[r][m] * inv8[c][m])
    if (tj == kb && ti > kb) {
      float iv[8][8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int c = 0; c < 8; ++c) iv[i][c] = inv8[i * 8 + c];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        float nw[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          float sum = 0.f;
#pragma unroll
          for (int m = 0; m <= c; ++m) sum = fmaf(a[r][m], iv[c][m], sum);
          nw[c] = sum;
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          a[r][c] = nw[c];
          pan[(ti * 8 + r) * 8 + c] = nw[c];
        }
      }
    }
    __syncthreads();
    // (c) trailing update of the lower tiles to the right of the panel: A_tile -= L_i * L_j^T
    if (tj > kb && ti >= tj) {
      float li[8][8], lj[8][8];
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          li[r][m] = pan[(ti * 8 + r) * 8 + m];
          lj[r][m] = pan[(tj * 8 + r) * 8 + m];
        }
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          float sum = a[r][c];
#pragma unroll
          for (int m = 0; m < 8; ++m) sum = fmaf(-li[r][m], lj[c][m], sum);
          a[r][c] = sum;
        }
    }
    // no barrier here: the next panel's diagonal tile is in the registers of the thread that just updated it, and
    // `pan` / `inv8` are rewritten only after the next __syncthreads(), which every thread reaches after these reads
  }
  if (bad && info) atomicMax(info, tag);
  // write L: registers -> LDS and -> global (zero above the diagonal)
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int i = ti * 8 + r, j = tj * 8 + c;
      const float v = (tj < ti || (tj == ti && c <= r)) ? a[r][c] : 0.f;
      Ls[i * CP + j] = v;
      if (i < n && j < n) A[(int64_t)i * lda + j] = v;
    }
  __syncthreads();
  // phase 2: X = L^-1 by recursive doubling inside LDS (the 8x8 diagonal blocks were inverted in phase 1).
  //   level b = 8, 16, 32, 64: every pair of inverted b-blocks [[A,0],[C,B]] gets X21 = -B^-1 (C A^-1): first
  //   T = C A^-1 is parked TRANSPOSED in the (unused) upper-triangle mirror of the X21 block, then X21 is formed from it.
  // Each level is two dense b x b x b products out of LDS, register-blocked 4 x 4 per thread (16 independent FMA chains;
  // a one-output-per-thread version was bound by the LDS latency of its single chain and took half of the kernel):
  //   T   = C * A^-1   -> parked transposed in the (otherwise unused, zero) upper mirror of the C block inside Ls
  //   X21 = -B^-1 * T  -> Xs lower block.  The upper triangle of Xs stays zero throughout, so no index conditions.
  for (int lb = 3; lb < 7; ++lb) {
    const int bsz = 1 << lb, nb4 = bsz >> 2, lbb = lb - 2;          // nb4 = 4x4 blocks per edge = 1 << lbb
    const int per_pair = nb4 * nb4, total = (CB / (2 * bsz)) * per_pair;
    const int pr = tid >> (2 * lbb), rem = tid & (per_pair - 1), br = (rem >> lbb) * 4, bc = (rem & (nb4 - 1)) * 4;
    const int s1 = pr * 2 * bsz, s2 = s1 + bsz;
    float o[4][4];
    if (tid < total) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) o[i][j] = 0.f;
#pragma unroll 4
      for (int k = 0; k < bsz; ++k) {
        float cv[4], xv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) cv[i] = Ls[(s2 + br + i) * CP + s1 + k];
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[j] = Xs[(s1 + k) * CP + s1 + bc + j];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) o[i][j] = fmaf(cv[i], xv[j], o[i][j]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Ls[(s1 + bc + j) * CP + s2 + br + i] = o[i][j];  // T^T into the upper mirror
    }
    __syncthreads();
    if (tid < total) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) o[i][j] = 0.f;
#pragma unroll 4
      for (int k = 0; k < bsz; ++k) {
        float bv[4], tv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[i] = Xs[(s2 + br + i) * CP + s2 + k];
#pragma unroll
        for (int j = 0; j < 4; ++j) tv[j] = Ls[(s1 + bc + j) * CP + s2 + k];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) o[i][j] = fmaf(bv[i], tv[j], o[i][j]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Xs[(s2 + br + i) * CP + s1 + bc + j] = -o[i][j];
    }
    __syncthreads();
  }
  for (int idx = tid; idx < CB * CB; idx += 256) {
    const int i = idx >> 7, j = idx & 127;
    if (i < n && j < n) Linv[(int64_t)i * ldi + j] = j <= i ? Xs[i * CP + j] : 0.f;
  }
}

}  // namespace

// launcher shared with ifac.hip (inc_gptq_inverse_factor issues one of these per 128 columns)
int inc_launch_chol_diag_block(float* A, int64_t lda, int n, float* Linv, int64_t ldi, int32_t* info, int tag, hipStream_t s) {
  const size_t smem = (size_t)(2 * CB * CP + CB * 8 + 64) * sizeof(float);
  static std::atomic<uint64_t> attr_set{0};
  if (inc_attr_needed(attr_set)) {
    (void)hipFuncSetAttribute((const void*)chol_diag_block_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    inc_attr_done(attr_set);
  }
  chol_diag_block_kernel<<<1, 256, smem, s>>>(A, lda, n, Linv, ldi, info, tag);
  INC_LAUNCH_RETURN();
}

extern "C" {

int inc_chol_diag_block(float* A, int64_t lda, int n, float* Linv, int64_t ldi, int32_t* info, int tag,
                        inc_stream_t stream) {
  INC_CHECK_ARG(A && Linv && n > 0 && n <= CB && lda >= n && ldi >= n);
  return inc_launch_chol_diag_block(A, lda, n, Linv, ldi, info, tag, inc_s(stream));
}

}  // extern "C"
