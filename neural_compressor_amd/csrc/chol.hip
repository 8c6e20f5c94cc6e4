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
#include <stdlib.h>

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


// ---- second generation of the diagonal-block kernel (round 4): 32-column panels, the serial part inside ONE wave ---------------
// The first generation (above, kept as the harness's A/B partner) spends 16 panel steps of ~7 us each: an 8 x 8 factorisation by
// a single thread, then two LDS round trips of 8 x 8 register tiles.  Here the block lives in LDS ([128][132] fp32, L in place) and
//   step 1   wave 0 factors the 32 x 32 diagonal block with lane = row: per column one v_readlane of the pivot, a refined
//            v_rsq_f32, and 31 - k (v_readlane + v_fma) pairs -- no LDS, no barrier inside the 32 columns;
//   step 2a  wave 0 inverts that factor with lane = column (forward substitution, L entries by broadcast LDS reads) while
//   step 2b  waves 1-3 solve the rows below it (lane = row, same broadcast reads) and leave the panel k-major in `Lt`;
//   step 3   the trailing 32 x 32 tiles take their rank-32 update on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, fragments
//            straight from `Lt`);
//   phase 2  X = L^-1 by recursive doubling (32 -> 64 -> 128), again MFMA products out of LDS.
// Twelve barriers per block instead of ~45, and the serial chain is 128 x (readlane + rsq + mul) instead of 16 x (8 x 8 factor +
// inverse by one thread).
typedef __attribute__((ext_vector_type(16))) float cf32x16;
constexpr int LP = 132;  // LDS pitch (floats): rows 16-byte aligned (ds_read_b128 row access), 4-bank skew per row

__device__ __forceinline__ float lane_bcast(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// 1 / sqrt(p) and sqrt(p) to fp32 rounding: v_rsq_f32 + one Newton step each (p > 0)
__device__ __forceinline__ void rsqrt_sqrt(float p, float& rs, float& sq) {
  float y = __builtin_amdgcn_rsqf(p);
  y = y * fmaf(-0.5f * p * y, y, 1.5f);
  float d = p * y;
  d = fmaf(fmaf(-d, d, p), 0.5f * y, d);
  y = fmaf(fmaf(-d, y, 1.f), y, y);
  rs = y;
  sq = d;
}
// accumulator register r of a 32 x 32 MFMA tile: row 8 (r >> 2) + 4 (lane >> 5) + (r & 3), column lane & 31
__device__ __forceinline__ int acc_row(int r, int lane) { return 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3); }

__global__ __launch_bounds__(256) void chol_diag_block_v2_kernel(float* __restrict__ A, int64_t lda, int n, float* __restrict__ Linv, int64_t ldi,
                                                                  int32_t* __restrict__ info, int tag) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* a = reinterpret_cast<float*>(smem_raw);  // [128][LP]: the block, L in place
  float* x = a + CB * LP;                          // [128][LP]: X = L^-1
  float* Lt = x + CB * LP;                         // [32][LP]: current panel, k-major (phase 1) / [64][68] products T (phase 2)
  float* rdiag = Lt + 64 * 68;                     // [128]: 1 / L[i][i]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31;
  // ---- load the lower triangle (rows / columns >= n: identity) -----------------------------------------------------------------
  for (int idx = tid; idx < CB * (CB / 4); idx += 256) {
    const int i = idx >> 5, j4 = (idx & 31) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n && j4 <= i) v = *reinterpret_cast<const float4*>(A + (int64_t)i * lda + j4);  // (columns up to j4 + 3 < 128 <= lda)
    float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = j4 + q;
      if (j > i || j >= n) e[q] = 0.f;
      if (i >= n && j == i) e[q] = 1.f;
    }
    *reinterpret_cast<float4*>(&a[i * LP + j4]) = make_float4(e[0], e[1], e[2], e[3]);
  }
  __syncthreads();
  bool bad = false;
  for (int p = 0; p < 4; ++p) {
    const int r0 = 32 * p;
    if (wave == 0) {
      // ---- step 1: 32 x 32 Cholesky of the diagonal block, lane (& 31) = row ------------------------------------------------------
      float v[32];
#pragma unroll
      for (int c4 = 0; c4 < 8; ++c4) {
        const float4 t = *reinterpret_cast<const float4*>(&a[(r0 + l31) * LP + r0 + 4 * c4]);
        v[4 * c4] = t.x; v[4 * c4 + 1] = t.y; v[4 * c4 + 2] = t.z; v[4 * c4 + 3] = t.w;
      }
      float myrd = 0.f;
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        const float pk = lane_bcast(v[k], k);
        bad = bad || !(pk > 0.f);
        float rs, sq;
        rsqrt_sqrt(pk, rs, sq);
        const float lk = (l31 == k) ? sq : v[k] * rs;
        v[k] = lk;
        myrd = (l31 == k) ? rs : myrd;
#pragma unroll
        for (int j = k + 1; j < 32; ++j) v[j] = fmaf(-lk, lane_bcast(lk, j), v[j]);
      }
#pragma unroll
      for (int c = 0; c < 32; ++c) v[c] = c <= l31 ? v[c] : 0.f;
#pragma unroll
      for (int c4 = 0; c4 < 8; ++c4)
        *reinterpret_cast<float4*>(&a[(r0 + l31) * LP + r0 + 4 * c4]) = make_float4(v[4 * c4], v[4 * c4 + 1], v[4 * c4 + 2], v[4 * c4 + 3]);
      rdiag[r0 + l31] = myrd;
    }
    __syncthreads();
    if (wave == 0) {
      // ---- step 2a: X_pp = L_pp^-1, lane (& 31) = column j: x_i = (delta_ij - sum_{k<i} L[i][k] x_k) / L[i][i] ------------------------
      float xv[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        float s = (i == l31) ? 1.f : 0.f;
#pragma unroll
        for (int k4 = 0; k4 < (i + 3) / 4; ++k4) {
          const float4 t = *reinterpret_cast<const float4*>(&a[(r0 + i) * LP + r0 + 4 * k4]);  // uniform address: broadcast
          const float e[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (4 * k4 + q < i) s = fmaf(-e[q], xv[4 * k4 + q], s);
        }
        xv[i] = s * rdiag[r0 + i];
      }
#pragma unroll
      for (int i = 0; i < 32; ++i) x[(r0 + i) * LP + r0 + l31] = xv[i];
    } else {
      // ---- step 2b: rows below the diagonal block, lane (& 31) = row: l_c = (a_c - sum_{m<c} l_m L_pp[c][m]) / L_pp[c][c] -------------
      const int row = r0 + 32 * wave + l31;  // waves 1..3 -> rows r0 + 32 .., r0 + 64 .., r0 + 96 ..
      if (r0 + 32 * wave < CB) {
        float u[32];
#pragma unroll
        for (int c4 = 0; c4 < 8; ++c4) {
          const float4 t = *reinterpret_cast<const float4*>(&a[row * LP + r0 + 4 * c4]);
          u[4 * c4] = t.x; u[4 * c4 + 1] = t.y; u[4 * c4 + 2] = t.z; u[4 * c4 + 3] = t.w;
        }
#pragma unroll
        for (int c = 0; c < 32; ++c) {
          float s = u[c];
#pragma unroll
          for (int m4 = 0; m4 < (c + 3) / 4; ++m4) {
            const float4 t = *reinterpret_cast<const float4*>(&a[(r0 + c) * LP + r0 + 4 * m4]);  // uniform address: broadcast
            const float e[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (4 * m4 + q < c) s = fmaf(-u[4 * m4 + q], e[q], s);
          }
          u[c] = s * rdiag[r0 + c];
        }
#pragma unroll
        for (int c4 = 0; c4 < 8; ++c4)
          *reinterpret_cast<float4*>(&a[row * LP + r0 + 4 * c4]) = make_float4(u[4 * c4], u[4 * c4 + 1], u[4 * c4 + 2], u[4 * c4 + 3]);
#pragma unroll
        for (int c = 0; c < 32; ++c) Lt[c * LP + row] = u[c];
      }
    }
    __syncthreads();
    // ---- step 3: trailing 32 x 32 tiles (bi >= bj > p) -= L[bi, p] L[bj, p]^T on the fp32 matrix cores ----------------------------
    {
      const int nb = 3 - p, ntile = nb * (nb + 1) / 2;
      for (int t = wave; t < ntile; t += 4) {
        // tile order: (p+1, p+1) first (wave 0: it factors it next), then row by row
        int bi = 0, bj = 0, acc_ = t;
        for (int ii = 0; ii < nb; ++ii) {
          if (acc_ <= ii) { bi = ii; bj = acc_; break; }
          acc_ -= ii + 1;
        }
        bi += p + 1;
        bj += p + 1;
        cf32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = a[(32 * bi + acc_row(r, lane)) * LP + 32 * bj + l31];
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
          const float af = -Lt[(2 * s2 + (lane >> 5)) * LP + 32 * bi + l31];
          const float bf = Lt[(2 * s2 + (lane >> 5)) * LP + 32 * bj + l31];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) a[(32 * bi + acc_row(r, lane)) * LP + 32 * bj + l31] = acc[r];
      }
    }
    __syncthreads();
  }
  if (bad && info) atomicMax(info, tag);
  // ---- phase 2: X = L^-1 by recursive doubling; the four 32 x 32 diagonal inverses are in place ------------------------------------
  float* Ts = Lt;  // [64][68]
  constexpr int TP = 68;
  if (wave < 2) {  // level 1: pair q = wave: blocks (2q, 2q+1)
    const int c0 = 64 * wave, r1 = c0 + 32;
    cf32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {  // T = L21 X11
      const int k = 2 * s2 + (lane >> 5);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(r1 + l31) * LP + c0 + k], x[(c0 + k) * LP + c0 + l31], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) Ts[(32 * wave + acc_row(r, lane)) * TP + l31] = acc[r];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {  // X21 = -X22 T   (same wave wrote Ts: LDS operations of a wave complete in order)
      const int k = 2 * s2 + (lane >> 5);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(-x[(r1 + l31) * LP + r1 + k], Ts[(32 * wave + k) * TP + l31], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) x[(r1 + acc_row(r, lane)) * LP + c0 + l31] = acc[r];
  }
  __syncthreads();
  {  // level 2: blocks (0..63 | 64..127); tile (mi, nj) per wave
    const int mi = wave >> 1, nj = wave & 1;
    cf32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int s2 = 16 * nj; s2 < 32; ++s2) {  // T = L21 X11, X11 lower: k >= 32 nj
      const int k = 2 * s2 + (lane >> 5);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(64 + 32 * mi + l31) * LP + k], x[k * LP + 32 * nj + l31], acc, 0, 0, 0);
    }
    __syncthreads();  // level 1 is done with Ts
#pragma unroll
    for (int r = 0; r < 16; ++r) Ts[(32 * mi + acc_row(r, lane)) * TP + 32 * nj + l31] = acc[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int s2 = 0; s2 < 16 * (mi + 1); ++s2) {  // X21 = -X22 T, X22 lower: k <= 32 mi + 31
      const int k = 2 * s2 + (lane >> 5);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(-x[(64 + 32 * mi + l31) * LP + 64 + k], Ts[k * TP + 32 * nj + l31], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) x[(64 + 32 * mi + acc_row(r, lane)) * LP + 32 * nj + l31] = acc[r];
  }
  __syncthreads();
  // ---- write L (lower, zero above the diagonal) and X = L^-1 ----------------------------------------------------------------------
  for (int idx = tid; idx < CB * (CB / 4); idx += 256) {
    const int i = idx >> 5, j4 = (idx & 31) * 4;
    if (i >= n || j4 >= n) continue;
    const float4 lv = *reinterpret_cast<const float4*>(&a[i * LP + j4]);
    const float4 xv = *reinterpret_cast<const float4*>(&x[i * LP + j4]);
    float le[4] = {lv.x, lv.y, lv.z, lv.w}, xe[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (j4 + q > i) { le[q] = 0.f; xe[q] = 0.f; }
    if (j4 + 3 < n) {
      *reinterpret_cast<float4*>(A + (int64_t)i * lda + j4) = make_float4(le[0], le[1], le[2], le[3]);
      *reinterpret_cast<float4*>(Linv + (int64_t)i * ldi + j4) = make_float4(xe[0], xe[1], xe[2], xe[3]);
    } else {
      for (int q = 0; q < 4 && j4 + q < n; ++q) {
        A[(int64_t)i * lda + j4 + q] = le[q];
        Linv[(int64_t)i * ldi + j4 + q] = xe[q];
      }
    }
  }
}

}  // namespace

// launcher shared with ifac.hip (inc_gptq_inverse_factor issues one of these per 128 columns)
int inc_launch_chol_diag_block(float* A, int64_t lda, int n, float* Linv, int64_t ldi, int32_t* info, int tag, hipStream_t s) {
  // full, 16-byte-aligned blocks (every block of inc_gptq_inverse_factor): the second-generation kernel; ragged / unaligned ones
  // (the stand-alone entry point on a small matrix): the first generation.  Harness flag 201 forces the first generation (A/B).
  if (n == CB && (lda % 4) == 0 && (ldi % 4) == 0 && ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(Linv)) & 15) == 0 &&
      inc_small_tiles_flag(-1) != 201) {
    const size_t smem2 = (size_t)(2 * CB * LP + 64 * 68 + CB) * sizeof(float);
    static std::atomic<uint64_t> attr2_set{0};
    if (inc_attr_needed(attr2_set)) {
      (void)hipFuncSetAttribute((const void*)chol_diag_block_v2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2);
      inc_attr_done(attr2_set);
    }
    chol_diag_block_v2_kernel<<<1, 256, smem2, s>>>(A, lda, n, Linv, ldi, info, tag);
    INC_LAUNCH_RETURN();
  }
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
