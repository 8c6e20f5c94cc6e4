// ifac.hip -- K6': the whole "inverse Cholesky factor" of GPTQ behind ONE C-ABI call (inc_gptq_inverse_factor).
//
// Reference (neural_compressor/torch/algorithms/weight_only/gptq.py:1228-1231):
//     H = torch.linalg.cholesky(H); H = torch.cholesky_inverse(H); H = torch.linalg.cholesky(H, upper=True); Hinv = H
// i.e. the upper Cholesky factor U of H^-1 (H^-1 = U^T U).  With J the index reversal,
//     J H J = L L^T (lower Cholesky of the index-reversed matrix)   =>   U = J L^-1 J,
// so ONE blocked Cholesky and ONE blocked triangular inverse replace the three LAPACK factorisations (half the flops, one
// rounding pass; at least as close to the fp64 result as the fp32 trio -- tests/test_gpu_parity.py).  Round 1-3 ran the blocked
// algorithm from Python with torch.mm for every product; this file owns all of it:
//   * f32gemm_kernel        exact-fp32 GEMM on the matrix cores (v_mfma_f32_32x32x2_f32: fmaf-chain semantics), NT and NN forms,
//                           triangular operands skipped by K-range per tile, lower-triangle-only outputs (syrk), batched pairs
//   * ifac_split3_kernel + bf16x3_gemm_kernel (flags bit 1: the Python driver's DEFAULT)  the large products of the MAIN stream with
//                           their operands pre-split into three bf16 planes: six bf16 MFMAs per fp32 product, LDS-DMA ring (see the
//                           comment in front of them); products issued on the optional second stream stay exact fp32 (one plane
//                           buffer), so one- and two-stream forms are bit-identical for flags = 0 only
//   * chol_diag_block_kernel (chol.hip) the 128 x 128 diagonal block: factor + inverse of the factor in one workgroup
//   * ifac_flip_* / ifac_copy_panel: index reversal in / out, panel write-back
//   * the host side below issues them on the caller's stream (+ an optional second stream for the look-ahead over outer blocks)
// Two-level blocking (128 inside IFAC_OUTER columns), only the LOWER triangle of the working copy is read or kept up to date.
#include <algorithm>
#include <vector>

#include "common.hpp"

int inc_launch_chol_diag_block(float* A, int64_t lda, int n, float* Linv, int64_t ldi, int32_t* info, int tag, hipStream_t s);  // chol.hip

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int IFAC_NB = 128;      // leaf (diagonal block) edge
constexpr int IFAC_OUTER = 1024;  // outer block of the two-level factorisation
constexpr int BK = 16;            // k per LDS stage

enum { KR_FULL = 0, KR_B_LOWER_NT = 1, KR_B_LOWER_NN = 2, KR_A_LOWER = 3 };

struct GemmArgs {
  const float* A;  // A(m, k) = A[m * lda + k]
  const float* B;  // NT: B(k, n) = B[n * ldb + k];  NN: B(k, n) = B[k * ldb + n]
  float* C;        // C(m, n) = C[m * ldc + n]
  int64_t lda, ldb, ldc;
  int64_t sA, sB, sC;  // element strides between the members of a batch (blockIdx.y)
  int M, N, K;
  float alpha, beta;
  int krange;      // which K-range a tile needs (triangular operands): see tile_k_range
  int lower_only;  // square C: only tiles that touch the lower triangle (tile column start <= tile row end)
};

// C = beta C + alpha A B on the fp32 matrix cores.  Workgroup = 4 waves as 2 x 2, wave tile (TM/2) x (TN/2) = MI x NI MFMA tiles of
// 32 x 32; operands staged k-major in LDS (As[k][m], Bs[k][n]: the fragment of v_mfma_f32_32x32x2_f32 is lane -> (row or column
// = lane & 31, k = lane >> 5), i.e. two consecutive 128-byte LDS rows: conflict-free ds_read_b32), two stages, register prefetch.
// The fp32 MFMA runs at 1/16 of the bf16 rate (256 flop / clk / CU): 16 KiB of operands per 2048 MFMA cycles -- every other
// cost is small beside it, which is why the loaders are simple.
// C <- alpha * acc + beta * C for a wave's MI x NI accumulators of 32 x 32: register r = 4 rq + e of a tile is row 8 rq + 4 (lane >> 5) + e,
// column lane & 31.  beta != 0: EVERY old value is requested before the first is used (clamped addresses, no branches around the
// loads) -- the obvious per-element form compiles to 16 MI NI serial load -> wait -> fma -> store round trips per lane, which was
// longer than the K-loop of the trailing updates (K = 1024).
template <int MI, int NI>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, float* __restrict__ C, f32x16 (&acc)[MI][NI], int row0, int col0, int lane) {
  const int cl = lane & 31, rh = 4 * (lane >> 5);
  const bool rmw = g.beta != 0.f;
  float old[MI][NI][16];
  if (rmw) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int col = min(col0 + 32 * j + cl, g.N - 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = min(row0 + 32 * i + 8 * (r >> 2) + rh + (r & 3), g.M - 1);
          old[i][j][r] = C[(int64_t)row * g.ldc + col];
        }
      }
  }
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int col = col0 + 32 * j + cl;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + 32 * i + 8 * (r >> 2) + rh + (r & 3);
        if (row < g.M && col < g.N) {
          float v = g.alpha * acc[i][j][r];
          if (rmw) v += g.beta * old[i][j][r];
          C[(int64_t)row * g.ldc + col] = v;
        }
      }
    }
}

template <int TM, int TN, bool B_NT>
__global__ __launch_bounds__(256) void f32gemm_kernel(GemmArgs g) {
  constexpr int MI = TM / 64, NI = TN / 64;
  constexpr int PA = TM + 4, PB = TN + 4;  // LDS pitches (floats): +4 keeps the two k rows of a fragment on different banks
  __shared__ __attribute__((aligned(16))) float As[2][BK * PA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * PB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (g.N + TN - 1) / TN;
  int ti, tj;
  if (g.lower_only) {  // linear index over the lower triangle of tiles (TM == TN): ti (ti + 1) / 2 + tj
    const int t = blockIdx.x;
    ti = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    while (ti * (ti + 1) / 2 > t) --ti;
    tj = t - ti * (ti + 1) / 2;
  } else {
    // tiles in order of DECREASING K-range (the hardware hands out workgroups in index order: longest first packs the last round)
    const int tiles_m = (g.M + TM - 1) / TM, b = blockIdx.x;
    if (g.krange == KR_B_LOWER_NN) {         // k_lo = n0: the leftmost tile columns are the longest
      tj = b / tiles_m;
      ti = b - tj * tiles_m;
    } else if (g.krange == KR_B_LOWER_NT) {  // k_hi = n0 + TN: the rightmost tile columns are the longest
      tj = b / tiles_m;
      ti = b - tj * tiles_m;
      tj = tiles_n - 1 - tj;
    } else if (g.krange == KR_A_LOWER) {     // k_hi = m0 + TM: the bottom tile rows are the longest
      ti = b / tiles_n;
      tj = b - ti * tiles_n;
      ti = tiles_m - 1 - ti;
    } else {
      ti = b / tiles_n;
      tj = b - ti * tiles_n;
    }
  }
  const int m0 = ti * TM, n0 = tj * TN;
  int k_lo = 0, k_hi = g.K;
  if (g.krange == KR_B_LOWER_NT) k_hi = min(g.K, n0 + TN);       // B[n, k] = 0 for k > n
  else if (g.krange == KR_B_LOWER_NN) k_lo = min(n0, g.K);       // B[k, n] = 0 for k < n
  else if (g.krange == KR_A_LOWER) k_hi = min(g.K, m0 + TM);     // A[m, k] = 0 for k > m
  k_lo &= ~(BK - 1);
  // (no __restrict__: the in-place panel solve passes C == A -- every load of a tile precedes its stores, see launch_f32gemm)
  const float* A = g.A + (int64_t)blockIdx.y * g.sA;
  const float* B = g.B + (int64_t)blockIdx.y * g.sB;
  float* C = g.C + (int64_t)blockIdx.y * g.sC;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- loaders: global -> registers (next stage) -> LDS --------------------------------------------------------------------
  constexpr int NA = TM / 64;  // float4 per thread of the A tile (TM rows x 16 k)
  constexpr int NB_ = TN / 64;
  float4 ra[NA], rb[NB_];
  auto load_a = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int idx = tid + 256 * i, row = idx >> 2, c4 = idx & 3;
      const int m = m0 + row;
      ra[i] = (m < g.M && k0 + 4 * c4 < g.K) ? *reinterpret_cast<const float4*>(A + (int64_t)m * g.lda + k0 + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_a = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int idx = tid + 256 * i, row = idx >> 2, c4 = idx & 3;
      float* d = &As[buf][(4 * c4) * PA + row];
      d[0] = ra[i].x; d[PA] = ra[i].y; d[2 * PA] = ra[i].z; d[3 * PA] = ra[i].w;
    }
  };
  auto load_b = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NB_; ++i) {
      const int idx = tid + 256 * i;
      if constexpr (B_NT) {
        const int row = idx >> 2, c4 = idx & 3, n = n0 + row;
        rb[i] = (n < g.N && k0 + 4 * c4 < g.K) ? *reinterpret_cast<const float4*>(B + (int64_t)n * g.ldb + k0 + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        const int k = idx / (TN / 4), n4 = idx - k * (TN / 4), n = n0 + 4 * n4;
        rb[i] = (n < g.N && k0 + k < g.K) ? *reinterpret_cast<const float4*>(B + (int64_t)(k0 + k) * g.ldb + n) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NB_; ++i) {
      const int idx = tid + 256 * i;
      if constexpr (B_NT) {
        const int row = idx >> 2, c4 = idx & 3;
        float* d = &Bs[buf][(4 * c4) * PB + row];
        d[0] = rb[i].x; d[PB] = rb[i].y; d[2 * PB] = rb[i].z; d[3 * PB] = rb[i].w;
      } else {
        const int k = idx / (TN / 4), n4 = idx - k * (TN / 4);
        *reinterpret_cast<float4*>(&Bs[buf][k * PB + 4 * n4]) = rb[i];
      }
    }
  };

  const int nsteps = (k_hi - k_lo + BK - 1) / BK;
  if (nsteps > 0) {
    load_a(k_lo);
    load_b(k_lo);
    store_a(0);
    store_b(0);
  }
  __syncthreads();
  const int fa = (lane >> 5) * PA + wm * (TM / 2) + (lane & 31);
  const int fb = (lane >> 5) * PB + wn * (TN / 2) + (lane & 31);
  for (int s = 0; s < nsteps; ++s) {
    const int buf = s & 1;
    if (s + 1 < nsteps) {
      load_a(k_lo + (s + 1) * BK);
      load_b(k_lo + (s + 1) * BK);
    }
    // fragments one k-pair ahead of the MFMAs that use them (the compiler's own order was read -> wait -> 4 MFMAs -> read ...)
    float a[2][MI], b[2][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) a[0][i] = As[buf][fa + 32 * i];
#pragma unroll
    for (int j = 0; j < NI; ++j) b[0][j] = Bs[buf][fb + 32 * j];
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const int cur = kk & 1, nxt = cur ^ 1;
      if (kk + 1 < BK / 2) {
#pragma unroll
        for (int i = 0; i < MI; ++i) a[nxt][i] = As[buf][2 * (kk + 1) * PA + fa + 32 * i];
#pragma unroll
        for (int j = 0; j < NI; ++j) b[nxt][j] = Bs[buf][2 * (kk + 1) * PB + fb + 32 * j];
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the reads in front of this k-pair's MFMAs
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
    }
    if (s + 1 < nsteps) {
      store_a(buf ^ 1);  // the other stage: its last readers passed the barrier at the end of step s - 1
      store_b(buf ^ 1);
    }
    __syncthreads();
  }
  gemm_epilogue<MI, NI>(g, C, acc, m0 + wm * (TM / 2), n0 + wn * (TN / 2), lane);
}


// ---- opt-in (flags bit 1): the LARGE products with fp32 operands split into three bf16 pieces ---------------------------------------
// a = a1 + a2 + a3 (each piece the bf16 rounding of what is left: 24+ bits together); a b ~ a1 b1 + a1 b2 + a2 b1 + a1 b3 + a3 b1 +
// a2 b2 -- six v_mfma_f32_32x32x16_bf16 per 32 x 32 x 16 block (the dropped terms are <= 2^-24 |a b|), fp32 accumulation: 768 matrix
// cycles per 16 k of a wave's 64 x 64 against 2048 for v_mfma_f32_32x32x2_f32.  NOT the exact-fp32 arithmetic of the default
// path: its distance to the fp64 factor is measured in tests / scripts/chol_time.py.  Splitting inside the GEMM (every tile
// re-splitting its operands) is VALU-bound and slower than the fp32 kernel (21.2 vs 17.3 ms at K = 11008): a pre-pass writes each
// operand ONCE as three bf16 planes in the layout the GEMM streams -- [plane][k / 16][row][k % 16], so a 128-row tile of one K-step
// is 4 KiB contiguous per plane -- and the GEMM itself has no VALU work besides addresses.
// compile-time loop: f(std::integral_constant<int, B>) ... f(std::integral_constant<int, E - 1>)
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}
constexpr int X3_KB = 16;
struct X3Planes {
  const uint16_t* Ap;
  const uint16_t* Bp;
  int64_t a_plane, b_plane;  // elements between the planes of one operand
  int64_t RpA, RpB;          // rows (padded to 128) of the operands
};
__device__ __forceinline__ uint32_t split_hi(float v, float& rest) {  // bf16 rounding (RNE) of v as its 16 bits, rest = v - that
  const uint32_t b = f32_to_bf16_bits(v);
  rest = v - bf16_bits_to_f32((uint16_t)b);
  return b;
}
// element (r, k) of the operand = src[r * ld + k] (TR = false) or src[k * ld + r] (TR = true: B of an NN product); rows >= `rows`
// and k >= `cols` are written as zeros up to the padded extents (Rp rows, kq = gridDim.x * 64 >= cols columns)
template <bool TR>
__global__ __launch_bounds__(256) void ifac_split3_kernel(const float* __restrict__ src, int64_t ld, int rows, int cols, uint16_t* __restrict__ dst,
                                                          int64_t plane, int64_t Rp, int nkb) {
  __shared__ float tile[TR ? 64 : 1][65];
  const int t = threadIdx.x, row = t >> 2, q = t & 3;
  const int r0 = blockIdx.y * 64, k0 = blockIdx.x * 64;
  float v[4][4];
  if constexpr (!TR) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + 16 * j + 4 * q;
      float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r0 + row < rows && k < cols) w = *reinterpret_cast<const float4*>(src + (int64_t)(r0 + row) * ld + k);
      v[j][0] = w.x; v[j][1] = w.y; v[j][2] = w.z; v[j][3] = w.w;
    }
  } else {
    const int kk = t >> 4, n4 = t & 15;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + kk + 16 * j;
      float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < cols && r0 + 4 * n4 < rows) w = *reinterpret_cast<const float4*>(src + (int64_t)k * ld + r0 + 4 * n4);
      tile[kk + 16 * j][4 * n4 + 0] = w.x; tile[kk + 16 * j][4 * n4 + 1] = w.y; tile[kk + 16 * j][4 * n4 + 2] = w.z; tile[kk + 16 * j][4 * n4 + 3] = w.w;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[j][e] = tile[16 * j + 4 * q + e][row];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int kb = (k0 >> 4) + j;
    if (kb >= nkb) break;
    uint32_t p1[4], p2[4], p3[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float r1, r2, r3;
      p1[e] = split_hi(v[j][e], r1);
      p2[e] = split_hi(r1, r2);
      p3[e] = split_hi(r2, r3);
    }
    // the two 8-k halves of a row's 32 bytes swap places on rows with bit 3 set (the GEMM's LDS image is a linear copy of this)
    uint16_t* d = dst + ((int64_t)kb * Rp + r0 + row) * X3_KB + 4 * (q ^ ((row >> 2) & 2));
    *reinterpret_cast<uint2*>(d) = make_uint2(p1[0] | (p1[1] << 16), p1[2] | (p1[3] << 16));
    *reinterpret_cast<uint2*>(d + plane) = make_uint2(p2[0] | (p2[1] << 16), p2[2] | (p2[3] << 16));
    *reinterpret_cast<uint2*>(d + 2 * plane) = make_uint2(p3[0] | (p3[1] << 16), p3[2] | (p3[3] << 16));
  }
}

typedef __attribute__((ext_vector_type(8))) __bf16 ibf16x8;
// C = alpha * A B^T + beta * C over the K-range of the tile (g.A / g.B unused: both operands come as planes, B always as [n][k]).
// TM x 128 per workgroup (TM = 128: four waves, two workgroups per CU; TM = 256: eight waves, one per CU), waves of 64 x 64; a K-step
// (16 k) of the tile is 1 KiB contiguous pieces (3 planes x TM / 32 of A, 3 x 4 of B), dealt round-robin to the waves and copied
// global -> LDS by LDS-DMA (no VGPR staging, no ds_write) into a ring of NS stages: the requests of step s + NS - 1 are issued at
// step s, behind ONE barrier per step (which also retires the stage they overwrite).  With the first, register-staged two-stage
// form (requests one step ahead of a 0.4 us step) the loop waited on L2 latency: 136 "fp32" TFLOP/s on the first trailing update of
// K = 11008 against 100 for the fp32-MFMA kernel.  The operand stream is 32 B/clk/CU at 128 x 128; the TM = 256 form (24 B/clk/CU,
// three steps in flight, one workgroup per CU) measured 2 % SLOWER (profiles/NOTES.md) and is not instantiated.
template <int TM>
__global__ __launch_bounds__(TM * 2) void bf16x3_gemm_kernel(GemmArgs g, X3Planes p) {
  constexpr int TN = 128, MI = 2, NI = 2;
  constexpr int NW = TM / 32;                         // waves: (TM / 64) x 2
  constexpr int NS = TM == 128 ? 3 : 4, D = NS - 1;   // stages, steps the DMA runs ahead
  constexpr int PA = 3 * (TM / 32), PB = 12, P = PA + PB;  // 1-KiB pieces of A, of B, per step
  constexpr int APLANE = TM * 32, BPLANE = 4096, STAGE = P * 1024;  // bytes
  constexpr int NPW = (P + NW - 1) / NW, NFULL = P % NW == 0 ? NW : P % NW;  // pieces per wave (waves >= NFULL have one less)
  __shared__ __attribute__((aligned(1024))) char smem[NS * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (g.N + TN - 1) / TN, tiles_m = (g.M + TM - 1) / TM;
  int ti, tj;
  if (g.lower_only) {
    const int t = blockIdx.x;
    if constexpr (TM == TN) {
      ti = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
      while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
      while (ti * (ti + 1) / 2 > t) --ti;
      tj = t - ti * (ti + 1) / 2;
    } else {  // 256-row blocks x 128-column blocks: row block ti holds column blocks 0 .. 2 ti + 1 -> ti (ti + 1) tiles before it
      ti = (int)((sqrtf(4.f * (float)t + 1.f) - 1.f) * 0.5f);
      while ((ti + 1) * (ti + 2) <= t) ++ti;
      while (ti * (ti + 1) > t) --ti;
      tj = t - ti * (ti + 1);
    }
  } else {
    const int b = blockIdx.x;
    if (g.krange == KR_B_LOWER_NN) { tj = b / tiles_m; ti = b - tj * tiles_m; }
    else if (g.krange == KR_B_LOWER_NT) { tj = b / tiles_m; ti = b - tj * tiles_m; tj = tiles_n - 1 - tj; }
    else if (g.krange == KR_A_LOWER) { ti = b / tiles_n; tj = b - ti * tiles_n; ti = tiles_m - 1 - ti; }
    else { ti = b / tiles_n; tj = b - ti * tiles_n; }
  }
  const int m0 = ti * TM, n0 = tj * TN;
  if (n0 >= g.N) return;  // (the last 256-row block of a ragged lower-only product)
  int k_lo = 0, k_hi = g.K;
  if (g.krange == KR_B_LOWER_NT) k_hi = min(g.K, n0 + TN);
  else if (g.krange == KR_B_LOWER_NN) k_lo = min(n0, g.K);
  else if (g.krange == KR_A_LOWER) k_hi = min(g.K, m0 + TM);
  const int kb_lo = k_lo / X3_KB, kb_hi = (k_hi + X3_KB - 1) / X3_KB;
  const int nsteps = kb_hi - kb_lo;
  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if (nsteps > 0) {
    // this wave's pieces: q = wave + NW i.  q < PA: A, plane q / (TM / 32), 32-row slice q % (TM / 32); else B, plane (q - PA) / 4, slice % 4
    const uint16_t* pq[NPW];
    int64_t stepq[NPW];
    uint32_t ldsq[NPW];
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      int q = wave + NW * i;
      if (q > P - 1) q = P - 1;  // (never issued: see NFULL)
      const bool isa = q < PA;
      const int pl = isa ? q / (TM / 32) : (q - PA) / 4, sl = isa ? q % (TM / 32) : (q - PA) % 4;
      stepq[i] = (isa ? p.RpA : p.RpB) * X3_KB;
      pq[i] = (isa ? p.Ap + (int64_t)m0 * X3_KB + pl * p.a_plane : p.Bp + (int64_t)n0 * X3_KB + pl * p.b_plane) + sl * 512 + kb_lo * stepq[i];
      ldsq[i] = __builtin_amdgcn_readfirstlane(lds0 + (isa ? pl * APLANE : 3 * APLANE + pl * BPLANE) + sl * 1024);
    }
    const uint32_t voff = lane * 16;
#define INC_X3_DMA(I, STG, STEP)                                                                                             \
  asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(pq[I] + (STEP) * stepq[I]), \
               "s"(ldsq[I]), "s"((uint32_t)((STG) * STAGE)) : "memory", "scc")
    auto issue = [&](int stg, int step) {  // step clamped: past-the-end requests re-read the last block into a stage nobody reads again
      const int64_t st = (int64_t)min(step, nsteps - 1);
      static_assert(NPW == 5 || NPW == 6, "pieces per wave");
      INC_X3_DMA(0, stg, st); INC_X3_DMA(1, stg, st); INC_X3_DMA(2, stg, st); INC_X3_DMA(3, stg, st);
      if constexpr (NPW == 6) INC_X3_DMA(4, stg, st);
      if (NFULL == NW || wave < NFULL) INC_X3_DMA(NPW - 1, stg, st);
    };
#pragma unroll
    for (int d = 0; d < D; ++d) issue(d, d);
    // fragment of row r, k-half h of a plane: 32 r + 16 (h ^ ((r >> 3) & 1)) -- the split pre-pass writes the halves in that order, so
    // that the linear DMA image is conflict-free for ds_read_b128
    const int fsw = ((lane & 31) * 32) + ((((lane >> 5) ^ ((lane >> 3) & 1)) & 1) * 16);
    const int fa = wm * 64 * 32 + fsw, fb = 3 * APLANE + wn * 64 * 32 + fsw;
    int st = 0, st2 = D;
    for (int s = 0; s < nsteps; ++s) {
      // the pieces of step s have landed (those of the D - 1 steps behind it may be in flight)
      if (NFULL == NW || wave < NFULL) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((D - 1) * NPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"i"((D - 1) * (NPW - 1)) : "memory");
      __syncthreads();  // ... every wave's; and every wave is done with stage st2 (read at step s - 1)
      issue(st2, s + D);
      const char* const sb = smem + st * STAGE;
      ibf16x8 a[3][MI], b[3][NI];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
        for (int i = 0; i < MI; ++i) a[pl][i] = *reinterpret_cast<const ibf16x8*>(sb + pl * APLANE + fa + 32 * i * 32);
#pragma unroll
        for (int j = 0; j < NI; ++j) b[pl][j] = *reinterpret_cast<const ibf16x8*>(sb + pl * BPLANE + fb + 32 * j * 32);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {  // small terms first
          f32x16 c = acc[i][j];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], c, 0, 0, 0);
          acc[i][j] = c;
        }
      st = st == NS - 1 ? 0 : st + 1;
      st2 = st2 == NS - 1 ? 0 : st2 + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped tail requests must not outlive the workgroup's LDS
#undef INC_X3_DMA
  }
  gemm_epilogue<MI, NI>(g, g.C, acc, m0 + wm * 64, n0 + wn * 64, lane);
}

// the split pre-pass of one operand into `dst` (3 planes of Rp x kq bf16); returns the bytes it occupies
int64_t launch_split3(const float* src, int64_t ld, int rows, int cols, bool tr, int row_pad, uint16_t* dst, int64_t* plane_out, int64_t* rp_out,
                      hipStream_t s) {
  const int64_t Rp = ceil_div64(rows, row_pad) * row_pad, nkb = ceil_div64(cols, X3_KB), plane = nkb * Rp * X3_KB;
  dim3 grid((unsigned)ceil_div64(cols, 64), (unsigned)(Rp / 64));
  if (tr) ifac_split3_kernel<true><<<grid, 256, 0, s>>>(src, ld, rows, cols, dst, plane, Rp, (int)nkb);
  else ifac_split3_kernel<false><<<grid, 256, 0, s>>>(src, ld, rows, cols, dst, plane, Rp, (int)nkb);
  *plane_out = plane;
  *rp_out = Rp;
  return 3 * plane * (int64_t)sizeof(uint16_t);
}

// `planes` (flags bit 1, `planes_bytes` of them): the large un-batched products whose split operands fit run as bf16 x 3 splits
int launch_f32gemm(const GemmArgs& g, bool b_nt, int batch, hipStream_t s, void* planes = nullptr, int64_t planes_bytes = 0) {
  if (g.M <= 0 || g.N <= 0 || batch <= 0) return INC_OK;
  // tile choice: 128 x 128 when that still gives the chip enough workgroups, else 64 x 64 (the chain's small products)
  auto ntiles = [&](int t) {
    const int64_t tm = (g.M + t - 1) / t, tn = (g.N + t - 1) / t;
    return g.lower_only ? tm * (tm + 1) / 2 : tm * tn;
  };
  if (g.C == g.A) {
    // in place (C == A): legal only when ONE tile spans every column and the whole K of its rows -- its K-loop has read the row
    // block completely before the epilogue stores it, and no other workgroup touches those rows.  The 128-column panel solves.
    if (!(b_nt && g.N <= 128 && g.K <= 128 && !g.lower_only && batch == 1)) return INC_ERR_BAD_ARG;
    dim3 grid((unsigned)((g.M + 63) / 64), 1);
    f32gemm_kernel<64, 128, true><<<grid, 256, 0, s>>>(g);
    return hipGetLastError() == hipSuccess ? INC_OK : INC_ERR_LAUNCH;
  }
  const bool big = ntiles(128) * batch >= 192;
  const int t = big ? 128 : 64;
  dim3 grid((unsigned)ntiles(t), (unsigned)batch);
  const bool syrk_form = b_nt && g.B == g.A && g.ldb == g.lda && g.N == g.M;
  // bytes of the split operands: 3 planes x rows (padded to 128) x k (padded to 16) x 2 bytes each
  const int64_t kpad = ceil_div64(g.K, X3_KB) * X3_KB;
  const int64_t need = 6 * kpad * (ceil_div64(g.M, 128) * 128 + (syrk_form ? 0 : ceil_div64(g.N, 128) * 128));
  if (big && planes && need <= planes_bytes && batch == 1 && (g.lda & 3) == 0 && (g.ldb & 3) == 0 && (g.K & 3) == 0 && (g.N & 3) == 0 &&
      ((reinterpret_cast<uintptr_t>(g.A) | reinterpret_cast<uintptr_t>(g.B)) & 15) == 0) {
    X3Planes p;
    uint16_t* pa = static_cast<uint16_t*>(planes);
    const int64_t abytes = launch_split3(g.A, g.lda, g.M, g.K, false, 128, pa, &p.a_plane, &p.RpA, s);
    p.Ap = pa;
    if (syrk_form) {  // one operand
      p.Bp = pa; p.b_plane = p.a_plane; p.RpB = p.RpA;
    } else {
      uint16_t* pb = reinterpret_cast<uint16_t*>(static_cast<char*>(planes) + abytes);
      (void)launch_split3(g.B, g.ldb, g.N, g.K, !b_nt, 128, pb, &p.b_plane, &p.RpB, s);
      p.Bp = pb;
    }
    bf16x3_gemm_kernel<128><<<grid, 256, 0, s>>>(g, p);
  } else if (big) {
    if (b_nt) f32gemm_kernel<128, 128, true><<<grid, 256, 0, s>>>(g);
    else f32gemm_kernel<128, 128, false><<<grid, 256, 0, s>>>(g);
  } else {
    if (b_nt) f32gemm_kernel<64, 64, true><<<grid, 256, 0, s>>>(g);
    else f32gemm_kernel<64, 64, false><<<grid, 256, 0, s>>>(g);
  }
  return hipGetLastError() == hipSuccess ? INC_OK : INC_ERR_LAUNCH;
}

// A[i, j] = H[K-1-i, K-1-j] for i, j < K; identity on the padding (chol(blockdiag(Hr, I)) = blockdiag(L, I))
__global__ __launch_bounds__(256) void ifac_flip_in_kernel(const float* __restrict__ H, int64_t K, float* __restrict__ A, int64_t Kp) {
  const int64_t i = blockIdx.y;
  for (int64_t j = blockIdx.x * 256 + threadIdx.x; j < Kp; j += (int64_t)gridDim.x * 256) {
    float v = i == j ? 1.f : 0.f;
    if (i < K && j < K) v = H[(K - 1 - i) * K + (K - 1 - j)];
    A[i * Kp + j] = v;
  }
}
// U[i, j] = X[K-1-i, K-1-j] for j >= i, zero below the diagonal
__global__ __launch_bounds__(256) void ifac_flip_out_kernel(const float* __restrict__ X, int64_t Kp, float* __restrict__ U, int64_t K) {
  const int64_t i = blockIdx.y;
  for (int64_t j = blockIdx.x * 256 + threadIdx.x; j < K; j += (int64_t)gridDim.x * 256)
    U[i * K + j] = j >= i ? X[(K - 1 - i) * Kp + (K - 1 - j)] : 0.f;
}
// dst[r, 0:cols] = src[r, 0:cols] (16-byte pieces; cols % 4 == 0)
__global__ __launch_bounds__(256) void ifac_copy_panel_kernel(const float* __restrict__ src, int64_t lds_, float* __restrict__ dst, int64_t ldd, int64_t rows, int cols) {
  const int c4 = cols >> 2;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < rows * c4; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / c4;
    const int c = (int)(i - r * c4) * 4;
    *reinterpret_cast<float4*>(dst + r * ldd + c) = *reinterpret_cast<const float4*>(src + r * lds_ + c);
  }
}

struct Seg {
  int64_t start, size;
};

}  // namespace

extern "C" {

// the split operands of the largest product: every product of the factorisation has k (rows_A + rows_B) <= Kp^2 elements (a merge of
// spans n1 + n2 <= Kp multiplies [n2, n1] by [n1, n1]; the trailing syrk is one operand [<= Kp, 1024]) -> 6 bytes per element, rows
// padded to 128 and k to 16
static int64_t ifac_plane_bytes(int64_t Kp) { return 6 * (Kp + 256) * (Kp + 128); }

// A [Kp, Kp] + X [Kp, Kp] + T [Kp, Kp] + 2 x P [Kp, IFAC_OUTER] fp32, Kp = K rounded up to 128; with flags bit 1 the three bf16 planes of
// the largest product's operands behind them (ifac_plane_bytes: 6 Kp^2; 12 Kp^2 until round 5)
int64_t inc_gptq_inverse_factor_workspace_bytes(int64_t K, int flags) {
  if (K <= 0) return 0;
  const int64_t Kp = ceil_div64(K, IFAC_NB) * IFAC_NB;
  return (3 * Kp * Kp + 2 * Kp * (int64_t)IFAC_OUTER) * (int64_t)sizeof(float) + ((flags & 2) ? ifac_plane_bytes(Kp) : 0);
}

int inc_gptq_inverse_factor(const float* H, int64_t K, float* U, void* workspace, int64_t workspace_bytes, int32_t* info, int flags,
                            inc_stream_t stream, inc_stream_t aux_stream) {
  INC_CHECK_ARG(H && U && workspace && info && K > 0 && K < (1ll << 30));
  if (workspace_bytes < inc_gptq_inverse_factor_workspace_bytes(K, flags)) return INC_ERR_WORKSPACE;
  if ((reinterpret_cast<uintptr_t>(workspace) & 15) != 0) return INC_ERR_BAD_ARG;
  hipStream_t s = inc_s(stream);
  const int64_t Kp = ceil_div64(K, IFAC_NB) * IFAC_NB, ld = Kp;
  float* A = static_cast<float*>(workspace);
  float* X = A + Kp * Kp;
  float* T = X + Kp * Kp;  // products C X11 of the doubling levels: the pair (s1, n1, s2, n2) keeps its n2 x n1 product in rows s2.. of T
  float* Pb[2] = {T + Kp * Kp, T + Kp * Kp + Kp * (int64_t)IFAC_OUTER};  // panel L[i > block, block] of outer block b in Pb[b & 1]
  void* const planes = (flags & 2) ? static_cast<void*>(Pb[1] + Kp * (int64_t)IFAC_OUTER) : nullptr;
  // Look-ahead (aux_stream given, flags bit 0 clear, more than two outer blocks): the main stream runs the CHAIN -- per outer block the
  // eight diagonal kernels with their small products, the block's inverse, the panel solve and the update of the NEXT block's columns
  // -- and the second stream everything the chain does not wait for: the rest of every trailing update and the doubling products of
  // the top level as soon as their operands are final.  Every memory location receives its updates in the same order either way
  // (rest(b - 1) is awaited before block b's update touches the same columns): identical results with and without the second stream.
  hipStream_t side = (aux_stream && !(flags & 1) && Kp > 2 * IFAC_OUTER) ? inc_s(aux_stream) : s;
  const bool two = side != s;
  hipEvent_t ev_main = nullptr, ev_side = nullptr;
  if (two && (hipEventCreateWithFlags(&ev_main, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ev_side, hipEventDisableTiming) != hipSuccess)) {
    if (ev_main) (void)hipEventDestroy(ev_main);
    return INC_ERR_LAUNCH;
  }
  if (hipMemsetAsync(info, 0, sizeof(int32_t), s) != hipSuccess) {
    if (ev_main) (void)hipEventDestroy(ev_main);
    if (ev_side) (void)hipEventDestroy(ev_side);
    return INC_ERR_LAUNCH;
  }
  // (X needs no initialisation: every product reads only blocks on or below the block diagonal, all written before they are read)
  {
    dim3 grid((unsigned)std::min<int64_t>(ceil_div64(Kp, 256), 64), (unsigned)Kp);
    ifac_flip_in_kernel<<<grid, 256, 0, s>>>(H, K, A, Kp);
  }
  int rc = INC_OK;
  auto gemm = [&](hipStream_t st, const float* a, int64_t lda, const float* b, int64_t ldb, float* c, int64_t ldc, int64_t M, int64_t N, int64_t Kd,
                  float alpha, float beta, int krange, bool lower_only, bool b_nt, int batch = 1, int64_t sa = 0, int64_t sb = 0, int64_t sc = 0) {
    GemmArgs g{a, b, c, lda, ldb, ldc, sa, sb, sc, (int)M, (int)N, (int)Kd, alpha, beta, krange, lower_only ? 1 : 0};
    const int r = launch_f32gemm(g, b_nt, batch, st, st == s ? planes : nullptr, planes ? ifac_plane_bytes(Kp) : 0);  // one plane buffer: the main stream's products only
    if (r != INC_OK) rc = r;
  };
  auto copy_panel = [&](hipStream_t st, const float* src, int64_t lds_, float* dst, int64_t ldd, int64_t rows, int cols) {
    const int64_t n4 = rows * (cols / 4);
    ifac_copy_panel_kernel<<<(unsigned)std::min<int64_t>(ceil_div64(n4, 256), 2048), 256, 0, st>>>(src, lds_, dst, ldd, rows, cols);
  };
  // one doubling step: inv([[A, 0], [C, B]]) = [[A^-1, 0], [-B^-1 C A^-1, B^-1]] for `batch` pairs (a_k, b_k) at a constant distance;
  // both factors of X21 = -X22 (C X11) are lower-triangular inverses (K-ranges skip their zero halves)
  auto merge_pairs = [&](hipStream_t st, Seg a, Seg b, int batch) {
    const int64_t n1 = a.size, n2 = b.size, step = (a.size + b.size) * (ld + 1), ldt = Kp, tstep = (a.size + b.size) * ldt;
    float* t0 = T + b.start * ldt;
    gemm(st, A + b.start * ld + a.start, ld, X + a.start * ld + a.start, ld, t0, ldt, n2, n1, n1, 1.f, 0.f, KR_B_LOWER_NN, false, false, batch, step, step, tstep);
    gemm(st, X + b.start * ld + b.start, ld, t0, ldt, X + b.start * ld + a.start, ld, n2, n1, n2, -1.f, 0.f, KR_A_LOWER, false, false, batch, step, tstep, step);
  };
  // X of the span covered by `segs` (whose diagonal blocks of X already hold the inverses), level by level; pairs of equal geometry
  // at a constant distance go out as ONE batched launch per product
  auto invert_by_doubling = [&](hipStream_t st, std::vector<Seg> segs) {
    while (segs.size() > 1) {
      std::vector<Seg> nxt;
      size_t p = 0;
      const size_t npairs = segs.size() / 2;
      while (p < npairs) {
        const Seg a = segs[2 * p], b = segs[2 * p + 1];
        size_t q = p + 1;
        while (q < npairs && segs[2 * q].size == a.size && segs[2 * q + 1].size == b.size &&
               segs[2 * q].start - segs[2 * (q - 1)].start == a.size + b.size)
          ++q;
        merge_pairs(st, a, b, (int)(q - p));
        for (size_t r = p; r < q; ++r) nxt.push_back(Seg{segs[2 * r].start, segs[2 * r].size + segs[2 * r + 1].size});
        p = q;
      }
      if (segs.size() & 1) nxt.push_back(segs.back());
      segs.swap(nxt);
    }
    return segs[0];
  };

  struct Top {
    Seg seg;
    int level;
  };
  std::vector<Top> stack;  // inverted spans of the top level, merged like a binary counter as the outer blocks complete
  int tag = 0, nblk = 0;
  bool rest_pending = false;
  for (int64_t Bo = 0; Bo < Kp; Bo += IFAC_OUTER, ++nblk) {
    const int64_t n2 = std::min<int64_t>(IFAC_OUTER, Kp - Bo);
    float* D = A + Bo * ld + Bo;
    float* XD = X + Bo * ld + Bo;
    std::vector<Seg> inner;
    for (int64_t j = 0; j < n2; j += IFAC_NB) {
      ++tag;
      if (inc_launch_chol_diag_block(D + j * ld + j, ld, IFAC_NB, XD + j * ld + j, ld, info, tag, s) != INC_OK) rc = INC_ERR_LAUNCH;
      inner.push_back(Seg{Bo + j, IFAC_NB});
      if (j + IFAC_NB < n2) {
        const int64_t m = n2 - (j + IFAC_NB);
        float* panel = D + (j + IFAC_NB) * ld + j;  // [m, 128]
        // L_panel = A_panel inv(L_jj)^T  (inv(L_jj) lower), in place: one 64 x 128 tile per row block
        gemm(s, panel, ld, XD + j * ld + j, ld, panel, ld, m, IFAC_NB, IFAC_NB, 1.f, 0.f, KR_FULL, false, true);
        // trailing update inside the outer block, lower triangle only
        gemm(s, panel, ld, panel, ld, D + (j + IFAC_NB) * (ld + 1), ld, m, m, IFAC_NB, -1.f, 1.f, KR_FULL, true, true);
      }
    }
    const Seg whole = invert_by_doubling(s, inner);
    float* P = Pb[nblk & 1];
    const int64_t Mr = Kp - (Bo + n2);
    if (Mr > 0) {
      float* panel = A + (Bo + n2) * ld + Bo;  // [Mr, n2]
      gemm(s, panel, ld, XD, ld, P, IFAC_OUTER, Mr, n2, n2, 1.f, 0.f, KR_B_LOWER_NT, false, true);  // L_panel = A_panel inv(L_DD)^T
      copy_panel(s, P, IFAC_OUTER, panel, ld, Mr, (int)n2);
    }
    if (two) (void)hipEventRecord(ev_main, s);  // the panel (P, written back) and this block's inverse are complete
    // chain: the NEXT outer block's columns of the trailing update (everything, without a second stream)
    float* Cn = A + (Bo + n2) * (ld + 1);  // trailing matrix [Mr, Mr]
    const int64_t c1 = two ? std::min<int64_t>(IFAC_OUTER, Mr) : Mr;
    if (Mr > 0) {
      if (rest_pending) {
        (void)hipStreamWaitEvent(s, ev_side, 0);  // rest(b - 1) accumulated into the same columns
        rest_pending = false;
      }
      gemm(s, P, IFAC_OUTER, P, IFAC_OUTER, Cn, ld, c1, c1, n2, -1.f, 1.f, KR_FULL, true, true);
      if (Mr > c1) gemm(s, P + c1 * IFAC_OUTER, IFAC_OUTER, P, IFAC_OUTER, Cn + c1 * ld, ld, Mr - c1, c1, n2, -1.f, 1.f, KR_FULL, false, true);
    }
    // second stream: the rest of the trailing update first (the chain waits for it one block later), then the top level: this block's
    // inverse joins the stack and equal-level neighbours merge (their C operand -- L of the rows below the left span -- became
    // final with the panel solves above; nothing on the chain reads these products)
    if (two) (void)hipStreamWaitEvent(side, ev_main, 0);
    if (Mr > c1) {
      gemm(side, P + c1 * IFAC_OUTER, IFAC_OUTER, P + c1 * IFAC_OUTER, IFAC_OUTER, Cn + c1 * (ld + 1), ld, Mr - c1, Mr - c1, n2, -1.f, 1.f, KR_FULL, true, true);
      (void)hipEventRecord(ev_side, side);
      rest_pending = true;
    }
    stack.push_back(Top{whole, 0});
    while (stack.size() > 1 && stack[stack.size() - 1].level == stack[stack.size() - 2].level) {
      const Top b = stack.back();
      stack.pop_back();
      const Top a = stack.back();
      stack.pop_back();
      merge_pairs(side, a.seg, b.seg, 1);
      stack.push_back(Top{Seg{a.seg.start, a.seg.size + b.seg.size}, a.level + 1});
    }
  }
  // what is left on the stack merges right to left (the last spans are the shortest)
  while (stack.size() > 1) {
    const Top b = stack.back();
    stack.pop_back();
    const Top a = stack.back();
    stack.pop_back();
    merge_pairs(side, a.seg, b.seg, 1);
    stack.push_back(Top{Seg{a.seg.start, a.seg.size + b.seg.size}, a.level + 1});
  }
  if (two) {
    (void)hipEventRecord(ev_side, side);
    (void)hipStreamWaitEvent(s, ev_side, 0);
  }
  {
    dim3 grid((unsigned)std::min<int64_t>(ceil_div64(K, 256), 64), (unsigned)K);
    ifac_flip_out_kernel<<<grid, 256, 0, s>>>(X, Kp, U, K);
  }
  if (ev_main) (void)hipEventDestroy(ev_main);
  if (ev_side) (void)hipEventDestroy(ev_side);
  if (rc != INC_OK) return rc;
  INC_LAUNCH_RETURN();
}

}  // extern "C"
