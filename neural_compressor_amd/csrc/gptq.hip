// gptq.hip -- K5/K6: GPTQ Hessian accumulation (MFMA syrk) and the block-column error-compensation loop.
//
// Reference (relative to /root/reference/neural_compressor/torch/algorithms/weight_only/gptq.py):
//   GPTQ.add_batch      :1111-1141   H <- H*n/(n+b) + (sqrt(2/(n+b)) X)^T (sqrt(2/(n+b)) X)
//   GPTQ.fasterquant    :1143-1351   dead columns, damping, the blocked column loop, lazy update
//   Quantizer.quantize  :1626-1637   q = clamp(round(x/scale)+zero, 0, maxq); scale*(q-zero)
//
// Kernels
//   hessian_syrk_16bit  bf16/f16 MFMA 32x32x16, fp32 accumulate, 128x128 tile of H per workgroup, upper
//                       triangle of tiles only.  X is [T,K] row-major (tokens x features) so both MFMA
//                       operands are X^T: each thread fetches an 8(token) x 8(feature) block with eight
//                       16-byte row loads (full 128-byte segments per row), transposes it in registers
//                       and writes eight 16-byte [feature][token] rows into LDS (pitch 144 B: both the
//                       ds_write_b128 and the fragment ds_read_b128 are bank-conflict free).
//   hessian_syrk_f32    exact fp32 MFMA 32x32x2 (A/B = one f32 per lane: no transpose needed).
//   gptq_quant_block    the serial 128-column chain.  One lane owns one weight row; the 128-wide row
//                       panel lives in 128 VGPRs (fully unrolled), the 128x128 Hinv tile is broadcast
//                       out of LDS.  Arithmetic is kept un-fused (mul then sub, true divisions) so that
//                       one block is bit-identical to the reference's torch ops.
//   gptq_lazy_update    W[:, i2:] -= Err1 @ Hinv[i1:i2, i2:] with the exact fp32 MFMA (first generation here; the product kernels -- tile and
//                       strip form -- live in gptq_lazy.hip, the Hessian syrk in gptq_hessian.hip).
#include <math.h>

#include <type_traits>

#include <algorithm>

#include "common.hpp"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

// W (any dtype) -> fp32 working copy with dead columns zeroed (gptq.py:1176, 1189)
template <int DT>
__global__ void gptq_prepare_weight_kernel(const void* __restrict__ w, float* __restrict__ out,
                                           const uint8_t* __restrict__ dead, int64_t N, int64_t K) {
  const int64_t total = N * K;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = i % K;
    out[i] = (dead && dead[k]) ? 0.f : load_as_f32<DT>(w, i);
  }
}
// K % 8 == 0 and 16-byte aligned rows: eight consecutive k of one row per thread -- one 16-byte load of a 16-bit weight (two for
// fp32), the eight `dead` flags as one 8-byte load, two 16-byte stores; the row / column split once per thread instead of a 64-bit
// modulo per element
template <int DT>
__global__ __launch_bounds__(256) void gptq_prepare_weight_vec_kernel(const void* __restrict__ w, float* __restrict__ out,
                                                                      const uint8_t* __restrict__ dead, int64_t N, int64_t K) {
  const int64_t k8 = K >> 3, total = N * k8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / k8, c = (i - n * k8) << 3, base = n * K + c;
    float v[8];
    if constexpr (DT == INC_F32) {
      const float4 a = reinterpret_cast<const float4*>(static_cast<const float*>(w) + base)[0];
      const float4 b = reinterpret_cast<const float4*>(static_cast<const float*>(w) + base)[1];
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
      const uint4 q = *reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(w) + base);
      const uint32_t u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if constexpr (DT == INC_BF16) {
          v[2 * e] = bf16_bits_to_f32((uint16_t)(u[e] & 0xffffu));
          v[2 * e + 1] = bf16_bits_to_f32((uint16_t)(u[e] >> 16));
        } else {
          v[2 * e] = f16_bits_to_f32((uint16_t)(u[e] & 0xffffu));
          v[2 * e + 1] = f16_bits_to_f32((uint16_t)(u[e] >> 16));
        }
      }
    }
    if (dead) {
      const uint2 d = *reinterpret_cast<const uint2*>(dead + c);
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (((e < 4 ? d.x : d.y) >> (8 * (e & 3))) & 0xffu) v[e] = 0.f;
    }
    reinterpret_cast<float4*>(out + base)[0] = make_float4(v[0], v[1], v[2], v[3]);
    reinterpret_cast<float4*>(out + base)[1] = make_float4(v[4], v[5], v[6], v[7]);
  }
}

// ---------------------------------------------------------------------------------------------
// the serial column chain for one 128-column block
// ---------------------------------------------------------------------------------------------
constexpr int QB = 128;         // columns per block
constexpr int QROWS = 64;       // rows per workgroup (one wave)
constexpr int QPITCH = QB + 1;  // fp32 staging pitch

#pragma clang fp contract(off)
template <int QDT>
__global__ __launch_bounds__(64) void gptq_quant_block_kernel(
    const float* __restrict__ w, const float* __restrict__ Hinv, const float* __restrict__ scale,
    const float* __restrict__ zero, uint8_t* __restrict__ codes, void* __restrict__ q_out,
    float* __restrict__ err, int64_t N, int64_t K, int64_t G, int64_t i1, int count, int gs, float maxq) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* hs = reinterpret_cast<float*>(smem_raw);     // [QB][QB] Hinv1 tile (upper part used)
  float* stage = hs + QB * QB;                        // [QROWS][QPITCH]
  uint8_t* cst = reinterpret_cast<uint8_t*>(stage + QROWS * QPITCH);  // [QROWS][QB+4] codes
  constexpr int CP = QB + 4;
  const int lane = threadIdx.x;
  const int64_t n0 = (int64_t)blockIdx.x * QROWS;
  const int64_t row = n0 + lane;

  // Hinv1 -> LDS (zero-padded), W panel -> LDS (coalesced) -> registers
  for (int idx = lane; idx < QB * QB; idx += 64) {
    const int r = idx >> 7, c = idx & 127;
    hs[idx] = (r < count && c < count) ? Hinv[(i1 + r) * K + i1 + c] : (r == c ? 1.f : 0.f);
  }
  for (int idx = lane; idx < QROWS * QB; idx += 64) {
    const int r = idx >> 7, c = idx & 127;
    stage[r * QPITCH + c] = (n0 + r < N && c < count) ? w[(n0 + r) * K + i1 + c] : 0.f;
  }
  __syncthreads();
  float wr[QB];
#pragma unroll
  for (int j = 0; j < QB; ++j) wr[j] = stage[lane * QPITCH + j];
  __syncthreads();

  const int64_t srow = (row < N ? row : N - 1) * G;
  // group bookkeeping without per-step divisions: `nb` = next in-block index that starts a group
  int g = gs > 0 ? (int)(i1 / gs) : 0;
  int nb = gs > 0 ? (int)(((int64_t)(g + 1)) * gs - i1) : (1 << 30);
  float s = scale[srow + g], z = zero[srow + g];
#pragma unroll
  for (int i = 0; i < QB; ++i) {
    if (i < count) {
      if (i == nb) {
        ++g;
        nb += gs;
        s = scale[srow + g];
        z = zero[srow + g];
      }
      const float d = hs[i * QB + i];
      const float x = wr[i];
      float t = rintf(x / s) + z;
      t = fminf(fmaxf(t, 0.f), maxq);
      const float q = s * (t - z);
      const float e = (x - q) / d;
      cst[lane * CP + i] = (uint8_t)t;
      stage[lane * QPITCH + i] = e;  // Err1 (the panel has been copied to registers)
      wr[i] = q;
#pragma unroll
      for (int j = i + 1; j < QB; ++j) {
        const float p = e * hs[i * QB + j];
        wr[j] = wr[j] - p;
      }
    } else {
      stage[lane * QPITCH + i] = 0.f;
      cst[lane * CP + i] = 0;
    }
  }
  __syncthreads();
  // Err1 out: [N][128] dense
  for (int idx = lane; idx < QROWS * QB; idx += 64) {
    const int r = idx >> 7, c = idx & 127;
    if (n0 + r < N) err[(n0 + r) * QB + c] = stage[r * QPITCH + c];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < QB; ++j) stage[lane * QPITCH + j] = wr[j];
  __syncthreads();
  for (int idx = lane; idx < QROWS * QB; idx += 64) {
    const int r = idx >> 7, c = idx & 127;
    if (n0 + r < N && c < count) {
      const int64_t o = (n0 + r) * K + i1 + c;
      if (q_out) store_from_f32<QDT>(q_out, o, stage[r * QPITCH + c]);
      if (codes) codes[o] = cst[r * CP + c];
    }
  }
}
#pragma clang fp contract(fast)

// ---------------------------------------------------------------------------------------------
// the serial column chain, second generation: FOUR lanes per weight row
// ---------------------------------------------------------------------------------------------
// The chain of 128 dependent steps cannot be shortened, so the kernel maximises how many chains run at
// once and minimises the work per step: a wave owns 16 rows, the 4 lanes of a quad share one row with the
// block's columns interleaved (lane q owns columns 4c+q).  At step i the owner lane (q = i&3) broadcasts
// the column value to its quad with one DPP quad_perm move; all 4 lanes then compute q/err redundantly
// (identical inputs -> identical bits, no second exchange) and each applies the rank-1 update to its own
// still-live columns (4c+q > i; column groups entirely behind i are skipped at compile time, which halves
// the average work).  N = 4096 rows -> 256 single-wave workgroups (one per CU) instead of 64, and a step
// costs ~16 updates instead of 128.  The arithmetic per element is unchanged (true divisions, mul-then-sub,
// contraction off), so results are bit-identical to the one-lane-per-row kernel and to the reference's ops.
constexpr int Q4R = 16;  // rows per wave
constexpr int Q4W = 4;   // waves per workgroup
// compile-time loop: f(integral_constant<int, I>) for I in [BEGIN, END) -- every register index in the step
// body is then a literal (a `#pragma unroll` loop this size is only partly unrolled and falls back to
// s_set_gpr_idx register indexing)
template <int I, int END, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < END) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, END>(f);
  }
}
// value of lane `src` (0..3) of each quad, in every lane of the quad: one v_mov_b32 with a DPP quad_perm
__device__ __forceinline__ float quad_bcast(float v, int src) {
  const int x = __float_as_int(v);
  int y;
  switch (src) {  // the control word must be a literal; `src` is a constant after unrolling
    case 0: y = __builtin_amdgcn_mov_dpp(x, 0x00, 0xf, 0xf, true); break;
    case 1: y = __builtin_amdgcn_mov_dpp(x, 0x55, 0xf, 0xf, true); break;
    case 2: y = __builtin_amdgcn_mov_dpp(x, 0xAA, 0xf, 0xf, true); break;
    default: y = __builtin_amdgcn_mov_dpp(x, 0xFF, 0xf, 0xf, true); break;
  }
  return __int_as_float(y);
}
#pragma clang fp contract(off)
// GPB = scale groups per 128-column block (1, 2 or 4: group_size >= 128 / 64 / 32 with i1 % 128 == 0), known
// at compile time so that the 128 steps contain no branch at all; count == 128 on this path.
// PARAMS: the kernel first computes the (scale, zero) of the block's own groups from the weights it has just loaded --
// Quantizer.find_params (gptq.py:1501-1571, perchannel, weight=True, no mse search) on W "as it is now" (gptq.py:1266-1272) --
// and writes them to scale / zero [N, G]: one launch less per 128 columns of the serial chain.  sym_flag as in find_params.
// WPW waves per workgroup share ONE Hinv1 tile in LDS (64 KiB): with a wave per workgroup the tile was fetched by every wave and
// LDS alone limited a CU to two of them -- 768 / 1376 single-wave workgroups for the stacked q/k/v and gate/up solves ran in
// 1.5 / 2.7 rounds, and the 256 of a 4096-row layer sat on every CU of the chip, where their 64 KiB + 304 registers kept the
// look-ahead loop's trailing update (two 64 KiB workgroups per CU) off the CU.  Four waves per workgroup: 64 / 192 / 344
// workgroups, one round, and the other CUs are free for the trailing update that runs underneath the chain.
template <int QDT, int GPB, bool PARAMS = false, int WPW = 4>
__global__ __launch_bounds__(64 * WPW) void gptq_quant_block_q4_kernel(
    const float* __restrict__ w, const float* __restrict__ Hinv, float* __restrict__ scale,
    float* __restrict__ zero, uint8_t* __restrict__ codes, void* __restrict__ q_out,
    float* __restrict__ err, int64_t N, int64_t K, int64_t G, int64_t i1, int64_t g0, float maxq, int sym_flag) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* hs = reinterpret_cast<float*>(smem_raw);  // [QB][QB] Hinv1 tile, later reused as the waves' output stages
  // The chain is the serial path of the column loop and, at 121 registers, shares its SIMDs with the trailing update's MFMA waves
  // (248 registers): it goes first whenever it has an instruction ready, the update fills the rest of the issue slots.
  __builtin_amdgcn_s_setprio(3);
  const int lane = threadIdx.x & 63, r = lane >> 2, q = lane & 3;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t n0 = ((int64_t)blockIdx.x * WPW + wave) * Q4R;
  const int64_t row = n0 + r, rowc = row < N ? row : N - 1;

  // Hinv1 tile -> LDS by LDS-DMA: 64 instructions of 1 KiB (two 512-byte rows each) shared out over the waves, all in flight at once
  {
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem_raw;
    const float* hb = Hinv + i1 * K + i1;
    const uint32_t v0 = (uint32_t)(((lane >> 5) * K + 4 * (lane & 31)) * 4);
    const uint32_t step = (uint32_t)(2 * K * 4);
#pragma unroll
    for (int jj = 0; jj < 16 / WPW; ++jj) {
      const int j = jj * WPW + wave;
      lds_dma_4x1k(hb, lds0 + j * 4096, v0 + (4 * j) * step, v0 + (4 * j + 1) * step, v0 + (4 * j + 2) * step, v0 + (4 * j + 3) * step);
    }
  }
  // this lane's 32 columns of its row: block columns 4c + q
  float wr[32], ev[32];
  uint32_t cw[8];
#pragma unroll
  for (int c = 0; c < 32; ++c) {
    wr[c] = w[rowc * K + i1 + 4 * c + q];
    ev[c] = 0.f;
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) cw[c] = 0u;
  float sg[GPB], zg[GPB];
  if constexpr (!PARAMS) {
#pragma unroll
    for (int g = 0; g < GPB; ++g) {
      sg[g] = scale[rowc * G + g0 + g];
      zg[g] = zero[rowc * G + g0 + g];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (PARAMS) {
    // group g of the block = block columns [g * 128 / GPB, (g + 1) * 128 / GPB): this lane holds 4c + q for c in [g * 32 / GPB, ...)
#pragma unroll
    for (int g = 0; g < GPB; ++g) {
      float vmax = -INFINITY, vmin = INFINITY;
#pragma unroll
      for (int c = g * (32 / GPB); c < (g + 1) * (32 / GPB); ++c) {
        vmax = fmaxf(vmax, wr[c]);
        vmin = fminf(vmin, wr[c]);
      }
      vmax = fmaxf(vmax, __shfl_xor(vmax, 1, 64));
      vmin = fminf(vmin, __shfl_xor(vmin, 1, 64));
      vmax = fmaxf(vmax, __shfl_xor(vmax, 2, 64));
      vmin = fminf(vmin, __shfl_xor(vmin, 2, 64));
      // gptq.py:1547-1571 (the arithmetic of gptq_find_params_kernel, quant.hip)
      float xmin = fminf(vmin, 0.f), xmax = fmaxf(vmax, 0.f);
      if (sym_flag) {
        xmax = fmaxf(fabsf(xmin), xmax);
        if (xmin < 0.f) xmin = -xmax;
      }
      if (xmin == 0.f && xmax == 0.f) { xmin = -1.f; xmax = 1.f; }
      const float sv = (xmax - xmin) / maxq;
      const float zv = sym_flag ? (maxq + 1.f) * 0.5f : rintf(-xmin / sv);
      sg[g] = sv;
      zg[g] = zv;
      if (q == 0 && row < N) {
        scale[row * G + g0 + g] = sv;
        zero[row * G + g0 + g] = zv;
      }
    }
  }
  __syncthreads();

  // The row of Hinv1 needed by step i+1 is read from LDS while step i computes: two register sets used
  // alternately (even / odd steps) so that no copies are needed.
  float ha[33], hb2[33];  // [c] = hs[i][4c+q] for the live c, [32] = hs[i][i]
  auto fetch_row = [&](auto ic, float (&h)[33]) {
    constexpr int i = decltype(ic)::value;
    if constexpr (i < QB) {
#pragma unroll
      for (int c = i >> 2; c < 32; ++c) h[c] = hs[i * QB + 4 * c + q];
      h[32] = hs[i * QB + i];
    }
  };
  auto step = [&](auto ic, float (&h)[33]) {
    constexpr int i = decltype(ic)::value;
    constexpr int qo = i & 3, ci = i >> 2, g = i / (QB / GPB);
    const float s = sg[g], z = zg[g];
    const float x = quad_bcast(wr[ci], qo);
    float t = rintf(x / s) + z;
    t = fminf(fmaxf(t, 0.f), maxq);
    const float qv = s * (t - z);
    const float e = (x - qv) / h[32];
    const bool own = q == qo;
#pragma unroll
    for (int c = ci; c < 32; ++c) {
      const float pr = e * h[c];  // exact zero below the diagonal (Hinv is upper triangular)
      wr[c] = wr[c] - pr;
    }
    wr[ci] = own ? qv : wr[ci];
    ev[ci] = own ? e : ev[ci];
    cw[ci >> 2] = own ? (cw[ci >> 2] | ((uint32_t)t << (8 * (ci & 3)))) : cw[ci >> 2];
    // Pinned HERE: left alone, the compiler sinks these three selects to the end of the chain -- nothing reads them before the output
    // stage -- and keeps t, e and qv of all 128 steps alive: 306 registers (50 of them AGPRs used as spill space) instead of 121,
    // i.e. one wave per SIMD, and no lazy-update workgroup (248 registers) beside a chain workgroup on the same CU.
    asm volatile("" : "+v"(wr[ci]), "+v"(ev[ci]), "+v"(cw[ci >> 2]));
  };
  fetch_row(std::integral_constant<int, 0>{}, ha);
  static_for<0, QB / 2>([&](auto pc) {
    constexpr int p = decltype(pc)::value;
    fetch_row(std::integral_constant<int, 2 * p + 1>{}, hb2);
    step(std::integral_constant<int, 2 * p>{}, ha);
    fetch_row(std::integral_constant<int, 2 * p + 2>{}, ha);
    step(std::integral_constant<int, 2 * p + 1>{}, hb2);
  });
  __syncthreads();
  // stage the three outputs through LDS (rows of 128) for coalesced stores: Err1, Q, codes -- every wave in its own 10 KiB
  float* st = reinterpret_cast<float*>(smem_raw + wave * (Q4R * QB * 5));  // [Q4R][QB] floats, then [Q4R][QB] bytes
#pragma unroll
  for (int c = 0; c < 32; ++c) st[r * QB + 4 * c + q] = ev[c];
  __syncthreads();
  for (int idx = lane; idx < Q4R * QB / 4; idx += 64) {
    const int rr = idx >> 5, c4 = (idx & 31) * 4;
    if (n0 + rr < N) *reinterpret_cast<float4*>(err + (n0 + rr) * QB + c4) = *reinterpret_cast<const float4*>(st + rr * QB + c4);
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 32; ++c) st[r * QB + 4 * c + q] = wr[c];
  uint8_t* cst = reinterpret_cast<uint8_t*>(st + Q4R * QB);  // [Q4R][QB] bytes
#pragma unroll
  for (int c = 0; c < 32; ++c) cst[r * QB + 4 * c + q] = (uint8_t)((cw[c >> 2] >> (8 * (c & 3))) & 0xffu);
  __syncthreads();
  for (int idx = lane; idx < Q4R * QB; idx += 64) {
    const int rr = idx >> 7, c = idx & 127;
    if (n0 + rr < N) {
      const int64_t o = (n0 + rr) * K + i1 + c;
      if (q_out) store_from_f32<QDT>(q_out, o, st[rr * QB + c]);
      if (codes) codes[o] = cst[rr * QB + c];
    }
  }
}
#pragma clang fp contract(fast)

// ---------------------------------------------------------------------------------------------
// lazy update: W[:, i2:] -= Err1[N,128] @ Hinv[i1:i1+128, i2:]
// ---------------------------------------------------------------------------------------------
constexpr int LT = 128;
constexpr int LAP = LT + 1;
__global__ __launch_bounds__(256) void gptq_lazy_update_kernel(float* __restrict__ w,
                                                               const float* __restrict__ Hinv,
                                                               const float* __restrict__ err, int64_t N,
                                                               int64_t K, int64_t i1, int count) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* As = reinterpret_cast<float*>(smem_raw);  // [LT rows][LAP]  (k along the row)
  float* Bs = As + LT * LAP;                        // [QB k][LT cols]
  const int64_t i2 = i1 + count;
  const int64_t c0 = i2 + (int64_t)blockIdx.x * LT, r0 = (int64_t)blockIdx.y * LT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  for (int idx = tid; idx < LT * QB; idx += 256) {
    const int r = idx >> 7, k = idx & 127;
    As[r * LAP + k] = (r0 + r < N && k < count) ? err[(r0 + r) * QB + k] : 0.f;
  }
  for (int idx = tid; idx < QB * LT; idx += 256) {
    const int k = idx >> 7, c = idx & 127;
    Bs[idx] = (k < count && c0 + c < K) ? Hinv[(i1 + k) * K + c0 + c] : 0.f;
  }
  __syncthreads();
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
#pragma unroll 4
  for (int s = 0; s < QB / 2; ++s) {
    const int k = 2 * s + (lane >> 5);
    float a[2], b[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      a[m] = As[(wr * 64 + m * 32 + (lane & 31)) * LAP + k];
      b[m] = Bs[k * LT + wc * 64 + m * 32 + (lane & 31)];
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[n], acc[m][n], 0, 0, 0);
  }
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int64_t col = c0 + wc * 64 + n * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = r0 + wr * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < N && col < K) {
          float* p = w + row * K + col;
          *p = *p - acc[m][n][r];
        }
      }
    }
}

constexpr int L2T = 128;  // rows / columns of a lazy-update tile

#ifdef INC_KBENCH  // superseded by the third generation below; harness flag 86, its bitwise A/B partner (tools/kbench colloop / qlayer)
#include "../../tools/kbench_gptq_2.inc"
#include "../../tools/kbench_gptq_3.inc"  // harness flag 106: the lazy update with split (bf16 x 3) products -- an experiment, not the product
#endif  // INC_KBENCH

// third / fourth generation of the lazy update: gptq_lazy.hip
}  // namespace
void inc_launch_lazy_update_v3(float* w, const float* Hinv, const float* err, int64_t N, int64_t K, int64_t i1, int64_t c_begin,
                               int64_t c_end, bool exclusive, hipStream_t s);
namespace {

}  // namespace

extern "C" {

int inc_gptq_prepare_weight(const void* w, int wdtype, float* out, const uint8_t* dead, int64_t N,
                            int64_t K, inc_stream_t stream) {
  INC_CHECK_ARG(w && out && N > 0 && K > 0);
  const bool vec = (K % 8) == 0 && ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(out)) & 15) == 0 &&
                   (!dead || (reinterpret_cast<uintptr_t>(dead) & 7) == 0);
  int64_t blocks = ceil_div64(vec ? N * (K / 8) : N * K, 256);
  if (blocks > 8192) blocks = 8192;
  INC_DISPATCH_DTYPE(wdtype, DT, {
    if (vec) gptq_prepare_weight_vec_kernel<DT><<<(unsigned)blocks, 256, 0, inc_s(stream)>>>(w, out, dead, N, K);
    else gptq_prepare_weight_kernel<DT><<<(unsigned)blocks, 256, 0, inc_s(stream)>>>(w, out, dead, N, K);
  })
  INC_LAUNCH_RETURN();
}

int inc_gptq_quant_block(const float* w, const float* Hinv, const float* scale, const float* zero,
                         uint8_t* codes, void* q_out, int q_dtype, float* err, int64_t N, int64_t K,
                         int64_t G, int64_t i1, int count, int group_size, int bits,
                         inc_stream_t stream) {
  INC_CHECK_ARG(w && Hinv && scale && zero && err && N > 0 && K > 0 && G > 0);
  INC_CHECK_ARG(i1 >= 0 && count > 0 && count <= QB && i1 + count <= K && bits >= 1 && bits <= 8);
  const size_t smem = (size_t)QB * QB * 4 + (size_t)QROWS * QPITCH * 4 + (size_t)QROWS * (QB + 4);
  const unsigned blocks = (unsigned)ceil_div64(N, QROWS);
  const float maxq = (float)((1 << bits) - 1);
  hipStream_t s = inc_s(stream);
  static std::atomic<uint64_t> attr_set{0};
  if (inc_attr_needed(attr_set)) {
    (void)hipFuncSetAttribute((const void*)gptq_quant_block_kernel<INC_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute((const void*)gptq_quant_block_kernel<INC_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute((const void*)gptq_quant_block_kernel<INC_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    inc_attr_done(attr_set);
  }
  // second-generation kernel: full 128-column block starting on a 128-column boundary, 1 / 2 / 4 groups per block
  int gpb = 0;
  if (group_size <= 0 || (group_size % QB) == 0) gpb = 1;
  else if (group_size == 64) gpb = 2;
  else if (group_size == 32) gpb = 4;
  if (gpb && count == QB && (i1 % QB) == 0 && (K % 4) == 0 && K * (int64_t)(QB + 1) * 4 < ((int64_t)1 << 32) && !inc_force_small_tiles()) {
    const size_t smem4 = (size_t)QB * QB * 4;  // 64 KiB: Hinv1 tile, reused as the output stage
    static std::atomic<uint64_t> attr4_set{0};
#define INC_Q4_ATTR(DT, GP) (void)hipFuncSetAttribute((const void*)gptq_quant_block_q4_kernel<DT, GP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem4)
    if (inc_attr_needed(attr4_set)) {
      INC_Q4_ATTR(INC_F32, 1); INC_Q4_ATTR(INC_F32, 2); INC_Q4_ATTR(INC_F32, 4);
      INC_Q4_ATTR(INC_F16, 1); INC_Q4_ATTR(INC_F16, 2); INC_Q4_ATTR(INC_F16, 4);
      INC_Q4_ATTR(INC_BF16, 1); INC_Q4_ATTR(INC_BF16, 2); INC_Q4_ATTR(INC_BF16, 4);
      inc_attr_done(attr4_set);
    }
#undef INC_Q4_ATTR
    const unsigned blocks4 = (unsigned)ceil_div64(N, Q4R * Q4W);
    const int64_t g0 = group_size > 0 ? i1 / group_size : 0;
#define INC_Q4(GP) gptq_quant_block_q4_kernel<DT, GP><<<blocks4, 64 * Q4W, smem4, s>>>(w, Hinv, const_cast<float*>(scale), const_cast<float*>(zero), codes, q_out, err, N, K, G, i1, g0, maxq, 0)
    INC_DISPATCH_DTYPE(q_dtype, DT, {
      if (gpb == 1) INC_Q4(1); else if (gpb == 2) INC_Q4(2); else INC_Q4(4);
    })
#undef INC_Q4
    INC_LAUNCH_RETURN();
  }
  INC_DISPATCH_DTYPE(q_dtype, DT, {
    gptq_quant_block_kernel<DT><<<blocks, 64, smem, s>>>(w, Hinv, scale, zero, codes, q_out, err, N, K, G, i1, count, group_size, maxq);
  })
  INC_LAUNCH_RETURN();
}

// inc_gptq_quant_block for a full 128-column block whose groups lie inside it (group_size 32 / 64 / 128, i1 on a 128-column
// boundary), computing the groups' (scale, zero) itself -- see the PARAMS form of the kernel.  INC_ERR_UNSUPPORTED otherwise.
int inc_gptq_quant_block_params(const float* w, const float* Hinv, float* scale, float* zero, uint8_t* codes, void* q_out,
                                int q_dtype, float* err, int64_t N, int64_t K, int64_t G, int64_t i1, int count, int group_size,
                                int bits, int sym, inc_stream_t stream) {
  INC_CHECK_ARG(w && Hinv && scale && zero && err && N > 0 && K > 0 && G > 0);
  INC_CHECK_ARG(i1 >= 0 && count > 0 && count <= QB && i1 + count <= K && bits >= 1 && bits <= 8);
  int gpb = 0;
  if (group_size == QB) gpb = 1;
  else if (group_size == 64) gpb = 2;
  else if (group_size == 32) gpb = 4;
  if (!(gpb && count == QB && (i1 % QB) == 0 && (K % 4) == 0 && K * (int64_t)(QB + 1) * 4 < ((int64_t)1 << 32))) return INC_ERR_UNSUPPORTED;
  const int64_t g0 = i1 / group_size;
  INC_CHECK_ARG(g0 + gpb <= G);
  const float maxq = (float)((1 << bits) - 1);
  hipStream_t s = inc_s(stream);
  const size_t smem4 = (size_t)QB * QB * 4;
  static std::atomic<uint64_t> attr_set{0};
#define INC_Q4P_ATTR(DT, GP) (void)hipFuncSetAttribute((const void*)gptq_quant_block_q4_kernel<DT, GP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem4)
  if (inc_attr_needed(attr_set)) {
    INC_Q4P_ATTR(INC_F32, 1); INC_Q4P_ATTR(INC_F32, 2); INC_Q4P_ATTR(INC_F32, 4);
    INC_Q4P_ATTR(INC_F16, 1); INC_Q4P_ATTR(INC_F16, 2); INC_Q4P_ATTR(INC_F16, 4);
    INC_Q4P_ATTR(INC_BF16, 1); INC_Q4P_ATTR(INC_BF16, 2); INC_Q4P_ATTR(INC_BF16, 4);
    inc_attr_done(attr_set);
  }
#undef INC_Q4P_ATTR
  const unsigned blocks4 = (unsigned)ceil_div64(N, Q4R * Q4W);
#define INC_Q4P(GP) gptq_quant_block_q4_kernel<DT, GP, true><<<blocks4, 64 * Q4W, smem4, s>>>(w, Hinv, scale, zero, codes, q_out, err, N, K, G, i1, g0, maxq, sym)
  INC_DISPATCH_DTYPE(q_dtype, DT, {
    if (gpb == 1) INC_Q4P(1); else if (gpb == 2) INC_Q4P(2); else INC_Q4P(4);
  })
#undef INC_Q4P
  INC_LAUNCH_RETURN();
}

int inc_gptq_lazy_update(float* w, const float* Hinv, const float* err, int64_t N, int64_t K,
                         int64_t i1, int count, inc_stream_t stream) {
  INC_CHECK_ARG(w && Hinv && err && N > 0 && K > 0 && i1 >= 0 && count > 0 && count <= QB);
  const int64_t i2 = i1 + count;
  if (i2 >= K) return INC_OK;  // nothing to the right of the block
  if (count == QB && (i1 % 4) == 0 && (K % 4) == 0 && K * (int64_t)(QB + 1) * 4 < ((int64_t)1 << 32) && !inc_force_small_tiles() && inc_small_tiles_flag(-1) != 86) {
    inc_launch_lazy_update_v3(w, Hinv, err, N, K, i1, i2, K, true, inc_s(stream));
    INC_LAUNCH_RETURN();
  }
#ifdef INC_KBENCH
  if (count == QB && (i1 % 4) == 0 && (K % 4) == 0 && K * (int64_t)(QB + 1) * 4 < ((int64_t)1 << 32) && !inc_force_small_tiles()) {  // harness flag 86: second generation
    const int ncol_tiles = (int)ceil_div64(K - i2, L2T);
    const int row_tiles = (int)ceil_div64(N, L2T);
    int nchunks = (int)ceil_div64(512, row_tiles);  // ~512 workgroups when the trailing matrix is wide enough
    if (nchunks > ncol_tiles) nchunks = ncol_tiles;
    const size_t smem2 = (size_t)2 * L2T * L2T * 4;  // 128 KiB
    static std::atomic<uint64_t> attr2_set{0};
    if (inc_attr_needed(attr2_set)) {
      (void)hipFuncSetAttribute((const void*)gptq_lazy_update_v2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2);
      inc_attr_done(attr2_set);
    }
    gptq_lazy_update_v2_kernel<true><<<dim3((unsigned)nchunks, (unsigned)row_tiles), 256, smem2, inc_s(stream)>>>(
        w, Hinv, err, N, K, i1, i2, nchunks, ncol_tiles);
    INC_LAUNCH_RETURN();
  }
#endif
  const size_t smem = (size_t)LT * LAP * 4 + (size_t)QB * LT * 4;
  static std::atomic<uint64_t> attr_set{0};
  if (inc_attr_needed(attr_set)) {
    (void)hipFuncSetAttribute((const void*)gptq_lazy_update_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    inc_attr_done(attr_set);
  }
  dim3 grid((unsigned)ceil_div64(K - i2, LT), (unsigned)ceil_div64(N, LT));
  gptq_lazy_update_kernel<<<grid, 256, smem, inc_s(stream)>>>(w, Hinv, err, N, K, i1, count);
  INC_LAUNCH_RETURN();
}

// The same update restricted to the columns [col_begin, col_end) of the trailing matrix, col_begin on the 128-column tile grid
// that starts at i2: the tiles, and therefore every sum, are those of the one-launch form (bit-identical W).
int inc_gptq_lazy_update_cols(float* w, const float* Hinv, const float* err, int64_t N, int64_t K, int64_t i1, int count,
                              int64_t col_begin, int64_t col_end, inc_stream_t stream) {
  INC_CHECK_ARG(w && Hinv && err && N > 0 && K > 0 && i1 >= 0 && count > 0 && count <= QB);
  const int64_t i2 = i1 + count;
  INC_CHECK_ARG(col_begin >= i2 && col_end <= K && col_begin <= col_end && ((col_begin - i2) % L2T) == 0 &&
                (col_end == K || ((col_end - col_begin) % L2T) == 0));
  if (col_begin == col_end) return INC_OK;
  if (!(count == QB && (i1 % 4) == 0 && (K % 4) == 0 && K * (int64_t)(QB + 1) * 4 < ((int64_t)1 << 32))) return INC_ERR_UNSUPPORTED;
#ifdef INC_KBENCH
  if (inc_small_tiles_flag(-1) == 106 && launch_lazy_update_x3(w, err, N, K, i1, col_begin, col_end, inc_s(stream))) INC_LAUNCH_RETURN();
#endif
  if (inc_small_tiles_flag(-1) != 86) {
    inc_launch_lazy_update_v3(w, Hinv, err, N, K, i1, col_begin, col_end, false, inc_s(stream));
    INC_LAUNCH_RETURN();
  }
#ifdef INC_KBENCH  // harness flag 86: second generation
  const int ncol_tiles = (int)ceil_div64(col_end - col_begin, L2T);
  const int row_tiles = (int)ceil_div64(N, L2T);
  int nchunks = (int)ceil_div64(512, row_tiles);
  if (nchunks > ncol_tiles) nchunks = ncol_tiles;
  const size_t smem2 = (size_t)2 * L2T * L2T * 4;  // 128 KiB
  static std::atomic<uint64_t> attr2_set{0};
  if (inc_attr_needed(attr2_set)) {
    (void)hipFuncSetAttribute((const void*)gptq_lazy_update_v2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2);
    inc_attr_done(attr2_set);
  }
  gptq_lazy_update_v2_kernel<true><<<dim3((unsigned)nchunks, (unsigned)row_tiles), 256, smem2, inc_s(stream)>>>(
      w, Hinv, err, N, K, i1, col_begin, nchunks, ncol_tiles);
#endif
  INC_LAUNCH_RETURN();
}


// ---- K6 as ONE call: the whole blocked column loop of GPTQ.fasterquant (gptq.py:1250-1304) -------------------------------
// The loop is a chain of launches per 128 columns -- [find_params] -> 128-step quantisation chain -> lazy update -- whose order
// and stream placement a caller would otherwise have to restate (the look-ahead below, the rule that find_params reads W "as
// it is now").  This entry point issues the chain itself on `stream`; with an `aux_stream` the bulk of every lazy update runs
// there underneath the next block's quantisation chain (look-ahead: the chain of block b+1 needs only the NEXT 128 columns of
// block b's update; rest(b-1) is awaited before next(b), so every column still receives its updates in block order from the
// same 128-column tiles: W, codes and Q are bit-identical to the one-stream loop).  Nothing persistent is allocated: two HIP
// events live for the duration of the call.
int inc_gptq_quantize_layer(float* w, const float* Hinv, float* scale, float* zero, int64_t G, const float* loop_scale,
                            const float* loop_zero, int64_t loop_G, uint8_t* codes, void* q_out, int q_dtype, float* err_ws,
                            int64_t N, int64_t K, int group_size, int kernel_group_size, int block_size, int bits, int sym, int flags,
                            inc_stream_t stream, inc_stream_t aux_stream) {
  INC_CHECK_ARG(w && Hinv && scale && zero && err_ws && N > 0 && K > 0 && G > 0 && bits >= 1 && bits <= 8 && group_size > 0);
  const bool dynamic_groups = (flags & INC_GPTQ_DYNAMIC_GROUPS) != 0, mse = (flags & INC_GPTQ_MSE) != 0;
  const bool own_tables = loop_scale == nullptr || loop_scale == scale;
  if (own_tables) { loop_scale = scale; loop_zero = zero; loop_G = G; }
  INC_CHECK_ARG(loop_zero && loop_G > 0);
  const int64_t blocksize = block_size > 0 ? block_size : K;
  hipStream_t main = inc_s(stream), side = inc_s(aux_stream);
  const bool lookahead = aux_stream != nullptr && (flags & INC_GPTQ_NO_LOOKAHEAD) == 0 && K % QB == 0 && blocksize % QB == 0 && K >= 3 * QB;
  // find_params inside the quantisation launch: only when the reference block IS the 128-column block and every group lies inside it
  const bool fuse_params = (flags & INC_GPTQ_NO_FUSED_PARAMS) == 0 && dynamic_groups && !mse && blocksize == QB &&
                           (group_size == 32 || group_size == 64 || group_size == QB) && K % QB == 0 && own_tables &&
                           (K % 4) == 0 && K * (int64_t)(QB + 1) * 4 < ((int64_t)1 << 32);
  float* errs[2] = {err_ws, err_ws + N * QB};
  hipEvent_t ready = nullptr, rest_done = nullptr;
  bool rest_pending = false;
  int rc = INC_OK;
#define INC_TRY(call)                  \
  {                                    \
    const int r_ = (call);             \
    if (r_ != INC_OK) { rc = r_; goto done; } \
  }
  if (lookahead) {
    if (hipEventCreateWithFlags(&ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&rest_done, hipEventDisableTiming) != hipSuccess) {
      rc = INC_ERR_LAUNCH;
      goto done;
    }
    // w / Hinv / the tables were produced on the main stream
    (void)hipEventRecord(ready, main);
    (void)hipStreamWaitEvent(side, ready, 0);
  }
  {
    int blk = 0;
    for (int64_t i1 = 0; i1 < K; ++blk) {
      const int64_t ref_end = std::min((i1 / blocksize + 1) * blocksize, K);  // end of the reference's block (gptq.py:1250)
      const int count = (int)std::min<int64_t>(QB, ref_end - i1);
      if (dynamic_groups && i1 % blocksize == 0 && !fuse_params) {
        // groups that START inside this reference block read the global W as it is now (gptq.py:1266-1272)
        const int64_t g_first = (i1 + group_size - 1) / group_size, g_last = (ref_end - 1) / group_size;
        if (g_last >= g_first) {
          if (lookahead && rest_pending && (g_last + 1) * group_size > i1 + QB) (void)hipStreamWaitEvent(main, rest_done, 0);
          if (mse) INC_TRY(inc_gptq_find_params_mse(w, N, K, g_first * group_size, group_size, (int)(g_last - g_first + 1), bits, sym, 100, 0.8f, 2.4f, scale, zero, G, g_first, stream))
          else INC_TRY(inc_gptq_find_params(w, N, K, g_first * group_size, group_size, (int)(g_last - g_first + 1), bits, sym, scale, zero, G, g_first, stream))
        }
      }
      float* e = errs[lookahead ? (blk & 1) : 0];
      if (fuse_params) INC_TRY(inc_gptq_quant_block_params(w, Hinv, scale, zero, codes, q_out, q_dtype, e, N, K, G, i1, count, group_size, bits, sym, stream))
      else INC_TRY(inc_gptq_quant_block(w, Hinv, loop_scale, loop_zero, codes, q_out, q_dtype, e, N, K, loop_G, i1, count, kernel_group_size, bits, stream))
      const int64_t i2 = i1 + count;
      if (!lookahead) {
        INC_TRY(inc_gptq_lazy_update(w, Hinv, e, N, K, i1, count, stream))
      } else if (i2 < K) {
        const int64_t nxt_end = std::min<int64_t>(i2 + QB, K);
        // Err1 of this block is complete once the chain is: rest(b) may follow rest(b-1) on the side stream at once, next to
        // next(b) (disjoint columns) -- when the remainder is the longer pole (wide layers) the side stream then never idles
        if (nxt_end < K) (void)hipEventRecord(ready, main);
        if (rest_pending) (void)hipStreamWaitEvent(main, rest_done, 0);  // rest(b-1) wrote the columns next(b) updates (and read Err of b-1)
        INC_TRY(inc_gptq_lazy_update_cols(w, Hinv, e, N, K, i1, count, i2, nxt_end, stream))
        if (nxt_end < K) {
          (void)hipStreamWaitEvent(side, ready, 0);
          INC_TRY(inc_gptq_lazy_update_cols(w, Hinv, e, N, K, i1, count, nxt_end, K, aux_stream))
          (void)hipEventRecord(rest_done, side);
          rest_pending = true;
        }
      }
      i1 = i2;
    }
  }
  if (lookahead && rest_pending) (void)hipStreamWaitEvent(main, rest_done, 0);  // w and both Err buffers are free again
done:
#undef INC_TRY
  if (ready) (void)hipEventDestroy(ready);
  if (rest_done) (void)hipEventDestroy(rest_done);
  return rc;
}

#ifdef INC_KBENCH
// harness only (tools/kbench qlayer): operand planes of the split-product lazy update.  `planes` must hold
// inc_debug_lazy_x3_bytes(N, K) bytes: Hinv^T planes (written here, once per layer) followed by scratch for two sets of Err1 planes.
int64_t inc_debug_lazy_x3_bytes(int64_t N, int64_t K) { return lazyx3_planes_bytes(K, K) + 2 * lazyx3_planes_bytes(N, 128); }
int inc_debug_lazy_x3_prepare(const float* Hinv, int64_t N, int64_t K, void* planes, inc_stream_t stream) {
  LazyX3State& g = g_lazyx3;
  if (!planes) { g = LazyX3State(); return INC_OK; }
  uint16_t* hp = static_cast<uint16_t*>(planes);
  lazyx3_split(Hinv, K, K, K, true, hp, &g.hinv_plane, &g.hinv_rp, inc_s(stream));
  g.hinv_planes = hp;
  g.err_planes = reinterpret_cast<uint16_t*>(static_cast<char*>(planes) + lazyx3_planes_bytes(K, K));
  INC_LAUNCH_RETURN();
}
#endif

}  // extern "C"
