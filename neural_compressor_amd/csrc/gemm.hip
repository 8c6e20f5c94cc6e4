// gemm.hip -- K4: fused INT4/INT8 unpack + group-wise dequant + bf16/f16 MFMA GEMM.
//
// Replaces INCWeightOnlyLinear.forward (reference neural_compressor/torch/algorithms/weight_only/
// modules.py:594-610), which dequantises the whole weight with a Python loop (recover, :413-443), caches
// it dense and calls F.linear.  Here the dense weight never exists: packed words are read from HBM once,
// unpacked and scaled in registers, and fed to the matrix cores.
//
//   y[M,N] = x[M,K] . W^T + bias,   W[n,k] = rn16( int8(q[n,k] - zp[n,g]) * scale[n,g] )
//
// where rn16 rounds to the compute dtype (bf16 / f16), exactly what inc_woq_dequant produces, so
// y == F.linear(x, recover_in_that_dtype) up to fp32 accumulation order.
//
// Layout facts that shape the kernels (optimum format, modules.py:254-267):
//   qweight [K/8, N] int32 -- one word = 8 consecutive k of ONE output column n.  That is exactly one
//   lane's B-operand of v_mfma_f32_32x32x16_bf16 / 16x16x32 (8 k-values of column j), so a word is
//   dequantised straight into a B fragment with no cross-lane movement, and words are contiguous in n
//   so wave loads of qweight are full-line.
//
// Kernels
//   woq_gemm_tile   M > 16 : 128x128x64 workgroup tile, 2x2 waves of 64x64 (MFMA 32x32x16), x and the
//                   dequantised weights double-buffered in LDS (pitch 144 B, conflict-free b128 access),
//                   next tile prefetched into registers while the current one is multiplied.
//   woq_gemm_small  M <= 16: HBM-bound.  MFMA 16x16x32 with the 16 rows of x as the A operand; each lane
//                   loads 16 B (4 columns) of qweight per packed row, the 4 waves of a workgroup split
//                   the K range, reduce through LDS, and write one fp32 partial per K-slice which the
//                   epilogue kernel sums (+bias) and converts.
#include "common.hpp"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <bool IS_BF16>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  if constexpr (IS_BF16) {
    return (uint32_t)f32_to_bf16_bits(lo) | ((uint32_t)f32_to_bf16_bits(hi) << 16);
  } else {
    return (uint32_t)f32_to_f16_bits(lo) | ((uint32_t)f32_to_f16_bits(hi) << 16);
  }
}

template <bool IS_BF16>
__device__ __forceinline__ float cvt16(uint16_t b) {
  if constexpr (IS_BF16) return bf16_bits_to_f32(b);
  else return f16_bits_to_f32(b);
}

// group parameters of one (group, column): fp32 scale and integer zero point
struct GroupQ {
  float s;
  int z;
};

template <int BITS>
__device__ __forceinline__ GroupQ load_group(const uint16_t* __restrict__ scales,
                                             const uint32_t* __restrict__ qzeros, int64_t g, int64_t n,
                                             int64_t N, int64_t NW) {
  constexpr int NP = 32 / BITS;
  constexpr uint32_t MASK = (1u << BITS) - 1u;
  GroupQ r;
  r.s = f16_bits_to_f32(scales[g * N + n]);
  uint32_t zz = ((qzeros[g * NW + n / NP] >> (BITS * (uint32_t)(n % NP))) & MASK) + 1u;  // modules.py:407-410
  r.z = zz > MASK ? 0 : (int)zz;
  return r;
}

// dequantise one packed word (NP consecutive k of one column) into NP/2 dwords of 16-bit pairs
template <int BITS, bool IS_BF16>
__device__ __forceinline__ void dequant_word(uint32_t word, const GroupQ& gq, uint32_t (&out)[16 / BITS]) {
  constexpr int NP = 32 / BITS;
  constexpr uint32_t MASK = (1u << BITS) - 1u;
#pragma unroll
  for (int h = 0; h < NP / 2; ++h) {
    const int q0 = (int)((word >> (BITS * (2 * h))) & MASK);
    const int q1 = (int)((word >> (BITS * (2 * h + 1))) & MASK);
    const float v0 = (float)(int8_t)(q0 - gq.z) * gq.s;
    const float v1 = (float)(int8_t)(q1 - gq.z) * gq.s;
    out[h] = pack2<IS_BF16>(v0, v1);
  }
}

template <bool IS_BF16>
__device__ __forceinline__ f32x16 mfma32(const uint4& a, const uint4& b, f32x16 c) {
  if constexpr (IS_BF16) {
    bf16x8 fa, fb;
    __builtin_memcpy(&fa, &a, 16);
    __builtin_memcpy(&fb, &b, 16);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, c, 0, 0, 0);
  } else {
    f16x8 fa, fb;
    __builtin_memcpy(&fa, &a, 16);
    __builtin_memcpy(&fb, &b, 16);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, c, 0, 0, 0);
  }
}
template <bool IS_BF16>
__device__ __forceinline__ f32x4 mfma16(const uint4& a, const uint4& b, f32x4 c) {
  if constexpr (IS_BF16) {
    bf16x8 fa, fb;
    __builtin_memcpy(&fa, &a, 16);
    __builtin_memcpy(&fb, &b, 16);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c, 0, 0, 0);
  } else {
    f16x8 fa, fb;
    __builtin_memcpy(&fa, &a, 16);
    __builtin_memcpy(&fb, &b, 16);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c, 0, 0, 0);
  }
}

// =============================================================================================
// large-M tile kernel
// =============================================================================================
constexpr int GM = 128, GN = 128, GK = 64;
constexpr int GP = GK + 8;  // LDS pitch (elements) = 144 B

template <int BITS, bool IS_BF16>
__global__ __launch_bounds__(256) void woq_gemm_tile_kernel(
    const uint16_t* __restrict__ x, const uint32_t* __restrict__ qweight,
    const uint16_t* __restrict__ scales, const uint32_t* __restrict__ qzeros,
    const int32_t* __restrict__ g_idx, const uint16_t* __restrict__ bias, uint16_t* __restrict__ y,
    int64_t M, int64_t N, int64_t K, int64_t KW, int64_t NW, int group_size, int x_vec_ok) {
  constexpr int NP = 32 / BITS;        // k per packed word
  constexpr int WPT = GK / NP;         // packed rows per K-step
  constexpr int BW = (WPT * GN) / 256; // words per thread per K-step (4-bit: 4, 8-bit: 8)
  constexpr int DW = NP / 2;           // dwords per dequantised word
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  uint16_t* smem = reinterpret_cast<uint16_t*>(smem_raw);
  constexpr int OPER = GM * GP;  // elements per operand stage (GM == GN)

  // XCD-aware tile order: consecutive workgroup ids land on different XCDs (id % 8), so give each
  // XCD a contiguous run of N-tiles of the same M-row-panel -> x panel and weight columns hit in L2.
  const int tiles_n = (int)((N + GN - 1) / GN);
  const int tiles_m = (int)((M + GM - 1) / GM);
  const int nwg = tiles_m * tiles_n;
  int wg = blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = wg % 8, idx = wg / 8;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;  // bijective remap
  }
  const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
  const int64_t m0 = (int64_t)tm * GM, n0 = (int64_t)tn * GN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // staging assignments
  //   x: 1024 16-byte chunks per K-step, 4 per thread: chunk c = tid + 256*i -> row c/8, k-chunk c%8
  //   w: WPT*128 words per K-step: n = tid & 127, packed row = (tid>>7) + 2*i
  uint4 xa[4];
  uint32_t wb[BW];
  GroupQ gq[BW];
  const int bn = tid & 127;
  const int64_t ncol = n0 + bn;

  auto fetch = [&](int kt) {
    const int64_t k0 = (int64_t)kt * GK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * i;
      const int64_t row = m0 + (c >> 3), k = k0 + (c & 7) * 8;
      if (row < M && x_vec_ok && k + 8 <= K) {
        xa[i] = *reinterpret_cast<const uint4*>(x + row * K + k);
      } else {
        uint16_t e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = (row < M && k + j < K) ? x[row * K + k + j] : (uint16_t)0;
        xa[i] = make_uint4((uint32_t)e[0] | ((uint32_t)e[1] << 16), (uint32_t)e[2] | ((uint32_t)e[3] << 16),
                           (uint32_t)e[4] | ((uint32_t)e[5] << 16), (uint32_t)e[6] | ((uint32_t)e[7] << 16));
      }
    }
#pragma unroll
    for (int i = 0; i < BW; ++i) {
      const int64_t kw = k0 / NP + (tid >> 7) + 2 * i;
      if (ncol < N && kw < KW) {
        wb[i] = qweight[kw * N + ncol];
        const int64_t kk = kw * NP;
        const int64_t g = g_idx ? (int64_t)g_idx[kk] : kk / group_size;
        gq[i] = load_group<BITS>(scales, qzeros, g, ncol, N, NW);
      } else {
        wb[i] = 0;
        gq[i].s = 0.f;
        gq[i].z = 0;
      }
    }
  };
  auto stash = [&](int stage) {
    uint16_t* As = smem + (stage * 2 + 0) * OPER;
    uint16_t* Bs = smem + (stage * 2 + 1) * OPER;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * i;
      *reinterpret_cast<uint4*>(As + (c >> 3) * GP + (c & 7) * 8) = xa[i];
    }
#pragma unroll
    for (int i = 0; i < BW; ++i) {
      const int kwl = (tid >> 7) + 2 * i;
      uint32_t d[DW];
      dequant_word<BITS, IS_BF16>(wb[i], gq[i], d);
      if constexpr (DW == 4) {
        *reinterpret_cast<uint4*>(Bs + bn * GP + kwl * NP) = make_uint4(d[0], d[1], d[2], d[3]);
      } else if constexpr (DW == 2) {
        *reinterpret_cast<uint2*>(Bs + bn * GP + kwl * NP) = make_uint2(d[0], d[1]);
      } else {
#pragma unroll
        for (int h = 0; h < DW; ++h) *reinterpret_cast<uint32_t*>(Bs + bn * GP + kwl * NP + 2 * h) = d[h];
      }
    }
  };

  const int nk = (int)((K + GK - 1) / GK);
  fetch(0);
  stash(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) fetch(kt + 1);
    const uint16_t* As = smem + (cur * 2 + 0) * OPER + (wr * 64) * GP;
    const uint16_t* Bs = smem + (cur * 2 + 1) * OPER + (wc * 64) * GP;
#pragma unroll
    for (int kk = 0; kk < GK / 16; ++kk) {
      const int koff = kk * 16 + 8 * (lane >> 5);
      uint4 a[2], b[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        a[m] = *reinterpret_cast<const uint4*>(As + (m * 32 + (lane & 31)) * GP + koff);
        b[m] = *reinterpret_cast<const uint4*>(Bs + (m * 32 + (lane & 31)) * GP + koff);
      }
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = mfma32<IS_BF16>(a[m], b[n], acc[m][n]);
    }
    if (kt + 1 < nk) stash(cur ^ 1);
    __syncthreads();
  }

  // epilogue: D col = lane&31 (n), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (m)
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int64_t col = n0 + wc * 64 + n * 32 + (lane & 31);
    const float bv = (bias && col < N) ? cvt16<IS_BF16>(bias[col]) : 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wr * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < M && col < N) {
          const float v = acc[m][n][r] + bv;
          y[row * N + col] = IS_BF16 ? f32_to_bf16_bits(v) : f32_to_f16_bits(v);
        }
      }
    }
  }
}

// =============================================================================================
// small-M (decode) kernel: M <= 16
// =============================================================================================
constexpr int SN = 64;  // columns per workgroup strip (16 lanes x 4 columns)

template <int BITS, bool IS_BF16>
__global__ __launch_bounds__(256) void woq_gemm_small_kernel(
    const uint16_t* __restrict__ x, const uint32_t* __restrict__ qweight,
    const uint16_t* __restrict__ scales, const uint32_t* __restrict__ qzeros,
    const int32_t* __restrict__ g_idx, float* __restrict__ partial, int64_t M, int64_t N, int64_t K,
    int64_t KW, int64_t NW, int group_size, int kw_per_slice) {
  constexpr int NP = 32 / BITS;
  constexpr int STEP_KW = 32 / NP;  // packed rows per MFMA K=32 step (4-bit: 4, 8-bit: 8)
  __shared__ float red[4][16][SN + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t n0 = (int64_t)blockIdx.x * SN;
  const int slice = blockIdx.y;
  // this wave's packed-row range inside the slice
  const int per_wave = kw_per_slice / 4;
  const int64_t kw_beg = (int64_t)slice * kw_per_slice + (int64_t)wave * per_wave;
  const int64_t kw_end = kw_beg + per_wave;

  const int jn = lane & 15, koct = lane >> 4;  // column quad index, k-octet index (0..3)
  const int64_t ncol = n0 + 4 * jn;            // first of this lane's 4 columns
  const int am = lane & 15;                    // A row (m)
  f32x4 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int64_t kw = kw_beg; kw < kw_end; kw += STEP_KW) {
    // A fragment: x[m = lane&15][k = 32*step + 8*koct .. +7]
    const int64_t ka = kw * NP + 8 * koct;
    uint4 a;
    if (am < M && ka + 8 <= K) {
      const uint16_t* p = x + (int64_t)am * K + ka;
      if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
        a = *reinterpret_cast<const uint4*>(p);
      } else {
        a = make_uint4((uint32_t)p[0] | ((uint32_t)p[1] << 16), (uint32_t)p[2] | ((uint32_t)p[3] << 16),
                       (uint32_t)p[4] | ((uint32_t)p[5] << 16), (uint32_t)p[6] | ((uint32_t)p[7] << 16));
      }
    } else {
      uint16_t e[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = (am < M && ka + j < K) ? x[(int64_t)am * K + ka + j] : (uint16_t)0;
      a = make_uint4((uint32_t)e[0] | ((uint32_t)e[1] << 16), (uint32_t)e[2] | ((uint32_t)e[3] << 16),
                     (uint32_t)e[4] | ((uint32_t)e[5] << 16), (uint32_t)e[6] | ((uint32_t)e[7] << 16));
    }
    // B fragments: this lane's k-octet of 4 adjacent columns.
    uint4 b[4];
    if constexpr (BITS == 4) {
      const int64_t kwr = kw + koct;  // one packed row holds the whole octet
      uint32_t w4[4] = {0, 0, 0, 0};
      if (kwr < KW) {
        if (ncol + 4 <= N && (N % 4 == 0)) {
          const uint4 v = *reinterpret_cast<const uint4*>(qweight + kwr * N + ncol);
          w4[0] = v.x; w4[1] = v.y; w4[2] = v.z; w4[3] = v.w;
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) w4[c] = (ncol + c < N) ? qweight[kwr * N + ncol + c] : 0u;
        }
      }
      const int64_t kk = kwr * NP;
      const int64_t g = (kwr < KW) ? (g_idx ? (int64_t)g_idx[kk] : kk / group_size) : 0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        GroupQ gq;
        if (ncol + c < N && kwr < KW) gq = load_group<4>(scales, qzeros, g, ncol + c, N, NW);
        else { gq.s = 0.f; gq.z = 0; }
        uint32_t d[4];
        dequant_word<4, IS_BF16>(w4[c], gq, d);
        b[c] = make_uint4(d[0], d[1], d[2], d[3]);
      }
    } else {  // 8-bit: an octet spans two packed rows
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        uint32_t d[4] = {0, 0, 0, 0};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int64_t kwr = kw + 2 * koct + h;
          if (kwr < KW && ncol + c < N) {
            const uint32_t word = qweight[kwr * N + ncol + c];
            const int64_t kk = kwr * NP;
            const int64_t g = g_idx ? (int64_t)g_idx[kk] : kk / group_size;
            const GroupQ gq = load_group<8>(scales, qzeros, g, ncol + c, N, NW);
            uint32_t dd[2];
            dequant_word<8, IS_BF16>(word, gq, dd);
            d[2 * h] = dd[0];
            d[2 * h + 1] = dd[1];
          }
        }
        b[c] = make_uint4(d[0], d[1], d[2], d[3]);
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = mfma16<IS_BF16>(a, b[c], acc[c]);
  }
  // D: col = lane&15 -> column quad jn, sub-column c; row m = 4*(lane>>4) + r
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][4 * koct + r][4 * jn + c] = acc[c][r];
  __syncthreads();
  for (int idx = tid; idx < 16 * SN; idx += 256) {
    const int m = idx / SN, c = idx - m * SN;
    if (m < M && n0 + c < N) {
      const float v = red[0][m][c] + red[1][m][c] + red[2][m][c] + red[3][m][c];
      partial[((int64_t)slice * M + m) * N + n0 + c] = v;
    }
  }
}

template <bool IS_BF16>
__global__ void splitk_reduce_kernel(const float* __restrict__ partial, const uint16_t* __restrict__ bias,
                                     uint16_t* __restrict__ y, int64_t M, int64_t N, int slices) {
  const int64_t total = M * N;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    float v = 0.f;
    for (int s = 0; s < slices; ++s) v += partial[(int64_t)s * total + i];
    if (bias) v += cvt16<IS_BF16>(bias[i % N]);
    y[i] = IS_BF16 ? f32_to_bf16_bits(v) : f32_to_f16_bits(v);
  }
}

// choose the number of K-slices for the small-M kernel: enough workgroups to cover the chip, each
// slice a multiple of 4 waves x one MFMA K=32 step
inline int small_slices(int64_t N, int64_t K, int bits, int* kw_per_slice_out) {
  const int np = 32 / bits;
  const int64_t KW = ceil_div64(K, np);
  const int step_kw = 32 / np;
  const int64_t strips = ceil_div64(N, SN);
  const int64_t unit = 4 * step_kw;              // packed rows per workgroup per MFMA round
  const int64_t units = ceil_div64(KW, unit);     // rounds available along K
  int64_t want = ceil_div64(1024, strips);        // ~1024 workgroups
  if (want < 1) want = 1;
  if (want > units) want = units;
  if (want > 64) want = 64;
  const int64_t units_per_slice = ceil_div64(units, want);
  const int slices = (int)ceil_div64(units, units_per_slice);
  *kw_per_slice_out = (int)(units_per_slice * unit);
  return slices;
}

}  // namespace

extern "C" {

int64_t inc_woq_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M > 16) return 0;
  (void)K;
  return (int64_t)64 * M * N * 4;  // <= 64 K-slices of fp32 partials
}

int inc_woq_gemm(const void* x, int xdtype, const int32_t* qweight, const uint16_t* scales,
                 const int32_t* qzeros, const int32_t* g_idx, const void* bias, void* y, int64_t M,
                 int64_t N, int64_t K, int64_t G, int group_size, int bits, void* workspace,
                 int64_t workspace_bytes, inc_stream_t stream) {
  INC_CHECK_ARG(x && qweight && scales && qzeros && y && M > 0 && N > 0 && K > 0 && G > 0 && group_size > 0);
  if (!(bits == 4 || bits == 8)) return INC_ERR_UNSUPPORTED;
  if (g_idx) return INC_ERR_UNSUPPORTED;  // per-element groups (act_order): not in this ABI version
  if (!(xdtype == INC_BF16 || xdtype == INC_F16)) return INC_ERR_UNSUPPORTED;
  const int np = 32 / bits;
  // a packed word must not straddle two groups unless g_idx is given per element... (word-granular
  // group lookup): require group boundaries on word boundaries.
  if (!g_idx && (group_size % np) != 0 && group_size < K) return INC_ERR_UNSUPPORTED;
  const int64_t KW = ceil_div64(K, np), NW = ceil_div64(N, np);
  hipStream_t s = inc_s(stream);
  const uint16_t* xp = (const uint16_t*)x;
  const uint32_t* qw = (const uint32_t*)qweight;
  const uint32_t* qz = (const uint32_t*)qzeros;
  const uint16_t* bp = (const uint16_t*)bias;
  uint16_t* yp = (uint16_t*)y;
  const bool bf = xdtype == INC_BF16;
  if (M > 16) {
    const int x_vec_ok = (K % 8 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    const size_t smem = (size_t)2 * 2 * GM * GP * sizeof(uint16_t);
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)woq_gemm_tile_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      (void)hipFuncSetAttribute((const void*)woq_gemm_tile_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      (void)hipFuncSetAttribute((const void*)woq_gemm_tile_kernel<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      (void)hipFuncSetAttribute((const void*)woq_gemm_tile_kernel<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      attr_set = true;
    }
    const unsigned grid = (unsigned)(ceil_div64(M, GM) * ceil_div64(N, GN));
#define INC_TILE(B, F) woq_gemm_tile_kernel<B, F><<<grid, 256, smem, s>>>(xp, qw, scales, qz, g_idx, bp, yp, M, N, K, KW, NW, group_size, x_vec_ok)
    if (bits == 4) { if (bf) INC_TILE(4, true); else INC_TILE(4, false); }
    else { if (bf) INC_TILE(8, true); else INC_TILE(8, false); }
#undef INC_TILE
  } else {
    int kw_per_slice = 0;
    const int slices = small_slices(N, K, bits, &kw_per_slice);
    if (!workspace || workspace_bytes < (int64_t)slices * M * N * 4) return INC_ERR_WORKSPACE;
    float* part = (float*)workspace;
    dim3 grid((unsigned)ceil_div64(N, SN), (unsigned)slices);
#define INC_SMALL(B, F) woq_gemm_small_kernel<B, F><<<grid, 256, 0, s>>>(xp, qw, scales, qz, g_idx, part, M, N, K, KW, NW, group_size, kw_per_slice)
    if (bits == 4) { if (bf) INC_SMALL(4, true); else INC_SMALL(4, false); }
    else { if (bf) INC_SMALL(8, true); else INC_SMALL(8, false); }
#undef INC_SMALL
    int64_t rb = ceil_div64(M * N, 256);
    if (rb > 2048) rb = 2048;
    if (bf) splitk_reduce_kernel<true><<<(unsigned)rb, 256, 0, s>>>(part, bp, yp, M, N, slices);
    else splitk_reduce_kernel<false><<<(unsigned)rb, 256, 0, s>>>(part, bp, yp, M, N, slices);
  }
  INC_LAUNCH_RETURN();
}

}  // extern "C"
