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
#include "gemm_common.hpp"

namespace {


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
  // 4 / 8 bits: a K-step is a whole number of words and a thread dequantises whole words.  Every other width the reference's
  // configs tune (1, 2, 3, 5, 6, 7: n_pack = 32 // bits, modules.py:231 -- 10 / 6 / 5 / 4 fields with unused high bits for 3 / 5 / 6
  // / 7) takes the ANYW form: a thread owns 32 consecutive k of one column, fetches the <= MAXW words they live in and places
  // every field by its own k (per-element group lookup, so any group_size and any g_idx).
  constexpr bool ANYW = !(BITS == 4 || BITS == 8);
  constexpr int MAXW = (31 + NP - 1) / NP + 1;  // words a run of 32 k can touch
  constexpr int WPT = ANYW ? 1 : GK / NP;         // packed rows per K-step
  constexpr int BW = ANYW ? MAXW : (WPT * GN) / 256; // words per thread per K-step (4-bit: 4, 8-bit: 8)
  constexpr int DW = ANYW ? 1 : NP / 2;           // dwords per dequantised word
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

  int64_t fetched_k0 = 0;  // K offset of the words held in wb (per-element g_idx lookups happen when they are dequantised)
  auto fetch = [&](int kt) {
    const int64_t k0 = (int64_t)kt * GK;
    fetched_k0 = k0;
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
    if constexpr (ANYW) {
      const int64_t kwf = (k0 + 32 * (tid >> 7)) / NP;  // first word of this thread's 32 k
#pragma unroll
      for (int i = 0; i < BW; ++i) wb[i] = (ncol < N && kwf + i < KW) ? qweight[(kwf + i) * N + ncol] : 0u;
    } else {
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
    if constexpr (ANYW) {
      constexpr uint32_t MASK = (1u << BITS) - 1u;
      const int kl0 = 32 * (tid >> 7);                 // first k of this thread inside the K-step
      const int64_t kbeg = fetched_k0 + kl0;
      const int64_t kwf = kbeg / NP;
      uint16_t* dst = Bs + bn * GP;
      if (ncol >= N || kbeg + 32 > K) {                // columns past N / the K tail multiply as zeros
#pragma unroll
        for (int q8 = 0; q8 < 4; ++q8) *reinterpret_cast<uint4*>(dst + kl0 + 8 * q8) = make_uint4(0u, 0u, 0u, 0u);
      }
      if (ncol < N) {
        int gprev = -1;
        GroupQ gcur;
        gcur.s = 0.f;
        gcur.z = 0;
#pragma unroll
        for (int i = 0; i < BW; ++i) {
#pragma unroll
          for (int e = 0; e < NP; ++e) {
            const int64_t k = (kwf + i) * NP + e;
            if (k >= kbeg && k < kbeg + 32 && k < K) {
              const int g = g_idx ? g_idx[k] : (int)((uint32_t)k / (uint32_t)group_size);  // (K < 2^31: inc_woq_gemm checks)
              if (g != gprev) {
                gcur = load_group<BITS>(scales, qzeros, g, ncol, N, NW);
                gprev = g;
              }
              const int q = (int)((wb[i] >> (BITS * e)) & MASK);
              const float v = (float)(int8_t)(q - gcur.z) * gcur.s;
              dst[k - fetched_k0] = IS_BF16 ? f32_to_bf16_bits(v) : f32_to_f16_bits(v);
            }
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < BW; ++i) {
        const int kwl = (tid >> 7) + 2 * i;
        uint32_t d[DW];
        if (g_idx && ncol < N && fetched_k0 / NP + kwl < KW)
          dequant_word_gidx<BITS, IS_BF16>(wb[i], scales, qzeros, g_idx, fetched_k0 + (int64_t)kwl * NP, K, ncol, N, NW, d);
        else
          dequant_word<BITS, IS_BF16>(wb[i], gq[i], d);
        if constexpr (DW == 4) {
          *reinterpret_cast<uint4*>(Bs + bn * GP + kwl * NP) = make_uint4(d[0], d[1], d[2], d[3]);
        } else {
          *reinterpret_cast<uint2*>(Bs + bn * GP + kwl * NP) = make_uint2(d[0], d[1]);
        }
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
// large-M fast path (4-bit, K % 64 == 0, group_size % 32 == 0): 256x256x64 workgroup tile
// =============================================================================================
// 512 threads = 8 waves as 2 (M) x 4 (N); a wave owns 128 (m) x 64 (n) = 4 x 2 MFMA 32x32x16 tiles
// (128 fp32 accumulators).  One workgroup per CU (128 KiB LDS, two stages):
//   x tile      256 rows x 128 B, brought in by LDS-DMA (global_load_lds_dwordx4): every DMA
//               instruction moves 8 full 128-byte rows.  The LDS image is row-major with the 16-byte
//               chunk index XOR-ed by ((row >> 1) & 7); the DMA destination is lane-linear, so the
//               permutation is applied to the per-lane SOURCE address and again on the ds_read_b128
//               side -> conflict-free fragment reads and full-line global reads.
//   W tile      each thread fetches 4 packed words (4 packed rows of ONE column: the wave reads 256
//               contiguous bytes per row), dequantises them once for the whole workgroup and writes
//               the 4 x 16 B in MFMA-fragment order [n-frag][k16][lane] -> both the ds_write_b128 and
//               the ds_read_b128 are lane-linear (conflict-free).
// The MFMA is issued with W as the A operand and x as the B operand: a lane then owns 4 consecutive
// output columns per accumulator quad -> 8-byte stores in the epilogue (4x fewer store instructions).

// two 8-bit packed words (k0..3, k4..7 of one column) -> 8 x rn16(int8(q - z) * s).  The difference wraps to int8 exactly like the
// reference's recover(), which unpacks 8-bit codes and zero points into int8 tensors and subtracts there (modules.py:377-443:
// an asymmetric code more than 127 away from its zero point flips sign) -- and like inc_woq_dequant / the first-generation
// kernel (dequant_word).  The product of an int8 and an 11-bit scale is exact in fp32: one rounding, in the 16-bit conversion.
template <bool IS_BF16>
__device__ __forceinline__ uint4 dequant8_from_bytes(uint32_t w0, uint32_t w1, float s, int z) {
  float f[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f[j] = (float)(int)(int8_t)(uint8_t)(((w0 >> (8 * j)) & 0xffu) - (uint32_t)z) * s;
    f[4 + j] = (float)(int)(int8_t)(uint8_t)(((w1 >> (8 * j)) & 0xffu) - (uint32_t)z) * s;
  }
  uint4 o;
  o.x = cvt_pair<IS_BF16>(f[0], f[1]);
  o.y = cvt_pair<IS_BF16>(f[2], f[3]);
  o.z = cvt_pair<IS_BF16>(f[4], f[5]);
  o.w = cvt_pair<IS_BF16>(f[6], f[7]);
  return o;
}

// ABL != 0: timing-only ablations for tools/kbench (results are WRONG): 1 = no dequant arithmetic, 2 = x fragments read
// once per K-step, 3 = no global traffic inside the K-loop, 4 = no MFMA
template <bool IS_BF16, int ABL = 0>
__global__ __launch_bounds__(512) void woq_gemm_w4_big_kernel(
    const uint16_t* __restrict__ x, const uint32_t* __restrict__ qweight,
    const uint16_t* __restrict__ scales, const uint32_t* __restrict__ qzeros,
    const uint16_t* __restrict__ bias, uint16_t* __restrict__ y, int64_t M, int64_t N, int64_t K,
    int64_t NW, int g_shift, int y_vec_ok) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Abase = smem;                 // 2 stages
  char* const Bbase = smem + 2 * T_ASTAGE;  // 2 stages

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
  const float inv_u = fp8_unit_inverse();
  const int wm = wave >> 2, wn = wave & 3;

  // ---- x staging (LDS-DMA): instruction i of this wave fills LDS rows (wave*4+i)*8 .. +7 -------
  uint32_t avoff[4];  // byte offset of this lane's 16-byte chunk from x + m0*K + kt*TK
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int R = (wave * 4 + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((R >> 1) & 7);
    int64_t row = m0 + R;
    if (row > M - 1) row = M - 1;  // rows past M are computed from a valid row and never stored
    avoff[i] = (uint32_t)(((row - m0) * K + 8 * c) * 2);
  }
  const uint16_t* const xtile = x + m0 * K;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;  // LDS byte address of the dynamic segment (low half of the flat address)
  // ---- W staging: this thread's column and packed-row half ---------------------------------------
  const int bcol = tid & 255, kwh = tid >> 8;
  int64_t ncol = n0 + bcol;
  if (ncol > N - 1) ncol = N - 1;
  const uint32_t* wsrc = qweight + (int64_t)(4 * kwh) * N + ncol;
  const int zshift = 4 * (int)(ncol & 7);
  const int64_t zcol = ncol >> 3;
  // LDS slot of word j: [nf = bcol>>5][kk = 2*kwh + (j>>1)][lane' = (bcol&31) + 32*(j&1)]
  const int bdst0 = (((bcol >> 5) * 4 + 2 * kwh) * 64 + (bcol & 31)) * 16;

  // this thread's share of the NEXT W tile, still packed (registers): 4 words + scale + zero word
  uint32_t raw[4], zw;
  uint16_t scb;
  const int voff = (4 * kwh) * (int)N + (int)ncol;  // element offset inside a K-tile of qweight (fits 32 bits)
  auto load_w = [&](int kt) {
    const uint32_t* tile = qweight + (int64_t)kt * (TK / 8) * N;  // wave-uniform base
#pragma unroll
    for (int j = 0; j < 4; ++j) raw[j] = tile[voff + j * (int)N];
    const int64_t g = g_shift >= 0 ? (((int64_t)kt * TK + 32 * kwh) >> g_shift) : 0;
    scb = scales[g * N + ncol];
    zw = qzeros[g * NW + zcol];
  };
  auto stash_regs = [&](int stage, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint16_t sb, uint32_t zword) {
    const float sc0 = f16_bits_to_f32(sb);
    uint32_t zz = ((zword >> zshift) & 15u) + 1u;  // modules.py:407-410 (stored zp-1; wraps above 15)
    zz = zz > 15u ? 0u : zz;
    const float nzs = -(float)zz * sc0;
    const float sc = sc0 * inv_u;
    char* dst = Bbase + stage * T_BSTAGE + bdst0;
    if constexpr (ABL == 1) {
      *reinterpret_cast<uint4*>(dst) = make_uint4(w0, w1, w2, w3);
      *reinterpret_cast<uint4*>(dst + 32 * 16) = make_uint4(w1, w2, w3, w0);
      *reinterpret_cast<uint4*>(dst + 64 * 16) = make_uint4(w2, w3, w0, w1);
      *reinterpret_cast<uint4*>(dst + (64 + 32) * 16) = make_uint4(w3, w0, w1, __float_as_uint(nzs));
      return;
    }
    *reinterpret_cast<uint4*>(dst) = dequant8<IS_BF16>(w0, sc, nzs);                    // kk = 2*kwh,   k-octet 0
    *reinterpret_cast<uint4*>(dst + 32 * 16) = dequant8<IS_BF16>(w1, sc, nzs);          //               k-octet 1
    *reinterpret_cast<uint4*>(dst + 64 * 16) = dequant8<IS_BF16>(w2, sc, nzs);          // kk = 2*kwh+1, k-octet 0
    *reinterpret_cast<uint4*>(dst + (64 + 32) * 16) = dequant8<IS_BF16>(w3, sc, nzs);   //               k-octet 1
  };
  auto stash_w = [&](int stage) { stash_regs(stage, raw[0], raw[1], raw[2], raw[3], scb, zw); };
  auto dma_x = [&](int kt, int stage) {
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + stage * T_ASTAGE + wave * 4096);
    lds_dma_4x1k(xtile + (int64_t)kt * TK, dst, avoff[0], avoff[1], avoff[2], avoff[3]);
  };
  // makes the compiler's own wait for the packed-W registers happen HERE (before the next DMA is
  // issued): its s_waitcnt accounting does not see the DMA and would otherwise drain it later
  auto settle_w = [&]() {
    asm volatile("" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(zw));
    uint32_t t = scb;
    asm volatile("" : "+v"(t));
    scb = (uint16_t)t;
  };

  f32x16 acc[2][4];  // [n-frag][m-frag]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // fragment read offsets
  const int a_row = wm * 128 + (lane & 31);         // + 32*mf
  const int a_sw = ((lane & 31) >> 1) & 7;           // (row >> 1) & 7 (tile bases are multiples of 32)
  const int a_hi = lane >> 5;                        // chunk = 2*kk + a_hi
  const int b_off = (wn * 2 * 4 * 64 + lane) * 16;   // + (nf*4 + kk) * 1024

  uint4 xa_keep[4];
  auto mma_step = [&](const char* As, const char* Bs, int kk) {
    uint4 xa[4], wb[2];
    const int chunk = ((2 * kk + a_hi) ^ a_sw) << 4;
#pragma unroll
    for (int mf = 0; mf < 4; ++mf) {
      if (ABL == 2 && kk != 0) xa[mf] = xa_keep[mf];
      else xa[mf] = *reinterpret_cast<const uint4*>(As + (a_row + 32 * mf) * 128 + chunk);
      if (ABL == 2 && kk == 0) xa_keep[mf] = xa[mf];
    }
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) wb[nf] = *reinterpret_cast<const uint4*>(Bs + (nf * 4 + kk) * 1024);
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
      for (int mf = 0; mf < 4; ++mf) {
        if constexpr (ABL == 4) {
          acc[nf][mf][0] += __uint_as_float(wb[nf].x ^ xa[mf].y);  // keeps the fragment reads alive without the matrix pipe
        } else {
          acc[nf][mf] = mfma32<IS_BF16>(wb[nf], xa[mf], acc[nf][mf]);
        }
      }
  };

  // Two-stage pipeline, one barrier per K-tile.  In iteration kt the LDS-DMA of x tile kt+1 and the
  // packed loads of W tile kt+2 are issued first and land under the 32 MFMAs; the dequantisation of W
  // tile kt+1 (registers -> other LDS stage) sits between the first and second MFMA group so that
  // its VALU work shares the issue slots the matrix pipe leaves free.  vmcnt(0) only at the barrier.
  const int nk = (int)(K / TK);
  dma_x(0, 0);
  load_w(0);
  stash_w(0);
  load_w(nk > 1 ? 1 : 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < nk - 1; ++kt) {
    const int cur = kt & 1;
    const char* As = Abase + cur * T_ASTAGE;
    const char* Bs = Bbase + cur * T_BSTAGE + b_off;
    settle_w();
    if (ABL != 3) dma_x(kt + 1, cur ^ 1);
    const uint32_t r0 = raw[0], r1 = raw[1], r2 = raw[2], r3 = raw[3], zcur = zw;
    const uint16_t scur = scb;
    if (ABL != 3) load_w(kt + 2 < nk ? kt + 2 : nk - 1);   // in flight for the whole K-step
    __builtin_amdgcn_sched_barrier(0);        // keep the loads up here (hipcc would sink them to their use)
    if constexpr (ABL == 5 || ABL == 6) {
      // one packed word per k16 group: ~19 VALU + 1 ds_write next to each group of 8 MFMAs instead of 76 VALU next to
      // the first group (the matrix pipe starves while a wave issues a long VALU run: both waves of a SIMD are in the
      // same phase, profiles/r1_pmc ablation)
      const float sc0 = f16_bits_to_f32(scur);
      uint32_t zz = ((zcur >> zshift) & 15u) + 1u;
      zz = zz > 15u ? 0u : zz;
      const float nzs = -(float)zz * sc0;
      const float sc = sc0 * inv_u;
      char* dst = Bbase + (cur ^ 1) * T_BSTAGE + bdst0;
      const uint32_t rw[4] = {r0, r1, r2, r3};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        mma_step(As, Bs, kk);
        *reinterpret_cast<uint4*>(dst + ((kk >> 1) * 64 + 32 * (kk & 1)) * 16) = dequant8<IS_BF16>(rw[kk], sc, nzs);
        if constexpr (ABL == 6) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);  // 3 VALU
          }
        }
      }
    } else {
    mma_step(As, Bs, 0);
    stash_regs(cur ^ 1, r0, r1, r2, r3, scur, zcur);
    mma_step(As, Bs, 1);
    mma_step(As, Bs, 2);
    mma_step(As, Bs, 3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  {
    const int cur = (nk - 1) & 1;
    const char* As = Abase + cur * T_ASTAGE;
    const char* Bs = Bbase + cur * T_BSTAGE + b_off;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) mma_step(As, Bs, kk);
  }

  // epilogue: D row i = n-offset (r&3) + 8*(r>>2) + 4*(lane>>5), col j = m-offset lane&31
#pragma unroll
  for (int nf = 0; nf < 2; ++nf) {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int64_t nb = n0 + wn * 64 + nf * 32 + 8 * rq + 4 * (lane >> 5);
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      if (bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (nb + e < N) bv[e] = cvt16<IS_BF16>(bias[nb + e]);
      }
#pragma unroll
      for (int mf = 0; mf < 4; ++mf) {
        const int64_t m = m0 + wm * 128 + mf * 32 + (lane & 31);
        if (m >= M) continue;
        const float v0 = acc[nf][mf][4 * rq + 0] + bv[0], v1 = acc[nf][mf][4 * rq + 1] + bv[1];
        const float v2 = acc[nf][mf][4 * rq + 2] + bv[2], v3 = acc[nf][mf][4 * rq + 3] + bv[3];
        uint16_t* dst = y + m * N + nb;
        if (y_vec_ok && nb + 4 <= N) {
          *reinterpret_cast<uint2*>(dst) = make_uint2(cvt_pair<IS_BF16>(v0, v1), cvt_pair<IS_BF16>(v2, v3));
        } else {
          const float vv[4] = {v0, v1, v2, v3};
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (nb + e < N) dst[e] = IS_BF16 ? f32_to_bf16_bits(vv[e]) : f32_to_f16_bits(vv[e]);
        }
      }
    }
  }
}

// =============================================================================================
// large-M path "3A2B" (4-bit, K % 128 == 0): 256x256x64 tile, THREE x stages + TWO W stages = all 160 KiB of LDS
// =============================================================================================
// Ablations (tools/kbench ablate, profiles/r1f): with two stages the DMA of x tile t+1 is issued at the start of step t
// and must have landed at its end, so a step can never be shorter than one loaded HBM/L2 round trip (~1.4 us measured
// against ~0.9 us of MFMA work) -- removing the global traffic alone gives +15-25 %, a deeper pipeline with shorter steps
// does not help because its loads still have only ~one round trip to land.  Here the x DMA runs TWO steps ahead (three 32
// KiB stages), the packed W words (8 KiB per step: registers are enough) also two steps ahead, and the dequantised W keeps
// its two 32 KiB stages: 3*32 + 2*32 = 160 KiB, exactly one CU's LDS.  All loop traffic is issued from asm and retired
// with counted waits: per step a wave issues 6 W requests then 4 DMAs; `vmcnt(14)` before the dequantisation leaves
// the previous step's 4 DMAs + this step's 10 requests in flight, `vmcnt(10)` before the barrier retires those 4 DMAs.
// The dequantisation is spread over the four k16 groups (one packed word next to each 8 MFMAs).
// SCHED selects the step's instruction schedule (same data flow, same results):
//   0  as written, the compiler orders the step (it sinks every fragment read to just before the MFMAs that use it and
//      so re-exposes the LDS latency four times per step -- see profiles/r1i)
//   1  sched_barrier fences pin the software pipeline: reads of k16 group g+1 are issued BEFORE the MFMAs of group g
//   3  "ping-pong": compute and load segments separated by barriers, partner waves half a step apart (see below)
//   10-16, 31-37  timing-only ablations (tools/kbench ablate)
// All of 0/1/3 give bit-identical outputs and, measured (profiles/r1i_kbench_*.log), the same speed within 10 %: the step is
// not bound by instruction placement -- a variant that spread the 10 VMEM requests between the carried group's MFMAs
// changed nothing either -- but by the sum of its parts.
#define INC_3A2B_DEFAULT_SCHED 1
// 1: the producer / consumer kernel (woq_gemm_w4_pc_kernel) takes the large-M 4-bit path whenever it applies
// 1: the direct-to-register kernel (gemm_d2r.hip) takes the large-M 4-bit path whenever it applies
#ifndef INC_GEMM_DEFAULT_D2R
#define INC_GEMM_DEFAULT_D2R 1
#endif
#ifndef INC_GEMM_DEFAULT_PC
#define INC_GEMM_DEFAULT_PC 1
#endif
// BITS = 8 (weight-only INT8, BASELINE config #1's packed layers): the same kernel with a 16 KiB packed W tile per step --
// a thread fetches 8 words (4 k each) of its column instead of 4 (8 k each), two words make one 16-byte fragment row, the
// integer -> float step is v_cvt_f32_ubyte*, a step has 14 VMEM requests instead of 10 (the counted waits follow).
template <bool IS_BF16, int SCHED, int BITS = 4>
__global__ __launch_bounds__(512) void woq_gemm_w4_3a2b_kernel(
    const uint16_t* __restrict__ x, const uint32_t* __restrict__ qweight,
    const uint16_t* __restrict__ scales, const uint32_t* __restrict__ qzeros,
    const uint16_t* __restrict__ bias, uint16_t* __restrict__ y, int64_t M, int64_t N, int64_t K,
    int64_t NW, int g_shift, int y_vec_ok, float* __restrict__ partial, int steps_per_split) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Abase = smem;                 // 3 stages
  char* const Bbase = smem + 3 * T_ASTAGE;  // 2 stages
  const int tiles_n = (int)((N + TN - 1) / TN);
  const int tiles_m = (int)((M + TM - 1) / TM);
  const int nwg = tiles_m * tiles_n;
  int wg = blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = wg % 8, idx = wg / 8;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;  // bijective XCD remap
  }
  // row-major inside an XCD's range: a 2 x 16 strip shares the 32 KiB x tiles 16 ways and the 8 KiB W tiles 2 ways -- 192 KiB of
  // unique operand bytes per K-step for 32 tiles; the 8 x 4 patch of banded_tile_decode needs 288 KiB and measured 15 % slower
  const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
  const int64_t m0 = (int64_t)tm * TM, n0 = (int64_t)tn * TN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float inv_u = fp8_unit_inverse();
  const int wm = wave >> 2, wn = wave & 3;
  // timing-only ablations (tools/kbench ablate; results are wrong by construction): which part of a step costs what
  // 30 + mask: the ping-pong schedule (3) minus {1: loads, 2: dequantisation + W writes, 4: fragment reads}
  constexpr bool PP = SCHED == 3 || SCHED >= 30;
  constexpr int PPM = SCHED >= 30 ? SCHED - 30 : 0;
  constexpr bool NO_DEQ = SCHED == 10 || SCHED == 15 || SCHED == 16 || (PPM & 2);  // no int4 -> bf16 arithmetic
  constexpr bool NO_WR = SCHED == 11 || SCHED == 15 || SCHED == 16 || (PPM & 2);   // no ds_write of the dequantised W
  constexpr bool NO_RD = SCHED == 12 || SCHED == 15 || SCHED == 16 || (PPM & 4);   // no fragment reads
  constexpr bool NO_LD = SCHED == 13 || SCHED == 15 || SCHED == 16 || (PPM & 1);   // no global loads / LDS-DMA
  constexpr bool NO_BAR = SCHED == 14 || SCHED == 16;                 // no per-step barrier
  constexpr bool NO_VMWAIT = SCHED == 17;                             // loads issued, their counted waits dropped
  constexpr bool FIXED_ADDR = SCHED == 18;                            // loads always fetch K-tile 0 (address arithmetic hoisted)
  constexpr bool NO_WLD = SCHED == 19;                                // x LDS-DMA kept, the 6 W-side loads dropped
  constexpr bool NO_DMA = SCHED == 20;                                // W-side loads kept, the 4 x LDS-DMAs dropped
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;

  uint32_t avoff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int R = (wave * 4 + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((R >> 1) & 7);
    int64_t row = m0 + R;
    if (row > M - 1) row = M - 1;
    avoff[i] = (uint32_t)(((row - m0) * K + 8 * c) * 2);
  }
  const uint16_t* const xtile = x + m0 * K;
  const int bcol = tid & 255, kwh = tid >> 8;
  int64_t ncol = n0 + bcol;
  if (ncol > N - 1) ncol = N - 1;
  constexpr int NPK = 32 / BITS;       // codes per packed word: 8 / 4
  constexpr int NWD = 32 / NPK;        // words of one column per thread and step (32 k): 4 / 8
  uint32_t wvoff[8];
#pragma unroll
  for (int j = 0; j < NWD; ++j) wvoff[j] = (uint32_t)(((int64_t)(NWD * kwh + j) * N + ncol) * 4);
  const uint32_t svoff = (uint32_t)(ncol * 2), zvoff = (uint32_t)((ncol / NPK) * 4);
  const int zshift = BITS * (int)(ncol % NPK);
  const int bdst0 = (((bcol >> 5) * 4 + 2 * kwh) * 64 + (bcol & 31)) * 16;
  const int kwh_s = __builtin_amdgcn_readfirstlane(tid >> 8);
  // split-K (medium M: fewer tiles than CUs): this workgroup multiplies K-tiles [kbase, kbase + nk) and, when `partial`
  // is given, stores its fp32 tile into slab blockIdx.y; inc_woq_gemm's finalize kernel adds the slabs in a fixed order
  const int nk_all = (int)(K / TK);
  const int kbase = blockIdx.y * steps_per_split;
  const int nk = min(steps_per_split, nk_all - kbase);

  // one step's requests: 6 for W (4 packed words, scale, zero word) FIRST, then 4 x DMAs
  auto issue_w = [&](int kt, uint32_t (&w)[8], uint32_t& sb, uint32_t& zw) {
    if (NO_LD || NO_WLD) {
      asm volatile("" : "=v"(w[0]), "=v"(w[1]), "=v"(w[2]), "=v"(w[3]), "=v"(sb), "=v"(zw));
      return;
    }
    kt = FIXED_ADDR ? 0 : kbase + (kt > nk - 1 ? nk - 1 : kt);
    const uint32_t* wbase = qweight + (int64_t)kt * (TK / NPK) * N;
    const int64_t g = g_shift >= 0 ? (((int64_t)kt * TK + 32 * kwh_s) >> g_shift) : 0;  // wave-uniform (kwh is)
    const uint16_t* sbase = scales + g * N;
    const uint32_t* zbase = qzeros + g * NW;
    asm volatile(
        "s_nop 4\n\t"
        "global_load_dword %0, %6, %12\n\t"
        "global_load_dword %1, %7, %12\n\t"
        "global_load_dword %2, %8, %12\n\t"
        "global_load_dword %3, %9, %12\n\t"
        "global_load_ushort %4, %10, %13\n\t"
        "global_load_dword %5, %11, %14"
        : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(sb), "=&v"(zw)
        : "v"(wvoff[0]), "v"(wvoff[1]), "v"(wvoff[2]), "v"(wvoff[3]), "v"(svoff), "v"(zvoff), "s"(wbase), "s"(sbase), "s"(zbase)
        : "memory");
    if constexpr (BITS == 8)
      asm volatile(
          "global_load_dword %0, %4, %8\n\t"
          "global_load_dword %1, %5, %8\n\t"
          "global_load_dword %2, %6, %8\n\t"
          "global_load_dword %3, %7, %8"
          : "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7])
          : "v"(wvoff[4]), "v"(wvoff[5]), "v"(wvoff[6]), "v"(wvoff[7]), "s"(wbase)
          : "memory");
  };
  auto issue_dma = [&](int kt, int astage) {
    if (NO_LD || NO_DMA) return;
    kt = FIXED_ADDR ? 0 : kbase + (kt > nk - 1 ? nk - 1 : kt);
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + astage * T_ASTAGE + wave * 4096);
    lds_dma_4x1k(xtile + (int64_t)kt * TK, dst, avoff[0], avoff[1], avoff[2], avoff[3]);
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int a_row = wm * 128 + (lane & 31);
  const int a_sw = ((lane & 31) >> 1) & 7;
  const int a_hi = lane >> 5;
  const int b_off = (wn * 2 * 4 * 64 + lane) * 16;
  auto read_frags = [&](const char* As, const char* Bs, int kk, uint4 (&xa)[4], uint4 (&wbv)[2]) {
    const int chunk = ((2 * kk + a_hi) ^ a_sw) << 4;
#pragma unroll
    for (int mf = 0; mf < 4; ++mf) xa[mf] = *reinterpret_cast<const uint4*>(As + (a_row + 32 * mf) * 128 + chunk);
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) wbv[nf] = *reinterpret_cast<const uint4*>(Bs + (nf * 4 + kk) * 1024);
  };
  auto mma8 = [&](const uint4 (&xa)[4], const uint4 (&wbv)[2]) {
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
      for (int mf = 0; mf < 4; ++mf) acc[nf][mf] = mfma32<IS_BF16>(wbv[nf], xa[mf], acc[nf][mf]);
  };
  auto mma1 = [&](int i, const uint4 (&xa)[4], const uint4 (&wbv)[2]) {
    acc[i >> 2][i & 3] = mfma32<IS_BF16>(wbv[i >> 2], xa[i & 3], acc[i >> 2][i & 3]);
  };
#define INC_SB() __builtin_amdgcn_sched_barrier(0)
  // the kk-th 8-k fragment row of this thread's column: one 4-bit word, or two 8-bit words
  auto dequant_into = [&](int bstage, int kk, const uint32_t (&wd)[8], float sc, float nzs) {
    char* dst = Bbase + bstage * T_BSTAGE + bdst0;
    uint4 v;
    if constexpr (BITS == 4) {
      const uint32_t word = wd[kk];
      v = NO_DEQ ? make_uint4(word, word, word, word) : dequant8<IS_BF16>(word, sc, nzs);
    } else {
      v = dequant8_from_bytes<IS_BF16>(wd[2 * kk], wd[2 * kk + 1], sc, (int)nzs);  // 8-bit: `nzs` carries the zero point itself
    }
    if (NO_WR) asm volatile("" : : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
    else *reinterpret_cast<uint4*>(dst + ((kk >> 1) * 64 + 32 * (kk & 1)) * 16) = v;
  };
  auto group_params = [&](uint32_t sb, uint32_t zw, float& sc, float& nzs) {
    const float sc0 = f16_bits_to_f32((uint16_t)sb);
    constexpr uint32_t qmax = (1u << BITS) - 1u;
    uint32_t zz = ((zw >> zshift) & qmax) + 1u;  // modules.py:407-410
    zz = zz > qmax ? 0u : zz;
    nzs = BITS == 4 ? -(float)zz * sc0 : (float)zz;  // 8-bit: the zero point itself (the int8 wrap of q - z needs it as an integer)
    sc = BITS == 4 ? sc0 * inv_u : sc0;               // the 4-bit path converts through the fp8 decoder (q * 2^-9)
  };

  // ---- prologue: x stage 0 and W stage 0 complete; queue = [W words of tile 1 (6), DMA of x tile 1 (4)] ---------------
  uint32_t wa[8], wsa, wza;  // W register set A: tiles with ODD index (4-bit: entries 0..3 only)
  uint32_t wb_[8], wsb, wzb; // W register set B: tiles with EVEN index >= 2
  {
    uint32_t w0[8], s0, z0;
    issue_w(0, w0, s0, z0);
    issue_dma(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(w0[0]), "+v"(w0[1]), "+v"(w0[2]), "+v"(w0[3]), "+v"(s0), "+v"(z0) : : "memory");
    if constexpr (BITS == 8) asm volatile("" : "+v"(w0[4]), "+v"(w0[5]), "+v"(w0[6]), "+v"(w0[7]) : : "memory");
    float sc, nzs;
    group_params(s0, z0, sc, nzs);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) dequant_into(0, kk, w0, sc, nzs);
  }
  if (PP && wm) issue_w(1, wb_, wsb, wzb);  // the second half enters the loop one load segment later: sets swapped
  else issue_w(1, wa, wsa, wza);
  issue_dma(1, 1);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // Fragment registers: X and Y alternate over the four k16 groups of a step.  The LAST group of step t is multiplied
  // AFTER the barrier that ends the step, while the first fragments of step t+1 are already on their way from LDS and the
  // next loads are being issued: the matrix pipe has work during what used to be a ~90-instruction bubble per step.
  uint4 xX[4], wX[2], xY[4], wY[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) xY[i] = make_uint4(0u, 0u, 0u, 0u);  // "previous step's last group" of step 0: adds zeros
  wY[0] = wY[1] = make_uint4(0u, 0u, 0u, 0u);
  read_frags(Abase, Bbase + b_off, 0, xX, wX);

  // step t: x stage t%3, W stage t&1.  Issues W words of tile t+2 and the DMA of x tile t+2 (stage (t+2)%3), dequantises
  // tile t+1 (words issued in step t-1) into W stage (t+1)&1.
  // SCHED 3 ("ping-pong"): a step is split into a compute segment (32 MFMAs + their 24 fragment reads) and a load segment
  // (10 VMEM requests, dequantisation, 4 LDS writes) with a barrier after each.  The waves of the second half (wm = 1:
  // wave w+4 shares a SIMD with wave w) run one extra load segment before the loop, so from then on one wave of every SIMD
  // computes while its partner loads -- the two instruction streams never compete for the matrix pipe and the VMEM/VALU
  // issue of one hides behind the MFMAs of the other.  Half-step h: wm 0 computes tile h/2 at even h, wm 1 at odd h.  To
  // keep every tile complete one half-step before its first reader, wm 1 works one tile further ahead in its load segment
  // (loads tile t+3, dequantises tile t+2); stage numbers become wave-uniform run-time values, the loop body is one code.
  // counted waits: a step issues NVM = NWD + 2 + 4 requests (W words, scale, zero word, 4 x DMAs); "words" retires the
  // W-side requests of the PREVIOUS step (that step's 4 DMAs and this step's NVM stay in flight), "tile" retires the
  // previous step's DMAs (this step's NVM stay in flight)
  auto wait_words = [&](uint32_t (&dw)[8], uint32_t& ds, uint32_t& dz) {
    if (NO_VMWAIT) asm volatile("" : "+v"(dw[0]), "+v"(dw[1]), "+v"(dw[2]), "+v"(dw[3]), "+v"(ds), "+v"(dz) : : "memory");
    else if constexpr (BITS == 4) asm volatile("s_waitcnt vmcnt(14)" : "+v"(dw[0]), "+v"(dw[1]), "+v"(dw[2]), "+v"(dw[3]), "+v"(ds), "+v"(dz) : : "memory");
    else asm volatile("s_waitcnt vmcnt(18)" : "+v"(dw[0]), "+v"(dw[1]), "+v"(dw[2]), "+v"(dw[3]), "+v"(ds), "+v"(dz) : : "memory");
    if constexpr (BITS == 8) asm volatile("" : "+v"(dw[4]), "+v"(dw[5]), "+v"(dw[6]), "+v"(dw[7]) : : "memory");
  };
  auto wait_tile = [&]() {
    if (NO_VMWAIT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else if constexpr (BITS == 4) asm volatile("s_waitcnt vmcnt(10)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(14)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
  };
#define INC_3A2B_COMPUTE(As, Bs)                                                                                   \
  {                                                                                                                \
    if (!NO_RD) read_frags(As, Bs, 0, xX, wX);                                                                                 \
    if (!NO_RD) read_frags(As, Bs, 1, xY, wY);                                                                                 \
    INC_SB();                                                                                                      \
    mma8(xX, wX);                                                                                                  \
    INC_SB();                                                                                                      \
    if (!NO_RD) read_frags(As, Bs, 2, xX, wX);                                                                                 \
    INC_SB();                                                                                                      \
    mma8(xY, wY);                                                                                                  \
    INC_SB();                                                                                                      \
    if (!NO_RD) read_frags(As, Bs, 3, xY, wY);                                                                                 \
    INC_SB();                                                                                                      \
    mma8(xX, wX);                                                                                                  \
    INC_SB();                                                                                                      \
    mma8(xY, wY);                                                                                                  \
    INC_SB();                                                                                                      \
  }
#define INC_3A2B_LOADSEG(T, LW, LWS, LWZ, DW, DWS, DWZ)                                                           \
  {                                                                                                                \
    issue_w((T) + 2, LW, LWS, LWZ);                                                                                \
    issue_dma((T) + 2, ((T) + 2) % 3);                                                                             \
    wait_words(DW, DWS, DWZ);                                                                                      \
    float sc_, nzs_;                                                                                               \
    group_params(DWS, DWZ, sc_, nzs_);                                                                             \
    dequant_into(((T) & 1) ^ 1, 0, DW, sc_, nzs_);                                                                 \
    dequant_into(((T) & 1) ^ 1, 1, DW, sc_, nzs_);                                                                 \
    dequant_into(((T) & 1) ^ 1, 2, DW, sc_, nzs_);                                                                 \
    dequant_into(((T) & 1) ^ 1, 3, DW, sc_, nzs_);                                                                 \
    INC_SB();                                                                                                      \
  }
#define INC_3A2B_STEP3(T, LW, LWS, LWZ, DW, DWS, DWZ)                                                              \
  {                                                                                                                \
    const int t_ = (T);                                                                                            \
    const char* As = Abase + (t_ % 3) * T_ASTAGE;                                                                  \
    const char* Bs = Bbase + (t_ & 1) * T_BSTAGE + b_off;                                                          \
    INC_3A2B_COMPUTE(As, Bs)                                                                                       \
    __builtin_amdgcn_s_barrier();                                                                                  \
    INC_SB(); /* keep the load segment's address arithmetic on its own side of the barrier */                     \
    INC_3A2B_LOADSEG(t_ + wm, LW, LWS, LWZ, DW, DWS, DWZ)                                                          \
    wait_tile();                                                                                                   \
    __builtin_amdgcn_s_barrier();                                                                                  \
  }
#define INC_3A2B_STEP(T, LW, LWS, LWZ, DW, DWS, DWZ)                                                              \
  {                                                                                                                \
    const int t_ = (T);                                                                                            \
    const int as_ = t_ % 3, bs_ = t_ & 1;                                                                          \
    const char* As = Abase + as_ * T_ASTAGE;                                                                       \
    const char* Bs = Bbase + bs_ * T_BSTAGE + b_off;                                                               \
    issue_w(t_ + 2, LW, LWS, LWZ);                                                                                 \
    issue_dma(t_ + 2, (t_ + 2) % 3);                                                                               \
    if (SCHED != 0) INC_SB();                                                                                      \
    mma8(xY, wY);                       /* group 3 of the previous step */                                         \
    if (SCHED != 0) INC_SB();                                                                                      \
    if (!NO_RD) read_frags(As, Bs, 1, xY, wY);                                                                               \
    if (SCHED != 0) INC_SB();                                                                                           \
    wait_words(DW, DWS, DWZ);                                                                                      \
    float sc_, nzs_;                                                                                               \
    group_params(DWS, DWZ, sc_, nzs_);                                                                             \
    if (SCHED != 0) dequant_into(bs_ ^ 1, 0, DW, sc_, nzs_);                                                         \
    mma8(xX, wX);                       /* group 0 */                                                              \
    if (SCHED == 0) dequant_into(bs_ ^ 1, 0, DW, sc_, nzs_);                                                        \
    if (SCHED != 0) INC_SB();                                                                                           \
    if (!NO_RD) read_frags(As, Bs, 2, xX, wX);                                                                                 \
    if (SCHED != 0) INC_SB();                                                                                           \
    if (SCHED != 0) dequant_into(bs_ ^ 1, 1, DW, sc_, nzs_);                                                         \
    mma8(xY, wY);                       /* group 1 */                                                              \
    if (SCHED == 0) dequant_into(bs_ ^ 1, 1, DW, sc_, nzs_);                                                        \
    if (SCHED != 0) INC_SB();                                                                                           \
    if (!NO_RD) read_frags(As, Bs, 3, xY, wY);                                                                                 \
    if (SCHED != 0) INC_SB();                                                                                           \
    if (SCHED != 0) { dequant_into(bs_ ^ 1, 2, DW, sc_, nzs_); dequant_into(bs_ ^ 1, 3, DW, sc_, nzs_); }         \
    mma8(xX, wX);                       /* group 2 */                                                              \
    if (SCHED == 0) { dequant_into(bs_ ^ 1, 2, DW, sc_, nzs_); dequant_into(bs_ ^ 1, 3, DW, sc_, nzs_); }        \
    if (SCHED != 0) INC_SB();                                                                                           \
    wait_tile();                                                                                                   \
    if (!NO_BAR) __builtin_amdgcn_s_barrier();                                                                                  \
    if (!NO_RD) read_frags(Abase + ((t_ + 1) % 3) * T_ASTAGE, Bbase + (bs_ ^ 1) * T_BSTAGE + b_off, 0, xX, wX);               \
    if (SCHED != 0) INC_SB();                                                                                           \
  }
  if (PP) {
    if (wm) {  // load segment "-1": tile 2 -> x stage 2 / set A, tile 1 (set B) -> W stage 1
      INC_3A2B_LOADSEG(0, wa, wsa, wza, wb_, wsb, wzb)
      wait_tile();
      __builtin_amdgcn_s_barrier();
    }
    for (int t0 = 0; t0 < nk; t0 += 2) {
      INC_3A2B_STEP3(t0, wb_, wsb, wzb, wa, wsa, wza)
      INC_3A2B_STEP3(t0 + 1, wa, wsa, wza, wb_, wsb, wzb)
    }
    if (!wm) __builtin_amdgcn_s_barrier();  // the first half has executed one barrier fewer
  } else {
    for (int t0 = 0; t0 < nk; t0 += 2) {
      INC_3A2B_STEP(t0, wb_, wsb, wzb, wa, wsa, wza)        // even step: load tile t+2 (even) -> set B, dequantise tile t+1 (odd) <- set A
      INC_3A2B_STEP(t0 + 1, wa, wsa, wza, wb_, wsb, wzb)    // odd step: the reverse
    }
    mma8(xY, wY);  // group 3 of the last step
  }
#undef INC_3A2B_STEP
#undef INC_3A2B_STEP3
#undef INC_3A2B_COMPUTE
#undef INC_3A2B_LOADSEG
#undef INC_SB
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // epilogue: D row i = n-offset (r&3) + 8*(r>>2) + 4*(lane>>5), col j = m-offset lane&31
  if (partial) {  // split-K: raw fp32 tile into this split's slab (bias and conversion happen in the finalize kernel)
    float* slab = partial + (int64_t)blockIdx.y * M * N;
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int64_t nb = n0 + wn * 64 + nf * 32 + 8 * rq + 4 * (lane >> 5);
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
          const int64_t m = m0 + wm * 128 + mf * 32 + (lane & 31);
          if (m >= M) continue;
          float* dst = slab + m * N + nb;
          if (nb + 4 <= N && (N % 4) == 0) {
            *reinterpret_cast<float4*>(dst) = make_float4(acc[nf][mf][4 * rq + 0], acc[nf][mf][4 * rq + 1], acc[nf][mf][4 * rq + 2], acc[nf][mf][4 * rq + 3]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (nb + e < N) dst[e] = acc[nf][mf][4 * rq + e];
          }
        }
      }
    return;
  }
#pragma unroll
  for (int nf = 0; nf < 2; ++nf) {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int64_t nb = n0 + wn * 64 + nf * 32 + 8 * rq + 4 * (lane >> 5);
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      if (bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (nb + e < N) bv[e] = cvt16<IS_BF16>(bias[nb + e]);
      }
#pragma unroll
      for (int mf = 0; mf < 4; ++mf) {
        const int64_t m = m0 + wm * 128 + mf * 32 + (lane & 31);
        if (m >= M) continue;
        const float v0 = acc[nf][mf][4 * rq + 0] + bv[0], v1 = acc[nf][mf][4 * rq + 1] + bv[1];
        const float v2 = acc[nf][mf][4 * rq + 2] + bv[2], v3 = acc[nf][mf][4 * rq + 3] + bv[3];
        uint16_t* dst = y + m * N + nb;
        if (y_vec_ok && nb + 4 <= N) {
          *reinterpret_cast<uint2*>(dst) = make_uint2(cvt_pair<IS_BF16>(v0, v1), cvt_pair<IS_BF16>(v2, v3));
        } else {
          const float vv[4] = {v0, v1, v2, v3};
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (nb + e < N) dst[e] = IS_BF16 ? f32_to_bf16_bits(vv[e]) : f32_to_f16_bits(vv[e]);
        }
      }
    }
  }
}

// =============================================================================================
// large-M path "PC" (4-bit, K % 128 == 0): the 3A2B tile with PRODUCER / CONSUMER wave specialisation
// =============================================================================================
// The 3A2B kernel above is issue-bound, not matrix-pipe-bound (profiles/r1_pmc: MFMA busy 0.49, 52 % of wave time
// issue-stalled): each of its 8 waves carries ~200 non-MFMA instructions per 32 MFMAs (10 VMEM requests, ~70 VALU of
// dequantisation, LDS writes, counted waits), and with two such waves per SIMD nothing is left to hide anything.  Here the
// SAME tile (256 x 256 x 64, three x stages + two dequantised-W stages = 160 KiB) is worked by 12 waves in two roles:
//   waves 0..7  CONSUMERS (2 x 4, 128 x 64 of the tile each, two per SIMD): per K-step 24 fragment ds_read_b128, 32 MFMAs and
//               the wave's share of the x tile's LDS-DMA (4 x 1 KiB, issued two steps ahead between the MFMA groups, retired
//               with a counted vmcnt before the step's barrier) -- no VGPR loads, no dequantisation;
//   waves 8..11 PRODUCERS (one per SIMD): the weight side.  Per wave and step 8 packed-word loads + scale + zero word of ONE
//               column per lane (two steps ahead), the int4 -> bf16 arithmetic (fp8-decoder trick of dequant8, bit-identical
//               to inc_woq_dequant) and 8 fragment-order ds_write_b128, each issued right behind the 15 VALU that produce it.
// tools/kbench pcablate (profiles/r2b): with everything on the producers their serial chain (0.79 us per step at full clock:
// x DMA issue 0.19, W loads 0.04, arithmetic 0.17, LDS writes 0.37) is as long as the MFMA work of the step and -- at the
// ~1.7 GHz the chip sustains under MFMA load -- longer; the consumers alone (reads + MFMA + barrier) run at 1.59 PFLOP/s.
// One s_barrier per K-step for all 12 waves: step t multiplies x stage t % 3 and W stage t & 1 while the producers dequantise
// tile t+1 into W stage (t+1) & 1 (read last in step t-1) and the DMA of x tile t+2 fills stage (t+2) % 3 (read last in
// step t-1).  A wave's 128 fp32 accumulators + one fragment set fit the 168-register budget of three waves per SIMD.
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#ifdef INC_KBENCH  // superseded by the direct-to-register kernel (gemm_d2r.hip); kept in the harness build as its bitwise A/B partner (tools/kbench d2r)
#include "../../tools/kbench_gemm_1.inc"
#endif  // INC_KBENCH

template <bool IS_BF16>
__global__ void splitk_slab_reduce_kernel(const float* __restrict__ partial, const uint16_t* __restrict__ bias,
                                          uint16_t* __restrict__ y, int64_t M, int64_t N, int splits) {
  const int64_t total4 = M * N / 4;  // N % 4 == 0 on this path
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 v = reinterpret_cast<const float4*>(partial)[i];
    for (int z = 1; z < splits; ++z) {  // fixed order: deterministic
      const float4 p = reinterpret_cast<const float4*>(partial + (int64_t)z * M * N)[i];
      v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    if (bias) {
      const int64_t n = (i * 4) % N;
      v.x += cvt16<IS_BF16>(bias[n]); v.y += cvt16<IS_BF16>(bias[n + 1]); v.z += cvt16<IS_BF16>(bias[n + 2]); v.w += cvt16<IS_BF16>(bias[n + 3]);
    }
    reinterpret_cast<uint2*>(y)[i] = make_uint2(cvt_pair<IS_BF16>(v.x, v.y), cvt_pair<IS_BF16>(v.z, v.w));
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
        if (g_idx && ncol + c < N && kwr < KW) dequant_word_gidx<4, IS_BF16>(w4[c], scales, qzeros, g_idx, kk, K, ncol + c, N, NW, d);
        else dequant_word<4, IS_BF16>(w4[c], gq, d);
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
            if (g_idx) dequant_word_gidx<8, IS_BF16>(word, scales, qzeros, g_idx, kk, K, ncol + c, N, NW, dd);
            else dequant_word<8, IS_BF16>(word, gq, dd);
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

// =============================================================================================
// small-M fast path (4-bit, M <= 16, K % 32 == 0, N % 4 == 0, group lookup by shift): HBM-bound
// =============================================================================================
// The whole packed matrix is only N*K/2 bytes (8 MiB at 4096^2), a few microseconds of HBM time, so the
// kernel is built around memory-level parallelism: a wave owns 64 columns x 256 k and issues ALL of its
// weight traffic (8 x 16 B per lane = 8 KiB per wave) before it touches any of it; 4 waves of a workgroup
// take 4 consecutive k-ranges of the same 64-column strip (1024 k), grid = strips x K/1024 workgroups
// (256 at 4096^2: one per CU).  x is the 16-row A operand of v_mfma_f32_16x16x32 (rows >= M zeroed), each
// lane's 16-byte weight load is 4 adjacent columns = 4 B operands (output column 4*(lane&15)+c).
// Reduction: the 4 waves add through LDS; across workgroups each writes its fp32 partial strip, then the
// LAST workgroup to arrive on the strip's counter (agent-scope release / acquire, cdna_hip_programming.md
// Guideline 16) sums the partials in a fixed order, adds the bias, converts and stores -> one launch,
// deterministic.  The counters live in the caller's workspace, must be zero on entry and are returned to
// zero by the last arriver.
// Hand-off of the split-K partials (MI355X_MICROARCH.md, "Valid forms besides R1/R2"): write-through (`sc1`) partial stores ->
// every wave drains them (`s_waitcnt vmcnt(0)`) -> barrier -> ONE relaxed agent-scope ticket; the last arriver reads the slabs
// with `sc1` loads (L1-bypassing), so neither side needs an agent-scope fence (the release / acquire pair this replaces cost
// ~3.4 us of an 8.4 us kernel).  VSTEPS = MFMA K=32 steps per wave: 8 (a wave streams 8 KiB of weights) for large matrices,
// 4 when that would leave CUs without a workgroup or SIMDs with a single wave (the dequantisation arithmetic of a wave is a
// serial ~60-instruction chain per step).
constexpr int VS = 8;   // largest VSTEPS (sizes the workspace)
// MB = 16-row blocks of x per workgroup (M <= 16 * MB): batched decode (16 < M <= 64) streams the packed weights ONCE like the
// M <= 16 case -- every dequantised B fragment feeds MB MFMAs -- instead of parking a 256-row tile that is mostly clamped rows.
// NT (harness A/B, same results): the packed-weight requests carry the non-temporal hint -- every word is read once by one CU
// The body of one (64-column strip, K-slice) workgroup: `counter` is the strip's arrival counter, `partial` the module's slabs.
// BITS = 8 (weight-only INT8, round 6): a step's 32 k of four columns are TWO packed rows per lane (a word = 4 k of one column), the
// integer -> float step is the int8 wrap of dequant8_from_bytes (bit-identical to inc_woq_dequant); always 4 steps per wave, so a wave
// streams the same 8 KiB as the 4-bit form with 8 steps.  `NW` = words per row of qzeros (N / 8 for 4 bits, N / 4 for 8).
template <bool IS_BF16, bool G128, int VSTEPS, int MB, bool NT = false, int BITS = 4>
__device__ __forceinline__ void woq_gemv_w4_body(
    const uint16_t* __restrict__ x, const uint32_t* __restrict__ qweight, const uint16_t* __restrict__ scales,
    const uint32_t* __restrict__ qzeros, const uint16_t* __restrict__ bias, uint16_t* __restrict__ y,
    float* __restrict__ partial, unsigned* __restrict__ counter, int M, int64_t N, int64_t K, int64_t NW,
    int g_shift, int splitk, int strip, int slice) {
  constexpr int VS = VSTEPS;  // shadows the file-level maximum inside this kernel
  constexpr int ROWS = 16 * MB;
  constexpr int NOUT = ROWS * 64 / 256;  // outputs per thread of the strip
  __shared__ float red[4 * ROWS * 65];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float inv_u = fp8_unit_inverse();
  const int jn = lane & 15, oct = lane >> 4;
  const int64_t n0 = (int64_t)strip * 64;
  int64_t ncol = n0 + 4 * jn;
  if (ncol > N - 4) ncol = N - 4;  // clamped lanes recompute valid columns; their results are not stored
  const int steps_total = (int)(K / 32);
  const int step0 = (slice * 4 + wave) * VS;

  // ---- issue every load of this wave up front ---------------------------------------------------
  constexpr int WPS = BITS == 8 ? 2 : 1;  // 16-byte weight requests per lane and step
  static_assert(BITS == 4 || (BITS == 8 && !NT), "4- or 8-bit words");
  uint4 w[VS * WPS], a[MB][VS];
#pragma unroll
  for (int s = 0; s < VS; ++s) {
    int st = step0 + s;
    if (st > steps_total - 1) st = steps_total - 1;  // past-the-end steps re-read the last one and are zeroed via A
    if constexpr (BITS == 8) {
      w[2 * s] = *reinterpret_cast<const uint4*>(qweight + ((int64_t)st * 8 + 2 * oct) * N + ncol);
      w[2 * s + 1] = *reinterpret_cast<const uint4*>(qweight + ((int64_t)st * 8 + 2 * oct + 1) * N + ncol);
    } else if constexpr (NT) {
      typedef uint32_t nt_u32x4 __attribute__((ext_vector_type(4)));
      const nt_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_u32x4*>(qweight + ((int64_t)st * 4 + oct) * N + ncol));
      w[s] = make_uint4(v.x, v.y, v.z, v.w);
    } else {
      w[s] = *reinterpret_cast<const uint4*>(qweight + ((int64_t)st * 4 + oct) * N + ncol);
    }
#pragma unroll
    for (int b = 0; b < MB; ++b) {
      const int am = 16 * b + jn < M ? 16 * b + jn : M - 1;  // A row (clamped; rows >= M are zeroed below)
      a[b][s] = *reinterpret_cast<const uint4*>(x + (int64_t)am * K + (int64_t)st * 32 + 8 * oct);
    }
  }
  // group parameters: G128 -> one group per 4 steps (step0 is a multiple of 4)
  constexpr int NG = G128 ? (VS + 3) / 4 : VS;
  uint2 sraw[NG];
  uint32_t zraw[NG];
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    int st = step0 + (G128 ? 4 * i : i);
    if (st > steps_total - 1) st = steps_total - 1;
    const int64_t g = g_shift >= 0 ? (((int64_t)st * 32) >> g_shift) : 0;
    sraw[i] = *reinterpret_cast<const uint2*>(scales + g * N + ncol);
    zraw[i] = qzeros[g * NW + (BITS == 8 ? (ncol >> 2) : (ncol >> 3))];
  }
  const int zsh = 4 * (int)(ncol & 7);  // ncol % 4 == 0: the 4 zero nibbles sit at bits zsh .. zsh+15
  // this thread's outputs of the strip: idx = tid + 256*i -> row idx>>6, column idx&63; bias fetched now
  uint16_t braw[NOUT];
  const uint16_t* const bsrc = bias ? bias : scales;  // always a valid address: the loads stay unconditional
  bool out_ok[NOUT];
  int64_t out_off[NOUT];
#pragma unroll
  for (int i = 0; i < NOUT; ++i) {
    const int idx = tid + 256 * i, m = idx >> 6, c = idx & 63;
    out_ok[i] = m < M && n0 + c < N;
    out_off[i] = out_ok[i] ? (int64_t)m * N + n0 + c : 0;
    braw[i] = bsrc[out_ok[i] ? n0 + c : 0];
  }
  __builtin_amdgcn_sched_barrier(0);  // everything above is in flight before the first use below

  f32x4 acc[MB][4];
#pragma unroll
  for (int b = 0; b < MB; ++b)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[b][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < VS; ++s) {
    const int gi = G128 ? (s >> 2) : s;
    const bool in_k = step0 + s < steps_total;
    uint4 av[MB];
#pragma unroll
    for (int b = 0; b < MB; ++b) {
      const bool live = in_k && (16 * b + jn < M);
      av[b] = a[b][s];
      av[b].x = live ? av[b].x : 0u; av[b].y = live ? av[b].y : 0u; av[b].z = live ? av[b].z : 0u; av[b].w = live ? av[b].w : 0u;
    }
    const uint32_t sw[2] = {sraw[gi].x, sraw[gi].y};
    const uint32_t ww[4] = {w[s * WPS].x, w[s * WPS].y, w[s * WPS].z, w[s * WPS].w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float sc = f16_bits_to_f32((uint16_t)(sw[c >> 1] >> (16 * (c & 1))));
      if constexpr (BITS == 8) {
        const uint32_t wh[4] = {w[s * WPS + WPS - 1].x, w[s * WPS + WPS - 1].y, w[s * WPS + WPS - 1].z, w[s * WPS + WPS - 1].w};
        uint32_t z8 = ((zraw[gi] >> (8 * c)) & 255u) + 1u;  // ncol % 4 == 0: the word holds exactly this lane's four zero points
        z8 = z8 > 255u ? 0u : z8;
        const uint4 bq8 = dequant8_from_bytes<IS_BF16>(ww[c], wh[c], sc, (int)z8);
#pragma unroll
        for (int b = 0; b < MB; ++b) acc[b][c] = mfma16<IS_BF16>(av[b], bq8, acc[b][c]);
        continue;
      }
      uint32_t zz = ((zraw[gi] >> (zsh + 4 * c)) & 15u) + 1u;
      zz = zz > 15u ? 0u : zz;
      const uint4 bq = dequant8<IS_BF16, 1>(ww[c], sc * inv_u, -(float)zz * sc);  // (packed fp32 FMAs: same values; with four MFMAs per step the VALU is the busy pipe here: - 5 % per launch, tools/gemv_lab)
#pragma unroll
      for (int b = 0; b < MB; ++b) acc[b][c] = mfma16<IS_BF16>(av[b], bq, acc[b][c]);
    }
  }
  // ---- reduce the 4 waves: D col = lane&15 -> column 4*jn + c, row m = 16*b + 4*oct + r ------------------
#pragma unroll
  for (int b = 0; b < MB; ++b)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * ROWS + 16 * b + 4 * oct + r) * 65 + 4 * jn + c] = acc[b][c][r];
  __syncthreads();
  float sum[NOUT];
#pragma unroll
  for (int i = 0; i < NOUT; ++i) {
    const int idx = tid + 256 * i, m = idx >> 6, c = idx & 63;
    sum[i] = red[(0 * ROWS + m) * 65 + c] + red[(1 * ROWS + m) * 65 + c] + red[(2 * ROWS + m) * 65 + c] + red[(3 * ROWS + m) * 65 + c];
  }
  if (splitk > 1) {
    const int64_t slab = (int64_t)M * N;
#pragma unroll
    for (int i = 0; i < NOUT; ++i)
      if (out_ok[i]) __hip_atomic_store(&partial[(int64_t)slice * slab + out_off[i]], sum[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1
    // publish: every wave drains its write-through stores, then one relaxed agent-scope ticket from lane 0
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const unsigned ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool last = ticket == (unsigned)(splitk - 1);
      if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm for the next call
      red[0] = last ? 1.f : 0.f;
    }
    __syncthreads();
    if (red[0] == 0.f) return;
    // last arriver: fixed-order sum over the slices, up to 32 partial loads of this thread in flight at a time
#pragma unroll
    for (int i = 0; i < NOUT; ++i) sum[i] = 0.f;
    constexpr int SB = 32 / NOUT;  // slices per batch (8 for M <= 16: one L2 round trip for up to 8 slices)
    for (int sl0 = 0; sl0 < splitk; sl0 += SB) {
      float pv[SB][NOUT];
#pragma unroll
      for (int d = 0; d < SB; ++d) {
        const int sl = sl0 + d < splitk ? sl0 + d : splitk - 1;
#pragma unroll
        for (int i = 0; i < NOUT; ++i) pv[d][i] = __hip_atomic_load(&partial[(int64_t)sl * slab + out_off[i]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1
      }
#pragma unroll
      for (int d = 0; d < SB; ++d)
#pragma unroll
        for (int i = 0; i < NOUT; ++i) sum[i] += (sl0 + d < splitk) ? pv[d][i] : 0.f;
    }
  }
#pragma unroll
  for (int i = 0; i < NOUT; ++i)
    if (out_ok[i]) {
      const float v = sum[i] + (bias ? cvt16<IS_BF16>(braw[i]) : 0.f);
      y[out_off[i]] = IS_BF16 ? f32_to_bf16_bits(v) : f32_to_f16_bits(v);
    }
}

template <bool IS_BF16, bool G128, int VSTEPS, int MB, bool NT = false, int BITS = 4>
__global__ __launch_bounds__(256) void woq_gemv_w4_kernel(
    const uint16_t* __restrict__ x, const uint32_t* __restrict__ qweight, const uint16_t* __restrict__ scales,
    const uint32_t* __restrict__ qzeros, const uint16_t* __restrict__ bias, uint16_t* __restrict__ y,
    float* __restrict__ partial, unsigned* __restrict__ counters, int M, int64_t N, int64_t K, int64_t NW,
    int64_t G, int g_shift, int splitk) {
  woq_gemv_w4_body<IS_BF16, G128, VSTEPS, MB, NT, BITS>(x, qweight, scales, qzeros, bias, y, partial, counters + blockIdx.x, M, N, K, NW, g_shift, splitk,
                                                 (int)blockIdx.x, (int)blockIdx.y);
}

// Several packed modules that multiply the SAME x (q / k / v of an attention block; gate / up of an MLP) in ONE launch
// (inc_woq_gemm_multi): a decode call of one module is ~2 us of streaming behind ~5 us of launch boundary, first-byte latency and
// split-K hand-off, and the modules of a group are independent given x.  The strips of the modules occupy consecutive ranges of
// blockIdx.x; every strip runs exactly the body above on its own module's tensors -> bit-identical to the single launches.
template <bool IS_BF16, bool G128, int VSTEPS, int MB, int BITS = 4>
__global__ __launch_bounds__(256) void woq_gemv_w4_multi_kernel(GemvBatch args, const uint16_t* __restrict__ x, float* __restrict__ partial,
                                                                unsigned* __restrict__ counters, int M, int64_t K, int g_shift, int splitk) {
  const int b = (int)blockIdx.x;
  int p = 0;
#pragma unroll
  for (int i = 1; i < GEMV_MAX_BATCH; ++i)
    if (i < args.n && b >= args.first[i]) p = i;
  p = __builtin_amdgcn_readfirstlane(p);
  const int64_t N = args.N[p];
  woq_gemv_w4_body<IS_BF16, G128, VSTEPS, MB, false, BITS>(x, args.qweight[p], args.scales[p], args.qzeros[p], args.bias[p], args.y[p],
                                                           partial + args.part_off[p], counters + b, M, N, K, BITS == 8 ? (N + 3) / 4 : (N + 7) / 8, g_shift,
                                                           splitk, b - args.first[p], (int)blockIdx.y);
}

// =============================================================================================
// mid-M strip kernel (round 2): 64 < M <= STRIP_MAX_M
// =============================================================================================
// The 256 x 256 tile needs split-K over 8-16 fp32 slabs to put such a problem on 256 CUs (M = 512, 4096^2: 32 tiles, 134 MB of
// slab traffic, 62 us; M = 128: 30 us).  This kernel extends the streaming kernel instead: a workgroup owns 64 rows x 128
// columns for the WHOLE of K (or 1/splitk of it when rows x columns alone leave CUs idle), its eight waves take an eighth of
// the K-steps each, and a wave's packed-weight loads ARE its MFMA B fragments (a lane's uint4 = 8 consecutive k of 4 adjacent
// columns), dequantised in registers and fed to four 16x16x32 MFMAs each; the waves' accumulators meet in LDS at the end (64 x 64
// outputs per pass).  Split-K (only when needed, <= 4 slices) hands over like the streaming kernel: write-through partials, one
// relaxed ticket, the last arriver sums in slice order -> deterministic.
constexpr int64_t STRIP_MAX_M = 1024;
constexpr int STRIP_WAVES = 8;                                          // waves per workgroup: eighths of the K range
constexpr int STRIP_RING = 3;                                           // operand slots (K-steps in flight) per wave
constexpr int STRIP_SMEM_BYTES = STRIP_WAVES * STRIP_RING * 6 * 1024;   // 144 KiB: the operand rings; the epilogue reuses them
static_assert(STRIP_SMEM_BYTES >= STRIP_WAVES * 64 * 68 * 4, "the reduction buffer (WAVES x 64 x 68 fp32) aliases the rings");

// ABL (harness build only, timing-only, WRONG results): bit 0 no x requests, bit 1 no W requests, bit 2 no MFMA / dequantisation
template <bool IS_BF16, int WAVES, int RING, int ABL = 0>
__global__ __launch_bounds__(64 * WAVES) void woq_gemm_w4_strip_kernel(
    const uint16_t* __restrict__ x, const uint32_t* __restrict__ qweight, const uint16_t* __restrict__ scales,
    const uint32_t* __restrict__ qzeros, const uint16_t* __restrict__ bias, uint16_t* __restrict__ y,
    float* __restrict__ partial, unsigned* __restrict__ counters, int M, int64_t N, int64_t K, int64_t NW, int g_shift, int splitk) {
  constexpr int MB = 4, NB = 2;  // 16-row blocks and 64-column groups of a wave: the operand requests below are written out for these
  constexpr int ROWS = 16 * MB, COLS = 64 * NB, NT = 64 * WAVES;
  extern __shared__ __attribute__((aligned(16))) char strip_smem[];
  float* const red = reinterpret_cast<float*>(strip_smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int jn = lane & 15, oct = lane >> 4;
  const float inv_u = fp8_unit_inverse();
  // XCD-aware tile order: workgroup L (dispatch order: x fastest, then y) runs on XCD L % 8.  The tiles are renumbered so that an
  // XCD owns a CONTIGUOUS range of (row strip, column strip) pairs, row strip major: the 32 column strips of one 64-row strip
  // share that strip's x rows (512 KiB at K = 4096) out of ONE XCD's L2 instead of every XCD streaming the whole of x (4 MiB at
  // M = 512, the size of an L2) from the Infinity Cache.
  int bx = (int)blockIdx.x, by = (int)blockIdx.y;
  {
    const int nx = (int)gridDim.x, nt = nx * (int)gridDim.y, L = by * nx + bx;
    const int q = nt / 8, r = nt % 8, xcd = L % 8, idx = L / 8;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;  // bijective (same form as the d2r kernel's)
    by = t / nx;
    bx = t - by * nx;
  }
  const int64_t n0 = (int64_t)bx * COLS;
  const int m0 = by * ROWS;
  const int slice = blockIdx.z;

  int64_t ncol[NB];
  int zsh[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    ncol[nb] = n0 + 64 * nb + 4 * jn;
    if (ncol[nb] > N - 4) ncol[nb] = N - 4;  // clamped lanes recompute valid columns; their results are not stored
    zsh[nb] = 4 * (int)(ncol[nb] & 7);
  }
  // this wave's K-steps (32 k each): an even share of the slice-and-wave grid
  const int steps_total = (int)(K / 32);
  const int Q = WAVES * splitk, q = slice * WAVES + wave;
  const int lo = (int)((int64_t)steps_total * q / Q), hi = (int)((int64_t)steps_total * (q + 1) / Q);

  // Operand pipeline.  hipcc sinks plain loads to their first use, and a wave then pays the full memory latency in every step
  // (measured: 1.4 us per step); pinned register prefetch one step ahead is all the register file allows next to 128
  // accumulators at two waves per SIMD, and was still latency-bound (a third of the wave time issuing).  So the fragments travel
  // through LDS without touching a register: every wave owns a ring of RING slots of 6 KiB; a step's four x fragments and two
  // weight fragments arrive by six LDS-DMA requests (lane-linear image = the fragment layout), RING steps ahead of the MFMAs, and
  // are picked up with six conflict-free ds_read_b128.  The raw group parameters (4 small requests per step) stay register
  // loads in the same in-order queue: one step = 10 requests.
  struct Step {  // x and weight fragments of one K-step
    uint4 a[4];
    uint4 w[2];
  };
  struct Par {  // raw scales / zero-point words of the step's group
    u32x2 s[2];
    uint32_t z[2];
  };
  constexpr int SLOT = 6 * 1024;
  const uint32_t ring0 = (uint32_t)(uintptr_t)strip_smem + (uint32_t)wave * (RING * SLOT);
  uint32_t xoff[4], woff[2], soff[2], zoff[2];
  // x fragments: FOUR ADJACENT LANES fetch the 64 contiguous bytes (32 k) of one row -- the memory pipeline coalesces adjacent
  // lanes only; with the MFMA operand's own lane order (adjacent lanes = adjacent rows, 8 KiB apart) every lane is its own
  // 16-byte request and the workgroup gets ~11 bytes per clock (measured).  Lane l lands at byte 16 l of the slot and carries row
  // l >> 2, 16-byte chunk (l & 3) ^ (row >> 2): the XOR makes the pick-up below (lane (jn, oct) reads row jn, chunk oct)
  // conflict-free.
  const int xr = lane >> 2, xc = (lane & 3) ^ (xr >> 2);
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    int am = m0 + 16 * b + xr;
    if (am > M - 1) am = M - 1;  // rows past M are computed from a valid row and never stored
    xoff[b] = (uint32_t)(((int64_t)am * K + 8 * xc) * 2);
  }
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    woff[nb] = (uint32_t)(((int64_t)oct * N + ncol[nb]) * 4);
    soff[nb] = (uint32_t)(ncol[nb] * 2);
    zoff[nb] = (uint32_t)((ncol[nb] >> 3) * 4);
  }
  auto issue = [&](int slot, Par& p, int st) {
    st = st > hi - 1 ? hi - 1 : st;  // the prefetch past the end re-reads the last step
    const int64_t g = g_shift >= 0 ? (((int64_t)st * 32) >> g_shift) : 0;
    const uint16_t* xb = x + (int64_t)st * 32;
    const uint32_t* wb = qweight + (int64_t)st * 4 * N;
    const uint16_t* sb = scales + g * N;
    const uint32_t* zb = qzeros + g * NW;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(ring0 + (uint32_t)slot * SLOT);
    uint32_t keep;
    if constexpr (ABL & 3) {  // timing-only: the same ten-request step with some requests left out (vmcnt bookkeeping is by count, so
                              // every omitted request is replaced by a 4-byte load of the parameter words)
      asm volatile("s_mov_b32 %0, m0" : "=&s"(keep));
#define INC_STRIP_REQ(COND, OFF, BASE, LDSOFF)                                                                                  \
  if constexpr (COND) asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(OFF), "s"(BASE), "s"(dst), "i"(LDSOFF) : "memory", "scc"); \
  else asm volatile("global_load_dword %0, %1, %2" : "=&v"(p.z[0]) : "v"(zoff[0]), "s"(zb) : "memory");
      INC_STRIP_REQ((ABL & 1) == 0, xoff[0], xb, 0x0)
      INC_STRIP_REQ((ABL & 1) == 0, xoff[1], xb, 0x400)
      INC_STRIP_REQ((ABL & 1) == 0, xoff[2], xb, 0x800)
      INC_STRIP_REQ((ABL & 1) == 0, xoff[3], xb, 0xc00)
      INC_STRIP_REQ((ABL & 2) == 0, woff[0], wb, 0x1000)
      INC_STRIP_REQ((ABL & 2) == 0, woff[1], wb, 0x1400)
#undef INC_STRIP_REQ
      asm volatile("global_load_dwordx2 %0, %4, %6\n\tglobal_load_dwordx2 %1, %5, %6\n\tglobal_load_dword %2, %7, %9\n\tglobal_load_dword %3, %8, %9\n\ts_mov_b32 m0, %10"
                   : "=&v"(p.s[0]), "=&v"(p.s[1]), "=&v"(p.z[0]), "=&v"(p.z[1])
                   : "v"(soff[0]), "v"(soff[1]), "s"(sb), "v"(zoff[0]), "v"(zoff[1]), "s"(zb), "s"(keep)
                   : "memory");
      return;
    }
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_nop 4\n\t"
        "s_mov_b32 m0, %19\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %5, %15\n\t"
        "s_add_u32 m0, %19, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %6, %15\n\t"
        "s_add_u32 m0, %19, 0x800\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %7, %15\n\t"
        "s_add_u32 m0, %19, 0xc00\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %8, %15\n\t"
        "s_add_u32 m0, %19, 0x1000\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %9, %16\n\t"
        "s_add_u32 m0, %19, 0x1400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %10, %16\n\t"
        "global_load_dwordx2 %1, %11, %17\n\t"
        "global_load_dwordx2 %2, %12, %17\n\t"
        "global_load_dword %3, %13, %18\n\t"
        "global_load_dword %4, %14, %18\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep), "=&v"(p.s[0]), "=&v"(p.s[1]), "=&v"(p.z[0]), "=&v"(p.z[1])
        : "v"(xoff[0]), "v"(xoff[1]), "v"(xoff[2]), "v"(xoff[3]), "v"(woff[0]), "v"(woff[1]), "v"(soff[0]), "v"(soff[1]), "v"(zoff[0]),
          "v"(zoff[1]), "s"(xb), "s"(wb), "s"(sb), "s"(zb), "s"(dst)
        : "memory", "scc");
  };
  // the oldest of RING steps in flight has landed in its slot / its parameter registers (the younger ones stay in flight)
  auto landed = [&](Par& p) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(p.s[0]), "+v"(p.s[1]), "+v"(p.z[0]), "+v"(p.z[1]) : "i"(10 * (RING - 1)) : "memory");
  };
  auto fetch_issue = [&](Step& t, int slot) {
    const char* base = strip_smem + wave * (RING * SLOT) + slot * SLOT;
    const int apos = (4 * jn + (oct ^ (jn >> 2))) * 16;  // where row jn, chunk oct of an x fragment landed
#pragma unroll
    for (int b = 0; b < 4; ++b) t.a[b] = *reinterpret_cast<const uint4*>(base + b * 1024 + apos);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) t.w[nb] = *reinterpret_cast<const uint4*>(base + 4096 + nb * 1024 + lane * 16);
  };
  auto fetch = [&](Step& t, int slot) {
    fetch_issue(t, slot);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slot is free for the next DMA once these have returned
  };

  f32x4 acc[MB][4 * NB];
#pragma unroll
  for (int b = 0; b < MB; ++b)
#pragma unroll
    for (int c = 0; c < 4 * NB; ++c) acc[b][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  // scale / zero point of this lane's 8 columns, refreshed when the K-step enters a new group (every 2^(g_shift-5) steps)
  float scu[4 * NB], nzs[4 * NB];
  const int gmask = g_shift < 0 ? 0x7fffffff : ((1 << (g_shift - 5)) - 1);
  auto refresh = [&](const Par& p, int st) {
    if (st == lo || (st & gmask) == 0) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float sc = f16_bits_to_f32((uint16_t)(p.s[nb][c >> 1] >> (16 * (c & 1))));
          uint32_t zz = ((p.z[nb] >> (zsh[nb] + 4 * c)) & 15u) + 1u;  // modules.py:407-410 (stored zp - 1; wraps above 15)
          zz = zz > 15u ? 0u : zz;
          scu[4 * nb + c] = sc * inv_u;
          nzs[4 * nb + c] = -(float)zz * sc;
        }
    }
  };
  auto compute = [&](const Step& t) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const uint32_t ww[4] = {t.w[nb].x, t.w[nb].y, t.w[nb].z, t.w[nb].w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint4 bq = dequant8<IS_BF16>(ww[c], scu[4 * nb + c], nzs[4 * nb + c]);
#pragma unroll
        for (int b = 0; b < MB; ++b) acc[b][4 * nb + c] = mfma16<IS_BF16>(t.a[b], bq, acc[b][4 * nb + c]);
      }
    }
  };

  static_assert(RING >= 2 && 10 * (RING - 1) < 64, "vmcnt counts 63 requests at most");
  // ABL bit 3 (harness A/B, CORRECT results): the LDS reads of step s + 1 are requested before the MFMAs of step s (two fragment
  // sets) instead of each step waiting for its own reads in front of its MFMAs; bit 4: s_setprio 1 around a step's dequantise + MFMA
  if constexpr ((ABL & 8) != 0) {
    if (lo < hi) {
      Par p[RING];
      Step t[2];
#pragma unroll
      for (int r = 0; r < RING; ++r) issue(r, p[r], lo + r);
      landed(p[0]);
      fetch_issue(t[0], 0);
      for (int st = lo; st < hi; st += 2 * RING) {  // two rounds of the ring per iteration: the fragment set index stays a constant
#pragma unroll
        for (int rr = 0; rr < 2 * RING; ++rr) {
          constexpr int dummy = 0;
          (void)dummy;
          const int r = rr % RING;
          if (st + rr < hi) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // t[rr & 1] has arrived: slot r is free
            refresh(p[r], st + rr);
            issue(r, p[r], st + rr + RING);
            if (st + rr + 1 < hi) {
              landed(p[(r + 1) % RING]);
              fetch_issue(t[(rr + 1) & 1], (r + 1) % RING);
            }
            if constexpr ((ABL & 16) != 0) __builtin_amdgcn_s_setprio(1);
            compute(t[rr & 1]);
            if constexpr ((ABL & 16) != 0) __builtin_amdgcn_s_setprio(0);
          }
        }
      }
    }
  } else if (lo < hi) {
    Par p[RING];  // (indexed by unrolled constants only: registers)
    Step t;
#pragma unroll
    for (int r = 0; r < RING; ++r) issue(r, p[r], lo + r);
    for (int st = lo; st < hi; st += RING) {
#pragma unroll
      for (int r = 0; r < RING; ++r) {
        if (st + r < hi) {
          landed(p[r]);
          fetch(t, r);
          refresh(p[r], st + r);
          issue(r, p[r], st + r + RING);
          if constexpr ((ABL & 16) != 0) __builtin_amdgcn_s_setprio(1);
          if constexpr ((ABL & 4) == 0) compute(t);
          if constexpr ((ABL & 16) != 0) __builtin_amdgcn_s_setprio(0);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped prefetches past the end
  __syncthreads();                                   // every wave's ring is dead: the reduction buffer takes their place

  // ---- the waves' accumulators meet in LDS, 64 rows x 64 columns per pass ---------------------------------------------
  // D of an MFMA: column = lane & 15 -> tile column 4*jn + c, row = 4*oct + r.  Every thread sums and stores FOUR adjacent columns at a
  // time: 16-byte LDS reads, 16-byte write-through partial stores or one 8-byte store of four outputs (round 6: the single-float form of
  // this epilogue was about a third of a mid-M launch, tools/midm_lab)
  const int64_t slab = (int64_t)M * N;
  constexpr int RP = 68;  // row pitch of the reduction buffer in floats
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    if (nb > 0) __syncthreads();  // the previous pass has been read
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r)  // the lane's 4 adjacent columns of one row: one 16-byte store (row pitch 68 floats = 17 x 16 B)
        *reinterpret_cast<float4*>(red + (wave * 64 + 16 * b + 4 * oct + r) * RP + 4 * jn) =
            make_float4(acc[b][4 * nb + 0][r], acc[b][4 * nb + 1][r], acc[b][4 * nb + 2][r], acc[b][4 * nb + 3][r]);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < (64 * 16 + NT - 1) / NT; ++i) {
      const int idx = tid + NT * i, rr = idx >> 4, c4 = (idx & 15) * 4;
      if (idx >= 64 * 16) continue;
      float4 v = *reinterpret_cast<const float4*>(red + rr * RP + c4);
#pragma unroll
      for (int wv = 1; wv < WAVES; ++wv) {  // fixed order
        const float4 u = *reinterpret_cast<const float4*>(red + (wv * 64 + rr) * RP + c4);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      }
      const int m = m0 + rr;
      const int64_t n = n0 + 64 * nb + c4;
      if (m < M && n < N) {  // N % 4 == 0 on this path: the four columns exist together
        if (splitk > 1) splitk_store16_sc1(partial + (int64_t)slice * slab + (int64_t)m * N + n, f32x4{v.x, v.y, v.z, v.w});
        else store_out4<IS_BF16>(y + (int64_t)m * N + n, v, bias ? bias + n : nullptr);
      }
    }
  }
  if (splitk <= 1) return;
  // publish: every wave drains its write-through stores, then one relaxed agent-scope ticket from thread 0
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  unsigned* const counter = counters + (by * gridDim.x + bx);
  if (tid == 0) {
    const unsigned ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = ticket == (unsigned)(splitk - 1);
    if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm for the next call
    red[0] = last ? 1.f : 0.f;
  }
  __syncthreads();
  if (red[0] == 0.f) return;
  // last arriver: fixed-order sum over the slices (sc1 16-byte loads: the partials were written through); 4 column quads x up to 4
  // slices of a thread are in flight together
  constexpr int QUADS = ROWS * COLS / 4 / NT;
  static_assert(ROWS * COLS / 4 % NT == 0 && QUADS % 4 == 0, "whole batches of four quads per thread");
  for (int i0 = 0; i0 < QUADS; i0 += 4) {
    f32x4 pv[4][4];
    int64_t off[4];
    bool ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + NT * (i0 + i), rr = idx / (COLS / 4), c4 = (idx % (COLS / 4)) * 4;
      const int m = m0 + rr;
      const int64_t n = n0 + c4;
      ok[i] = m < M && n < N;
      off[i] = ok[i] ? (int64_t)m * N + n : 0;
    }
    const float* sb[4];
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {  // slab bases are wave-uniform (SGPR pairs); slices past splitk re-read the last one and are not added
      const uint64_t a = (uint64_t)(uintptr_t)(partial + (int64_t)(sl < splitk ? sl : splitk - 1) * slab);
      sb[sl] = reinterpret_cast<const float*>((uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(a >> 32)) << 32) |
                                                          (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)a)));
    }
    splitk_load16x16_sc1(pv, sb[0], sb[1], sb[2], sb[3], (uint32_t)(off[0] * 4), (uint32_t)(off[1] * 4), (uint32_t)(off[2] * 4), (uint32_t)(off[3] * 4));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 v = pv[0][i];
#pragma unroll
      for (int sl = 1; sl < 4; ++sl)
        if (sl < splitk) v += pv[sl][i];
      if (ok[i]) {
        const int64_t n = n0 + ((tid + NT * (i0 + i)) % (COLS / 4)) * 4;
        store_out4<IS_BF16>(y + off[i], make_float4(v[0], v[1], v[2], v[3]), bias ? bias + n : nullptr);
      }
    }
  }
}

// K-slices of the strip kernel for (M, N, K): split-K only to fill the chip (one workgroup = two waves per SIMD per CU)
static int strip_splitk(int64_t M, int64_t N, int64_t K) {
  const int64_t wgs = ceil_div64(M, 64) * ceil_div64(N, 128);
  int sk = 1;
  if (wgs < 192) {
    sk = (int)(256 / wgs);
    if (sk > 4) sk = 4;                                                   // the last arriver sums <= 4 slabs
    while (sk > 1 && (K / 32) / (STRIP_WAVES * sk) < 4) --sk;            // >= 4 steps per wave
  }
  return sk;
}

// =============================================================================================
// decode kernel without split-K (round 2): M <= 16, K <= GEMV16_MAX_K
// =============================================================================================
// The streaming kernel above splits K over workgroups to put 512+ of them on the chip, and pays for it after the last MFMA:
// write-through partials, a drain, a ticket, and the last arriver's reload -- about 2 us of a 4.5 us kernel.  Here a workgroup
// owns only 16 columns but ALL of K: sixteen waves take a sixteenth of the K-steps each, a lane's packed word (8 k of one
// column) is the B operand of v_mfma_f32_16x16x32 once dequantised, and the sixteen accumulators meet in LDS -- one hop, inside
// the workgroup.  N / 16 workgroups (256 at N = 4096), every byte of W requested before the first use.
constexpr int GEMV16_WAVES = 16;
constexpr int GEMV16_CH = 12;                                              // K-steps per wave and pass whose loads are issued up front
constexpr int64_t GEMV16_MAX_K = (int64_t)32 * GEMV16_WAVES * 2 * GEMV16_CH;  // two passes: K <= 12288

template <bool IS_BF16, bool NT = false>
__global__ __launch_bounds__(64 * GEMV16_WAVES) void woq_gemv16_w4_kernel(
    const uint16_t* __restrict__ x, const uint32_t* __restrict__ qweight, const uint16_t* __restrict__ scales,
    const uint32_t* __restrict__ qzeros, const uint16_t* __restrict__ bias, uint16_t* __restrict__ y, int M, int64_t N, int64_t K,
    int64_t NW, int g_shift) {
  constexpr int WAVES = GEMV16_WAVES, CH = GEMV16_CH;
  __shared__ float red[WAVES * 16 * 17];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int jn = lane & 15, kg = lane >> 4;
  const float inv_u = fp8_unit_inverse();
  const int64_t n0 = (int64_t)blockIdx.x * 16;
  int64_t ncol = n0 + jn;
  if (ncol > N - 1) ncol = N - 1;  // clamped lanes recompute a valid column; their results are not stored
  const int zsh = 4 * (int)(ncol & 7);
  const int am = jn < M ? jn : M - 1;  // A row (clamped; rows >= M only feed outputs that are never stored)
  const uint16_t* const xrow = x + (int64_t)am * K + 8 * kg;
  const uint32_t* const wcol = qweight + (int64_t)kg * N + ncol;
  const int steps_total = (int)(K / 32);
  const int lo = (int)((int64_t)steps_total * wave / WAVES), hi = (int)((int64_t)steps_total * (wave + 1) / WAVES);

  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int c0 = lo; c0 < hi; c0 += CH) {
    uint32_t w[CH], zraw[CH];
    uint16_t sraw[CH];
    uint4 a[CH];
#pragma unroll
    for (int s = 0; s < CH; ++s) {
      int st = c0 + s;
      if (st > hi - 1) st = hi - 1;  // steps past the end re-read the last one and are skipped below
      const int64_t g = g_shift >= 0 ? (((int64_t)st * 32) >> g_shift) : 0;
      w[s] = NT ? __builtin_nontemporal_load(wcol + (int64_t)st * 4 * N) : wcol[(int64_t)st * 4 * N];
      sraw[s] = scales[g * N + ncol];
      zraw[s] = qzeros[g * NW + (ncol >> 3)];
      a[s] = *reinterpret_cast<const uint4*>(xrow + (int64_t)st * 32);
    }
    __builtin_amdgcn_sched_barrier(0);  // everything above is in flight before the first use below
#pragma unroll
    for (int s = 0; s < CH; ++s) {
      if (c0 + s < hi) {
        const float sc = f16_bits_to_f32(sraw[s]);
        uint32_t zz = ((zraw[s] >> zsh) & 15u) + 1u;  // modules.py:407-410 (stored zp - 1; wraps above 15)
        zz = zz > 15u ? 0u : zz;
        const uint4 bq = dequant8<IS_BF16, 1>(w[s], sc * inv_u, -(float)zz * sc);
        acc = mfma16<IS_BF16>(a[s], bq, acc);
      }
    }
  }
  // D of the MFMA: column = lane & 15 (= jn), row = 4 * kg + r
#pragma unroll
  for (int r = 0; r < 4; ++r) red[(wave * 16 + 4 * kg + r) * 17 + jn] = acc[r];
  __syncthreads();
  if (tid < 256) {
    const int m = tid >> 4, c = tid & 15;
    float v = red[m * 17 + c];
#pragma unroll
    for (int wv = 1; wv < WAVES; ++wv) v += red[(wv * 16 + m) * 17 + c];  // fixed order
    const int64_t n = n0 + c;
    if (m < M && n < N) {
      v += bias ? cvt16<IS_BF16>(bias[n]) : 0.f;
      y[(int64_t)m * N + n] = IS_BF16 ? f32_to_bf16_bits(v) : f32_to_f16_bits(v);
    }
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

// workspace layout : [0, 16 KiB) arrival counters of the fast GEMV (uint32 per 64-column
// strip; MUST be zero on first use, the kernel re-arms them), then fp32 split-K partials.
constexpr int64_t WS_COUNTER_BYTES = 16384;
// split-K plan of the 3A2B kernel for medium M: enough workgroups to cover the chip, slabs of >= 4 K-steps
static int big_splitk(int64_t M, int64_t N, int64_t K, int* steps_out) {
  const int64_t tiles = ceil_div64(M, TM) * ceil_div64(N, TN);
  const int nk = (int)(K / TK);
  int splits = 1;
  if (tiles < 192 && (K % 128) == 0 && (N % 4) == 0) {
    splits = (int)(256 / tiles);
    const int cap = tiles <= 16 ? 16 : 8;
    if (splits > cap) splits = cap;
    while (splits > 1 && (nk / splits) < 4) --splits;
  }
  int steps = (int)ceil_div64(nk, splits);
  steps += steps & 1;  // the K-loop is unrolled by two
  splits = (int)ceil_div64(nk, steps);
  *steps_out = steps;
  return splits;
}

constexpr int64_t GEMV_MAX_M = 64;  // M <= 64 streams the weights once (woq_gemv_w4_kernel with 1 / 2 / 4 row blocks)
// the same for 8-bit words: up to 32 rows everywhere, up to 64 rows while N * K <= 2^25 -- measured against the 8-bit tile kernel
// (scripts/w8_gemm_time.py: 4096^2 M = 48 / 64: 17.3 / 18.5 vs 24.4 us; 11008 x 4096 M = 32: 18.7 vs 32.7 us, M = 48 / 64: 40.8 / 43.8 vs 32.7 us)
constexpr int64_t GEMV8_MAX_M = 64, GEMV8_WIDE_MAX_M = 32, GEMV8_WIDE_ELEMS = (int64_t)1 << 25;

int64_t inc_woq_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  // an upper bound over the routes inc_woq_gemm can take for (M, N, K) (it does not know bits / group size here)
  int64_t need = 0;
  auto at_least = [&](int64_t v) { if (v > need) need = v; };
  if (M > 16) {
    if (N >= 64) {  // 256-row tiles (producer / consumer, 3A2B incl. its 8-bit form): split-K slabs behind the counter block
      int steps;
      const int splits = big_splitk(M, N, K, &steps);
      if (splits > 1) at_least(WS_COUNTER_BYTES + (int64_t)splits * M * N * 4);
    }
    // the strip kernel, where inc_woq_gemm routes to it, with the K-slices it would use
    const bool strip_m = M > GEMV_MAX_M || (M > 32 && N * K > ((int64_t)24 << 20));
    if (strip_m && M <= STRIP_MAX_M && N >= 64 && (K % 32) == 0 && ceil_div64(M, TM) * ceil_div64(N, TN) <= 64) {
      const int sk = strip_splitk(M, N, K);
      if (sk > 1) at_least(WS_COUNTER_BYTES + (int64_t)sk * M * N * 4);
    }
    if (M <= GEMV_MAX_M) at_least(WS_COUNTER_BYTES + ceil_div64(K, 32 * 4 * 4) * M * N * 4);  // streaming kernel, 4 steps per wave
    return need;
  }
  int64_t slices = ceil_div64(K, 32 * 4 * 4);  // the streaming kernel at 4 steps per wave
  if (slices < 64) slices = 64;                 // the generic split-K path uses up to 64 slices
  return WS_COUNTER_BYTES + slices * M * N * 4;
}

int inc_woq_gemm(const void* x, int xdtype, const int32_t* qweight, const uint16_t* scales,
                 const int32_t* qzeros, const int32_t* g_idx, const void* bias, void* y, int64_t M,
                 int64_t N, int64_t K, int64_t G, int group_size, int bits, void* workspace,
                 int64_t workspace_bytes, inc_stream_t stream) {
  INC_CHECK_ARG(x && qweight && scales && qzeros && y && M > 0 && N > 0 && K > 0 && G > 0 && group_size > 0);
  if (bits < 1 || bits > 8) return INC_ERR_UNSUPPORTED;
  // g_idx (per-element groups: GPTQ act_order / HF desc_act): the general tile kernels below look the group up per k; the
  // 256-row kernels assume contiguous groups (the module sorts K by group once and calls them without g_idx, modules.py)
  if (!(xdtype == INC_BF16 || xdtype == INC_F16)) return INC_ERR_UNSUPPORTED;
  const int np = 32 / bits;
  // a packed word must not straddle two groups unless g_idx is given per element... (word-granular
  // group lookup): require group boundaries on word boundaries.
  const bool anyw = !(bits == 4 || bits == 8);  // 1 / 2 / 3 / 5 / 6 / 7 bits: the tile kernel's per-element form, any group_size
  if (!anyw && !g_idx && (group_size % np) != 0 && group_size < K) return INC_ERR_UNSUPPORTED;
  if (anyw && K >= ((int64_t)1 << 31)) return INC_ERR_UNSUPPORTED;
  const int64_t KW = ceil_div64(K, np), NW = ceil_div64(N, np);
  hipStream_t s = inc_s(stream);
  const uint16_t* xp = (const uint16_t*)x;
  const uint32_t* qw = (const uint32_t*)qweight;
  const uint32_t* qz = (const uint32_t*)qzeros;
  const uint16_t* bp = (const uint16_t*)bias;
  uint16_t* yp = (uint16_t*)y;
  const bool bf = xdtype == INC_BF16;
  // group lookup by shift: group_size a power of two >= 32, or a single group (group_size >= K)
  int g_shift = -2;
  if (group_size >= K) g_shift = -1;
  else if (group_size >= 32 && (group_size & (group_size - 1)) == 0) { g_shift = 0; while ((1 << g_shift) < group_size) ++g_shift; }
  // 16 < M < 128 (batched decode) runs the same 256-row tile with most rows clamped: with split-K over up to 16 slabs that is
  // 30 us at 64 x 4096 x 4096 where the 128x128 register-staged kernel needed 208 us (and 615 us at 17 x 4096 x 11008)
  const bool gemv_ok = !g_idx && bits == 4 && g_shift != -2 && (K % 32) == 0 && (N % 4) == 0 && N >= 64 && ceil_div64(N, 64) * 4 <= WS_COUNTER_BYTES &&
                       (reinterpret_cast<uintptr_t>(x) & 15) == 0 && !inc_force_small_tiles() && M <= GEMV_MAX_M && inc_small_tiles_flag(-1) != 42;
  const bool big_ok = !g_idx && bits == 4 && (K % TK) == 0 && g_shift != -2 && M > 16 && N >= 64 && !(gemv_ok && ceil_div64(K, 32 * 4 * 4) <= 64) &&
                      (reinterpret_cast<uintptr_t>(x) & 15) == 0 && N * (K / 8) < (int64_t)1 << 31;
  const int dbg = inc_small_tiles_flag(-1);
  // weight-only INT8 (BASELINE config #1's layers): the 3A2B kernel's 8-bit instantiation, same tiling and split-K plan
  // weight-only INT8 at M <= 64: the streaming kernel's 8-bit form (M <= 16 whatever K; 16 < M <= 64 while its K-slices fit the counters' plan)
  const bool gemv8_ok = !g_idx && bits == 8 && g_shift != -2 && (K % 32) == 0 && (N % 4) == 0 && N >= 64 && M <= GEMV8_MAX_M && dbg == 0 &&
                        (M <= GEMV8_WIDE_MAX_M || N * K <= GEMV8_WIDE_ELEMS) && (M <= 16 || ceil_div64(K, 32 * 4 * 4) <= 64) && ceil_div64(N, 64) * 4 <= WS_COUNTER_BYTES && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  const bool big8_ok = !gemv8_ok && !g_idx && bits == 8 && (K % 128) == 0 && g_shift != -2 && M > 16 && N >= 64 && dbg == 0 &&
                       (reinterpret_cast<uintptr_t>(x) & 15) == 0 && N * (K / 4) < (int64_t)1 << 31;
  // 64 < M <= 1024 with at most 64 tiles of 256 x 256: the strip kernel (no 256-row tiles, no slab passes).  With more tiles the
  // producer / consumer kernel fills the chip with <= 2 slabs and wins (M = 512, N = 11008: 71 vs 81 us; tools/kbench strip).
  // 32 < M <= 64 on the larger layers too (M = 64, 11008 x 4096: 21 vs 29 us for the streaming kernel, whose x fragments are
  // per-lane 16-byte gathers; at 4096^2 the streaming kernel keeps a 1 us lead).  Harness flags 42 / 40 / 4 / 6 select the tile paths, 83 this kernel for any M > 16.
  if (anyw) {
    const int x_vec_ok = (K % 8 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    const size_t smem = (size_t)2 * 2 * GM * GP * sizeof(uint16_t);
    const unsigned grid = (unsigned)(ceil_div64(M, GM) * ceil_div64(N, GN));
#define INC_TILE_W(B)                                                                                                                 \
  {                                                                                                                                   \
    static std::atomic<uint64_t> aset{0};                                                                                             \
    if (inc_attr_needed(aset)) {                                                                                                      \
      (void)hipFuncSetAttribute((const void*)woq_gemm_tile_kernel<B, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);   \
      (void)hipFuncSetAttribute((const void*)woq_gemm_tile_kernel<B, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);  \
      inc_attr_done(aset);                                                                                                            \
    }                                                                                                                                 \
    if (bf) woq_gemm_tile_kernel<B, true><<<grid, 256, smem, s>>>(xp, qw, scales, qz, g_idx, bp, yp, M, N, K, KW, NW, group_size, x_vec_ok);  \
    else woq_gemm_tile_kernel<B, false><<<grid, 256, smem, s>>>(xp, qw, scales, qz, g_idx, bp, yp, M, N, K, KW, NW, group_size, x_vec_ok);    \
  }
    switch (bits) {
      case 1: INC_TILE_W(1) break;
      case 2: INC_TILE_W(2) break;
      case 3: INC_TILE_W(3) break;
      case 5: INC_TILE_W(5) break;
      case 6: INC_TILE_W(6) break;
      default: INC_TILE_W(7) break;
    }
#undef INC_TILE_W
    INC_LAUNCH_RETURN();
  }
  const bool strip_ok = !g_idx && bits == 4 && g_shift != -2 && (K % 32) == 0 && (N % 4) == 0 && N >= 64 && (M > GEMV_MAX_M || (M > 32 && N * K > ((int64_t)24 << 20)) || (dbg == 83 && M > 16)) && M <= STRIP_MAX_M &&
                        ceil_div64(M, TM) * ceil_div64(N, TN) <= 64 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 7) == 0 &&
                        (reinterpret_cast<uintptr_t>(bias) & 7) == 0 && (dbg == 0 || (dbg >= 83 && dbg <= 89) || (dbg >= 100 && dbg <= 102));
  if (strip_ok && dbg == 0 && M > 128 && M * N < ((int64_t)1 << 30) && inc_woq_gemm_strip8_splitk(M, N, K) == 1) {
    // enough 128 x 128 tiles to fill the chip without K-slices (>= 192): the four-wave kernel that dequantises every weight once per
    // 128 rows (gemm_strip8.hip).  tools/midm_lab, 4096 x 4096: M = 1024 46 vs 57 us; 11008 x 4096, M = 256: 40 vs 48 us.  With K-slices
    // its larger partial tiles lose to the 64-row strips below (M = 512: 34 vs 30 us), so those keep this kernel's predecessor.
    return inc_launch_woq_gemm_strip8(xp, qw, scales, qz, bp, yp, M, N, K, NW, g_shift, nullptr, nullptr, 1, bf, s);
  }
  if (strip_ok && M * N < ((int64_t)1 << 30)) {
    int splitk = strip_splitk(M, N, K);
    const int64_t wgs = ceil_div64(M, 64) * ceil_div64(N, 128);
    if (splitk > 1 && (!workspace || workspace_bytes < WS_COUNTER_BYTES + (int64_t)splitk * M * N * 4 || wgs * 4 > WS_COUNTER_BYTES)) splitk = 1;
    static std::atomic<uint64_t> strip_attr_set{0};
    if (inc_attr_needed(strip_attr_set)) {
      (void)hipFuncSetAttribute((const void*)woq_gemm_w4_strip_kernel<true, STRIP_WAVES, STRIP_RING>, hipFuncAttributeMaxDynamicSharedMemorySize, STRIP_SMEM_BYTES);
      (void)hipFuncSetAttribute((const void*)woq_gemm_w4_strip_kernel<false, STRIP_WAVES, STRIP_RING>, hipFuncAttributeMaxDynamicSharedMemorySize, STRIP_SMEM_BYTES);
      inc_attr_done(strip_attr_set);
    }
    unsigned* counters = (unsigned*)workspace;
    float* part = splitk > 1 ? (float*)((char*)workspace + WS_COUNTER_BYTES) : nullptr;
    dim3 grid((unsigned)ceil_div64(N, 128), (unsigned)ceil_div64(M, 64), (unsigned)splitk);
#ifdef INC_KBENCH
    {
      const int f = inc_small_tiles_flag(-1);
      if (bf && f >= 85 && f <= 89) {  // harness: timing-only ablations of the strip step
#define INC_STRIP_ABL(A)                                                                                                              \
  {                                                                                                                                   \
    (void)hipFuncSetAttribute((const void*)woq_gemm_w4_strip_kernel<true, STRIP_WAVES, STRIP_RING, A>, hipFuncAttributeMaxDynamicSharedMemorySize, STRIP_SMEM_BYTES); \
    woq_gemm_w4_strip_kernel<true, STRIP_WAVES, STRIP_RING, A><<<grid, 64 * STRIP_WAVES, STRIP_SMEM_BYTES, s>>>(xp, qw, scales, qz, bp, yp, part, counters, (int)M, N, K, NW, g_shift, splitk); \
  }
        if (f == 85) INC_STRIP_ABL(1) else if (f == 86) INC_STRIP_ABL(2) else if (f == 87) INC_STRIP_ABL(3) else if (f == 88) INC_STRIP_ABL(4) else INC_STRIP_ABL(7)
        INC_LAUNCH_RETURN();
      }
      if (bf && f >= 100 && f <= 102) {  // harness A/B with CORRECT results: 100 LDS reads one step ahead, 101 s_setprio around the compute, 102 both
        if (f == 100) INC_STRIP_ABL(8) else if (f == 101) INC_STRIP_ABL(16) else INC_STRIP_ABL(24)
        INC_LAUNCH_RETURN();
      }
#undef INC_STRIP_ABL
    }
    if (bf && inc_small_tiles_flag(-1) == 84) {  // harness A/B: four waves (one per SIMD) with a six-deep ring
      (void)hipFuncSetAttribute((const void*)woq_gemm_w4_strip_kernel<true, 4, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 6 * 6 * 1024);
      woq_gemm_w4_strip_kernel<true, 4, 6><<<grid, 64 * 4, 4 * 6 * 6 * 1024, s>>>(xp, qw, scales, qz, bp, yp, part, counters, (int)M, N, K, NW, g_shift, splitk);
      INC_LAUNCH_RETURN();
    }
#endif
    if (bf) woq_gemm_w4_strip_kernel<true, STRIP_WAVES, STRIP_RING><<<grid, 64 * STRIP_WAVES, STRIP_SMEM_BYTES, s>>>(xp, qw, scales, qz, bp, yp, part, counters, (int)M, N, K, NW, g_shift, splitk);
    else woq_gemm_w4_strip_kernel<false, STRIP_WAVES, STRIP_RING><<<grid, 64 * STRIP_WAVES, STRIP_SMEM_BYTES, s>>>(xp, qw, scales, qz, bp, yp, part, counters, (int)M, N, K, NW, g_shift, splitk);
    INC_LAUNCH_RETURN();
  }
  if (big8_ok) {
    const size_t smem = (size_t)3 * T_ASTAGE + 2 * T_BSTAGE;
    static std::atomic<uint64_t> a8_attr_set{0};
    if (inc_attr_needed(a8_attr_set)) {
      (void)hipFuncSetAttribute((const void*)woq_gemm_w4_3a2b_kernel<true, INC_3A2B_DEFAULT_SCHED, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      (void)hipFuncSetAttribute((const void*)woq_gemm_w4_3a2b_kernel<false, INC_3A2B_DEFAULT_SCHED, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      inc_attr_done(a8_attr_set);
    }
    const unsigned grid = (unsigned)(ceil_div64(M, TM) * ceil_div64(N, TN));
    const int y_vec_ok = (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) & 7) == 0);
    int steps = (int)(K / TK);
    int splits = big_splitk(M, N, K, &steps);
    float* part = nullptr;
    if (splits > 1) {
      if (y_vec_ok && workspace && workspace_bytes >= WS_COUNTER_BYTES + (int64_t)splits * M * N * 4)
        part = (float*)((char*)workspace + WS_COUNTER_BYTES);
      else { splits = 1; steps = (int)(K / TK); }
    }
    dim3 g2(grid, (unsigned)splits);
    if (bf) woq_gemm_w4_3a2b_kernel<true, INC_3A2B_DEFAULT_SCHED, 8><<<g2, 512, smem, s>>>(xp, qw, scales, qz, bp, yp, M, N, K, NW, g_shift, y_vec_ok, part, steps);
    else woq_gemm_w4_3a2b_kernel<false, INC_3A2B_DEFAULT_SCHED, 8><<<g2, 512, smem, s>>>(xp, qw, scales, qz, bp, yp, M, N, K, NW, g_shift, y_vec_ok, part, steps);
    if (part) {
      int64_t rb = ceil_div64(M * N / 4, 256);
      if (rb > 4096) rb = 4096;
      if (bf) splitk_slab_reduce_kernel<true><<<(unsigned)rb, 256, 0, s>>>(part, bp, yp, M, N, splits);
      else splitk_slab_reduce_kernel<false><<<(unsigned)rb, 256, 0, s>>>(part, bp, yp, M, N, splits);
    }
  } else if (big_ok && (K % 128) == 0 && (g_shift == -1 || g_shift >= 6) && (N % 2) == 0 && ((dbg == 0 && INC_GEMM_DEFAULT_D2R) || (dbg >= 90 && dbg <= 99))) {
    // weights direct to registers (gemm_d2r.hip): four waves, one per SIMD, no dequantised tile in LDS
    const int y_vec_ok = (((N % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) & 7) == 0)) ? 1 : 0) |
                         (((N % 8 == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0)) ? 2 : 0);
    int steps = (int)(K / TK);
    int splits = big_splitk(M, N, K, &steps);
    float* part = nullptr;
    if (splits > 1) {
      if ((y_vec_ok & 1) && workspace && workspace_bytes >= WS_COUNTER_BYTES + (int64_t)splits * M * N * 4)
        part = (float*)((char*)workspace + WS_COUNTER_BYTES);
      else { splits = 1; steps = (int)(K / TK); }
    }
    static const int d2r_abl[10] = {0, 0, 4, 8, 12, 76, 128, 0, 256, 0};  // harness flags 90..96 (91: three x stages; 92..96 timing-only), 98: time stamps
    const int abl = dbg >= 90 ? d2r_abl[dbg - 90] : 0;
#ifdef INC_KBENCH
    // harness flag 97: the eight-wave form (gemm_d2r8.hip, two waves per SIMD with redundant dequantisation): bit-identical and 6 %
    // SLOWER (1203 vs 1284 TFLOP/s at 4096^3, profiles/r3h_kbench_d2r8.log) -- built into the harness library only
    if (dbg == 97)
      (void)inc_launch_woq_gemm_d2r8(xp, qw, scales, qz, bp, yp, M, N, K, NW, g_shift, y_vec_ok, part, steps, splits, bf, s);
    else
#endif
      (void)inc_launch_woq_gemm_d2r(xp, qw, scales, qz, bp, yp, M, N, K, NW, g_shift, y_vec_ok, part, steps, splits, bf, dbg == 91 ? 3 : 4, abl, s);
    if (part) {
      int64_t rb = ceil_div64(M * N / 4, 256);
      if (rb > 4096) rb = 4096;
      if (bf) splitk_slab_reduce_kernel<true><<<(unsigned)rb, 256, 0, s>>>(part, bp, yp, M, N, splits);
      else splitk_slab_reduce_kernel<false><<<(unsigned)rb, 256, 0, s>>>(part, bp, yp, M, N, splits);
    }
#ifdef INC_KBENCH
#include "../../tools/kbench_gemm_2.inc"
#endif  // INC_KBENCH
  } else if (big_ok && (K % 128) == 0 && (dbg == 0 || dbg == 40 || dbg == 4 || dbg == 6 || (dbg >= 20 && dbg <= 30) || (dbg >= 31 && dbg <= 37))) {
    const size_t smem = (size_t)3 * T_ASTAGE + 2 * T_BSTAGE;  // 160 KiB: the whole LDS of a CU
    static std::atomic<uint64_t> a3_attr_set{0};
    if (inc_attr_needed(a3_attr_set)) {
#define INC_A3_ATTR(B, S) (void)hipFuncSetAttribute((const void*)woq_gemm_w4_3a2b_kernel<B, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)
      INC_A3_ATTR(true, INC_3A2B_DEFAULT_SCHED); INC_A3_ATTR(false, INC_3A2B_DEFAULT_SCHED);
#ifdef INC_KBENCH  // harness build: the other schedules and the timing-only ablations (tools/kbench)
      INC_A3_ATTR(true, 1 - INC_3A2B_DEFAULT_SCHED);
      INC_A3_ATTR(true, 10); INC_A3_ATTR(true, 11); INC_A3_ATTR(true, 12); INC_A3_ATTR(true, 13); INC_A3_ATTR(true, 14);
      INC_A3_ATTR(true, 15); INC_A3_ATTR(true, 16); INC_A3_ATTR(true, 3); INC_A3_ATTR(true, 17); INC_A3_ATTR(true, 18); INC_A3_ATTR(true, 19); INC_A3_ATTR(true, 20);
      INC_A3_ATTR(true, 31); INC_A3_ATTR(true, 32); INC_A3_ATTR(true, 34); INC_A3_ATTR(true, 37);
#endif
#undef INC_A3_ATTR
      inc_attr_done(a3_attr_set);
    }
    const unsigned grid = (unsigned)(ceil_div64(M, TM) * ceil_div64(N, TN));
    const int y_vec_ok = (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) & 7) == 0);
    int steps = (int)(K / TK);
    int splits = big_splitk(M, N, K, &steps);
    float* part = nullptr;
    if (splits > 1) {
      if (y_vec_ok && workspace && workspace_bytes >= WS_COUNTER_BYTES + (int64_t)splits * M * N * 4)
        part = (float*)((char*)workspace + WS_COUNTER_BYTES);  // never touch the GEMV's arrival counters
      else { splits = 1; steps = (int)(K / TK); }  // no workspace given: single pass (still correct, fewer workgroups)
    }
    dim3 g2(grid, (unsigned)splits);
#define INC_A3(B, S) woq_gemm_w4_3a2b_kernel<B, S><<<g2, 512, smem, s>>>(xp, qw, scales, qz, bp, yp, M, N, K, NW, g_shift, y_vec_ok, part, steps)
    if (!bf) INC_A3(false, INC_3A2B_DEFAULT_SCHED);
#ifdef INC_KBENCH
    // harness build: flag 4 / 6 time the other schedules of the bf16 kernel, 20..37 its timing-only ablations
    else if (dbg == 4) INC_A3(true, 1 - INC_3A2B_DEFAULT_SCHED);
    else if (dbg == 6) INC_A3(true, 3);
    else if (dbg == 31) INC_A3(true, 31);
    else if (dbg == 32) INC_A3(true, 32);
    else if (dbg == 34) INC_A3(true, 34);
    else if (dbg == 37) INC_A3(true, 37);
    else if (dbg == 20) INC_A3(true, 10);  // 20..26: timing-only ablations of schedule 1 (wrong results by construction)
    else if (dbg == 21) INC_A3(true, 11);
    else if (dbg == 22) INC_A3(true, 12);
    else if (dbg == 23) INC_A3(true, 13);
    else if (dbg == 24) INC_A3(true, 14);
    else if (dbg == 25) INC_A3(true, 15);
    else if (dbg == 26) INC_A3(true, 16);
    else if (dbg == 27) INC_A3(true, 17);
    else if (dbg == 28) INC_A3(true, 18);
    else if (dbg == 29) INC_A3(true, 19);
    else if (dbg == 30) INC_A3(true, 20);
#endif
    else INC_A3(true, INC_3A2B_DEFAULT_SCHED);
#undef INC_A3
    if (part) {
      int64_t rb = ceil_div64(M * N / 4, 256);
      if (rb > 4096) rb = 4096;
      if (bf) splitk_slab_reduce_kernel<true><<<(unsigned)rb, 256, 0, s>>>(part, bp, yp, M, N, splits);
      else splitk_slab_reduce_kernel<false><<<(unsigned)rb, 256, 0, s>>>(part, bp, yp, M, N, splits);
    }
  } else if (big_ok && !inc_force_small_tiles()) {
    const size_t smem = (size_t)2 * T_ASTAGE + 2 * T_BSTAGE;  // 128 KiB
    static std::atomic<uint64_t> big_attr_set{0};
    if (inc_attr_needed(big_attr_set)) {
      (void)hipFuncSetAttribute((const void*)woq_gemm_w4_big_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      (void)hipFuncSetAttribute((const void*)woq_gemm_w4_big_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      inc_attr_done(big_attr_set);
    }
    const unsigned grid = (unsigned)(ceil_div64(M, TM) * ceil_div64(N, TN));
    const int y_vec_ok = (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) & 7) == 0);
    if (bf) woq_gemm_w4_big_kernel<true><<<grid, 512, smem, s>>>(xp, qw, scales, qz, bp, yp, M, N, K, NW, g_shift, y_vec_ok);
    else woq_gemm_w4_big_kernel<false><<<grid, 512, smem, s>>>(xp, qw, scales, qz, bp, yp, M, N, K, NW, g_shift, y_vec_ok);
  } else if (M > 16 && !gemv8_ok && !(gemv_ok && ceil_div64(K, 32 * 4 * 4) <= 64)) {
    const int x_vec_ok = (K % 8 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    const size_t smem = (size_t)2 * 2 * GM * GP * sizeof(uint16_t);
    static std::atomic<uint64_t> attr_set{0};
    if (inc_attr_needed(attr_set)) {
      (void)hipFuncSetAttribute((const void*)woq_gemm_tile_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      (void)hipFuncSetAttribute((const void*)woq_gemm_tile_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      (void)hipFuncSetAttribute((const void*)woq_gemm_tile_kernel<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      (void)hipFuncSetAttribute((const void*)woq_gemm_tile_kernel<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      inc_attr_done(attr_set);
    }
    const unsigned grid = (unsigned)(ceil_div64(M, GM) * ceil_div64(N, GN));
#define INC_TILE(B, F) woq_gemm_tile_kernel<B, F><<<grid, 256, smem, s>>>(xp, qw, scales, qz, g_idx, bp, yp, M, N, K, KW, NW, group_size, x_vec_ok)
    if (bits == 4) { if (bf) INC_TILE(4, true); else INC_TILE(4, false); }
    else { if (bf) INC_TILE(8, true); else INC_TILE(8, false); }
#undef INC_TILE
#ifdef INC_KBENCH
  } else if (gemv_ok && M <= 16 && K <= GEMV16_MAX_K && bf && dbg == 103) {  // harness: the no-split decode kernel with non-temporal weight loads
    woq_gemv16_w4_kernel<true, true><<<(unsigned)ceil_div64(N, 16), 64 * GEMV16_WAVES, 0, s>>>(xp, qw, scales, qz, bp, yp, (int)M, N, K, NW, g_shift);
  } else if (gemv_ok && M <= 16 && bf && dbg == 104 && (g_shift == -1 || g_shift >= 7)) {  // harness: the streaming kernel, non-temporal weight loads
    const bool vs4 = ceil_div64(N, 64) * ceil_div64(K, 32 * 8 * 4) < 512 && ceil_div64(K, 32 * 4 * 4) <= 64;
    const int splitk = (int)ceil_div64(K, 32 * (vs4 ? 4 : 8) * 4);
    if (!workspace || workspace_bytes < WS_COUNTER_BYTES + (int64_t)splitk * M * N * 4) return INC_ERR_WORKSPACE;
    unsigned* counters = (unsigned*)workspace;
    float* part = (float*)((char*)workspace + WS_COUNTER_BYTES);
    dim3 grid((unsigned)ceil_div64(N, 64), (unsigned)splitk);
    if (vs4) woq_gemv_w4_kernel<true, true, 4, 1, true><<<grid, 256, 0, s>>>(xp, qw, scales, qz, bp, yp, part, counters, (int)M, N, K, NW, G, g_shift, splitk);
    else woq_gemv_w4_kernel<true, true, 8, 1, true><<<grid, 256, 0, s>>>(xp, qw, scales, qz, bp, yp, part, counters, (int)M, N, K, NW, G, g_shift, splitk);
#endif
  } else if (gemv_ok && M <= 16 && K <= GEMV16_MAX_K && (dbg == 85 || (dbg == 0 && M <= 4 && N <= 4096 && K <= 4096))) {
    // decode without split-K: one workgroup of 16 waves per 16 columns, the whole of K.  Wins where its N / 16 workgroups are a
    // single round on the chip and x is <= 4 rows (M = 1, 4096^2: 5.9 vs 6.8 us); elsewhere the streaming kernel's 64-column
    // requests and K-slices use the HBM better (M = 1, 11008 x 4096: 8.5 vs 15.9 us; tools/kbench decode; harness flag 85 forces it)
    const unsigned grid = (unsigned)ceil_div64(N, 16);
    if (bf) woq_gemv16_w4_kernel<true><<<grid, 64 * GEMV16_WAVES, 0, s>>>(xp, qw, scales, qz, bp, yp, (int)M, N, K, NW, g_shift);
    else woq_gemv16_w4_kernel<false><<<grid, 64 * GEMV16_WAVES, 0, s>>>(xp, qw, scales, qz, bp, yp, (int)M, N, K, NW, g_shift);
  } else if (gemv_ok && (M <= 16 || ceil_div64(K, 32 * 4 * 4) <= 64)) {
    // 8 steps per wave when that still gives every SIMD two waves (>= 512 workgroups), else 4; row-blocked (M > 16): always 4
    const bool vs4 = M > 16 || (ceil_div64(N, 64) * ceil_div64(K, 32 * 8 * 4) < 512 && ceil_div64(K, 32 * 4 * 4) <= 64);
    const int vsteps = vs4 ? 4 : 8;
    const int splitk = (int)ceil_div64(K, 32 * vsteps * 4);
    if (!workspace || workspace_bytes < WS_COUNTER_BYTES + (int64_t)splitk * M * N * 4) return INC_ERR_WORKSPACE;
    unsigned* counters = (unsigned*)workspace;
    float* part = (float*)((char*)workspace + WS_COUNTER_BYTES);
    dim3 grid((unsigned)ceil_div64(N, 64), (unsigned)splitk);
    const bool g128 = g_shift == -1 || g_shift >= 7;
#define INC_GEMV(F, GG, V, B) woq_gemv_w4_kernel<F, GG, V, B><<<grid, 256, 0, s>>>(xp, qw, scales, qz, bp, yp, part, counters, (int)M, N, K, NW, G, g_shift, splitk)
#define INC_GEMV2(F, GG) { if (M > 32) INC_GEMV(F, GG, 4, 4); else if (M > 16) INC_GEMV(F, GG, 4, 2); else if (vs4) INC_GEMV(F, GG, 4, 1); else INC_GEMV(F, GG, 8, 1); }
    if (bf) { if (g128) INC_GEMV2(true, true) else INC_GEMV2(true, false) }
    else { if (g128) INC_GEMV2(false, true) else INC_GEMV2(false, false) }
#undef INC_GEMV2
#undef INC_GEMV
  } else if (gemv8_ok) {
    // weight-only INT8 decode (BASELINE config #1's format): the streaming kernel's 8-bit form -- 64-column strips x K-slices of 512 k,
    // every wave's 8 KiB of packed weights requested before the first use, same hand-off.  (The generic split-K kernel it replaces
    // here read 16.8 MB in 20.6 us at 4096^2 and 45 MB in 46.7 us at 11008 x 4096: scripts/w8_gemm_time.py.)
    const int splitk = (int)ceil_div64(K, 32 * 4 * 4);
    if (!workspace || workspace_bytes < WS_COUNTER_BYTES + (int64_t)splitk * M * N * 4) return INC_ERR_WORKSPACE;
    unsigned* counters = (unsigned*)workspace;
    float* part = (float*)((char*)workspace + WS_COUNTER_BYTES);
    dim3 grid((unsigned)ceil_div64(N, 64), (unsigned)splitk);
    const bool g128 = g_shift == -1 || g_shift >= 7;
    const int64_t NW8 = ceil_div64(N, 4);
#define INC_GEMV8B(F, GG, B) woq_gemv_w4_kernel<F, GG, 4, B, false, 8><<<grid, 256, 0, s>>>(xp, qw, scales, qz, bp, yp, part, counters, (int)M, N, K, NW8, G, g_shift, splitk)
#define INC_GEMV8(F, GG) { if (M > 32) INC_GEMV8B(F, GG, 4); else if (M > 16) INC_GEMV8B(F, GG, 2); else INC_GEMV8B(F, GG, 1); }
    if (bf) { if (g128) INC_GEMV8(true, true) else INC_GEMV8(true, false) }
    else { if (g128) INC_GEMV8(false, true) else INC_GEMV8(false, false) }
#undef INC_GEMV8
#undef INC_GEMV8B
  } else {
    int kw_per_slice = 0;
    const int slices = small_slices(N, K, bits, &kw_per_slice);
    if (!workspace || workspace_bytes < WS_COUNTER_BYTES + (int64_t)slices * M * N * 4) return INC_ERR_WORKSPACE;
    float* part = (float*)((char*)workspace + WS_COUNTER_BYTES);
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

// ---- modules that share x, one launch (decode: q / k / v, gate / up) -----------------------------------------------------------
// plan of the batched streaming launch: strips of 64 columns per module, VSTEPS by the rule inc_woq_gemm applies to ONE module with
// the modules' columns together -> the launch is bit-identical to inc_woq_gemm on the N-concatenated module (strips are independent)
static bool gemv_multi_plan(int n, int64_t M, const int64_t* N, int64_t K, int group_size, int bits, int* g_shift_out, int* vsteps_out,
                            int* splitk_out, int64_t* strips_out) {
  if (n < 2 || n > GEMV_MAX_BATCH || !(bits == 4 || (bits == 8 && M <= 16)) || M < 1 || M > GEMV_MAX_M || K <= 0 || (K % 32) != 0) return false;
  int g_shift = -2;
  if (group_size >= K) g_shift = -1;
  else if (group_size >= 32 && (group_size & (group_size - 1)) == 0) { g_shift = 0; while ((1 << g_shift) < group_size) ++g_shift; }
  if (g_shift == -2) return false;
  int64_t strips = 0;
  for (int i = 0; i < n; ++i) {
    if (N[i] < 64 || (N[i] % 4) != 0) return false;
    strips += ceil_div64(N[i], 64);
  }
  if (strips * 4 > WS_COUNTER_BYTES) return false;
  if (M > 16 && ceil_div64(K, 32 * 4 * 4) > 64) return false;  // (row-blocked form: the single call takes a tile kernel there)
  int64_t ntot = 0;
  for (int i = 0; i < n; ++i) ntot += N[i];
  if (M > 32 && ntot * K > ((int64_t)24 << 20)) return false;  // inc_woq_gemm prefers the strip kernel there (21 vs 29 us at 64 x 11008 x 4096)
  const bool vs4 = bits == 8 || M > 16 || (strips * ceil_div64(K, 32 * 8 * 4) < 512 && ceil_div64(K, 32 * 4 * 4) <= 64);
  *g_shift_out = g_shift;
  *vsteps_out = vs4 ? 4 : 8;
  *splitk_out = (int)ceil_div64(K, 32 * (vs4 ? 4 : 8) * 4);
  *strips_out = strips;
  return true;
}

int64_t inc_woq_gemm_multi_workspace_bytes(int n, int64_t M, const int64_t* N, int64_t K) {
  if (n < 1 || !N) return 0;
  int64_t ntot = 0;
  for (int i = 0; i < n; ++i) ntot += N[i];
  return WS_COUNTER_BYTES + ceil_div64(K, 32 * 4 * 4) * M * ntot * 4;  // 4 steps per wave: the most slices either form uses
}

int inc_woq_gemm_multi(int n, const void* x, int xdtype, const int32_t* const* qweight, const uint16_t* const* scales,
                       const int32_t* const* qzeros, const void* const* bias, void* const* y, int64_t M, const int64_t* N, int64_t K,
                       int group_size, int bits, void* workspace, int64_t workspace_bytes, inc_stream_t stream) {
  INC_CHECK_ARG(x && qweight && scales && qzeros && y && N && n > 0 && M > 0 && K > 0 && group_size > 0);
  if (!(xdtype == INC_BF16 || xdtype == INC_F16)) return INC_ERR_UNSUPPORTED;
  int g_shift, vsteps, splitk;
  int64_t strips;
  if (!gemv_multi_plan(n, M, N, K, group_size, bits, &g_shift, &vsteps, &splitk, &strips) || (reinterpret_cast<uintptr_t>(x) & 15) != 0)
    return INC_ERR_UNSUPPORTED;  // nothing launched: the caller issues inc_woq_gemm per module
  GemvBatch args;
  args.n = n;
  int64_t off = 0;
  int first = 0;
  for (int i = 0; i < n; ++i) {
    INC_CHECK_ARG(qweight[i] && scales[i] && qzeros[i] && y[i]);
    args.qweight[i] = (const uint32_t*)qweight[i];
    args.scales[i] = scales[i];
    args.qzeros[i] = (const uint32_t*)qzeros[i];
    args.bias[i] = bias ? (const uint16_t*)bias[i] : nullptr;
    args.y[i] = (uint16_t*)y[i];
    args.N[i] = N[i];
    args.part_off[i] = off;
    args.first[i] = first;
    off += (int64_t)splitk * M * N[i];
    first += (int)ceil_div64(N[i], 64);
  }
  for (int i = n; i <= GEMV_MAX_BATCH; ++i) args.first[i] = first;
  for (int i = n; i < GEMV_MAX_BATCH; ++i) { args.qweight[i] = nullptr; args.scales[i] = nullptr; args.qzeros[i] = nullptr; args.bias[i] = nullptr; args.y[i] = nullptr; args.N[i] = 0; args.part_off[i] = 0; }
  if (!workspace || workspace_bytes < WS_COUNTER_BYTES + off * 4) return INC_ERR_WORKSPACE;
  hipStream_t s = inc_s(stream);
  unsigned* counters = (unsigned*)workspace;
  float* part = (float*)((char*)workspace + WS_COUNTER_BYTES);
  const uint16_t* xp = (const uint16_t*)x;
  dim3 grid((unsigned)strips, (unsigned)splitk);
  const bool bf = xdtype == INC_BF16, g128 = g_shift == -1 || g_shift >= 7;
#define INC_GEMVM(F, GG, V, B) woq_gemv_w4_multi_kernel<F, GG, V, B><<<grid, 256, 0, s>>>(args, xp, part, counters, (int)M, K, g_shift, splitk)
#define INC_GEMVM2(F, GG) { if (M > 32) INC_GEMVM(F, GG, 4, 4); else if (M > 16) INC_GEMVM(F, GG, 4, 2); else if (vsteps == 4) INC_GEMVM(F, GG, 4, 1); else INC_GEMVM(F, GG, 8, 1); }
  if (bits == 8) {
#define INC_GEMVM8(F, GG) woq_gemv_w4_multi_kernel<F, GG, 4, 1, 8><<<grid, 256, 0, s>>>(args, xp, part, counters, (int)M, K, g_shift, splitk)
    if (bf) { if (g128) INC_GEMVM8(true, true); else INC_GEMVM8(true, false); }
    else { if (g128) INC_GEMVM8(false, true); else INC_GEMVM8(false, false); }
#undef INC_GEMVM8
    INC_LAUNCH_RETURN();
  }
  if (bf) { if (g128) INC_GEMVM2(true, true) else INC_GEMVM2(true, false) }
  else { if (g128) INC_GEMVM2(false, true) else INC_GEMVM2(false, false) }
#undef INC_GEMVM2
#undef INC_GEMVM
  INC_LAUNCH_RETURN();
}

}  // extern "C"
