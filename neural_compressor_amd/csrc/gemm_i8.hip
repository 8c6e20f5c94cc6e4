// gemm_i8.hip -- K14: W8A8 GEMM for SmoothQuant (BASELINE config #4), v_mfma_i32_32x32x32_i8, MFMA-bound.
//
//   y[m, n] = alpha[n] * ( sum_k xq[m,k] * wq[n,k]  +  corr[n] ) + bias[n]
//
// xq [M, K] int8 (the uint8 activation codes minus 128, inc_sq_quant_act), wq [N, K] int8 per-output-channel codes
// (inc_sq_quant_weight), alpha[n] = s_x * s_w[n], corr[n] = (128 - zp_x) * rowsum(wq)[n]: together the exact integer form
// of  F.linear(quant_dequant_x_v1(X * input_scale), quant_dequant_w_v1(W))  -- the fake-quant pair the reference keeps
// in tree (smooth_quant/utility.py:652-755); the reference itself executes W8A8 through intel_extension_for_pytorch
// (smooth_quant/smooth_quant.py:105-125), which is not part of /root/reference, so this formula is the specification.
//
// Tile 256 (M) x 256 (N) x 128 (K bytes) per workgroup, 512 threads = 8 waves as 2 (M) x 4 (N), a wave owns 128 x 64 = 4 x 2
// MFMA tiles (128 int32 accumulators).  Both operands are K-contiguous int8, so both tiles are 256 rows x 128 B and use the
// SAME path as the x tile of the 4-bit kernel (gemm.hip): LDS-DMA, 8 full 128-byte rows per instruction, 16-byte chunk
// index XOR-ed with (row >> 1) & 7 on the source address and again on the ds_read_b128 side (no bank conflicts).  Two 64 KiB
// stages: the DMA of tile t+1 is issued at the top of step t.  W is the A operand and x the B operand, as in gemm.hip, so a
// lane owns 4 consecutive output columns and the epilogue stores 8 bytes.
// Tile quantisation: the tiles of the last, partly filled round of workgroups (or all of them when there are fewer tiles
// than CUs) are split along K into `split` workgroups each; every split stores its int32 tile into its own slab of the
// caller's workspace and a second, small kernel adds the slabs and runs the epilogue for those tiles.  (A single-kernel
// variant in which the last split to arrive finished the tile -- agent-scope release / acquire around an atomic ticket --
// was measured 1.5x SLOWER than not splitting at all: every release writes back the XCD's whole dirty L2, which at that
// moment holds the other workgroups' output tiles.)  Integer addition is associative: bit-identical to the unsplit kernel.
// Algorithmic work per launch: 2*M*N*K int8 op; bytes M*K + N*K + 2*M*N (+ 8*N).
#include "common.hpp"

namespace {

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) int i32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2_;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_;

constexpr int IM = 256, IN = 256, IK = 128;
constexpr int I_TILE = 256 * IK;        // 32 KiB: one operand tile
constexpr int I_STAGE = 2 * I_TILE;     // x tile + w tile
constexpr int I_CUS = 256;              // workgroups per round (one per CU: 128 KiB of LDS each)
constexpr int64_t I_COUNTER_BYTES = 0;              // (no tickets: the tail tiles are finished by a second kernel)
constexpr int64_t I_SLAB_BYTES = (int64_t)IM * IN * 4;  // one int32 tile

// how the tail round is split: returns the number of K-splits (1 = none) and the count of unsplit ("full") tiles
static inline int i8_tail_plan(int64_t tiles, int nk, int64_t* full_tiles) {
  const int64_t tail = tiles % I_CUS;
  *full_tiles = tiles - tail;
  if (tail == 0) { return 1; }
  int split = (int)(I_CUS / tail);
  if (split > 8) split = 8;
  // measured (profiles/r1j): 27 K-steps per split gain 14 %, 10 per split lose 8 % (prologue + 256 KiB slab + second kernel).
  // With fewer tiles than a quarter of the CUs (medium M: batched decode) most of the chip would sit idle, and only the
  // 32-row blocks that hold real rows are parked in the slabs, so short splits pay there.
  const int min_steps = tiles < I_CUS / 4 ? 4 : 24;
  while (split > 1 && nk / split < min_steps) --split;
  if (split <= 1) { *full_tiles = tiles; return 1; }
  return split;
}

template <bool IS_BF16>
__device__ __forceinline__ uint32_t cvt_pair16(float a, float b) {
  f32x2_ f = {a, b};
  uint32_t r;
  if constexpr (IS_BF16) {
    bf16x2_ h = __builtin_convertvector(f, bf16x2_);
    __builtin_memcpy(&r, &h, 4);
  } else {
    f16x2_ h = __builtin_convertvector(f, f16x2_);
    __builtin_memcpy(&r, &h, 4);
  }
  return r;
}

__device__ __forceinline__ i32x16 mfma_i8(const uint4& a, const uint4& b, i32x16 c) {
  i32x4 av, bv;
  __builtin_memcpy(&av, &a, 16);
  __builtin_memcpy(&bv, &b, 16);
  return __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, c, 0, 0, 0);
}

template <bool IS_BF16>
__global__ __launch_bounds__(512) void w8a8_gemm_kernel(const int8_t* __restrict__ xq, const int8_t* __restrict__ wq,
                                                        const float* __restrict__ alpha, const int32_t* __restrict__ corr,
                                                        const uint16_t* __restrict__ bias, uint16_t* __restrict__ y,
                                                        int64_t M, int64_t N, int64_t K, int y_vec_ok, int full_tiles,
                                                        int split, int* __restrict__ slabs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles_n = (int)((N + IN - 1) / IN);
  const int tiles_m = (int)((M + IM - 1) / IM);
  int wg = blockIdx.x, ks = 0, tail_id = -1;
  if (wg < full_tiles) {
    const int q = full_tiles / 8, xcd = wg % 8, idx = wg / 8;  // full_tiles is a multiple of 256 (or all tiles when no split)
    const int r = full_tiles % 8;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;  // bijective XCD remap (block b runs on XCD b % 8)
  } else {
    const int j = wg - full_tiles;
    tail_id = j / split;
    ks = j - tail_id * split;
    wg = full_tiles + tail_id;
  }
  int tm, tn;
  banded_tile_decode(wg, tiles_m, tiles_n, tm, tn);
  const int64_t m0 = (int64_t)tm * IM, n0 = (int64_t)tn * IN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;

  // DMA source offsets: instruction i of this wave brings rows (wave*4+i)*8 .. +7; a lane brings chunk (lane&7)^sw of row lane>>3
  uint32_t xoff[4], woff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int R = (wave * 4 + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((R >> 1) & 7);
    int64_t xr = m0 + R, wr = n0 + R;
    if (xr > M - 1) xr = M - 1;
    if (wr > N - 1) wr = N - 1;
    xoff[i] = (uint32_t)((xr - m0) * K + 16 * c);
    woff[i] = (uint32_t)((wr - n0) * K + 16 * c);
  }
  const int8_t* const xtile = xq + m0 * K;
  const int8_t* const wtile = wq + n0 * K;
  const int nk_all = (int)(K / IK);
  const int steps = tail_id >= 0 ? (nk_all + split - 1) / split : nk_all;
  const int kbase = ks * steps;
  const int nk = min(steps, nk_all - kbase);
  auto issue = [&](int kt, int stage) {
    kt = kbase + (kt > nk - 1 ? nk - 1 : kt);
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + stage * I_STAGE + wave * 4096);
    lds_dma_4x1k(xtile + (int64_t)kt * IK, dst, xoff[0], xoff[1], xoff[2], xoff[3]);
    lds_dma_4x1k(wtile + (int64_t)kt * IK, dst + I_TILE, woff[0], woff[1], woff[2], woff[3]);
  };

  i32x16 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0;

  const int sw = ((lane & 31) >> 1) & 7, hi = lane >> 5;
  const int x_row = wm * 128 + (lane & 31);
  const int w_row = wn * 64 + (lane & 31);
  auto read_frags = [&](const char* S, int kk, uint4 (&xa)[4], uint4 (&wa)[2]) {
    const int chunk = ((2 * kk + hi) ^ sw) << 4;
#pragma unroll
    for (int mf = 0; mf < 4; ++mf) xa[mf] = *reinterpret_cast<const uint4*>(S + (x_row + 32 * mf) * 128 + chunk);
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) wa[nf] = *reinterpret_cast<const uint4*>(S + I_TILE + (w_row + 32 * nf) * 128 + chunk);
  };
  auto mma8 = [&](const uint4 (&xa)[4], const uint4 (&wa)[2]) {
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
      for (int mf = 0; mf < 4; ++mf) acc[nf][mf] = mfma_i8(wa[nf], xa[mf], acc[nf][mf]);
  };

  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  uint4 xX[4], wX[2], xY[4], wY[2];
#define INC_SB() __builtin_amdgcn_sched_barrier(0)
  for (int t = 0; t < nk; ++t) {
    const char* S = smem + (t & 1) * I_STAGE;
    issue(t + 1, (t + 1) & 1);  // the other stage was last read in step t-1, which every wave left through the barrier
    INC_SB();
    read_frags(S, 0, xX, wX);
    read_frags(S, 1, xY, wY);
    INC_SB();
    mma8(xX, wX);
    INC_SB();
    read_frags(S, 2, xX, wX);
    INC_SB();
    mma8(xY, wY);
    INC_SB();
    read_frags(S, 3, xY, wY);
    INC_SB();
    mma8(xX, wX);
    INC_SB();
    mma8(xY, wY);
    INC_SB();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tile t+1 has landed (this wave's share; the barrier covers the rest)
    __builtin_amdgcn_s_barrier();
  }
#undef INC_SB

  if (tail_id >= 0) {
    // K-split tile: park this split's int32 tile (register order: coalesced); w8a8_tail_finish_kernel does the rest
    int* const mine = slabs + ((int64_t)tail_id * split + ks) * (IM * IN);
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
      for (int mf = 0; mf < 4; ++mf) {
        if (m0 + wm * 128 + mf * 32 >= M) continue;  // a 32-row block without a real row is neither stored nor read back
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[((nf * 4 + mf) * 16 + r) * 512 + tid] = acc[nf][mf][r];
      }
    return;
  }

  // epilogue: D row i = n-offset (r&3) + 8*(r>>2) + 4*(lane>>5), col j = m-offset lane&31
#pragma unroll
  for (int nf = 0; nf < 2; ++nf) {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int64_t nb = n0 + wn * 64 + nf * 32 + 8 * rq + 4 * (lane >> 5);
      float al[4] = {0.f, 0.f, 0.f, 0.f}, bv[4] = {0.f, 0.f, 0.f, 0.f};
      int cr[4] = {0, 0, 0, 0};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (nb + e < N) {
          al[e] = alpha[nb + e];
          if (corr) cr[e] = corr[nb + e];
          if (bias) bv[e] = IS_BF16 ? bf16_bits_to_f32(bias[nb + e]) : f16_bits_to_f32(bias[nb + e]);
        }
#pragma unroll
      for (int mf = 0; mf < 4; ++mf) {
        const int64_t m = m0 + wm * 128 + mf * 32 + (lane & 31);
        if (m >= M) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = al[e] * (float)(acc[nf][mf][4 * rq + e] + cr[e]) + bv[e];
        uint16_t* dst = y + m * N + nb;
        if (y_vec_ok && nb + 4 <= N) {
          *reinterpret_cast<uint2*>(dst) = make_uint2(cvt_pair16<IS_BF16>(v[0], v[1]), cvt_pair16<IS_BF16>(v[2], v[3]));
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (nb + e < N) dst[e] = IS_BF16 ? f32_to_bf16_bits(v[e]) : f32_to_f16_bits(v[e]);
        }
      }
    }
  }
}

// ---- M <= 16 (decode): HBM-bound on the int8 weights, v_dot4_i32_i8 on the vector ALUs -------------------------------------
// A wave streams GV_ROWS weight rows at a time, 16 bytes per lane per row per iteration (1 KiB contiguous per row: full
// lines); the activations of the current K chunk (M x 4 KiB) sit in LDS and are read once per iteration for all rows
// (one ds_read_b128 per activation row).  int32 partial sums stay in registers, a xor-butterfly folds the 64 lanes at
// the end.  Algorithmic bytes: N*K (weights once) + M*K + 2*M*N.
constexpr int GV_ROWS = 4;     // weight rows per wave
constexpr int GV_KC = 4096;    // bytes of K staged in LDS per chunk
constexpr int GV_MAXM = 16;

template <bool IS_BF16, int MT>
__global__ __launch_bounds__(256) void w8a8_gemv_kernel(const int8_t* __restrict__ xq, const int8_t* __restrict__ wq,
                                                        const float* __restrict__ alpha, const int32_t* __restrict__ corr,
                                                        const uint16_t* __restrict__ bias, uint16_t* __restrict__ y, int M,
                                                        int64_t N, int64_t K) {
  __shared__ __attribute__((aligned(16))) int8_t xs[MT * GV_KC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t n0 = ((int64_t)blockIdx.x * 4 + wave) * GV_ROWS;
  const int8_t* wrow[GV_ROWS];
#pragma unroll
  for (int r = 0; r < GV_ROWS; ++r) {
    int64_t n = n0 + r;
    if (n > N - 1) n = N - 1;
    wrow[r] = wq + n * K;
  }
  int acc[GV_ROWS][MT];
#pragma unroll
  for (int r = 0; r < GV_ROWS; ++r)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[r][m] = 0;

  for (int64_t kc = 0; kc < K; kc += GV_KC) {
    const int len = (int)((K - kc) < GV_KC ? (K - kc) : GV_KC);  // multiple of 16
    __syncthreads();
    for (int i = tid; i < MT * (len / 16); i += 256) {
      const int m = i / (len / 16), c = i - m * (len / 16);
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (m < M) v = *reinterpret_cast<const uint4*>(xq + (int64_t)m * K + kc + 16 * c);
      *reinterpret_cast<uint4*>(xs + m * GV_KC + 16 * c) = v;
    }
    __syncthreads();
    for (int k0 = lane * 16; k0 < len; k0 += 1024) {
      uint4 w[GV_ROWS];
#pragma unroll
      for (int r = 0; r < GV_ROWS; ++r) {
        const i32x4 t = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(wrow[r] + kc + k0));  // streamed once
        w[r] = make_uint4((uint32_t)t[0], (uint32_t)t[1], (uint32_t)t[2], (uint32_t)t[3]);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const uint4 xv = *reinterpret_cast<const uint4*>(xs + m * GV_KC + k0);
#pragma unroll
        for (int r = 0; r < GV_ROWS; ++r) {
          int a = acc[r][m];
          a = __builtin_amdgcn_sdot4((int)w[r].x, (int)xv.x, a, false);
          a = __builtin_amdgcn_sdot4((int)w[r].y, (int)xv.y, a, false);
          a = __builtin_amdgcn_sdot4((int)w[r].z, (int)xv.z, a, false);
          a = __builtin_amdgcn_sdot4((int)w[r].w, (int)xv.w, a, false);
          acc[r][m] = a;
        }
      }
    }
  }
  // fold the 64 lanes; afterwards every lane holds every sum and lane (r * MT + m) stores y[m, n0 + r]
  int mine = 0;
#pragma unroll
  for (int r = 0; r < GV_ROWS; ++r)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      int v = acc[r][m];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if (lane == r * MT + m) mine = v;
    }
  if (lane < GV_ROWS * MT) {
    const int r = lane / MT, m = lane - r * MT;
    const int64_t n = n0 + r;
    if (n < N && m < M) {
      float v = alpha[n] * (float)(mine + (corr ? corr[n] : 0));
      if (bias) v += IS_BF16 ? bf16_bits_to_f32(bias[n]) : f16_bits_to_f32(bias[n]);
      y[(int64_t)m * N + n] = IS_BF16 ? f32_to_bf16_bits(v) : f32_to_f16_bits(v);
    }
  }
}

// tail tiles: sum the `split` slabs of a tile (same register-order layout: thread tid of the tile's workgroup owns element
// ((nf*4+mf)*16 + r)*512 + tid) and run the GEMM's epilogue.  One workgroup of 512 threads per tail tile.
template <bool IS_BF16>
__global__ __launch_bounds__(512) void w8a8_tail_finish_kernel(const int* __restrict__ slabs, const float* __restrict__ alpha,
                                                               const int32_t* __restrict__ corr, const uint16_t* __restrict__ bias,
                                                               uint16_t* __restrict__ y, int64_t M, int64_t N, int y_vec_ok,
                                                               int full_tiles, int split) {
  const int tiles_n = (int)((N + IN - 1) / IN);
  const int tiles_m = (int)((M + IM - 1) / IM);
  const int tail_id = blockIdx.x, wg = full_tiles + tail_id;
  int tm, tn;
  banded_tile_decode(wg, tiles_m, tiles_n, tm, tn);
  const int64_t m0 = (int64_t)tm * IM, n0 = (int64_t)tn * IN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int* base = slabs + (int64_t)tail_id * split * (IM * IN);
#pragma unroll
  for (int nf = 0; nf < 2; ++nf) {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int64_t nb = n0 + wn * 64 + nf * 32 + 8 * rq + 4 * (lane >> 5);
      float al[4] = {0.f, 0.f, 0.f, 0.f}, bv[4] = {0.f, 0.f, 0.f, 0.f};
      int cr[4] = {0, 0, 0, 0};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (nb + e < N) {
          al[e] = alpha[nb + e];
          if (corr) cr[e] = corr[nb + e];
          if (bias) bv[e] = IS_BF16 ? bf16_bits_to_f32(bias[nb + e]) : f16_bits_to_f32(bias[nb + e]);
        }
#pragma unroll
      for (int mf = 0; mf < 4; ++mf) {
        if (m0 + wm * 128 + mf * 32 >= M) continue;
        const int64_t m = m0 + wm * 128 + mf * 32 + (lane & 31);
        // all loads of a thread are independent: issue them together (split <= 8), then add in slab order
        int part[8][4];
#pragma unroll
        for (int o = 0; o < 8; ++o)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            part[o][e] = o < split ? __builtin_nontemporal_load(base + (int64_t)o * (IM * IN) + ((nf * 4 + mf) * 16 + 4 * rq + e) * 512 + tid) : 0;
        int a4[4] = {0, 0, 0, 0};
#pragma unroll
        for (int o = 0; o < 8; ++o)
#pragma unroll
          for (int e = 0; e < 4; ++e) a4[e] += part[o][e];
        if (m >= M) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = al[e] * (float)(a4[e] + cr[e]) + bv[e];
        uint16_t* dst = y + m * N + nb;
        if (y_vec_ok && nb + 4 <= N) {
          *reinterpret_cast<uint2*>(dst) = make_uint2(cvt_pair16<IS_BF16>(v[0], v[1]), cvt_pair16<IS_BF16>(v[2], v[3]));
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (nb + e < N) dst[e] = IS_BF16 ? f32_to_bf16_bits(v[e]) : f32_to_f16_bits(v[e]);
        }
      }
    }
  }
}

}  // namespace

extern "C" {

int64_t inc_w8a8_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= GV_MAXM || N <= 0 || K <= 0 || (K % IK) != 0) return 0;
  const int64_t tiles = ceil_div64(M, IM) * ceil_div64(N, IN);
  int64_t full;
  const int split = i8_tail_plan(tiles, (int)(K / IK), &full);
  if (split <= 1) return 0;
  return I_COUNTER_BYTES + (tiles - full) * split * I_SLAB_BYTES;
}

int inc_w8a8_gemm(const int8_t* xq, const int8_t* wq, const float* alpha, const int32_t* corr, const void* bias, void* y,
                  int ydtype, int64_t M, int64_t N, int64_t K, void* workspace, int64_t workspace_bytes,
                  inc_stream_t stream) {
  INC_CHECK_ARG(xq && wq && alpha && y && M > 0 && N > 0 && K > 0);
  if (ydtype != INC_BF16 && ydtype != INC_F16) return INC_ERR_UNSUPPORTED;
  if ((K % IK) != 0) return INC_ERR_UNSUPPORTED;  // the module pads K to a multiple of 128 with zero weight codes
  INC_CHECK_ARG(((reinterpret_cast<uintptr_t>(xq) | reinterpret_cast<uintptr_t>(wq)) & 15) == 0);
  INC_CHECK_ARG((int64_t)IM * K < ((int64_t)1 << 32));
  const size_t smem = (size_t)2 * I_STAGE;  // 128 KiB
  static std::atomic<uint64_t> attr_set{0};
  if (inc_attr_needed(attr_set)) {
    (void)hipFuncSetAttribute((const void*)w8a8_gemm_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute((const void*)w8a8_gemm_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    inc_attr_done(attr_set);
  }
  if (M <= GV_MAXM) {  // decode: stream the weights once, dot products on the vector ALUs
    const unsigned g = (unsigned)ceil_div64(N, 4 * GV_ROWS);
    hipStream_t sv = inc_s(stream);
    const bool bf = ydtype == INC_BF16;
#define INC_GV(B, MT) w8a8_gemv_kernel<B, MT><<<g, 256, 0, sv>>>(xq, wq, alpha, corr, (const uint16_t*)bias, (uint16_t*)y, (int)M, N, K)
    if (M == 1) { if (bf) INC_GV(true, 1); else INC_GV(false, 1); }
    else if (M <= 4) { if (bf) INC_GV(true, 4); else INC_GV(false, 4); }
    else if (M <= 8) { if (bf) INC_GV(true, 8); else INC_GV(false, 8); }
    else { if (bf) INC_GV(true, 16); else INC_GV(false, 16); }
#undef INC_GV
    INC_LAUNCH_RETURN();
  }
  const int64_t tiles = ceil_div64(M, IM) * ceil_div64(N, IN);
  int64_t full = tiles;
  int split = i8_tail_plan(tiles, (int)(K / IK), &full);
  const int64_t need = split > 1 ? I_COUNTER_BYTES + (tiles - full) * split * I_SLAB_BYTES : 0;
  if (split > 1 && (!workspace || workspace_bytes < need)) {
    split = 1;  // no (or too small a) workspace: one workgroup per tile, still correct
    full = tiles;
  }
  const unsigned grid = (unsigned)(full + (tiles - full) * split);
  const int y_vec_ok = (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) & 7) == 0);
  int* slabs = (int*)workspace;
  hipStream_t s = inc_s(stream);
  const unsigned tail = (unsigned)(tiles - full);
  if (ydtype == INC_BF16) {
    w8a8_gemm_kernel<true><<<grid, 512, smem, s>>>(xq, wq, alpha, corr, (const uint16_t*)bias, (uint16_t*)y, M, N, K, y_vec_ok,
                                                   (int)full, split, slabs);
    if (split > 1)
      w8a8_tail_finish_kernel<true><<<tail, 512, 0, s>>>(slabs, alpha, corr, (const uint16_t*)bias, (uint16_t*)y, M, N, y_vec_ok, (int)full, split);
  } else {
    w8a8_gemm_kernel<false><<<grid, 512, smem, s>>>(xq, wq, alpha, corr, (const uint16_t*)bias, (uint16_t*)y, M, N, K, y_vec_ok,
                                                    (int)full, split, slabs);
    if (split > 1)
      w8a8_tail_finish_kernel<false><<<tail, 512, 0, s>>>(slabs, alpha, corr, (const uint16_t*)bias, (uint16_t*)y, M, N, y_vec_ok, (int)full, split);
  }
  INC_LAUNCH_RETURN();
}

}  // extern "C"
