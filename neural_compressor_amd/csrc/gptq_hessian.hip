// gptq_hessian.hip -- K5: GPTQ Hessian accumulation (MFMA syrk) and its finalisation (mirror, dead columns, damping).
//
// Reference (relative to /root/reference/neural_compressor/torch/algorithms/weight_only/gptq.py):
//   GPTQ.add_batch      :1111-1141   H <- H*n/(n+b) + (sqrt(2/(n+b)) X)^T (sqrt(2/(n+b)) X)
//   GPTQ.fasterquant    :1186-1189, 1221-1227   dead columns, damping
//
// Kernels
//   hessian_syrk_16bit  bf16/f16 MFMA 32x32x16, fp32 accumulate, 128x128 tile of H per workgroup, upper
//                       triangle of tiles only.  X is [T,K] row-major (tokens x features) so both MFMA
//                       operands are X^T: each thread fetches an 8(token) x 8(feature) block with eight
//                       16-byte row loads (full 128-byte segments per row), transposes it in registers
//                       and writes eight 16-byte [feature][token] rows into LDS (pitch 144 B: both the
//                       ds_write_b128 and the fragment ds_read_b128 are bank-conflict free).
//   hessian_syrk_tr_256 the product tile: 256 x 256 of H per workgroup, X staged untouched by LDS-DMA, fragments by
//                       ds_read_b64_tr_b16 (see below); single and batched (all Hessians of a forward) launches.
//   hessian_syrk_f32    exact fp32 MFMA 32x32x2 (A/B = one f32 per lane: no transpose needed).
// (The column loop lives in gptq.hip; the two were one translation unit until round 6 -- this half alone compiles in a minute.)
#include <math.h>

#include <type_traits>

#include <algorithm>

#include "common.hpp"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// ---------------------------------------------------------------------------------------------
// Hessian: H <- beta*H + alpha*X^T X   (16-bit inputs)
// ---------------------------------------------------------------------------------------------
constexpr int HB = 128;          // H tile edge (features)
constexpr int HK = 64;           // tokens per pipeline step
constexpr int HPITCH = HK + 8;   // LDS row pitch in elements (144 B)

__device__ __forceinline__ void tri_decode(int idx, int nt, int& ti, int& tj) {
  // idx -> (ti, tj), tj >= ti, row-major over the upper triangle
  int t = 0, rem = idx;
  while (rem >= nt - t) { rem -= nt - t; ++t; }
  ti = t;
  tj = t + rem;
}

// Tile order for the 256x256 syrk: workgroup b runs on XCD b % 8 (private 4 MiB L2 each).  Give every XCD a
// CONTIGUOUS run of the logical tile sequence (bijective remap), and make that sequence walk the upper triangle in
// 8x8-tile super-tiles, so the ~32 tiles resident on one XCD at a time share <= 8 + 8 feature panels of X instead of
// touching ~40 different ones: the panels are then re-read from that XCD's L2, not from HBM / Infinity Cache
// (profiles/r1_pmc: 0.5-1 GB fetched per launch for 45 MB of X with the plain row-major order).
__device__ __forceinline__ void xcd_supertile_decode(int b, int n, int nt, int& ti, int& tj) {
  const int q = n / 8, r = n % 8, xcd = b % 8, t = b / 8;
  int rem = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + t;  // logical index
  constexpr int S = 8;
  const int ns = (nt + S - 1) / S;
  for (int si = 0; si < ns; ++si) {
    const int h = min(S, nt - si * S);
    for (int sj = si; sj < ns; ++sj) {
      const int w = min(S, nt - sj * S);
      const int count = si == sj ? h * (h + 1) / 2 : h * w;
      if (rem < count) {
        if (si != sj) {
          ti = si * S + rem / w;
          tj = sj * S + rem % w;
        } else {
          int row = 0;
          while (rem >= h - row) { rem -= h - row; ++row; }
          ti = si * S + row;
          tj = si * S + row + rem;
        }
        return;
      }
      rem -= count;
    }
  }
  ti = tj = 0;  // unreachable for b < n
}

// one thread's 8(token) x 8(feature) block of X, zero-filled out of range
__device__ __forceinline__ void load_block8x8(const uint16_t* __restrict__ x, int64_t T, int64_t K,
                                              int64_t ldx, int64_t t0, int64_t f0, bool vec,
                                              uint4 (&r)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t t = t0 + i;
    if (t < T && vec && f0 + 8 <= K) {
      r[i] = *reinterpret_cast<const uint4*>(x + t * ldx + f0);
    } else {
      uint16_t e[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) e[c] = (t < T && f0 + c < K) ? x[t * ldx + f0 + c] : (uint16_t)0;
      r[i] = make_uint4((uint32_t)e[0] | ((uint32_t)e[1] << 16), (uint32_t)e[2] | ((uint32_t)e[3] << 16),
                        (uint32_t)e[4] | ((uint32_t)e[5] << 16), (uint32_t)e[6] | ((uint32_t)e[7] << 16));
    }
  }
}

// transpose the 8x8 16-bit block held as r[token][4 dwords] and store 8 rows [feature][8 tokens]
__device__ __forceinline__ void store_block_transposed(uint16_t* lds, int f_local, int t_local,
                                                       const uint4 (&r)[8]) {
  const uint32_t* w = reinterpret_cast<const uint32_t*>(&r[0]);  // w[token*4 + m]
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    uint32_t lo[4], hi[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const uint32_t a = w[(2 * p) * 4 + m], b = w[(2 * p + 1) * 4 + m];
      lo[p] = (a & 0xffffu) | (b << 16);
      hi[p] = (a >> 16) | (b & 0xffff0000u);
    }
    *reinterpret_cast<uint4*>(lds + (f_local + 2 * m) * HPITCH + t_local) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    *reinterpret_cast<uint4*>(lds + (f_local + 2 * m + 1) * HPITCH + t_local) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
  }
}

template <bool IS_BF16>
__device__ __forceinline__ f32x16 mfma16(const uint4& a, const uint4& b, f32x16 c) {
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
__global__ __launch_bounds__(256) void hessian_syrk_16bit_kernel(const uint16_t* __restrict__ x,
                                                                 int64_t T, int64_t K, int64_t ldx,
                                                                 float* __restrict__ H, float beta,
                                                                 float alpha, int nt, int vec_ok) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  uint16_t* smem = reinterpret_cast<uint16_t*>(smem_raw);
  // [stage][operand A/B][HB][HPITCH]
  constexpr int OPER = HB * HPITCH;
  int ti, tj;
  tri_decode(blockIdx.x, nt, ti, tj);
  const int64_t i0 = (int64_t)ti * HB, j0 = (int64_t)tj * HB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;  // 2x2 waves, 64x64 each
  // staging role: threads 0..127 fetch the A (i) tile, 128..255 the B (j) tile
  const int oper = tid >> 7, tt = tid & 127;
  const int t_chunk = tt & 7, f_chunk = tt >> 3;
  const int64_t fbase = (oper == 0 ? i0 : j0) + f_chunk * 8;
  const bool vec = vec_ok != 0;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nk = (int)((T + HK - 1) / HK);
  uint4 regs[8];
  load_block8x8(x, T, K, ldx, (int64_t)t_chunk * 8, fbase, vec, regs);
  store_block_transposed(smem + (0 * 2 + oper) * OPER, f_chunk * 8, t_chunk * 8, regs);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_block8x8(x, T, K, ldx, (int64_t)(kt + 1) * HK + t_chunk * 8, fbase, vec, regs);
    const uint16_t* As = smem + (cur * 2 + 0) * OPER + (wr * 64) * HPITCH;
    const uint16_t* Bs = smem + (cur * 2 + 1) * OPER + (wc * 64) * HPITCH;
#pragma unroll
    for (int kk = 0; kk < HK / 16; ++kk) {
      const int koff = kk * 16 + 8 * (lane >> 5);
      uint4 a[2], b[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        a[m] = *reinterpret_cast<const uint4*>(As + (m * 32 + (lane & 31)) * HPITCH + koff);
        b[m] = *reinterpret_cast<const uint4*>(Bs + (m * 32 + (lane & 31)) * HPITCH + koff);
      }
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = mfma16<IS_BF16>(a[m], b[n], acc[m][n]);
    }
    if (kt + 1 < nk) store_block_transposed(smem + ((cur ^ 1) * 2 + oper) * OPER, f_chunk * 8, t_chunk * 8, regs);
    __syncthreads();
  }

  // epilogue: D[row][col], col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int64_t col = j0 + wc * 64 + n * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = i0 + wr * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < K && col < K) {
          float* p = H + row * K + col;
          *p = beta * (*p) + alpha * acc[m][n][r];
        }
      }
    }
}

constexpr int H2 = 256;  // H tile edge of the 256 x 256 kernels

#ifdef INC_KBENCH  // first 256 x 256 generation (operands transposed in registers): harness flag 45, A/B partner of the kernel below
#include "../../tools/kbench_gptq_1.inc"
#endif  // INC_KBENCH

// ---- 256x256 syrk tile, transpose-read generation (round 2) --------------------------------------------------------
// The kernel above moves X through registers: per 64-token step every thread issues eight 16-byte loads, transposes an 8x8
// block with 32 v_perm_b32 and writes eight 16-byte rows to LDS -- more VALU / VMEM / LDS-store issue than the 32 MFMAs they
// feed (PMC: matrix pipe busy 0.39).  Here the X tile goes to LDS untouched and untransposed: one LDS-DMA instruction per
// token row ([i-panel 512 B][j-panel 512 B], lanes 0-31 / 32-63 read the two panels), no VGPR, no VALU, no ds_write; the
// fragments are read with ds_read_b64_tr_b16, which hands lane (4a + e) of a 16-lane group element e of lanes a, a+4, a+8,
// a+12 (tools/kbench probe): with lane (a + 4b) pointing at token row b, feature block 4a, a lane receives 4 consecutive
// tokens of ONE feature -- the K-contiguous operand of v_mfma_f32_16x16x32 -- for both operands of X^T X.  Row pitch 1056 B
// (1024 + 32 = 8 dwords past a multiple of 64 banks): a 32-lane half of a transpose-read touches 8 token rows x 32 B, which
// are conflict-free when the rows are CONSECUTIVE (8 different 32-byte bank slots).  The MFMA K index only has to pair the
// same token in both operands, so group g of 16 lanes takes tokens 4g .. 4g+3 (first read) and 16+4g .. 16+4g+3 (second
// read) of a 32-token sub-step instead of the 8g .. 8g+7 of the operand's natural order, whose two halves (rows 0-3 and
// 8-11) share bank slots.  Stages: TOK tokens each, NST of them, the DMA runs NST-1 steps ahead with counted vmcnt waits;
// (64, 2) is the product configuration (132 KiB; (32, 4) pays twice the barriers and measured 5 % slower).
constexpr int TR_PITCH = 1056;
constexpr int TR_TOK = 64;
constexpr int TR_NST = 2;
constexpr int TR_STAGE = TR_TOK * TR_PITCH;  // 67 584 B
// The tile below is the product form (rounds 4-5: tools/kbench hessian / hpf, profiles/r5/kbench_hessian*.log): the step's 16 MFMA rows run
// as ONE rolling fragment pipeline, waves 4-7 carry static priority, and a step's eight LDS-DMA pieces are issued from one asm block
// (per-lane row offsets computed once per tile; ~45 instead of ~200 instructions per wave and step: 9.79 -> 9.22 ms per batched launch).
// The other generations of this tile -- stage shapes, issue orders, and the timing-only ablations that produce WRONG results -- are
// harness code: tools/kbench_gptq_tile_lab.inc (same requests and MFMAs in the same order wherever results are correct: bit-identical H).
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

// `slab` != nullptr: the raw sums of this token range go to slab[256][256] instead of into H (a tile of the launch's last, partly
// filled round computed by several workgroups: hessian_tail_finalize_kernel adds the ranges in order)
template <bool IS_BF16>
__device__ __forceinline__ void hessian_syrk_tr_tile(const uint16_t* __restrict__ x, int64_t T, int64_t K, int64_t ldx,
                                                     float* __restrict__ H, float beta, float alpha, int nt, int block, int nblocks,
                                                     float* __restrict__ slab = nullptr) {
  constexpr int TOK = TR_TOK, NST = TR_NST;
  static_assert(NST == 2 && TOK == 64, "two 64-token stages: the DMA runs one step ahead, eight token rows per wave and step");
  constexpr int RPW = TOK / 8;      // DMA requests (token rows) per wave and step
  constexpr int STAGE = TOK * TR_PITCH;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  int ti, tj;
  xcd_supertile_decode(block, nblocks, nt, ti, tj);
  const int64_t i0 = (int64_t)ti * H2, j0 = (int64_t)tj * H2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem_raw;

  // DMA source: this lane's 16-byte chunk (8 features) of the i panel (lanes 0..31) or the j panel (lanes 32..63)
  int64_t f = (lane < 32 ? i0 : j0) + 8 * (lane & 31);
  if (f > K - 8) f = K - 8;  // K % 8 == 0 on this path: a chunk past K only feeds rows / columns that are never stored
  const uint32_t voff = (uint32_t)(f * 2);
  const int nk = (int)((T + TOK - 1) / TOK);
  // a step whose TOK token rows all exist (every step but a ragged last one) issues its eight pieces from ONE asm block -- the row stride
  // sits in per-lane offsets computed once per tile, so a piece is an M0 update and the request instead of ~25 instructions (a clamped
  // 64-bit row address, M0 save / restore, wait states); the ragged last step keeps the clamped per-piece form
  uint32_t voffr[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) voffr[i] = voff + (uint32_t)((int64_t)i * ldx * 2);
  auto issue = [&](int kt) {
    const int stage = kt & (NST - 1);
    if ((int64_t)(kt + 1) * TOK <= T) {
      const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + stage * STAGE + wave * RPW * TR_PITCH);
      lds_dma_8x1k<TR_PITCH>(x + ((int64_t)kt * TOK + wave * RPW) * ldx, dst, voffr[0], voffr[1], voffr[2], voffr[3], voffr[4], voffr[5],
                             voffr[6], voffr[7]);
      return;
    }
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int r = wave * RPW + i;
      int64_t t = (int64_t)kt * TOK + r;
      if (t > T - 1) t = T - 1;  // rows past T are zeroed in LDS before they are multiplied (below)
      const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + stage * STAGE + r * TR_PITCH);
      lds_dma_1k(x + t * ldx, dst, voff);
    }
  };

  f32x4_t acc[8][4];  // [i fragment of 16 rows][j fragment of 16 columns]
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment read address: lane (a + 4b) of 16-lane group g -> token row 4g + b (second read: 16 + 4g + b), feature block 4a
  const int s16 = lane & 15, fa = s16 & 3, fb = s16 >> 2, fg = lane >> 4;
  const uint32_t rbase = (uint32_t)((4 * fg + fb) * TR_PITCH + (wm * 128 + 4 * fa) * 2);        // + mt * 32
  const uint32_t cbase = (uint32_t)((4 * fg + fb) * TR_PITCH + 512 + (wn * 64 + 4 * fa) * 2);  // + nt * 32
  typedef __attribute__((address_space(3))) s16x4_t* lds_ptr_t;
  auto frag = [&](uint32_t byte_off) {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(uintptr_t)(byte_off));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(uintptr_t)(byte_off + 16 * TR_PITCH));
    uint4 v;
    __builtin_memcpy(&v, &lo, 8);
    __builtin_memcpy(reinterpret_cast<char*>(&v) + 8, &hi, 8);
    return v;
  };

  // static priority for the second-dispatched half of the workgroup (the younger wave of every SIMD loses each arbitration to the older
  // one, MI355X_MICROARCH.md "Two waves per SIMD")
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
  if (nk > 0) issue(0);
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & (NST - 1);
    if (kt + 1 < nk) issue(kt + 1);  // its stage was last read in step kt - 1 (barrier passed)
    if (kt == nk - 1 && (T % TOK) != 0) {
      // token tail: zero the rows past T of this (last) stage; its DMA has landed (counted wait + barrier of the previous step)
      const int first = (int)(T - (int64_t)kt * TOK);
      for (int idx = tid; idx < (TOK - first) * 64; idx += 512) {
        const int r = first + idx / 64, c = idx % 64;
        *reinterpret_cast<uint4*>(smem_raw + cur * STAGE + r * TR_PITCH + c * 16) = make_uint4(0u, 0u, 0u, 0u);
      }
      __syncthreads();
    }
    // ONE rolling pipeline over the step's 8 * TOK / 32 MFMA rows -- the row fragment of row r + 2 and (at row 4 of a 32-token sub-step)
    // the column fragments of the NEXT sub-step are requested before row r's MFMAs, so only the step's first six fragment reads are exposed
    {
      constexpr int NKK = TOK / 32, ROWS = 8 * NKK;
      const uint32_t st0 = lds0 + cur * STAGE;
      uint4 bq[2][4], aq[3];
#pragma unroll
      for (int n = 0; n < 4; ++n) bq[0][n] = frag(st0 + cbase + n * 32);
      aq[0] = frag(st0 + rbase);
      aq[1] = frag(st0 + rbase + 32);
#pragma unroll
      for (int row = 0; row < ROWS; ++row) {
        const int kk = row / 8, m = row % 8;
        if (row + 2 < ROWS) aq[(row + 2) % 3] = frag(st0 + ((row + 2) / 8) * 32 * TR_PITCH + rbase + ((row + 2) % 8) * 32);
        if (m == 4 && kk + 1 < NKK) {
#pragma unroll
          for (int n = 0; n < 4; ++n) bq[(kk + 1) & 1][n] = frag(st0 + (kk + 1) * 32 * TR_PITCH + cbase + n * 32);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          if constexpr (IS_BF16) {
            bf16x8 fa8, fb8;
            __builtin_memcpy(&fa8, &aq[row % 3], 16);
            __builtin_memcpy(&fb8, &bq[kk & 1][n], 16);
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa8, fb8, acc[m][n], 0, 0, 0);
          } else {
            f16x8 fa8, fb8;
            __builtin_memcpy(&fa8, &aq[row % 3], 16);
            __builtin_memcpy(&fb8, &bq[kk & 1][n], 16);
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa8, fb8, acc[m][n], 0, 0, 0);
          }
        }
      }
    }
    // every fragment read of this step has returned (the compiler sinks the last MFMAs below the barrier, so this is not
    // implied by program order) and step kt + 1 has landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
  }

  // epilogue: D[row i][col j] of a 16x16 fragment: col = lane & 15, row = 4 * (lane >> 4) + r
  if (slab) {
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          slab[(wm * 128 + m * 16 + 4 * (lane >> 4) + r) * H2 + wn * 64 + n * 16 + (lane & 15)] = acc[m][n][r];
    return;
  }
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int64_t col = j0 + wn * 64 + n * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = i0 + wm * 128 + m * 16 + 4 * (lane >> 4) + r;
        if (row < K && col < K) {
          float* p = H + row * K + col;
          *p = beta * (*p) + alpha * acc[m][n][r];
        }
      }
    }
}

template <bool IS_BF16>
__global__ __launch_bounds__(512) void hessian_syrk_tr_256_kernel(const uint16_t* __restrict__ x, int64_t T, int64_t K, int64_t ldx,
                                                                  float* __restrict__ H, float beta, float alpha, int nt) {
  hessian_syrk_tr_tile<IS_BF16>(x, T, K, ldx, H, beta, alpha, nt, (int)blockIdx.x, (int)gridDim.x);
}

#ifdef INC_KBENCH  // the tile's other generations and its timing-only ablations: harness code
#include "../../tools/kbench_gptq_tile_lab.inc"
#endif

#ifdef INC_KBENCH
template <bool IS_BF16, bool TAIL>
__global__ __launch_bounds__(512) void hessian_syrk_16bit_256_kernel(const uint16_t* __restrict__ x, int64_t T,
                                                                     int64_t K, int64_t ldx, float* __restrict__ H,
                                                                     float beta, float alpha, int nt) {
  hessian_syrk_256_tile<IS_BF16, TAIL>(x, T, K, ldx, H, beta, alpha, nt, (int)blockIdx.x, (int)gridDim.x);
}
#endif

// Several Hessians of ONE calibration forward in a single launch (same token count T): the three K = 4096 Hessians of a Llama
// block have 136 tiles each -- alone they leave 120 of the 256 CUs idle for the whole launch (0.395 of peak by the 2*T*K^2
// convention against 0.64 at K = 11008, profiles/r1l); together with the K = 11008 one they are 1354 tiles of equal length
// that the dispatcher spreads over the chip.  Problems occupy contiguous block ranges; inside its range a problem keeps its
// XCD-local super-tile order (the decode only needs the block's index modulo 8 to be constant per XCD, which a shifted range
// preserves).  Each tile is computed exactly as in the single-problem launch: bit-identical H.
constexpr int HESSIAN_MAX_BATCH = 8;
constexpr int HESSIAN_TAIL_UNITS = 256;  // workgroups of a split tail at most (= the workspace: 64 MiB of 256 x 256 fp32 tiles)
struct HessianBatch {
  const uint16_t* x[HESSIAN_MAX_BATCH];
  float* H[HESSIAN_MAX_BATCH];
  int64_t K[HESSIAN_MAX_BATCH];
  int64_t ldx[HESSIAN_MAX_BATCH];
  float beta[HESSIAN_MAX_BATCH];
  float alpha[HESSIAN_MAX_BATCH];
  int nt[HESSIAN_MAX_BATCH];
  int first[HESSIAN_MAX_BATCH + 1];  // first block of every problem, then the number of tiles
  int n;
  // the launch's last, partly filled round: tiles [full, first[n]) are computed by `nseg` workgroups each (token ranges, raw sums into
  // `slab`), hessian_tail_finalize_kernel folds them into H.  nseg == 1: every tile is one workgroup (full == first[n]).
  int full, nseg;
  float* slab;
  int block0;  // first workgroup of this launch (the call may issue its rounds of one-tile-per-CU as separate launches)
};

// token range of segment `seg` of `nseg`: whole TOK-token steps, the first (steps % nseg) segments one step longer
__device__ __forceinline__ void hessian_segment(int64_t T, int tok, int seg, int nseg, int64_t& t0, int64_t& tcount) {
  const int steps = (int)((T + tok - 1) / tok), q = steps / nseg, r = steps % nseg;
  const int s0 = seg * q + min(seg, r), s1 = s0 + q + (seg < r ? 1 : 0);
  t0 = (int64_t)s0 * tok;
  const int64_t t1 = min((int64_t)s1 * tok, T);
  tcount = t1 > t0 ? t1 - t0 : 0;
}

template <bool IS_BF16>
__global__ __launch_bounds__(512) void hessian_syrk_tr_256_multi_kernel(HessianBatch args, int64_t T) {
  const int gb = (int)blockIdx.x + args.block0;  // index in the whole call's grid
  int b = gb, seg = 0;
  const bool split = b >= args.full;
  if (split) {  // a unit of the split tail: tile full + u / nseg, token range u % nseg
    const int u = b - args.full;
    seg = u % args.nseg;
    b = args.full + u / args.nseg;
  }
  int p = 0;
#pragma unroll
  for (int i = 1; i < HESSIAN_MAX_BATCH; ++i)
    if (i < args.n && b >= args.first[i]) p = i;
  p = __builtin_amdgcn_readfirstlane(p);
  if (!split) {
    hessian_syrk_tr_tile<IS_BF16>(args.x[p], T, args.K[p], args.ldx[p], args.H[p], args.beta[p], args.alpha[p], args.nt[p],
                                  b - args.first[p], args.first[p + 1] - args.first[p]);
  } else {
    int64_t t0, tc;
    hessian_segment(T, TR_TOK, seg, args.nseg, t0, tc);
    float* slab = args.slab + ((int64_t)(gb - args.full)) * (H2 * H2);
    if (tc > 0)
      hessian_syrk_tr_tile<IS_BF16>(args.x[p] + t0 * args.ldx[p], tc, args.K[p], args.ldx[p], args.H[p], args.beta[p], args.alpha[p],
                                    args.nt[p], b - args.first[p], args.first[p + 1] - args.first[p], slab);
    else
      for (int i = threadIdx.x; i < H2 * H2; i += 512) slab[i] = 0.f;
  }
}

#ifdef INC_KBENCH  // the batched launch over the harness generations of the tile
template <bool IS_BF16, int TOK = TR_TOK, int NST = TR_NST, int ABL = TR_LAB_ABL>
__global__ __launch_bounds__(512) void hessian_syrk_tr_256_multi_lab_kernel(HessianBatch args, int64_t T) {
  const int gb = (int)blockIdx.x + args.block0;  // index in the whole call's grid
  int b = gb, seg = 0;
  const bool split = b >= args.full;
  if (split) {  // a unit of the split tail: tile full + u / nseg, token range u % nseg
    const int u = b - args.full;
    seg = u % args.nseg;
    b = args.full + u / args.nseg;
  }
  int p = 0;
#pragma unroll
  for (int i = 1; i < HESSIAN_MAX_BATCH; ++i)
    if (i < args.n && b >= args.first[i]) p = i;
  p = __builtin_amdgcn_readfirstlane(p);
  if (!split) {
    hessian_syrk_tr_tile_lab<IS_BF16, TOK, NST, ABL>(args.x[p], T, args.K[p], args.ldx[p], args.H[p], args.beta[p], args.alpha[p], args.nt[p],
                                  b - args.first[p], args.first[p + 1] - args.first[p]);
  } else {
    int64_t t0, tc;
    hessian_segment(T, TOK, seg, args.nseg, t0, tc);
    float* slab = args.slab + ((int64_t)(gb - args.full)) * (H2 * H2);
    if (tc > 0)
      hessian_syrk_tr_tile_lab<IS_BF16, TOK, NST, ABL>(args.x[p] + t0 * args.ldx[p], tc, args.K[p], args.ldx[p], args.H[p], args.beta[p], args.alpha[p],
                                    args.nt[p], b - args.first[p], args.first[p + 1] - args.first[p], slab);
    else
      for (int i = threadIdx.x; i < H2 * H2; i += 512) slab[i] = 0.f;
  }
}
#endif

// H tile <- beta * H + alpha * (range 0 + range 1 + ...), ranges added in order: the tiles of the split tail
// (four workgroups per tile, 64 rows each: a tile per workgroup left 182 of the 256 CUs without work for 0.1 ms per launch)
__global__ __launch_bounds__(512) void hessian_tail_finalize_kernel(HessianBatch args) {
  const int tile = (int)blockIdx.x >> 2, quarter = (int)blockIdx.x & 3;
  const int b = args.full + tile;
  int p = 0;
#pragma unroll
  for (int i = 1; i < HESSIAN_MAX_BATCH; ++i)
    if (i < args.n && b >= args.first[i]) p = i;
  p = __builtin_amdgcn_readfirstlane(p);
  int ti, tj;
  xcd_supertile_decode(b - args.first[p], args.first[p + 1] - args.first[p], args.nt[p], ti, tj);
  const int64_t K = args.K[p], i0 = (int64_t)ti * H2, j0 = (int64_t)tj * H2;
  float* __restrict__ H = args.H[p];
  const float beta = args.beta[p], alpha = args.alpha[p];
  const float* __restrict__ sl = args.slab + (int64_t)tile * args.nseg * (H2 * H2);
  for (int idx = quarter * (H2 * H2 / 4) + threadIdx.x * 4; idx < (quarter + 1) * (H2 * H2 / 4); idx += 512 * 4) {
    const int r = idx / H2, c = idx % H2;
    float4 sum = *reinterpret_cast<const float4*>(sl + idx);
    for (int s2 = 1; s2 < args.nseg; ++s2) {
      const float4 v = *reinterpret_cast<const float4*>(sl + (int64_t)s2 * (H2 * H2) + idx);
      sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
    const int64_t row = i0 + r, col = j0 + c;
    if (row < K) {
      float* hp = H + row * K + col;
      const float sv[4] = {sum.x, sum.y, sum.z, sum.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (col + e < K) hp[e] = beta * hp[e] + alpha * sv[e];
    }
  }
}

#ifdef INC_KBENCH
template <bool IS_BF16, bool TAIL>
__global__ __launch_bounds__(512) void hessian_syrk_16bit_256_multi_kernel(HessianBatch args, int64_t T) {
  const int b = (int)blockIdx.x;
  int p = 0;
#pragma unroll
  for (int i = 1; i < HESSIAN_MAX_BATCH; ++i)
    if (i < args.n && b >= args.first[i]) p = i;
  p = __builtin_amdgcn_readfirstlane(p);
  hessian_syrk_256_tile<IS_BF16, TAIL>(args.x[p], T, args.K[p], args.ldx[p], args.H[p], args.beta[p], args.alpha[p], args.nt[p],
                                       b - args.first[p], args.first[p + 1] - args.first[p]);
}
#endif

// ---- fp32 inputs: exact fp32 MFMA 32x32x2 ------------------------------------------------------
constexpr int FK = 32;  // tokens per step
__global__ __launch_bounds__(256) void hessian_syrk_f32_kernel(const float* __restrict__ x, int64_t T,
                                                               int64_t K, int64_t ldx,
                                                               float* __restrict__ H, float beta,
                                                               float alpha, int nt) {
  __shared__ float As[FK * HB];
  __shared__ float Bs[FK * HB];
  int ti, tj;
  tri_decode(blockIdx.x, nt, ti, tj);
  const int64_t i0 = (int64_t)ti * HB, j0 = (int64_t)tj * HB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int nk = (int)((T + FK - 1) / FK);
  for (int kt = 0; kt < nk; ++kt) {
    // stage [FK tokens][128 features] of both operands (coalesced along features)
    for (int idx = tid; idx < FK * HB; idx += 256) {
      const int t = idx / HB, f = idx - t * HB;
      const int64_t tg = (int64_t)kt * FK + t;
      As[idx] = (tg < T && i0 + f < K) ? x[tg * ldx + i0 + f] : 0.f;
      Bs[idx] = (tg < T && j0 + f < K) ? x[tg * ldx + j0 + f] : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int s = 0; s < FK / 2; ++s) {
      const int k = 2 * s + (lane >> 5);
      float a[2], b[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        a[m] = As[k * HB + wr * 64 + m * 32 + (lane & 31)];
        b[m] = Bs[k * HB + wc * 64 + m * 32 + (lane & 31)];
      }
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[n], acc[m][n], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int64_t col = j0 + wc * 64 + n * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = i0 + wr * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < K && col < K) {
          float* p = H + row * K + col;
          *p = beta * (*p) + alpha * acc[m][n][r];
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------
// finalize: dead columns + damping on the diagonal, then mirror upper -> lower
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void hessian_diag_kernel(float* __restrict__ H, int64_t K,
                                                            float percdamp, uint8_t* __restrict__ dead,
                                                            float* __restrict__ ws) {
  __shared__ float part[16];
  __shared__ float damp_s;
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < K; i += blockDim.x) {
    float d = H[i * K + i];
    const bool is_dead = (d == 0.f);
    if (is_dead) { d = 1.f; H[i * K + i] = 1.f; }
    if (dead) dead[i] = is_dead ? 1 : 0;
    acc += d;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += part[i];
    damp_s = percdamp * (s / (float)K);
    if (ws) ws[0] = damp_s;
  }
  __syncthreads();
  const float damp = damp_s;
  for (int64_t i = threadIdx.x; i < K; i += blockDim.x) H[i * K + i] += damp;
}

__global__ __launch_bounds__(256) void hessian_mirror_kernel(float* __restrict__ H, int64_t K) {
  __shared__ float tile[32][33];
  const int ti = blockIdx.y, tj = blockIdx.x;
  if (tj < ti) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int64_t row = (int64_t)ti * 32 + r, col = (int64_t)tj * 32 + tx;
    tile[r][tx] = (row < K && col < K) ? H[row * K + col] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int64_t row = (int64_t)tj * 32 + r, col = (int64_t)ti * 32 + tx;  // transposed position
    if (row < K && col < K && row > col) H[row * K + col] = tile[tx][r];
  }
}

}  // namespace

extern "C" {

int inc_gptq_hessian_accum(const void* x, int xdtype, int64_t T, int64_t K, int64_t ldx, float* H,
                           float beta, float alpha, inc_stream_t stream) {
  INC_CHECK_ARG(x && H && T > 0 && K > 0 && ldx >= K);
  const int nt = (int)ceil_div64(K, HB);
  const int ntiles = nt * (nt + 1) / 2;
  hipStream_t s = inc_s(stream);
  if (xdtype == INC_F32) {
    hessian_syrk_f32_kernel<<<ntiles, 256, 0, s>>>((const float*)x, T, K, ldx, H, beta, alpha, nt);
  } else if (xdtype == INC_BF16 || xdtype == INC_F16) {
    const int vec_ok = (ldx % 8 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    const size_t smem = (size_t)2 * 2 * HB * HPITCH * sizeof(uint16_t);
    static std::atomic<uint64_t> attr_set{0};
    if (inc_attr_needed(attr_set)) {
      (void)hipFuncSetAttribute((const void*)hessian_syrk_16bit_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      (void)hipFuncSetAttribute((const void*)hessian_syrk_16bit_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      inc_attr_done(attr_set);
    }
    if (K >= H2 && vec_ok && (K % 8) == 0 && !inc_force_small_tiles()) {
      const int nt2 = (int)ceil_div64(K, H2);
      const int ntiles2 = nt2 * (nt2 + 1) / 2;
      const uint16_t* xp = (const uint16_t*)x;
#ifdef INC_KBENCH
      if (inc_small_tiles_flag(-1) == 45) {  // harness flag 45: the register-transposing generation
        const size_t smem2 = (size_t)2 * H2_STAGE * sizeof(uint16_t);  // 144 KiB
        const bool tail = (T % HK) != 0;
        (void)hipFuncSetAttribute((const void*)hessian_syrk_16bit_256_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2);
        (void)hipFuncSetAttribute((const void*)hessian_syrk_16bit_256_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2);
        (void)hipFuncSetAttribute((const void*)hessian_syrk_16bit_256_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2);
        (void)hipFuncSetAttribute((const void*)hessian_syrk_16bit_256_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2);
#define INC_H2(B, TL) hessian_syrk_16bit_256_kernel<B, TL><<<ntiles2, 512, smem2, s>>>(xp, T, K, ldx, H, beta, alpha, nt2)
        if (xdtype == INC_BF16) { if (tail) INC_H2(true, true); else INC_H2(true, false); }
        else { if (tail) INC_H2(false, true); else INC_H2(false, false); }
#undef INC_H2
        INC_LAUNCH_RETURN();
      }
#endif
      {  // transpose-read generation
        const size_t smem3 = (size_t)TR_NST * TR_STAGE;  // 132 KiB
        static std::atomic<uint64_t> attr3_set{0};
        if (inc_attr_needed(attr3_set)) {
          (void)hipFuncSetAttribute((const void*)hessian_syrk_tr_256_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem3);
          (void)hipFuncSetAttribute((const void*)hessian_syrk_tr_256_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem3);
          inc_attr_done(attr3_set);
        }
#ifdef INC_KBENCH
        if (inc_small_tiles_flag(-1) == 46 && xdtype == INC_BF16) {  // timing A/B: four 32-token stages
          (void)hipFuncSetAttribute((const void*)hessian_syrk_tr_256_lab_kernel<true, 32, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem3);
          hessian_syrk_tr_256_lab_kernel<true, 32, 4><<<ntiles2, 512, smem3, s>>>(xp, T, K, ldx, H, beta, alpha, nt2);
          INC_LAUNCH_RETURN();
        }
        const int habl = inc_small_tiles_flag(-1) - 46;  // 47 / 48 / 49: timing-only, no LDS-DMA / no MFMA + fragment reads / neither
        if (habl >= 1 && habl <= 3 && xdtype == INC_BF16) {
#define INC_HABL(A) { (void)hipFuncSetAttribute((const void*)hessian_syrk_tr_256_lab_kernel<true, TR_TOK, TR_NST, A>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem3); \
                      hessian_syrk_tr_256_lab_kernel<true, TR_TOK, TR_NST, A><<<ntiles2, 512, smem3, s>>>(xp, T, K, ldx, H, beta, alpha, nt2); }
          if (habl == 1) INC_HABL(1) else if (habl == 2) INC_HABL(2) else INC_HABL(3)
          INC_LAUNCH_RETURN();
        }
        if (habl == 4 && xdtype == INC_BF16) {  // 50: CORRECT results, the DMA requests spread over the step's MFMA rows
          INC_HABL(4)
          INC_LAUNCH_RETURN();
        }
        if ((habl == 8 || habl == 9) && xdtype == INC_BF16) {  // 54: static priority for waves 4-7; 55: that + spread + rolling fragments
          if (habl == 8) INC_HABL(16) else INC_HABL(28)
          INC_LAUNCH_RETURN();
        }
        if ((habl == 6 || habl == 7) && xdtype == INC_BF16) {  // 52: spread + row fragments two rows ahead; 53: the rolling fragments alone
          if (habl == 6) INC_HABL(12) else INC_HABL(8)
          INC_LAUNCH_RETURN();
        }
        if (habl == 13 && xdtype == INC_BF16) {  // 59: the round-4 form of the tile (A/B partner of TR_ABL)
          INC_HABL(0)
          INC_LAUNCH_RETURN();
        }
        if (habl == 14 && xdtype == INC_BF16) {  // 60: the product tile with its pieces issued from one asm block per step
          INC_HABL(176)
          INC_LAUNCH_RETURN();
        }
        if (habl >= 10 && habl <= 12 && xdtype == INC_BF16) {  // 56: rolling fragments + priority; 57: one rolling pipeline per step; 58: that + priority
          if (habl == 10) INC_HABL(24) else if (habl == 11) INC_HABL(32) else INC_HABL(48)
          INC_LAUNCH_RETURN();
        }
        if (habl == 5 && xdtype == INC_BF16) {  // 51: the same with four 32-token stages (three steps in flight)
          (void)hipFuncSetAttribute((const void*)hessian_syrk_tr_256_lab_kernel<true, 32, 4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem3);
          hessian_syrk_tr_256_lab_kernel<true, 32, 4, 4><<<ntiles2, 512, smem3, s>>>(xp, T, K, ldx, H, beta, alpha, nt2);
          INC_LAUNCH_RETURN();
        }
#undef INC_HABL
#endif
        if (xdtype == INC_BF16) hessian_syrk_tr_256_kernel<true><<<ntiles2, 512, smem3, s>>>(xp, T, K, ldx, H, beta, alpha, nt2);
        else hessian_syrk_tr_256_kernel<false><<<ntiles2, 512, smem3, s>>>(xp, T, K, ldx, H, beta, alpha, nt2);
        INC_LAUNCH_RETURN();
      }
    } else if (xdtype == INC_BF16)
      hessian_syrk_16bit_kernel<true><<<ntiles, 256, smem, s>>>((const uint16_t*)x, T, K, ldx, H, beta, alpha, nt, vec_ok);
    else
      hessian_syrk_16bit_kernel<false><<<ntiles, 256, smem, s>>>((const uint16_t*)x, T, K, ldx, H, beta, alpha, nt, vec_ok);
  } else {
    return INC_ERR_UNSUPPORTED;
  }
  INC_LAUNCH_RETURN();
}

// Bytes of scratch with which inc_gptq_hessian_accum_multi can split the tiles of its last, partly filled round over idle CUs
// (at most one round of 256 x 256 fp32 tiles).
int64_t inc_gptq_hessian_accum_multi_workspace_bytes(void) { return (int64_t)HESSIAN_TAIL_UNITS * H2 * H2 * 4; }

int inc_gptq_hessian_accum_multi(int n, const void* const* xs, int xdtype, int64_t T, const int64_t* Ks, const int64_t* ldxs,
                                 float* const* Hs, const float* betas, const float* alphas, void* workspace, int64_t workspace_bytes,
                                 inc_stream_t stream) {
  INC_CHECK_ARG(n > 0 && xs && Ks && ldxs && Hs && betas && alphas && T > 0);
  if (n > HESSIAN_MAX_BATCH || !(xdtype == INC_BF16 || xdtype == INC_F16) || inc_force_small_tiles()) return INC_ERR_UNSUPPORTED;
  HessianBatch a;
  int first = 0;
  for (int i = 0; i < n; ++i) {
    INC_CHECK_ARG(xs[i] && Hs[i] && Ks[i] > 0 && ldxs[i] >= Ks[i]);
    const bool vec_ok = (ldxs[i] % 8 == 0) && ((reinterpret_cast<uintptr_t>(xs[i]) & 15) == 0);
    if (!(Ks[i] >= H2 && vec_ok && (Ks[i] % 8) == 0)) return INC_ERR_UNSUPPORTED;  // the caller falls back to single launches
    a.x[i] = (const uint16_t*)xs[i];
    a.H[i] = Hs[i];
    a.K[i] = Ks[i];
    a.ldx[i] = ldxs[i];
    a.beta[i] = betas[i];
    a.alpha[i] = alphas[i];
    a.nt[i] = (int)ceil_div64(Ks[i], H2);
    a.first[i] = first;
    first += a.nt[i] * (a.nt[i] + 1) / 2;
  }
  for (int i = n; i <= HESSIAN_MAX_BATCH; ++i) a.first[i] = first;
  for (int i = n; i < HESSIAN_MAX_BATCH; ++i) { a.x[i] = a.x[0]; a.H[i] = a.H[0]; a.K[i] = a.K[0]; a.ldx[i] = a.ldx[0]; a.beta[i] = 1.f; a.alpha[i] = 0.f; a.nt[i] = a.nt[0]; }
  a.n = n;
  a.full = first;
  a.nseg = 1;
  a.slab = nullptr;
  a.block0 = 0;
  hipStream_t s = inc_s(stream);
  // Tile quantisation: `first` equal tiles on `cus` CUs (one workgroup per CU: 132 KiB of LDS) run in ceil(first / cus) rounds; when the
  // last round fills less than half of the chip its tiles are cut into nseg = cus / tail token ranges, one workgroup each (a Llama
  // block's launch: 1354 tiles = 5 rounds + 74 tiles -> 222 units of a third: 5.4 rounds instead of 6).  The ranges' raw sums go to
  // the caller's workspace and a second, small launch adds them into H in range order: deterministic, and every tile outside the tail
  // is computed exactly as before.
  {
    int dev = 0, cus = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int tail = cus > 0 ? first % cus : 0;
    const int steps = (int)ceil_div64(T, TR_TOK);
    if (workspace && first > cus && tail > 0 && 2 * tail <= cus && inc_small_tiles_flag(-1) != 44) {
      int nseg = cus / tail;
      if (nseg > 4) nseg = 4;
      if (nseg > steps / 16) nseg = steps / 16;  // a range is at least 16 steps long
      if (nseg >= 2 && tail * nseg <= HESSIAN_TAIL_UNITS && workspace_bytes >= (int64_t)tail * nseg * H2 * H2 * 4 &&
          (reinterpret_cast<uintptr_t>(workspace) & 15) == 0) {
        a.full = first - tail;
        a.nseg = nseg;
        a.slab = (float*)workspace;
      }
    }
  }
  const int grid = a.full + (first - a.full) * a.nseg;
#ifdef INC_KBENCH
  if (inc_small_tiles_flag(-1) == 45) {  // harness flag 45: the register-transposing generation
    a.full = first; a.nseg = 1;
    const size_t smem2 = (size_t)2 * H2_STAGE * sizeof(uint16_t);  // 144 KiB
    const bool tail = (T % HK) != 0;
    (void)hipFuncSetAttribute((const void*)hessian_syrk_16bit_256_multi_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2);
    (void)hipFuncSetAttribute((const void*)hessian_syrk_16bit_256_multi_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2);
    (void)hipFuncSetAttribute((const void*)hessian_syrk_16bit_256_multi_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2);
    (void)hipFuncSetAttribute((const void*)hessian_syrk_16bit_256_multi_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2);
#define INC_HM(B, TL) hessian_syrk_16bit_256_multi_kernel<B, TL><<<first, 512, smem2, s>>>(a, T)
    if (xdtype == INC_BF16) { if (tail) INC_HM(true, true); else INC_HM(true, false); }
    else { if (tail) INC_HM(false, true); else INC_HM(false, false); }
#undef INC_HM
    INC_LAUNCH_RETURN();
  }
#endif
  {  // transpose-read generation
    const size_t smem3 = (size_t)TR_NST * TR_STAGE;
    static std::atomic<uint64_t> attr3_set{0};
    if (inc_attr_needed(attr3_set)) {
      (void)hipFuncSetAttribute((const void*)hessian_syrk_tr_256_multi_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem3);
      (void)hipFuncSetAttribute((const void*)hessian_syrk_tr_256_multi_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem3);
      inc_attr_done(attr3_set);
    }
#ifdef INC_KBENCH
    if (inc_small_tiles_flag(-1) == 46 && xdtype == INC_BF16) {  // timing A/B: four 32-token stages
      a.full = first; a.nseg = 1;
      (void)hipFuncSetAttribute((const void*)hessian_syrk_tr_256_multi_lab_kernel<true, 32, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem3);
      hessian_syrk_tr_256_multi_lab_kernel<true, 32, 4><<<first, 512, smem3, s>>>(a, T);
      INC_LAUNCH_RETURN();
    }
#endif
#ifdef INC_KBENCH
    {  // harness flags 53 / 54 / 56 / 57 / 58 / 59: the tile variants of the single-problem launch, in the batched launch
      const int f = inc_small_tiles_flag(-1);
      const int mabl = f == 53 ? 8 : f == 54 ? 16 : f == 56 ? 24 : f == 57 ? 32 : f == 58 ? 48 : f == 59 ? 64 : f == 60 ? 176 : 0;  // (64 = ABL 0)
      if (mabl && xdtype == INC_BF16) {
#define INC_HMV(A) { (void)hipFuncSetAttribute((const void*)hessian_syrk_tr_256_multi_lab_kernel<true, TR_TOK, TR_NST, A>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem3); \
                     a.block0 = 0; hessian_syrk_tr_256_multi_lab_kernel<true, TR_TOK, TR_NST, A><<<grid, 512, smem3, s>>>(a, T); }
        if (mabl == 8) INC_HMV(8) else if (mabl == 16) INC_HMV(16) else if (mabl == 24) INC_HMV(24) else if (mabl == 32) INC_HMV(32) else if (mabl == 64) INC_HMV(0) else if (mabl == 176) INC_HMV(176) else INC_HMV(48)
#undef INC_HMV
        if (a.nseg > 1) hessian_tail_finalize_kernel<<<4 * (first - a.full), 512, 0, s>>>(a);
        INC_LAUNCH_RETURN();
      }
    }
#endif
    // (one launch per round of one-tile-per-CU, so that the tiles sharing X panels in an XCD's L2 restart together, measured 1 %
    // slower than this single launch: profiles/NOTES.md round 4; harness flag 43 keeps it as an A/B partner)
    int chunk = grid;
#ifdef INC_KBENCH
    if (inc_small_tiles_flag(-1) == 43) {
      int dev = 0, cus = 256;
      (void)hipGetDevice(&dev);
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      chunk = (cus <= 0 || (cus % 8) != 0) ? 256 : cus;
    }
#endif
    for (int b0 = 0; b0 < grid; b0 += chunk) {
      a.block0 = b0;
      const int g = grid - b0 < chunk ? grid - b0 : chunk;
      if (xdtype == INC_BF16) hessian_syrk_tr_256_multi_kernel<true><<<g, 512, smem3, s>>>(a, T);
      else hessian_syrk_tr_256_multi_kernel<false><<<g, 512, smem3, s>>>(a, T);
    }
    if (a.nseg > 1) hessian_tail_finalize_kernel<<<4 * (first - a.full), 512, 0, s>>>(a);
    INC_LAUNCH_RETURN();
  }
}

int inc_gptq_hessian_finalize(float* H, int64_t K, float percdamp, uint8_t* dead, void* workspace,
                              inc_stream_t stream) {
  INC_CHECK_ARG(H && K > 0);
  hipStream_t s = inc_s(stream);
  hessian_diag_kernel<<<1, 1024, 0, s>>>(H, K, percdamp, dead, (float*)workspace);
  const unsigned nt = (unsigned)ceil_div64(K, 32);
  hessian_mirror_kernel<<<dim3(nt, nt), 256, 0, s>>>(H, K);
  INC_LAUNCH_RETURN();
}

}  // extern "C"
