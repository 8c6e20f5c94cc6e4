// gptq_lazy.hip -- K6's trailing update W[:, c_begin:c_end] -= Err1 @ Hinv[i1:i1+128, c_begin:c_end] (gptq.py:1304) with the exact fp32 MFMA:
// the third generation (one tile per workgroup: the "next 128 columns" on the critical path, and small remainders) and the fourth
// (a workgroup owns a 128-row strip and walks column tiles: the bulk of the update).  Launched from gptq.hip (inc_gptq_lazy_update[_cols]);
// its own translation unit so that the column loop's chain kernels (minutes of compile time) are not rebuilt with it.
#include <algorithm>
#include <type_traits>

#include "common.hpp"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int QB = 128;   // columns per block (gptq.hip)
constexpr int L2T = 128;  // rows / columns of a lazy-update tile (gptq.hip)

// ---------------------------------------------------------------------------------------------
// lazy update, third generation: ONE [128 rows x CW columns] tile per workgroup, two workgroups per CU
// ---------------------------------------------------------------------------------------------
// The second generation parks a 128 KiB double buffer per workgroup: one workgroup of four waves per CU, one wave per SIMD, and
// each of them runs its phases back to back -- Err slice in, Hinv tile in, W tile in, 256 dependent-by-four fp32 MFMAs (7.5 us),
// W tile out -- so the matrix pipe idles through every load and the loads idle through every MFMA: 40 TFLOP/s of the 157 the
// exact-fp32 MFMA has at 4096^2 (tools/kbench colloop), and the 128 KiB exclude the quantisation chain's workgroups (64 KiB) from
// the CU, which serialised the look-ahead loop's two streams (kernel trace profiles/r3d: the "rest" update of block b-1 ran for
// 110-130 us and the chain of block b+1 could not start under it).  Here a workgroup owns one tile and stages only that tile's
// slice of Hinv ([128 k][CW] fp32: 64 KiB at CW = 128, 16 KiB at CW = 32): two workgroups share a CU (<= 256 registers per wave), one
// multiplies while the other loads or stores, and a chain workgroup still fits next to one of them.  CW = 32 serves the
// look-ahead loop's "next 128 columns" update, which sits on the critical path with only N / 128 row tiles to spread: four times
// the workgroups, a quarter of the dependent MFMAs each.  Per output element the products are added in the order of the
// second generation (k = 2s + (lane >> 5), s ascending, acc from 0, then W - acc): bit-identical W.
template <int CW>
__global__ __launch_bounds__(256, 2) void gptq_lazy_update_v3_kernel(float* __restrict__ w, const float* __restrict__ Hinv,
                                                                     const float* __restrict__ err, int64_t N, int64_t K,
                                                                     int64_t i1, int64_t c_begin) {
  constexpr int NF = CW / 32;
  // LDS: [0, 64 KiB) the four waves' Err1 slices (16 KiB each), then the Hinv slice [128 k][CW].  At CW = 128 the Hinv slice
  // REUSES the first 64 KiB: wave w's share of it (k rows 32w .. 32w + 31) lands exactly on wave w's own Err1 slice, which that
  // wave has finished reading by then -- no workgroup barrier in between.
  constexpr uint32_t HS_OFF = CW == 128 ? 0u : 65536u;
  // (the quarter tiles serve the "next 128 columns" update, which the next chain waits for: ahead of the rest of the trailing update,
  // behind the chain itself)
  if constexpr (CW != 128) __builtin_amdgcn_s_setprio(2);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem_raw;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t r0 = (int64_t)blockIdx.y * L2T + wave * 32;  // first row of this wave
  const int64_t c0 = c_begin + (int64_t)blockIdx.x * CW;
  const float* const hbase = Hinv + i1 * K;

  // Hinv[i1 + k][c0 .. c0 + CW) -> LDS [k][CW]; this wave moves rows wave * 32 .. + 31 (16 KiB / 4 KiB)
  auto dma_hinv = [&]() {
    if constexpr (CW == 128) {
      int64_t col = c0 + 4 * (lane & 31);
      if (col > K - 4) col = K - 4;  // partial last tile: clamped columns are never stored
      const uint32_t v = (uint32_t)(((wave * 32 + (lane >> 5)) * K + col) * 4);
      const uint32_t step = (uint32_t)(2 * K * 4);
      const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + HS_OFF + wave * 16384);
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4)
        lds_dma_4x1k(hbase, dst + q4 * 4096, v + (4 * q4) * step, v + (4 * q4 + 1) * step, v + (4 * q4 + 2) * step, v + (4 * q4 + 3) * step);
    } else {
      int64_t col = c0 + 4 * (lane & 7);
      if (col > K - 4) col = K - 4;
      const uint32_t v = (uint32_t)(((wave * 32 + (lane >> 3)) * K + col) * 4);
      const uint32_t step = (uint32_t)(8 * K * 4);
      const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + HS_OFF + wave * 4096);
      lds_dma_4x1k(hbase, dst, v, v + step, v + 2 * step, v + 3 * step);
    }
  };
  // Err1 slice of this wave (32 rows x 512 B) -> LDS by LDS-DMA, two full rows per instruction.  A lane of the MFMA wants ONE row
  // (A operand: row lane & 31, k = 2s + (lane >> 5)); fetched that way from global memory every load instruction touches 32
  // different lines and the 16 KiB slice costs 128 KiB of L2 -> CU traffic per wave (four times the tile's Hinv and W bytes
  // together; the second generation did exactly that).  Through LDS the global side is coalesced, and the 16-byte chunk index is
  // XOR-ed with (row & 7) on the SOURCE side so that the row-per-lane ds_read_b128 below is conflict-free.
  {
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + wave * 16384);
    const int64_t rows_here = N - r0 < 32 ? N - r0 : 32;  // >= 1: the grid has no workgroup without rows; a wave may have none
    const float* ebase = err + (rows_here > 0 ? r0 : N - 1) * QB;
    uint32_t v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int R = 2 * j + (lane >> 5);
      const int Rc = rows_here > 0 ? (R < rows_here ? R : (int)rows_here - 1) : 0;  // rows past N: a valid row's bytes, never stored
      v[j] = (uint32_t)(Rc * (QB * 4)) + (uint32_t)(((lane & 31) ^ (R & 7)) * 16);
    }
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) lds_dma_4x1k(ebase, dst + q4 * 4096, v[4 * q4], v[4 * q4 + 1], v[4 * q4 + 2], v[4 * q4 + 3]);
  }
  if constexpr (CW != 128) dma_hinv();  // separate LDS region: both transfers in flight together
  // W tile -> registers (D layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)); rows / columns past the edge clamped
  float wt[NF][16];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    int64_t col = c0 + nf * 32 + (lane & 31);
    if (col > K - 1) col = K - 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int64_t row = r0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row > N - 1) row = N - 1;
      wt[nf][r] = w[row * K + col];  // (stays between the DMA asm statements around it: they are compiler memory barriers)
    }
  }
  f32x16 acc[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nf][r] = 0.f;
  // the Err1 DMA is the oldest part of the in-order queue: what was issued after it (the Hinv slice at CW = 32, the W loads) may stay out
  if constexpr (CW == 128) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
  float a[64];
  {
    const int R = lane & 31;
    const char* eb = smem_raw + wave * 16384 + R * 512;
    const bool hi = (lane >> 5) != 0;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const float4 v4 = *reinterpret_cast<const float4*>(eb + ((j ^ (R & 7)) * 16));
      a[2 * j] = hi ? v4.y : v4.x;
      a[2 * j + 1] = hi ? v4.w : v4.z;
    }
  }
  if constexpr (CW == 128) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's slice is in registers: its LDS space takes the wave's Hinv rows
    dma_hinv();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const float* hs = reinterpret_cast<const float*>(smem_raw + HS_OFF) + (lane >> 5) * CW + (lane & 31);
  constexpr int BS = NF == 4 ? 4 : 8;  // k-pairs per batch: the B operands of a batch are read together, then multiplied
#pragma unroll
  for (int g = 0; g < 64 / BS; ++g) {
    float b[BS * NF];
#pragma unroll
    for (int sb = 0; sb < BS; ++sb)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) b[sb * NF + nf] = hs[(2 * (BS * g + sb)) * CW + nf * 32];
#pragma unroll
    for (int sb = 0; sb < BS; ++sb)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        acc[nf] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[BS * g + sb], b[sb * NF + nf], acc[nf], 0, 0, 0);
      }
  }
  if (r0 + 32 <= N && c0 + CW <= K) {  // wave-uniform: interior tile, unguarded stores
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      float* wp = w + (r0 + 4 * (lane >> 5)) * K + c0 + nf * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) wp[((r & 3) + 8 * (r >> 2)) * K] = wt[nf][r] - acc[nf][r];
    }
  } else {
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      const int64_t col = c0 + nf * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = r0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < N && col < K) w[row * K + col] = wt[nf][r] - acc[nf][r];
      }
    }
  }
}

#ifdef INC_KBENCH  // the same kernel with its timing-only ablations: harness code
#include "../../tools/kbench_gptq_lazy_lab.inc"
#endif

// ---------------------------------------------------------------------------------------------
// lazy update, fourth generation (the "rest" of the trailing update): a workgroup OWNS a 128-row strip and walks column tiles
// ---------------------------------------------------------------------------------------------
// The third generation brings Err1 in once per TILE (64 KiB through LDS for 4.2 MFLOP) and runs its phases back to back -- loads,
// 256 MFMAs, stores -- so the matrix pipe is busy only while the CU's other workgroup happens to be loading: 78 TFLOP/s alone on the
// chip, of the 155 the bare exact-fp32 MFMA sustains (tools/f32mfma_lab; profiles/NOTES.md).  Here a workgroup keeps its rows' Err1 fragment in registers
// (the A operand: 64 VGPRs, loaded ONCE) and walks `tiles_per_wg` column tiles of 32 columns.  Everything a tile needs arrives by
// LDS-DMA one tile ahead: the Hinv slice ([128 k][32] fp32 = 16 KiB, wave w moves k rows 32w .. 32w + 31) and the wave's own 32 x 32
// patch of W (4 KiB; read back in the MFMA's D layout) into the other half of two double buffers, requested before the 64 MFMAs of
// the current tile start; the 16 stores of a tile drain under the next tile -- one barrier and one counted `s_waitcnt vmcnt` per
// tile, and no register-destination load the compiler's own wait-count bookkeeping could serialise (a form with W in registers got
// `vmcnt(6)` from it behind every DMA burst).  One accumulator: the 32x32x2 fp32 MFMA issues every 64 cycles, which is also its
// dependent latency.  All buffers live in what was the wave's own Err1 slice (no barrier between reading Err1 and the first
// request): 64 KiB of LDS and 180 registers, two of these or one and a chain workgroup per CU.  Per output element the sum is the
// third generation's (k = 2s + (lane >> 5), s ascending, acc from 0, then W - acc): bit-identical W.
// (A form with two accumulators and the neighbours' epilogue / requests spread over the MFMA shadows is faster alone on the chip and slower
// inside the step -- 214 registers against 180: tools/rejected/gptq_lazy_strip_pipelined.inc, profiles/NOTES.md.)
constexpr int L4W = 32;  // columns of a fourth-generation tile
#ifndef INC_LAZY_STRIP_TILES
#define INC_LAZY_STRIP_TILES 2048
#endif
#ifndef INC_LAZY_STRIP_CAP
#define INC_LAZY_STRIP_CAP 8
#endif

__global__ __launch_bounds__(256, 2) void gptq_lazy_update_v4_kernel(float* __restrict__ w, const float* __restrict__ Hinv,
                                                                     const float* __restrict__ err, int64_t N, int64_t K, int64_t i1,
                                                                     int64_t c_begin, int64_t c_end, int tiles_per_wg) {
  // LDS, per wave a 16 KiB slice: first its Err1 rows; afterwards [0, 8 KiB) its k rows of the two Hinv buffers, [8, 16 KiB) its two W patches
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem_raw;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t r0 = (int64_t)blockIdx.y * L2T + wave * 32;  // first row of this wave
  const int64_t col0 = c_begin + (int64_t)blockIdx.x * tiles_per_wg * L4W;
  int tiles = (int)((c_end - col0 + L4W - 1) / L4W);  // >= 1 by the grid's construction
  if (tiles > tiles_per_wg) tiles = tiles_per_wg;
  const float* const hbase = Hinv + i1 * K;
  const int rows_here = (int)(N - r0 < 32 ? (N - r0 > 0 ? N - r0 : 0) : 32);
  const float* const wbase = w + (rows_here > 0 ? r0 : N - 1) * K;  // a wave without rows reads a valid row's bytes and never stores
  const uint32_t slice = __builtin_amdgcn_readfirstlane(lds0 + wave * 16384);
  // a 1-KiB piece = eight rows of 128 B: lane -> row lane >> 3, 16-byte chunk lane & 7
  const uint32_t h_step = (uint32_t)(8 * K * 4);
  const uint32_t h_row = (uint32_t)((wave * 32 + (lane >> 3)) * K * 4);
  uint32_t w_row[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int R = 8 * q + (lane >> 3);
    R = rows_here > 0 ? (R < rows_here ? R : rows_here - 1) : 0;
    w_row[q] = (uint32_t)(R * K * 4);
  }
  auto request = [&](int64_t c0, int b) {  // tile at column c0 -> buffers b
    int64_t col = c0 + 4 * (lane & 7);
    if (col > K - 4) col = K - 4;  // partial last tile: clamped columns are never stored
    const uint32_t cb = (uint32_t)(col * 4);
    lds_dma_4x1k(hbase, slice + b * 4096, h_row + cb, h_row + h_step + cb, h_row + 2 * h_step + cb, h_row + 3 * h_step + cb);
    lds_dma_4x1k(wbase, slice + 8192 + b * 4096, w_row[0] + cb, w_row[1] + cb, w_row[2] + cb, w_row[3] + cb);
  };

  // Err1 slice of this wave (32 rows x 512 B) -> LDS, 16-byte chunk index XOR-ed with (row & 7) on the source side (third generation)
  {
    const float* ebase = err + (rows_here > 0 ? r0 : N - 1) * QB;
    uint32_t v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int R = 2 * j + (lane >> 5);
      const int Rc = rows_here > 0 ? (R < rows_here ? R : rows_here - 1) : 0;
      v[j] = (uint32_t)(Rc * (QB * 4)) + (uint32_t)(((lane & 31) ^ (R & 7)) * 16);
    }
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) lds_dma_4x1k(ebase, slice + q4 * 4096, v[4 * q4], v[4 * q4 + 1], v[4 * q4 + 2], v[4 * q4 + 3]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float a[64];
  {
    const int R = lane & 31;
    const char* eb = smem_raw + wave * 16384 + R * 512;
    const bool hi = (lane >> 5) != 0;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const float4 v4 = *reinterpret_cast<const float4*>(eb + ((j ^ (R & 7)) * 16));
      a[2 * j] = hi ? v4.y : v4.x;
      a[2 * j + 1] = hi ? v4.w : v4.z;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slice is in registers: its LDS space takes the wave's buffers
  request(col0, 0);

  const char* const hs0 = smem_raw + lane * 4;  // B operand: k = 2s + (lane >> 5), column lane & 31 -> byte (lane >> 5) * 128 + (lane & 31) * 4
  const char* const ws0 = smem_raw + wave * 16384 + 8192 + (lane >> 5) * 512 + (lane & 31) * 4;  // D layout: + ((r & 3) + 8 (r >> 2)) * 128
  float* const wp0 = w + (r0 + 4 * (lane >> 5)) * K + (lane & 31);
  bool counted = false;  // the previous tile issued exactly 16 stores (an interior tile)
  for (int t = 0; t < tiles; ++t) {
    const int b = t & 1;
    const int64_t c0 = col0 + (int64_t)t * L4W;
    // in-order queue here: [Hinv(t) 4 pieces] [W(t) 4 pieces] [the stores of tile t - 1]
    if (counted) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // every wave's share of Hinv(t) has landed, and every wave is done reading the buffer Hinv(t + 1) goes to
    if (t + 1 < tiles) request(c0 + L4W, b ^ 1);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const char* hs = hs0 + b * 4096;
    constexpr int BS = 8;  // k-pairs per batch: the B operands of the next batch are read before this batch is multiplied
    float bq[2][BS];
#pragma unroll
    for (int sb = 0; sb < BS; ++sb) bq[0][sb] = *reinterpret_cast<const float*>(hs + ((2 * sb) >> 5) * 16384 + ((2 * sb) & 31) * 128);
#pragma unroll
    for (int g = 0; g < 64 / BS; ++g) {
      if (g + 1 < 64 / BS) {
#pragma unroll
        for (int sb = 0; sb < BS; ++sb) {
          const int s2 = 2 * (BS * (g + 1) + sb);
          bq[(g + 1) & 1][sb] = *reinterpret_cast<const float*>(hs + (s2 >> 5) * 16384 + (s2 & 31) * 128);
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // (left alone the scheduler sinks every read to just before its MFMA and waits for it there)
#pragma unroll
      for (int sb = 0; sb < BS; ++sb) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[BS * g + sb], bq[g & 1][sb], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    const char* ws = ws0 + b * 4096;
    float wt[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) wt[r] = *reinterpret_cast<const float*>(ws + ((r & 3) + 8 * (r >> 2)) * 128);
    counted = r0 + 32 <= N && c0 + L4W <= K;  // wave-uniform: an interior tile
    if (counted) {
      float* wp = wp0 + c0;
#pragma unroll
      for (int r = 0; r < 16; ++r) wp[((r & 3) + 8 * (r >> 2)) * K] = wt[r] - acc[r];
    } else {
      const int64_t col = c0 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = r0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < N && col < K) w[row * K + col] = wt[r] - acc[r];
      }
    }
    asm volatile("" ::: "memory");
  }
}

}  // namespace

// launch over the columns [c_begin, c_end) of the trailing matrix (c_begin on the 128-column tile grid that starts at i2): the strip
// form for the bulk, whole tiles of the third generation below that, quarter tiles when whole tiles would leave most CUs without a workgroup
// `exclusive`: the update has the chip to itself (the one-stream loop's whole-range call).  In the look-ahead loop the rest of the update runs
// beside the next block's chain, whose workgroups (64 KiB of LDS each) must find room on CUs that two strips would fill: short strips there.
void inc_launch_lazy_update_v3(float* w, const float* Hinv, const float* err, int64_t N, int64_t K, int64_t i1, int64_t c_begin,
                               int64_t c_end, bool exclusive, hipStream_t s) {
  static std::atomic<uint64_t> attr_set{0};
  if (inc_attr_needed(attr_set)) {
    (void)hipFuncSetAttribute((const void*)gptq_lazy_update_v3_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)gptq_lazy_update_v3_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + L2T * 32 * 4);
    inc_attr_done(attr_set);
  }
  const int64_t row_tiles = ceil_div64(N, L2T);
  const int64_t col_tiles = ceil_div64(c_end - c_begin, L2T);
#ifdef INC_KBENCH
  const int abl = inc_small_tiles_flag(-1) - 86;  // 87 / 88 / 90: timing-only (no MFMAs / no loads / no stores)
  if (abl == 1 || abl == 2 || abl == 4) {
#define INC_L3A(A) { (void)hipFuncSetAttribute((const void*)gptq_lazy_update_v3_lab_kernel<128, A>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); \
                     gptq_lazy_update_v3_lab_kernel<128, A><<<dim3((unsigned)col_tiles, (unsigned)row_tiles), 256, 65536, s>>>(w, Hinv, err, N, K, i1, c_begin); }
    if (abl == 1) INC_L3A(1) else if (abl == 2) INC_L3A(2) else INC_L3A(4)
#undef INC_L3A
    return;
  }
#endif
  // the strip form when every one of ~512 workgroups (two per CU) gets at least four 32-column tiles
  const int64_t nt4 = ceil_div64(c_end - c_begin, L4W);
  if (row_tiles * nt4 >= (int64_t)INC_LAZY_STRIP_TILES && inc_small_tiles_flag(-1) != 107) {  // (harness flag 107: third generation everywhere, the A/B partner)
    static std::atomic<uint64_t> attr4_set{0};
    if (inc_attr_needed(attr4_set)) {
      (void)hipFuncSetAttribute((const void*)gptq_lazy_update_v4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
      inc_attr_done(attr4_set);
    }
    // Strip length.  A strip's start-up (Err1 in, fragment out, first requests: two dependent round trips) costs about three tiles, so
    // long strips are better -- but only when the grid is ONE round of at most two workgroups per CU (a few left-over workgroups in a
    // second round cost a whole strip) AND nothing else needs the CUs meanwhile: 92 against 78 TFLOP/s alone on the chip at 4096 x 11008,
    // but the look-ahead loop got SLOWER with them (9.45 vs 9.27 ms: the next chain's workgroups wait for a strip to end).  Otherwise
    // short strips (<= 8 tiles) that the dispatcher balances over several rounds; the busiest CU's work in tiles decides.  (Timing-only
    // ablation, tools/kbench colloop: the strip's MFMA + LDS skeleton alone runs at ~100 TFLOP/s with 8-tile strips against 155 for the
    // bare instruction, tools/f32mfma_lab -- the start-up is the difference; profiles/NOTES.md.)
    constexpr int64_t SLOTS = 512, CUS = 256, STARTUP = 3;
    const int64_t chunks_one = SLOTS / row_tiles > 0 ? SLOTS / row_tiles : 1;
    int64_t tpw_one = ceil_div64(nt4, chunks_one);
    if (tpw_one < 4) tpw_one = 4;
    const int64_t cost_one = ceil_div64(ceil_div64(nt4, tpw_one) * row_tiles, CUS) * (tpw_one + STARTUP);
    int64_t cap = INC_LAZY_STRIP_CAP;
#ifdef INC_KBENCH
    if (inc_small_tiles_flag(-1) == 108) cap = 4;
    if (inc_small_tiles_flag(-1) == 109) cap = 16;
#endif
    int64_t tpw_short = nt4 / ceil_div64(SLOTS, row_tiles);
    if (tpw_short > cap) tpw_short = cap;
    if (tpw_short < 4) tpw_short = 4;
    const int64_t cost_short = ceil_div64(ceil_div64(nt4, tpw_short) * row_tiles, CUS) * (2 * tpw_short + STARTUP) / 2;
    bool one_round = exclusive && ceil_div64(nt4, tpw_one) * row_tiles <= SLOTS && cost_one <= cost_short;
#ifdef INC_KBENCH
    if (inc_small_tiles_flag(-1) == 108 || inc_small_tiles_flag(-1) == 109 || inc_small_tiles_flag(-1) == 110) one_round = false;  // 110: short strips everywhere
#endif
    const int64_t tpw = one_round ? tpw_one : tpw_short;
    const int64_t chunks = ceil_div64(nt4, tpw);
    gptq_lazy_update_v4_kernel<<<dim3((unsigned)chunks, (unsigned)row_tiles), 256, 65536, s>>>(w, Hinv, err, N, K, i1, c_begin, c_end, (int)tpw);
    return;
  }
  if (row_tiles * col_tiles <= 128)
    gptq_lazy_update_v3_kernel<32><<<dim3((unsigned)ceil_div64(c_end - c_begin, 32), (unsigned)row_tiles), 256, 65536 + L2T * 32 * 4, s>>>(w, Hinv, err, N, K, i1, c_begin);
  else
    gptq_lazy_update_v3_kernel<128><<<dim3((unsigned)col_tiles, (unsigned)row_tiles), 256, 65536, s>>>(w, Hinv, err, N, K, i1, c_begin);
}

