// Lab harness (test infrastructure, not part of the library): candidate forms of the 256 x 256 syrk tile, checked and timed against the
// product's single-problem launch (inc_gptq_hessian_accum of libinc_mi355x.so).  Build: make -C tools hess_lab; run: tools/hess_lab
// A candidate that wins moves into neural_compressor_amd/csrc/gptq.hip; one that loses stays here with its numbers in profiles/NOTES.md.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <algorithm>
#include <type_traits>
#include <vector>

#include "../include/inc_mi355x.h"
#include "../neural_compressor_amd/csrc/common.hpp"

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

namespace {

constexpr int H2 = 256;

// (same order as gptq.hip's xcd_supertile_decode)
__device__ __forceinline__ void xcd_supertile_decode(int b, int n, int nt, int& ti, int& tj) {
  const int q = n / 8, r = n % 8, xcd = b % 8, t = b / 8;
  int rem = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + t;
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
  ti = tj = 0;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// ---- candidate A: the product's ring (two 64-token stages, one-block issue, eight waves) with v_mfma_f32_32x32x16_bf16 ------------
// A wave owns 128 x 64 of the tile = 4 x 2 blocks of 32 x 32.  Operand of the 32 x 32 x 16 MFMA: lane l holds row (l % 32), k-block
// l / 32 (8 consecutive k).  A transpose-read hands lane (4a + e) of 16-lane group g element e of lanes a, a+4, a+8, a+12: with lane
// (a + 4b) of group g pointing at token row 8 * (g / 2) + b (second read: + 4), feature block 16 * (g % 2) + 4a, groups 0 / 1 receive
// features 0-15 / 16-31 of tokens 0-7 and groups 2 / 3 the same features of tokens 8-15.  A 32-lane half touches 4 token rows x 64 B:
// conflict-free when a row advances 16 banks -> pitch 1088 B.
constexpr int P32 = 1088;
constexpr int TOK = 64;
constexpr int STAGE32 = TOK * P32;

template <int VARIANT>
__global__ __launch_bounds__(512) void syrk32_kernel(const uint16_t* __restrict__ x, int64_t T, int64_t K, int64_t ldx, float* __restrict__ H,
                                                     float beta, float alpha, int nt) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  int ti, tj;
  xcd_supertile_decode((int)blockIdx.x, (int)gridDim.x, nt, ti, tj);
  const int64_t i0 = (int64_t)ti * H2, j0 = (int64_t)tj * H2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem_raw;
  int64_t f = (lane < 32 ? i0 : j0) + 8 * (lane & 31);
  if (f > K - 8) f = K - 8;
  const uint32_t voff = (uint32_t)(f * 2);
  const int nk = (int)(T / TOK);  // (lab: T % 64 == 0)
  uint32_t voffr[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) voffr[i] = voff + (uint32_t)((int64_t)i * ldx * 2);
  auto issue = [&](int kt) {
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (kt & 1) * STAGE32 + wave * 8 * P32);
    lds_dma_8x1k<P32>(x + ((int64_t)kt * TOK + wave * 8) * ldx, dst, voffr[0], voffr[1], voffr[2], voffr[3], voffr[4], voffr[5], voffr[6], voffr[7]);
  };
  f32x16_t acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int s16 = lane & 15, fa = s16 & 3, fb = s16 >> 2, fg = lane >> 4;
  const uint32_t rbase = (uint32_t)((8 * (fg >> 1) + fb) * P32 + (wm * 128 + 16 * (fg & 1) + 4 * fa) * 2);        // + m * 64 (32 features)
  const uint32_t cbase = (uint32_t)((8 * (fg >> 1) + fb) * P32 + 512 + (wn * 64 + 16 * (fg & 1) + 4 * fa) * 2);  // + n * 64
  typedef __attribute__((address_space(3))) s16x4_t* lds_ptr_t;
  auto frag = [&](uint32_t byte_off) {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(uintptr_t)(byte_off));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(uintptr_t)(byte_off + 4 * P32));
    uint4 v;
    __builtin_memcpy(&v, &lo, 8);
    __builtin_memcpy(reinterpret_cast<char*>(&v) + 8, &hi, 8);
    return v;
  };
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
  issue(0);
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  for (int kt = 0; kt < nk; ++kt) {
    const uint32_t st0 = lds0 + (kt & 1) * STAGE32;
    if (kt + 1 < nk) issue(kt + 1);
    // one rolling pipeline over the step's 16 rows (4 k-steps of 16 tokens x 4 row blocks): the row fragment two rows ahead and, at
    // row block 2, the column fragments of the next k-step are requested before this row's two MFMAs
    constexpr int ROWS = 16;
    uint4 bq[2][2], aq[4];
    bq[0][0] = frag(st0 + cbase);
    bq[0][1] = frag(st0 + cbase + 64);
    aq[0] = frag(st0 + rbase);
    aq[1] = frag(st0 + rbase + 64);
#pragma unroll
    for (int row = 0; row < ROWS; ++row) {
      const int ks = row / 4, m = row % 4, r2 = row + 2;
      if (r2 < ROWS) aq[r2 & 3] = frag(st0 + (r2 / 4) * 16 * P32 + rbase + (r2 % 4) * 64);
      if (m == 2 && ks + 1 < 4) {
        bq[(ks + 1) & 1][0] = frag(st0 + (ks + 1) * 16 * P32 + cbase);
        bq[(ks + 1) & 1][1] = frag(st0 + (ks + 1) * 16 * P32 + cbase + 64);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        bf16x8 fa8, fb8;
        __builtin_memcpy(&fa8, &aq[row & 3], 16);
        __builtin_memcpy(&fb8, &bq[ks & 1][n], 16);
        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa8, fb8, acc[m][n], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
  }
  // epilogue: D[i][j] of a 32 x 32 block: j = lane % 32, i = (r % 4) + 8 * (r / 4) + 4 * (lane / 32)
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int64_t col = j0 + wn * 64 + n * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = i0 + wm * 128 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < K && col < K) {
          float* p = H + row * K + col;
          *p = beta * (*p) + alpha * acc[m][n][r];
        }
      }
    }
}

// ---- candidate B: four waves, one per SIMD, wave tile 128 x 128 (a third fewer fragment reads per MFMA), 256 accumulators per lane
// kept in AGPRs through "+a" asm operands, v_mfma_f32_16x16x32_bf16, the product's ring (two 64-token stages, pitch 1056).  Same MFMAs
// per output element in the same token order as the product: bit-identical H expected.
constexpr int P16 = 1056;
constexpr int STAGE16 = TOK * P16;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;

template <bool FIRST>
__device__ __forceinline__ void mfma_acc(f32x4_t& c, const uint4& a, const uint4& b) {
  i32x4_t av, bv;
  __builtin_memcpy(&av, &a, 16);
  __builtin_memcpy(&bv, &b, 16);
  if constexpr (FIRST) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(av), "v"(bv));  // (C = 0: no zero-fill of 256 AGPRs)
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(av), "v"(bv));
}

template <int VARIANT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void syrk4w_kernel(const uint16_t* __restrict__ x, int64_t T, int64_t K, int64_t ldx, float* __restrict__ H,
                                                     float beta, float alpha, int nt) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  int ti, tj;
  xcd_supertile_decode((int)blockIdx.x, (int)gridDim.x, nt, ti, tj);
  const int64_t i0 = (int64_t)ti * H2, j0 = (int64_t)tj * H2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem_raw;
  int64_t f = (lane < 32 ? i0 : j0) + 8 * (lane & 31);
  if (f > K - 8) f = K - 8;
  const uint32_t voff = (uint32_t)(f * 2);
  const int nk = (int)(T / TOK);
  uint32_t voffr[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) voffr[i] = voff + (uint32_t)((int64_t)i * ldx * 2);
  auto issue_half = [&](int kt, int h) {  // rows wave * 16 + 8 h .. + 7 of step kt
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (kt & 1) * STAGE16 + (wave * 16 + 8 * h) * P16);
    lds_dma_8x1k<P16>(x + ((int64_t)kt * TOK + wave * 16 + 8 * h) * ldx, dst, voffr[0], voffr[1], voffr[2], voffr[3], voffr[4], voffr[5], voffr[6], voffr[7]);
  };
  f32x4_t acc[8][8];
  const int s16 = lane & 15, fa = s16 & 3, fb = s16 >> 2, fg = lane >> 4;
  const uint32_t rbase = (uint32_t)((4 * fg + fb) * P16 + (wm * 128 + 4 * fa) * 2);        // + m * 32
  const uint32_t cbase = (uint32_t)((4 * fg + fb) * P16 + 512 + (wn * 128 + 4 * fa) * 2);  // + n * 32
  typedef __attribute__((address_space(3))) s16x4_t* lds_ptr_t;
  auto frag = [&](uint32_t byte_off) {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(uintptr_t)(byte_off));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(uintptr_t)(byte_off + 16 * P16));
    uint4 v;
    __builtin_memcpy(&v, &lo, 8);
    __builtin_memcpy(reinterpret_cast<char*>(&v) + 8, &hi, 8);
    return v;
  };
  issue_half(0, 0);
  issue_half(0, 1);
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  auto step = [&](auto first_c, int kt) {
    constexpr bool FIRST = decltype(first_c)::value;
    const uint32_t st0 = lds0 + (kt & 1) * STAGE16;
    const bool more = kt + 1 < nk;
    if (VARIANT == 0 && more) { issue_half(kt + 1, 0); issue_half(kt + 1, 1); }
    constexpr int ROWS = 16;  // 2 sub-steps of 32 tokens x 8 row fragments, 8 MFMAs per row
    uint4 bq[2][8], aq[4];
#pragma unroll
    for (int n = 0; n < 8; ++n) bq[0][n] = frag(st0 + cbase + n * 32);
    aq[0] = frag(st0 + rbase);
    aq[1] = frag(st0 + rbase + 32);
#pragma unroll
    for (int row = 0; row < ROWS; ++row) {
      const int kk = row / 8, m = row % 8, r2 = row + 2;
      if (r2 < ROWS) aq[r2 & 3] = frag(st0 + (r2 / 8) * 32 * P16 + rbase + (r2 % 8) * 32);
      if (kk == 0 && m >= 4) {  // the next sub-step's column fragments, two per row
        bq[1][2 * (m - 4)] = frag(st0 + 32 * P16 + cbase + (2 * (m - 4)) * 32);
        bq[1][2 * (m - 4) + 1] = frag(st0 + 32 * P16 + cbase + (2 * (m - 4) + 1) * 32);
      }
      if (VARIANT == 1 && more && (row == 0 || row == 1)) {  // variant 1: the pieces behind the first fragment requests of the step
        __builtin_amdgcn_sched_barrier(0);
        issue_half(kt + 1, row);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        if (FIRST && row < 8) mfma_acc<true>(acc[m][n], aq[row & 3], bq[kk][n]);
        else mfma_acc<false>(acc[m][n], aq[row & 3], bq[kk][n]);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
  };
  step(std::true_type{}, 0);
  for (int kt = 1; kt < nk; ++kt) step(std::false_type{}, kt);
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      const int64_t col = j0 + wn * 128 + n * 16 + (lane & 15);
      f32x4_t v = acc[m][n];
      asm volatile("s_nop 7\n\ts_nop 7" : "+a"(v));  // (the last MFMA's result before it is read back)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = i0 + wm * 128 + m * 16 + 4 * (lane >> 4) + r;
        if (row < K && col < K) {
          float* p = H + row * K + col;
          *p = beta * (*p) + alpha * v[r];
        }
      }
    }
}

__global__ void fill_bf16_kernel(uint16_t* x, size_t n, uint32_t seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u ^ (uint32_t)(i >> 32) * 40503u ^ seed;
    float a = 0.f;
    for (int r = 0; r < 4; ++r) {
      h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
      a += (float)(h & 0xffffff) * (1.f / 16777216.f) - 0.5f;
    }
    const uint32_t u = __float_as_uint(a * 1.7f);
    x[i] = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
  }
}

// upper-triangular tiles only: max |a - b| / max |b| and the number of differing elements
__global__ void compare_kernel(const float* a, const float* b, int64_t K, float* maxdiff, float* maxabs, unsigned long long* ndiff) {
  float md = 0.f, ma = 0.f;
  unsigned long long nd = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < K * K; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / K, c = i % K;
    if (c / 256 < r / 256) continue;
    const float d = fabsf(a[i] - b[i]);
    md = fmaxf(md, d);
    ma = fmaxf(ma, fabsf(b[i]));
    nd += __float_as_uint(a[i]) != __float_as_uint(b[i]);
  }
  atomicMax((int*)maxdiff, __float_as_int(md));
  atomicMax((int*)maxabs, __float_as_int(ma));
  if (nd) atomicAdd(ndiff, nd);
}

struct Timer {
  hipEvent_t a, b;
  Timer() { HIPCHECK(hipEventCreate(&a)); HIPCHECK(hipEventCreate(&b)); }
  void start() { HIPCHECK(hipEventRecord(a, 0)); }
  float stop_ms() { HIPCHECK(hipEventRecord(b, 0)); HIPCHECK(hipEventSynchronize(b)); float ms; HIPCHECK(hipEventElapsedTime(&ms, a, b)); return ms; }
};

}  // namespace

int main(int argc, char** argv) {
  const int64_t T = argc > 1 ? atoll(argv[1]) : 16384, K = argc > 2 ? atoll(argv[2]) : 11008;
  uint16_t* x;
  float *Href, *Hc;
  HIPCHECK(hipMalloc(&x, (size_t)T * K * 2));
  HIPCHECK(hipMalloc(&Href, (size_t)K * K * 4));
  HIPCHECK(hipMalloc(&Hc, (size_t)K * K * 4));
  fill_bf16_kernel<<<2048, 256>>>(x, (size_t)T * K, 0x1234567u);
  HIPCHECK(hipMemset(Href, 0, (size_t)K * K * 4));
  HIPCHECK(hipMemset(Hc, 0, (size_t)K * K * 4));
  const int nt = (int)((K + 255) / 256), ntiles = nt * (nt + 1) / 2;
  const size_t smem = (size_t)2 * STAGE32;
  HIPCHECK(hipFuncSetAttribute((const void*)syrk32_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  auto product = [&](float* H, float beta, float alpha) {
    if (inc_gptq_hessian_accum(x, INC_BF16, T, K, K, H, beta, alpha, nullptr) != INC_OK) { fprintf(stderr, "product launch failed\n"); exit(1); }
  };
  const int which_cand = argc > 3 ? atoi(argv[3]) : 0;  // 0: 32x32x16 MFMA (8 waves); 1 / 2: four waves with AGPR accumulators (issue at the top / behind the first reads)
  const size_t smem4 = (size_t)2 * STAGE16;
  HIPCHECK(hipFuncSetAttribute((const void*)syrk4w_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem4));
  HIPCHECK(hipFuncSetAttribute((const void*)syrk4w_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem4));
  auto cand = [&](float* H, float beta, float alpha) {
    if (which_cand == 0) syrk32_kernel<0><<<ntiles, 512, smem, 0>>>(x, T, K, K, H, beta, alpha, nt);
    else if (which_cand == 1) syrk4w_kernel<0><<<ntiles, 256, smem4, 0>>>(x, T, K, K, H, beta, alpha, nt);
    else syrk4w_kernel<1><<<ntiles, 256, smem4, 0>>>(x, T, K, K, H, beta, alpha, nt);
  };
  product(Href, 0.f, 1.f);
  product(Href, 0.5f, 0.25f);
  cand(Hc, 0.f, 1.f);
  cand(Hc, 0.5f, 0.25f);
  HIPCHECK(hipDeviceSynchronize());
  float *dmd, *dma;
  unsigned long long* dnd;
  HIPCHECK(hipMalloc(&dmd, 4)); HIPCHECK(hipMalloc(&dma, 4)); HIPCHECK(hipMalloc(&dnd, 8));
  HIPCHECK(hipMemset(dmd, 0, 4)); HIPCHECK(hipMemset(dma, 0, 4)); HIPCHECK(hipMemset(dnd, 0, 8));
  compare_kernel<<<2048, 256>>>(Hc, Href, K, dmd, dma, dnd);
  float md, ma;
  unsigned long long nd;
  HIPCHECK(hipMemcpy(&md, dmd, 4, hipMemcpyDeviceToHost)); HIPCHECK(hipMemcpy(&ma, dma, 4, hipMemcpyDeviceToHost)); HIPCHECK(hipMemcpy(&nd, dnd, 8, hipMemcpyDeviceToHost));
  printf("HESS_LAB T=%ld K=%ld (%d tiles): candidate %d vs product: max |diff| %.3g, max |H| %.3g (rel %.2e), %llu elements differ  %s\n", (long)T, (long)K,
         ntiles, which_cand, md, ma, md / ma, nd, md <= 1e-5f * ma ? "OK" : "FAIL");
  Timer t;
  std::vector<float> tp, tc;
  for (int pass = 0; pass < 6; ++pass)
    for (int which = 0; which < 2; ++which) {
      const bool c = (which ^ (pass & 1)) != 0;
      if (c) cand(Hc, 0.5f, 0.5f); else product(Href, 0.5f, 0.5f);
      t.start();
      for (int i = 0; i < 5; ++i) { if (c) cand(Hc, 0.5f, 0.5f); else product(Href, 0.5f, 0.5f); }
      (c ? tc : tp).push_back(t.stop_ms() / 5);
    }
  std::sort(tp.begin(), tp.end()); std::sort(tc.begin(), tc.end());
  const double fl = 2.0 * T * K * K;
  printf("  product   median %8.4f ms (min %8.4f)  %7.1f TFLOP/s (2*T*K^2)\n", tp[tp.size() / 2], tp[0], fl / tp[tp.size() / 2] / 1e9);
  printf("  candidate median %8.4f ms (min %8.4f)  %7.1f TFLOP/s (2*T*K^2)\n", tc[tc.size() / 2], tc[0], fl / tc[tc.size() / 2] / 1e9);
  return md <= 1e-5f * ma ? 0 : 1;
}
