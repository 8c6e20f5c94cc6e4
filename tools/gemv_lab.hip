// gemv_lab -- the decode (M = 1) streaming body of csrc/gemm.hip with timing-only ablations (WRONG results unless GL_ABL = 0):
//   GL_ABL bit 0: no dequantise + MFMA; bit 1: no split-K hand-off (every slice writes y itself); bit 2: all weight requests hit one row.
// tools/gemv_lab_body.inc is generated from csrc/gemm.hip (see NOTES round 6).  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DGL_ABL=<n> -Iinclude -Ineural_compressor_amd/csrc tools/gemv_lab.hip -o tools/gemv_lab_a<n>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gemm_common.hpp"
#ifndef GL_FORM
#define GL_FORM 0
#endif
#ifndef GL_ABL
#define GL_ABL 0
#endif
namespace {
#include "gemv_lab_body.inc"
constexpr int GL_MAX = 8;
struct Batch {
  const uint32_t* qweight[GL_MAX];
  const uint16_t* scales[GL_MAX];
  const uint32_t* qzeros[GL_MAX];
  uint16_t* y[GL_MAX];
  int64_t N[GL_MAX];
  int64_t part_off[GL_MAX];
  int first[GL_MAX + 1];
  int n;
};
template <int VSTEPS>
__global__ __launch_bounds__(256) void lab_kernel(Batch args, const uint16_t* __restrict__ x, float* __restrict__ partial, unsigned* __restrict__ counters,
                                                  int M, int64_t K, int g_shift, int splitk) {
  const int b = (int)blockIdx.x;
  int p = 0;
#pragma unroll
  for (int i = 1; i < GL_MAX; ++i)
    if (i < args.n && b >= args.first[i]) p = i;
  p = __builtin_amdgcn_readfirstlane(p);
  const int64_t N = args.N[p];
  woq_gemv_w4_body<true, true, VSTEPS, 1>(x, args.qweight[p], args.scales[p], args.qzeros[p], nullptr, args.y[p], partial + args.part_off[p], counters + b, M, N, K,
                                          (N + 7) / 8, g_shift, splitk, b - args.first[p], (int)blockIdx.y);
}
}  // namespace

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 3;
  const int64_t N = argc > 2 ? atol(argv[2]) : 4096, K = argc > 3 ? atol(argv[3]) : 4096;
  const int vsteps = argc > 4 ? atoi(argv[4]) : 8;
  const int ring = (int)std::max<int64_t>(2, (int64_t)(600ll << 20) / (n * N * K / 2));
  const int64_t G = K / 128, NW = N / 8;
  std::vector<Batch> bs(ring);
  const int splitk = (int)((K + 32 * vsteps * 4 - 1) / (32 * vsteps * 4));
  uint16_t* x; CK(hipMalloc(&x, K * 2)); CK(hipMemset(x, 0x3c, K * 2));
  void* ws; const int64_t wsb = 16384 + (int64_t)splitk * n * N * 4; CK(hipMalloc(&ws, wsb)); CK(hipMemset(ws, 0, wsb));
  for (int r = 0; r < ring; ++r) {
    Batch& b = bs[r]; b.n = n; int first = 0; int64_t off = 0;
    for (int i = 0; i < n; ++i) {
      void *qw, *sc, *qz, *y;
      CK(hipMalloc(&qw, K / 8 * N * 4)); CK(hipMemset(qw, 0x5a, K / 8 * N * 4));
      CK(hipMalloc(&sc, G * N * 2)); CK(hipMemset(sc, 0x2c, G * N * 2));
      CK(hipMalloc(&qz, G * NW * 4)); CK(hipMemset(qz, 0x77, G * NW * 4));
      CK(hipMalloc(&y, N * 2));
      b.qweight[i] = (const uint32_t*)qw; b.scales[i] = (const uint16_t*)sc; b.qzeros[i] = (const uint32_t*)qz; b.y[i] = (uint16_t*)y; b.N[i] = N;
      b.part_off[i] = off; b.first[i] = first; off += (int64_t)splitk * N; first += (int)(N / 64);
    }
    for (int i = n; i <= GL_MAX; ++i) b.first[i] = first;
  }
  hipStream_t s; CK(hipStreamCreate(&s));
  dim3 grid((unsigned)(n * N / 64), (unsigned)splitk);
  auto launch = [&](int r) {
    if (vsteps == 8) lab_kernel<8><<<grid, 256, 0, s>>>(bs[r], x, (float*)((char*)ws + 16384), (unsigned*)ws, 1, K, 7, splitk);
    else lab_kernel<4><<<grid, 256, 0, s>>>(bs[r], x, (float*)((char*)ws + 16384), (unsigned*)ws, 1, K, 7, splitk);
  };
  for (int w = 0; w < 2; ++w) for (int r = 0; r < ring; ++r) launch(r);
  CK(hipStreamSynchronize(s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> t;
  for (int rep = 0; rep < 7; ++rep) {
    CK(hipEventRecord(e0, s)); for (int r = 0; r < ring; ++r) launch(r); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms / ring * 1e3f);
  }
  std::sort(t.begin(), t.end());
  const double bytes = n * (N * K / 2.0 + G * N * 2.0 + G * NW * 4.0);
  printf("GL_ABL=%d n=%d N=%ld K=%ld vsteps=%d splitk=%d grid=%ux%u ring=%d: %.2f us per launch (eager back-to-back, cold ring), %.0f GB/s = %.3f of 8 TB/s\n", GL_ABL, n, (long)N, (long)K,
         vsteps, splitk, grid.x, grid.y, ring, t[3], bytes / t[3] / 1e3, bytes / t[3] / 1e3 / 8000.0);
  return 0;
}
