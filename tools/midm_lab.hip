// midm_lab -- the mid-M fused dequant-GEMM candidates against the product route (inc_woq_gemm), results compared and timed interleaved.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ineural_compressor_amd/csrc tools/midm_lab.hip neural_compressor_amd/csrc/gemm_strip8.hip \
//         -Lneural_compressor_amd -linc_mi355x -Wl,-rpath,'$ORIGIN/../neural_compressor_amd' -o tools/midm_lab
//   tools/midm_lab [M N K ...]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "inc_mi355x.h"

int inc_woq_gemm_strip8_splitk(int64_t M, int64_t N, int64_t K);
int inc_launch_woq_gemm_strip8(const uint16_t* x, const uint32_t* qw, const uint16_t* scales, const uint32_t* qz, const uint16_t* bias,
                               uint16_t* y, int64_t M, int64_t N, int64_t K, int64_t NW, int g_shift, float* part, unsigned* counters,
                               int splitk, bool bf, hipStream_t s);

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t b; memcpy(&b, &h, 2); return b; }

static int run(int64_t M, int64_t N, int64_t K) {
  const int gs = 128;
  const int64_t G = K / gs, KW = K / 8, NW = (N + 7) / 8;
  std::mt19937 rng(1234);
  std::vector<uint32_t> qw(KW * N), qz(G * NW);
  std::vector<uint16_t> sc(G * N), xh(M * K);
  for (auto& v : qw) v = rng();
  for (auto& v : qz) v = rng();
  std::uniform_real_distribution<float> us(0.005f, 0.02f);
  for (auto& v : sc) v = f2h(us(rng));
  std::normal_distribution<float> nx(0.f, 1.f);
  for (auto& v : xh) v = f2bf(nx(rng));
  uint32_t *dqw, *dqz; uint16_t *dsc, *dx, *y0, *y1; void* ws; void* ws2;
  CK(hipMalloc(&dqw, qw.size() * 4)); CK(hipMalloc(&dqz, qz.size() * 4)); CK(hipMalloc(&dsc, sc.size() * 2)); CK(hipMalloc(&dx, xh.size() * 2));
  CK(hipMalloc(&y0, M * N * 2)); CK(hipMalloc(&y1, M * N * 2));
  CK(hipMemcpy(dqw, qw.data(), qw.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dqz, qz.data(), qz.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsc, sc.data(), sc.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dx, xh.data(), xh.size() * 2, hipMemcpyHostToDevice));
  const int64_t wsb = std::max<int64_t>(inc_woq_gemm_workspace_bytes(M, N, K), 16384 + 4 * M * N * 4) + 1024;
  CK(hipMalloc(&ws, wsb)); CK(hipMemset(ws, 0, wsb)); CK(hipMalloc(&ws2, wsb)); CK(hipMemset(ws2, 0, wsb));
  hipStream_t s; CK(hipStreamCreate(&s));
  const int sk = inc_woq_gemm_strip8_splitk(M, N, K);
  auto old_route = [&]() { return inc_woq_gemm(dx, INC_BF16, (const int32_t*)dqw, dsc, (const int32_t*)dqz, nullptr, nullptr, y0, M, N, K, G, gs, 4, ws, wsb, s); };
  auto new_route = [&]() {
    return inc_launch_woq_gemm_strip8(dx, dqw, dsc, dqz, nullptr, y1, M, N, K, NW, 7, sk > 1 ? (float*)((char*)ws2 + 16384) : nullptr, (unsigned*)ws2, sk, true, s);
  };
  if (old_route() != 0 || new_route() != 0) { printf("launch failed\n"); return 1; }
  CK(hipStreamSynchronize(s));
  std::vector<uint16_t> h0(M * N), h1(M * N);
  CK(hipMemcpy(h0.data(), y0, M * N * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), y1, M * N * 2, hipMemcpyDeviceToHost));
  double num = 0, den = 0, worst = 0; int64_t diff = 0;
  for (int64_t i = 0; i < M * N; ++i) {
    const double a = bf2f(h0[i]), b = bf2f(h1[i]);
    num += (a - b) * (a - b); den += a * a; worst = std::max(worst, std::fabs(a - b)); diff += h0[i] != h1[i];
  }
  // second launch: determinism and re-armed counters
  new_route(); CK(hipStreamSynchronize(s));
  std::vector<uint16_t> h2(M * N);
  CK(hipMemcpy(h2.data(), y1, M * N * 2, hipMemcpyDeviceToHost));
  const bool same = h1 == h2;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> to, tn;
  for (int i = 0; i < 30; ++i) { old_route(); new_route(); }
  for (int r = 0; r < 7; ++r) {
    float ms;
    CK(hipEventRecord(e0, s)); for (int i = 0; i < 20; ++i) old_route(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); to.push_back(ms / 20 * 1e3f);
    CK(hipEventRecord(e0, s)); for (int i = 0; i < 20; ++i) new_route(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); tn.push_back(ms / 20 * 1e3f);
  }
  std::sort(to.begin(), to.end()); std::sort(tn.begin(), tn.end());
  const double fl = 2.0 * M * N * K;
  printf("M=%ld N=%ld K=%ld splitk=%d: rel-Frobenius new vs product %.2e, max abs diff %.3g, bf16 outputs that differ %ld of %ld, repeat identical %s | product %.1f us (%.0f TFLOP/s), strip8 %.1f us (%.0f TFLOP/s, %.3f of 2500)\n",
         (long)M, (long)N, (long)K, sk, std::sqrt(num / den), worst, (long)diff, (long)(M * N), same ? "yes" : "NO", to[3], fl / to[3] / 1e6, tn[3], fl / tn[3] / 1e6,
         fl / tn[3] / 1e6 / 2500.0);
  hipFree(dqw); hipFree(dqz); hipFree(dsc); hipFree(dx); hipFree(y0); hipFree(y1); hipFree(ws); hipFree(ws2);
  return (std::sqrt(num / den) < 3e-3 && same) ? 0 : 1;
}

int main(int argc, char** argv) {
  int fails = 0;
  if (argc >= 4) {
    for (int i = 1; i + 2 < argc; i += 3) fails += run(atol(argv[i]), atol(argv[i + 1]), atol(argv[i + 2]));
  } else {
    const int64_t cases[][3] = {{256, 4096, 4096}, {512, 4096, 4096}, {128, 4096, 4096}, {192, 4096, 4096}, {1024, 4096, 4096}, {256, 11008, 4096},
                                {256, 4096, 11008}, {200, 1000, 4096}, {384, 4096, 4096}};
    for (auto& c : cases) fails += run(c[0], c[1], c[2]);
  }
  printf("%s\n", fails ? "FAIL" : "ok");
  return fails;
}
