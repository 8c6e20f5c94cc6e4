// tools/f32mfma_lab.hip -- what the exact-fp32 matrix instruction (v_mfma_f32_32x32x2_f32, the column loop's trailing update and the
// factorisation's exact products) sustains on this chip: a bare loop, operands in registers (random, non-zero data: the chip's power draw
// depends on the bits), NACC independent accumulators per wave, WPS waves per SIMD.  Test infrastructure (profiles/NOTES.md round 6).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <int NACC>
__global__ __launch_bounds__(256) void loop_kernel(const float* __restrict__ src, float* __restrict__ sink, int iters) {
  const int tid = threadIdx.x + blockIdx.x * 256;
  float a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = src[(tid * 16 + i) & 0xfffff]; b[i] = src[(tid * 16 + 8 + i) & 0xfffff]; }
  f32x16 acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[(i + j) & 7], acc[j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NACC; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  if (s == 123.456f) sink[tid] = s;
}
// the same loop with the B operand read from LDS for every MFMA (4 KiB of random data per wave, a new word per MFMA) and the A operand
// rotating through 64 registers: what a real kernel's operand delivery looks like to the power management
template <int NACC>
__global__ __launch_bounds__(256) void loop_lds_kernel(const float* __restrict__ src, float* __restrict__ sink, int iters) {
  __shared__ float lds[4 * 64 * 64];
  const int tid = threadIdx.x + blockIdx.x * 256, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4 * 64 * 64; i += 256) lds[i] = src[(blockIdx.x * 16384 + i) & 0xfffff];
  float a[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) a[i] = src[(tid * 64 + i) & 0xfffff];
  __syncthreads();
  f32x16 acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const float* lp = lds + wave * 4096 + lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 64 / NACC; ++i)
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i * NACC + j], lp[(i * NACC + j) * 64], acc[j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NACC; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  if (s == 123.456f) sink[tid] = s;
}
template <int NACC>
static void run_lds(const float* src, float* sink, int blocks, int iters, const char* what) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  loop_lds_kernel<NACC><<<blocks, 256>>>(src, sink, iters / 8);
  CK(hipDeviceSynchronize());
  float best = 1e9f, last = 0.f;
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipEventRecord(e0));
    loop_lds_kernel<NACC><<<blocks, 256>>>(src, sink, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best; last = ms;
  }
  const double flops = (double)blocks * 4 * iters * 64 * (32.0 * 32 * 2 * 2);
  printf("%-44s blocks %4d  %d accumulators: best %.3f ms = %.1f TFLOP/s, sixth back-to-back launch %.3f ms = %.1f TFLOP/s\n", what, blocks, NACC, best,
         flops / best / 1e9, last, flops / last / 1e9);
}
template <int NACC>
static void run(const float* src, float* sink, int blocks, int iters, const char* what) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  loop_kernel<NACC><<<blocks, 256>>>(src, sink, iters / 8);
  CK(hipDeviceSynchronize());
  float best = 1e9f, last = 0.f;
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipEventRecord(e0));
    loop_kernel<NACC><<<blocks, 256>>>(src, sink, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best; last = ms;
  }
  const double flops = (double)blocks * 4 * iters * 8 * NACC * (32.0 * 32 * 2 * 2);
  printf("%-44s blocks %4d  %d accumulators: best %.3f ms = %.1f TFLOP/s, sixth back-to-back launch %.3f ms = %.1f TFLOP/s\n", what, blocks, NACC, best,
         flops / best / 1e9, last, flops / last / 1e9);
}
int main() {
  float *src, *sink;
  std::vector<float> h(1 << 20);
  srand(1);
  for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
  CK(hipMalloc(&src, h.size() * 4)); CK(hipMalloc(&sink, 4 << 20));
  CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  run<1>(src, sink, 256, 40000, "one wave per SIMD");
  run<2>(src, sink, 256, 20000, "one wave per SIMD");
  run<4>(src, sink, 256, 10000, "one wave per SIMD");
  run<1>(src, sink, 512, 20000, "two waves per SIMD");
  run<4>(src, sink, 512, 5000, "two waves per SIMD");
  run<1>(src, sink, 1024, 10000, "four waves per SIMD");
  run_lds<1>(src, sink, 256, 5000, "B from LDS per MFMA, one wave per SIMD");
  run_lds<2>(src, sink, 256, 5000, "B from LDS per MFMA, one wave per SIMD");
  run_lds<1>(src, sink, 512, 2500, "B from LDS per MFMA, two waves per SIMD");
  run_lds<4>(src, sink, 512, 2500, "B from LDS per MFMA, two waves per SIMD");
  return 0;
}
