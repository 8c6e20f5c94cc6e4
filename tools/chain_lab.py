#!/usr/bin/env python3
"""Lab harness for the serial column chain (test infrastructure): cuts gptq_quant_block_q4_kernel out of
neural_compressor_amd/csrc/gptq.hip into a stand-alone program with three forms of the step's output selects --
  orig  the three selects left to the compiler (it sinks them to the end of the chain: 306 registers)
  pinv  pinned with an `asm volatile` (the product form: 121 registers)
  pinn  pinned + the rank-1 updates as packed fp32 operations on register pairs
-- times 32 launches of each at N rows x 4096 columns (alone on the chip; 64 KiB and 82 KiB of LDS: two / one workgroup per CU)
and compares the emitted codes.  usage: tools/chain_lab.py [build-dir]   ->  <build-dir>/chain_time ; run it with N as argument.
Numbers: profiles/r5/chain_pin_timing.log."""
import os, subprocess, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "tools")
src = open(os.path.join(root, "neural_compressor_amd", "csrc", "gptq.hip")).read().split("\n")
start = next(i for i, l in enumerate(src) if "the serial column chain, second generation" in l) - 3
end = next(i for i, l in enumerate(src) if l.startswith("// lazy update: W[:, i2:]")) - 2
body = "\n".join(src[start:end])
pin = '    asm volatile("" : "+v"(wr[ci]), "+v"(ev[ci]), "+v"(cw[ci >> 2]));'
assert pin in body, "the product's pin statement moved: update this script"
body = body.replace("  __builtin_amdgcn_s_setprio(3);\n", "")
def packed(b):
    """the rank-1 updates as v_pk_mul_f32 / v_pk_add_f32 on register pairs (same roundings: un-fused multiply then subtract)"""
    b = b.replace("  float wr[32], ev[32];", "  typedef float f32x2 __attribute__((ext_vector_type(2)));\n  f32x2 wr2[16];\n  float ev[32];")
    b = b.replace("    wr[c] = w[rowc * K + i1 + 4 * c + q];", "    wr2[c >> 1][c & 1] = w[rowc * K + i1 + 4 * c + q];")
    b = b.replace("vmax = fmaxf(vmax, wr[c]);", "vmax = fmaxf(vmax, wr2[c >> 1][c & 1]);").replace("vmin = fminf(vmin, wr[c]);", "vmin = fminf(vmin, wr2[c >> 1][c & 1]);")
    b = b.replace("  float ha[33], hb2[33];", "  f32x2 ha[17], hb2[17];")
    b = b.replace("auto fetch_row = [&](auto ic, float (&h)[33]) {", "auto fetch_row = [&](auto ic, f32x2 (&h)[17]) {")
    b = b.replace("      for (int c = i >> 2; c < 32; ++c) h[c] = hs[i * QB + 4 * c + q];\n      h[32] = hs[i * QB + i];",
                  "      for (int c = i >> 2; c < 32; ++c) h[c >> 1][c & 1] = hs[i * QB + 4 * c + q];\n      h[16][0] = hs[i * QB + i];")
    b = b.replace("auto step = [&](auto ic, float (&h)[33]) {", "auto step = [&](auto ic, f32x2 (&h)[17]) {")
    b = b.replace("const float x = quad_bcast(wr[ci], qo);", "const float x = quad_bcast(wr2[ci >> 1][ci & 1], qo);")
    b = b.replace("const float e = (x - qv) / h[32];", "const float e = (x - qv) / h[16][0];")
    old_loop = """#pragma unroll
    for (int c = ci; c < 32; ++c) {
      const float pr = e * h[c];  // exact zero below the diagonal (Hinv is upper triangular)
      wr[c] = wr[c] - pr;
    }"""
    new_loop = """    if constexpr ((ci & 1) != 0) {
      const float pr = e * h[ci >> 1][1];
      wr2[ci >> 1][1] = wr2[ci >> 1][1] - pr;
    }
    {
      const f32x2 e2 = {e, e};
#pragma unroll
      for (int c2 = (ci + 1) >> 1; c2 < 16; ++c2) {
        const f32x2 pr = e2 * h[c2];  // exact zero below the diagonal (Hinv is upper triangular)
        wr2[c2] = wr2[c2] - pr;
      }
    }"""
    assert old_loop in b
    b = b.replace(old_loop, new_loop)
    b = b.replace("    wr[ci] = own ? qv : wr[ci];", "    wr2[ci >> 1][ci & 1] = own ? qv : wr2[ci >> 1][ci & 1];")
    b = b.replace('"+v"(wr[ci]), "+v"(ev[ci])', '"+v"(wr2[ci >> 1]), "+v"(ev[ci])')
    b = b.replace("st[r * QB + 4 * c + q] = wr[c];", "st[r * QB + 4 * c + q] = wr2[c >> 1][c & 1];")
    assert "wr[" not in b.replace("wr2[", ""), [l for l in b.split("\n") if "wr[" in l.replace("wr2[", "")][:5]
    return b


variants = {"orig": body.replace(pin, ""), "pinv": body, "pinn": packed(body)}
code = """#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <type_traits>
#include <vector>
#include <algorithm>
#include "%(root)s/include/inc_mi355x.h"
#include "%(root)s/neural_compressor_amd/csrc/common.hpp"
#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%%s:%%d %%s\\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
""" % dict(root=root)
for name, b in variants.items():
    code += f"namespace {name} {{\nconstexpr int QB = 128;\nconstexpr int QROWS = 64;\nconstexpr int QPITCH = QB + 1;\n{b}\n}}\n"
code += r"""
__global__ void fill(float* p, size_t n, float scale, uint32_t seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u ^ seed; h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    p[i] = scale * ((float)(h & 0xffffff) / 16777216.f - 0.5f);
  }
}
__global__ void fix_hinv(float* H, int64_t K) {  // upper triangular, positive diagonal
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= K * K) return;
  int64_t r = i / K, c = i % K;
  if (c < r) H[i] = 0.f; else if (c == r) H[i] = 0.75f + 0.25f * fabsf(H[i]) * 50.f;
}
template <typename KF>
float time_kernel(KF launch, int blocks) {
  hipEvent_t a, b; HIPCHECK(hipEventCreate(&a)); HIPCHECK(hipEventCreate(&b));
  for (int i = 0; i < blocks; ++i) launch(i);
  HIPCHECK(hipEventRecord(a, 0));
  for (int i = 0; i < blocks; ++i) launch(i);
  HIPCHECK(hipEventRecord(b, 0)); HIPCHECK(hipEventSynchronize(b));
  float ms; HIPCHECK(hipEventElapsedTime(&ms, a, b)); return ms / blocks;
}
int main(int argc, char** argv) {
  const int64_t N = argc > 1 ? atoll(argv[1]) : 4096, K = 4096, G = K / 128;
  float *w, *Hinv, *scale, *zero, *err; uint8_t* codes[3]; uint16_t* q;
  HIPCHECK(hipMalloc(&w, N * K * 4)); HIPCHECK(hipMalloc(&Hinv, K * K * 4)); HIPCHECK(hipMalloc(&scale, N * G * 4)); HIPCHECK(hipMalloc(&zero, N * G * 4));
  HIPCHECK(hipMalloc(&err, N * 128 * 4)); HIPCHECK(hipMalloc(&q, N * K * 2));
  for (int i = 0; i < 3; ++i) { HIPCHECK(hipMalloc(&codes[i], N * K)); HIPCHECK(hipMemset(codes[i], 0, N * K)); }
  fill<<<1024, 256>>>(w, (size_t)N * K, 0.05f, 1u); fill<<<1024, 256>>>(Hinv, (size_t)K * K, 0.04f, 2u); fix_hinv<<<(unsigned)((K * K + 255) / 256), 256>>>(Hinv, K);
  HIPCHECK(hipDeviceSynchronize());
  const unsigned blocks4 = (unsigned)((N + 63) / 64);
  auto run = [&](int which, size_t smem, uint8_t* cd, const char* label) {
    auto launch = [&](int b) {
      const int64_t i1 = (int64_t)(b % 32) * 128;
#define ARGS w, Hinv, scale, zero, cd, (void*)q, err, N, K, G, i1, i1 / 128, 15.f, 1
      if (which == 0) orig::gptq_quant_block_q4_kernel<2, 1, true, 4><<<blocks4, 256, smem, 0>>>(ARGS);
      else if (which == 1) pinv::gptq_quant_block_q4_kernel<2, 1, true, 4><<<blocks4, 256, smem, 0>>>(ARGS);
      else pinn::gptq_quant_block_q4_kernel<2, 1, true, 4><<<blocks4, 256, smem, 0>>>(ARGS);
    };
    std::vector<float> t;
    for (int rep = 0; rep < 5; ++rep) t.push_back(time_kernel(launch, 32));
    std::sort(t.begin(), t.end());
    printf("  %-44s smem %6zu: median %7.2f us per launch (min %7.2f)\n", label, smem, 1000.f * t[2], 1000.f * t[0]);
  };
#define SETATTR(NS) HIPCHECK(hipFuncSetAttribute((const void*)NS::gptq_quant_block_q4_kernel<2, 1, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304))
  SETATTR(orig); SETATTR(pinv); SETATTR(pinn);
  printf("CHAIN_LAB N=%ld K=%ld (%u workgroups)\n", (long)N, (long)K, blocks4);
  for (int pass = 0; pass < 2; ++pass) {
    run(0, 65536, codes[0], "original (selects sunk, 306 registers)");
    run(1, 65536, codes[1], "pinned, asm volatile");
    run(1, 83968, codes[1], "pinned, asm volatile, one per CU");
    run(2, 65536, codes[2], "pinned + packed rank-1 updates");
    run(2, 83968, codes[2], "pinned + packed updates, one per CU");
  }
  std::vector<uint8_t> h0((size_t)N * K), h1((size_t)N * K), h2((size_t)N * K);
  HIPCHECK(hipMemcpy(h0.data(), codes[0], N * K, hipMemcpyDeviceToHost)); HIPCHECK(hipMemcpy(h1.data(), codes[1], N * K, hipMemcpyDeviceToHost)); HIPCHECK(hipMemcpy(h2.data(), codes[2], N * K, hipMemcpyDeviceToHost));
  size_t d1 = 0, d2 = 0; for (size_t i = 0; i < h0.size(); ++i) { d1 += h0[i] != h1[i]; d2 += h0[i] != h2[i]; }
  printf("  codes differing from the original: %zu (pinned), %zu (pinned + packed)\n", d1, d2);
  return 0;
}
"""
path = os.path.join(out_dir, "chain_time.hip")
open(path, "w").write(code)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-w", path, "-o", os.path.join(out_dir, "chain_time")])
print("built", os.path.join(out_dir, "chain_time"))
