// probe.hip -- measured ceilings for bench.py's `ceilings` object ("confirm the peaks with a stream triad and an
// MFMA-loop microbenchmark and state the measured ceilings next to the paper ones").  Measurement helpers, not part of the hot path:
// nothing in the product calls them.
#include "common.hpp"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// a[i] = b[i] + s * c[i], 16 bytes per lane and access, grid-stride: 3 * n * 4 bytes of HBM traffic
__global__ __launch_bounds__(256) void probe_triad_kernel(float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ c, float s, int64_t n4) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 x = b[i], y = c[i];
    a[i] = make_float4(x.x + s * y.x, x.y + s * y.y, x.z + s * y.z, x.w + s * y.w);
  }
}

// dst[i] = src[i], 16 bytes per lane and access: n bytes read + n bytes written.  UNROLL independent loads per lane are issued before
// the first store (memory-level parallelism is what a streaming kernel lives on); NT: non-temporal hints on both sides (every byte is
// touched once).  The guide's ceiling for this pattern is 6.29 TB/s (float4 copy, MI355X_MICROARCH.md "Chip-level parameters").
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void probe_copy_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, int64_t n16) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const int64_t chunk = (int64_t)256 * UNROLL;
  for (int64_t base = (int64_t)blockIdx.x * chunk; base < n16; base += (int64_t)gridDim.x * chunk) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t i = base + (int64_t)u * 256 + threadIdx.x;
      const int64_t ic = i < n16 ? i : n16 - 1;
      if constexpr (NT) v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src) + ic);
      else v[u] = reinterpret_cast<const u32x4*>(src)[ic];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t i = base + (int64_t)u * 256 + threadIdx.x;
      if (i < n16) {
        if constexpr (NT) __builtin_nontemporal_store(v[u], reinterpret_cast<u32x4*>(dst) + i);
        else reinterpret_cast<u32x4*>(dst)[i] = v[u];
      }
    }
  }
}

// every wave: `iters` x 8 v_mfma_f32_32x32x16_bf16 on four independent accumulators, operands from memory (random data: the chip
// clocks to its power budget, zero-filled operands would overstate the ceiling -- MI355X guide, DVFS)
__global__ __launch_bounds__(256) void probe_mfma_kernel(const uint4* __restrict__ src, float* __restrict__ sink, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 a[2], b[2];
  for (int i = 0; i < 2; ++i) {
    const uint4 va = src[(threadIdx.x * 4 + i) & 4095], vb = src[(threadIdx.x * 4 + 2 + i) & 4095];
    __builtin_memcpy(&a[i], &va, 16);
    __builtin_memcpy(&b[i], &vb, 16);
  }
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc[3], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  if (s == 12345.678f) sink[blockIdx.x * 256 + threadIdx.x] = s + lane;  // keeps the MFMAs alive, (almost) never stores
}

// empty kernel whose GRID SIZE is the message: rocprofv3 --kernel-trace lists Grid_Size_X = 64 * id for it, on the stream it was launched on
__global__ __launch_bounds__(64) void inc_trace_marker_kernel() {}

}  // namespace

extern "C" {

// phase marker for kernel-trace timelines (scripts/step_timeline.py): an empty launch of `id` workgroups of 64 threads on `stream`
int inc_trace_marker(int id, inc_stream_t stream) {
  INC_CHECK_ARG(id > 0 && id <= 4096);
  inc_trace_marker_kernel<<<id, 64, 0, inc_s(stream)>>>();
  INC_LAUNCH_RETURN();
}

// a <- b + s * c on n fp32 elements (n % 4 == 0, 16-byte aligned): HBM traffic 12 n bytes
int inc_probe_hbm_triad(float* a, const float* b, const float* c, float s, int64_t n, inc_stream_t stream) {
  INC_CHECK_ARG(a && b && c && n > 0 && (n % 4) == 0);
  probe_triad_kernel<<<256 * 16, 256, 0, inc_s(stream)>>>((float4*)a, (const float4*)b, (const float4*)c, s, n / 4);
  INC_LAUNCH_RETURN();
}

// dst <- src, `bytes` bytes (a multiple of 16, both 16-byte aligned): `bytes` read + `bytes` written.  variant = 4 * log2(UNROLL) + 2 * NT
// + (one workgroup per chunk instead of a grid-stride loop over 8 workgroups per CU); bench.py times all of them and reports the best
int inc_probe_hbm_copy(void* dst, const void* src, int64_t bytes, int variant, inc_stream_t stream) {
  INC_CHECK_ARG(dst && src && bytes > 0 && (bytes % 16) == 0 && variant >= 0 && variant < 16);
  const int64_t n16 = bytes / 16;
  const int lg = variant >> 2, unroll = 1 << lg;
  const bool nt = (variant & 2) != 0, flat = (variant & 1) != 0;
  const int64_t chunks = ceil_div64(n16, (int64_t)256 * unroll);
  int64_t grid = flat ? chunks : 256 * 8;
  if (grid > chunks) grid = chunks;
  if (grid > 0x7fffffff) return INC_ERR_UNSUPPORTED;
#define INC_COPY(U, T) probe_copy_kernel<U, T><<<(unsigned)grid, 256, 0, inc_s(stream)>>>((uint4*)dst, (const uint4*)src, n16)
  switch (lg) {
    case 0: if (nt) INC_COPY(1, true); else INC_COPY(1, false); break;
    case 1: if (nt) INC_COPY(2, true); else INC_COPY(2, false); break;
    case 2: if (nt) INC_COPY(4, true); else INC_COPY(4, false); break;
    default: if (nt) INC_COPY(8, true); else INC_COPY(8, false); break;
  }
#undef INC_COPY
  INC_LAUNCH_RETURN();
}

// `blocks` workgroups of 4 waves, each wave `iters` x 8 MFMA 32x32x16 bf16; src = 64 KiB of operand data; *flops_out (host) = flops launched
int inc_probe_mfma_bf16(const void* src, float* sink, int blocks, int iters, double* flops_out, inc_stream_t stream) {
  INC_CHECK_ARG(src && sink && blocks > 0 && iters > 0);
  probe_mfma_kernel<<<blocks, 256, 0, inc_s(stream)>>>((const uint4*)src, sink, iters);
  if (flops_out) *flops_out = (double)blocks * 4.0 * iters * 8.0 * 2.0 * 32 * 32 * 16;
  INC_LAUNCH_RETURN();
}

}  // extern "C"
