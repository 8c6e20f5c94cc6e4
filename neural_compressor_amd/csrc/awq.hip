// awq.hip -- K8: AWQ statistics (HBM-bound column reductions).
//
// Reference (relative to /root/reference/neural_compressor/torch/algorithms/weight_only/awq.py):
//   _get_act_scale    :151-154   mean over tokens of |x|            -> inc_awq_act_abs_sum (sum; host /T)
//   _get_weight_scale :131-147   mean over rows of |w| / groupmax|w| -> inc_awq_weight_scale
//
// Both walk a [rows, K] matrix once with 16-byte loads per lane (8 x 16-bit or 2 x 4 fp32), keep eight
// per-column partial sums in registers across a strip of rows, fold the four waves of a workgroup through
// LDS and finish with one fp32 atomicAdd per column per workgroup (Guideline 12).
#include <math.h>

#include "common.hpp"

namespace {

constexpr int CW = 512;    // columns per workgroup (64 lanes x 8)
constexpr int RSTRIP = 256;  // rows per workgroup

template <int DT>
__device__ __forceinline__ void load8c(const void* p, int64_t idx, int nvalid, bool vec, float (&v)[8]) {
  if (vec && nvalid >= 8) {
    if constexpr (DT == INC_F32) {
      const float4 a = *reinterpret_cast<const float4*>(static_cast<const float*>(p) + idx);
      const float4 b = *reinterpret_cast<const float4*>(static_cast<const float*>(p) + idx + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
      const uint4 u = *reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(p) + idx);
      const uint32_t r[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (DT == INC_BF16) {
          v[2 * i] = __uint_as_float(r[i] << 16);
          v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u);
        } else {
          v[2 * i] = f16_bits_to_f32((uint16_t)(r[i] & 0xffffu));
          v[2 * i + 1] = f16_bits_to_f32((uint16_t)(r[i] >> 16));
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (i < nvalid) ? load_as_f32<DT>(p, idx + i) : 0.f;
  }
}

// out[k] += sum over the strip's rows of f(x[r,k]); MODE 0: |x|, MODE 1: |x| / amax[r, g(k)]
template <int DT, int MODE>
__global__ __launch_bounds__(256) void col_reduce_kernel(const void* __restrict__ x, int64_t R, int64_t K,
                                                         const float* __restrict__ amax, int64_t G, int gs,
                                                         float* __restrict__ out, int vec_ok) {
  __shared__ float red[4][CW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t c0 = (int64_t)blockIdx.x * CW + lane * 8;
  const int64_t r0 = (int64_t)blockIdx.y * RSTRIP;
  const int64_t r1 = (r0 + RSTRIP < R) ? r0 + RSTRIP : R;
  const int nv = (int)((K - c0) < 8 ? (K - c0 < 0 ? 0 : K - c0) : 8);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (nv > 0) {
    for (int64_t r = r0 + wave; r < r1; r += 4) {
      float v[8];
      load8c<DT>(x, r * K + c0, nv, vec_ok != 0, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (i < nv) {
          float a = fabsf(v[i]);
          if constexpr (MODE == 1) a = round_to<DT>(a / amax[r * G + (c0 + i) / gs]);
          acc[i] += a;
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) red[wave][lane * 8 + i] = acc[i];
  __syncthreads();
  for (int c = threadIdx.x; c < CW; c += 256) {
    const int64_t col = (int64_t)blockIdx.x * CW + c;
    if (col < K) atomicAdd(out + col, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
  }
}

// amax[r, g] = max |w[r, g*gs : (g+1)*gs]|
template <int DT>
__global__ __launch_bounds__(256) void group_amax_kernel(const void* __restrict__ w, int64_t N, int64_t K,
                                                         int64_t G, int gs, int L, float* __restrict__ amax) {
  const int lane = threadIdx.x & 63;
  const int tl = lane & (L - 1), team = lane / L, tpw = 64 / L;
  const int64_t wave_global = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int64_t pair = wave_global * tpw + team;
  const bool active = pair < N * G;
  if (!active) pair = N * G - 1;
  const int64_t n = pair / G, g = pair - n * G;
  const int64_t kbeg = g * gs;
  const int klen = (int)((K - kbeg) < gs ? (K - kbeg) : gs);
  float m = 0.f;
  for (int k = tl; k < klen; k += L) m = fmaxf(m, fabsf(load_as_f32<DT>(w, n * K + kbeg + k)));
  for (int o = L >> 1; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if (active && tl == 0) amax[pair] = m;
}

}  // namespace

extern "C" {

int inc_awq_act_abs_sum(const void* x, int xdtype, int64_t T, int64_t K, float* out,
                        inc_stream_t stream) {
  INC_CHECK_ARG(x && out && T > 0 && K > 0);
  const int vec_ok = (K % 8 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  dim3 grid((unsigned)ceil_div64(K, CW), (unsigned)ceil_div64(T, RSTRIP));
  INC_DISPATCH_DTYPE(xdtype, DT, {
    col_reduce_kernel<DT, 0><<<grid, 256, 0, inc_s(stream)>>>(x, T, K, nullptr, 1, 1, out, vec_ok);
  })
  INC_LAUNCH_RETURN();
}

int64_t inc_awq_weight_scale_workspace_bytes(int64_t N, int64_t K, int group_size) {
  const int64_t gs = (group_size <= 0 || group_size > K) ? K : group_size;
  return N * ceil_div64(K, gs) * 4;
}

int inc_awq_weight_scale(const void* w, int wdtype, int64_t N, int64_t K, int group_size, float* out,
                         void* workspace, int64_t workspace_bytes, inc_stream_t stream) {
  INC_CHECK_ARG(w && out && workspace && N > 0 && K > 0);
  const int gs = (group_size <= 0 || group_size > K) ? (int)K : group_size;
  if (K % gs != 0) return INC_ERR_UNSUPPORTED;  // weight.view(-1, gs) requires it (awq.py:143)
  const int64_t G = K / gs;
  if (workspace_bytes < N * G * 4) return INC_ERR_WORKSPACE;
  float* amax = static_cast<float*>(workspace);
  int L = 1;
  while (L < gs && L < 64) L <<= 1;
  const int64_t waves = ceil_div64(N * G, 64 / L);
  const int vec_ok = (K % 8 == 0) && ((reinterpret_cast<uintptr_t>(w) & 15) == 0);
  dim3 grid((unsigned)ceil_div64(K, CW), (unsigned)ceil_div64(N, RSTRIP));
  hipStream_t s = inc_s(stream);
  INC_DISPATCH_DTYPE(wdtype, DT, {
    group_amax_kernel<DT><<<(unsigned)ceil_div64(waves, 4), 256, 0, s>>>(w, N, K, G, gs, L, amax);
    col_reduce_kernel<DT, 1><<<grid, 256, 0, s>>>(w, N, K, amax, G, gs, out, vec_ok);
  })
  INC_LAUNCH_RETURN();
}

}  // extern "C"
