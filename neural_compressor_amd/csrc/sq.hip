// sq.hip -- K10..K13: SmoothQuant W8A8 calibration and quantisation kernels (HBM-bound streaming / reductions).
//
// Reference (relative to /root/reference/neural_compressor/torch/algorithms/smooth_quant/utility.py):
//   Calibration._save_input_pc_hook :858-883  per-channel running min / max of a Linear's input  -> inc_sq_channel_minmax
//   cal_scale                       :605-626  s = clip(amax_x^a / clip(amax_w, 1e-5)^(1-a), 1e-5), s[amax_x^a == 0] = 1
//                                             -> inc_sq_weight_col_absmax + inc_sq_cal_scale
//   quant_dequant_w_v1 (Linear)     :652-695  per-output-channel int8 of the weight               -> inc_sq_quant_weight
//   SQLinearWrapper.forward         :2591-2605 X * input_scale, then quant_dequant_x_v1 :726-755   -> inc_sq_quant_act
// The INT8 GEMM that consumes these lives in gemm_i8.hip.  (The reference hands W8A8 execution to
// intel_extension_for_pytorch, which is not in /root/reference: the in-tree fake-quant functions above are the spec.)
#include <float.h>
#include <math.h>

#include "common.hpp"

namespace {

__device__ __forceinline__ void atomic_min_f32(float* addr, float v) {
  // ordered-int trick: for non-negative floats the int order equals the float order, for negative ones the uint order is
  // reversed (calibration data has no NaN: a NaN would simply never win)
  if (v >= 0.f) atomicMin(reinterpret_cast<int*>(addr), __float_as_int(v));
  else atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}
__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
  if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

// ---- K10: per-channel min / max over tokens -------------------------------------------------------------------------
// x [T, K] (row stride ld) -> mn[k] = min(mn[k], min_t x[t,k]), mx[k] likewise.  Grid (K/256, T/512): a workgroup owns 256
// columns (a lane: 4 consecutive columns -> 8- or 16-byte loads) and a strip of 512 tokens split over its 4 waves; waves
// fold through LDS, then one float atomic per column per workgroup.
template <int DT>
__global__ __launch_bounds__(256) void channel_minmax_kernel(const void* __restrict__ x, int64_t T, int64_t K, int64_t ld,
                                                             float* __restrict__ mn, float* __restrict__ mx) {
  __shared__ float smn[4][256], smx[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t c0 = (int64_t)blockIdx.x * 256 + lane * 4;
  const int64_t t0 = (int64_t)blockIdx.y * 512;
  const int64_t t1 = t0 + 512 < T ? t0 + 512 : T;
  float lo[4] = {FLT_MAX, FLT_MAX, FLT_MAX, FLT_MAX}, hi[4] = {-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
  const bool vec = c0 + 4 <= K && (ld % 4) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  for (int64_t t = t0 + wave; t < t1; t += 4) {
    float v[4];
    if (vec) {
      if constexpr (DT == INC_F32) {
        const float4 a = *reinterpret_cast<const float4*>(static_cast<const float*>(x) + t * ld + c0);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
      } else {
        const uint2 u = *reinterpret_cast<const uint2*>(static_cast<const uint16_t*>(x) + t * ld + c0);
        if constexpr (DT == INC_BF16) {
          v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
          v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
        } else {
          v[0] = f16_bits_to_f32((uint16_t)(u.x & 0xffffu)); v[1] = f16_bits_to_f32((uint16_t)(u.x >> 16));
          v[2] = f16_bits_to_f32((uint16_t)(u.y & 0xffffu)); v[3] = f16_bits_to_f32((uint16_t)(u.y >> 16));
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = (c0 + i < K) ? load_as_f32<DT>(x, t * ld + c0 + i) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      lo[i] = fminf(lo[i], v[i]);
      hi[i] = fmaxf(hi[i], v[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    smn[wave][lane * 4 + i] = lo[i];
    smx[wave][lane * 4 + i] = hi[i];
  }
  __syncthreads();
  const int c = threadIdx.x;
  const int64_t col = (int64_t)blockIdx.x * 256 + c;
  if (col < K && t0 < T) {
    const float a = fminf(fminf(smn[0][c], smn[1][c]), fminf(smn[2][c], smn[3][c]));
    const float b = fmaxf(fmaxf(smx[0][c], smx[1][c]), fmaxf(smx[2][c], smx[3][c]));
    atomic_min_f32(mn + col, a);
    atomic_max_f32(mx + col, b);
  }
}

// ---- K11a: column-wise abs-max of a weight [N, K] (the `torch.max(torch.abs(cat(weights)), dim=0)` of cal_scale) ---------
template <int DT>
__global__ __launch_bounds__(256) void col_absmax_kernel(const void* __restrict__ w, int64_t N, int64_t K,
                                                         float* __restrict__ out) {
  __shared__ float red[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t c0 = (int64_t)blockIdx.x * 256 + lane * 4;
  const int64_t r0 = (int64_t)blockIdx.y * 256;
  const int64_t r1 = r0 + 256 < N ? r0 + 256 : N;
  float m[4] = {0.f, 0.f, 0.f, 0.f};
  for (int64_t r = r0 + wave; r < r1; r += 4)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (c0 + i < K) m[i] = fmaxf(m[i], fabsf(load_as_f32<DT>(w, r * K + c0 + i)));
#pragma unroll
  for (int i = 0; i < 4; ++i) red[wave][lane * 4 + i] = m[i];
  __syncthreads();
  const int c = threadIdx.x;
  const int64_t col = (int64_t)blockIdx.x * 256 + c;
  if (col < K) {
    const float a = fmaxf(fmaxf(red[0][c], red[1][c]), fmaxf(red[2][c], red[3][c]));
    atomicMax(reinterpret_cast<int*>(out + col), __float_as_int(a));  // a >= 0
  }
}

// ---- K11b: the smoothing scale ----------------------------------------------------------------------------------------
__global__ void cal_scale_kernel(const float* __restrict__ amax_x, const float* __restrict__ amax_w, int64_t K, float alpha,
                                 float weight_max_lb, float* __restrict__ scale) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  const float wm = fmaxf(amax_w[k], weight_max_lb);       // torch.clip(weight_max, weight_max_lb)   :619
  const float ip = powf(amax_x[k], alpha);                // torch.pow(input_max_abs, alpha)         :620
  const float wp = powf(wm, 1.f - alpha);                 // torch.pow(weight_max, 1 - alpha)        :622
  float s = fmaxf(ip / wp, 1e-5f);                        // torch.clip(input_power / weight_power, min=1e-5)
  if (ip == 0.f) s = 1.f;                                 // weight_scale[input_power == 0] = 1.0    :624
  scale[k] = s;
}

// ---- K12: per-output-channel int8 of (W * smooth) -------------------------------------------------------------------------
// One workgroup per row: pass 1 abs-max (sym) or min / max (asym is not used by the W8A8 module and not built), pass 2
// quantise.  scale = clip(absmax / 127.5, eps); q = clamp(rint(w / scale), -128, 127)   (quant_dequant_w_v1 :669-690).
// Also emits rowsum[n] = sum_k q[n,k] (int32), which the GEMM epilogue needs for the activation zero point.
template <int DT>
__global__ __launch_bounds__(256) void quant_weight_kernel(const void* __restrict__ w, int64_t N, int64_t K, int64_t Kp,
                                                           const float* __restrict__ smooth, int8_t* __restrict__ qw,
                                                           float* __restrict__ w_scale, int32_t* __restrict__ rowsum) {
  __shared__ float redf[4];
  __shared__ int redi[4];
  const int64_t n = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float amax = 0.f;
  for (int64_t k = tid; k < K; k += 256) {
    float v = load_as_f32<DT>(w, n * K + k);
    if (smooth) v = v * smooth[k];
    amax = fmaxf(amax, fabsf(v));
  }
  amax = wave_max(amax);
  if (lane == 0) redf[wave] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
  float scale = amax / 127.5f;  // x_max / (float(q_max - q_min) / 2)
  scale = fmaxf(scale, FLT_EPSILON);
  int sum = 0;
  for (int64_t k = tid; k < Kp; k += 256) {
    int q = 0;
    if (k < K) {
      float v = load_as_f32<DT>(w, n * K + k);
      if (smooth) v = v * smooth[k];
      float t = rintf(v / scale);
      t = fminf(fmaxf(t, -128.f), 127.f);
      q = (int)t;
    }
    qw[n * Kp + k] = (int8_t)q;
    sum += q;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  if (lane == 0) redi[wave] = sum;
  __syncthreads();
  if (tid == 0) {
    w_scale[n] = scale;
    rowsum[n] = redi[0] + redi[1] + redi[2] + redi[3];
  }
}

// ---- K13: activation quantisation: q = clamp(rint(x * in_scale / sx + zp), 0, 255) - 128  (int8 for the signed MFMA) -------
// 16 values per lane: two 16-byte loads of a 16-bit input (or four of fp32), one 16-byte store.
template <int DT>
__global__ __launch_bounds__(256) void quant_act_kernel(const void* __restrict__ x, int64_t M, int64_t K, int64_t Kp,
                                                        const float* __restrict__ in_scale, float sx, float zp,
                                                        int8_t* __restrict__ out) {
  const int64_t chunks = Kp / 16;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * chunks) return;
  const int64_t m = i / chunks, k0 = (i - m * chunks) * 16;
  uint32_t packed[4] = {0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};  // q = 0 - 128 for the K padding
  const bool vec = (K % 8) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int64_t kb = k0 + 8 * h;
    if (kb >= K) continue;
    float v[8];
    if (vec) {
      if constexpr (DT == INC_F32) {
        const float4 a = *reinterpret_cast<const float4*>(static_cast<const float*>(x) + m * K + kb);
        const float4 b = *reinterpret_cast<const float4*>(static_cast<const float*>(x) + m * K + kb + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
      } else {
        const uint4 u = *reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(x) + m * K + kb);
        const uint32_t r[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (DT == INC_BF16) {
            v[2 * j] = __uint_as_float(r[j] << 16);
            v[2 * j + 1] = __uint_as_float(r[j] & 0xffff0000u);
          } else {
            v[2 * j] = f16_bits_to_f32((uint16_t)(r[j] & 0xffffu));
            v[2 * j + 1] = f16_bits_to_f32((uint16_t)(r[j] >> 16));
          }
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (kb + j < K) ? load_as_f32<DT>(x, m * K + kb + j) : 0.f;
    }
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint32_t b = 0x80u;
      if (kb + j < K) {
        float t = v[j];
        if (in_scale) t = t * in_scale[kb + j];   // torch.mul(X, self.input_scale)                 :2602
        t = rintf(t / sx + zp);                   // torch.round(x / scale + bias)                  :752
        t = fminf(fmaxf(t, 0.f), 255.f);          // q_x.clamp_(q_min, q_max)
        b = ((uint32_t)(int)t - 128u) & 0xffu;    // uint8 code -> signed operand of v_mfma_i32_*_i8
      }
      if (j < 4) lo |= b << (8 * j);
      else hi |= b << (8 * (j - 4));
    }
    packed[2 * h] = lo;
    packed[2 * h + 1] = hi;
  }
  *reinterpret_cast<uint4*>(out + m * Kp + k0) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
}

}  // namespace

extern "C" {

int inc_sq_channel_minmax(const void* x, int xdtype, int64_t T, int64_t K, int64_t ld, float* mn, float* mx,
                          inc_stream_t stream) {
  INC_CHECK_ARG(x && mn && mx && T > 0 && K > 0 && ld >= K);
  const dim3 grid((unsigned)ceil_div64(K, 256), (unsigned)ceil_div64(T, 512));
  INC_DISPATCH_DTYPE(xdtype, DT, { channel_minmax_kernel<DT><<<grid, 256, 0, inc_s(stream)>>>(x, T, K, ld, mn, mx); })
  INC_LAUNCH_RETURN();
}

int inc_sq_weight_col_absmax(const void* w, int wdtype, int64_t N, int64_t K, float* out, inc_stream_t stream) {
  INC_CHECK_ARG(w && out && N > 0 && K > 0);
  const dim3 grid((unsigned)ceil_div64(K, 256), (unsigned)ceil_div64(N, 256));
  INC_DISPATCH_DTYPE(wdtype, DT, { col_absmax_kernel<DT><<<grid, 256, 0, inc_s(stream)>>>(w, N, K, out); })
  INC_LAUNCH_RETURN();
}

int inc_sq_cal_scale(const float* amax_x, const float* amax_w, int64_t K, float alpha, float weight_max_lb, float* scale,
                     inc_stream_t stream) {
  INC_CHECK_ARG(amax_x && amax_w && scale && K > 0);
  cal_scale_kernel<<<(unsigned)ceil_div64(K, 256), 256, 0, inc_s(stream)>>>(amax_x, amax_w, K, alpha, weight_max_lb, scale);
  INC_LAUNCH_RETURN();
}

int inc_sq_quant_weight(const void* w, int wdtype, int64_t N, int64_t K, int64_t Kp, const float* smooth, int8_t* qw,
                        float* w_scale, int32_t* rowsum, inc_stream_t stream) {
  INC_CHECK_ARG(w && qw && w_scale && rowsum && N > 0 && K > 0 && Kp >= K);
  INC_DISPATCH_DTYPE(wdtype, DT, {
    quant_weight_kernel<DT><<<(unsigned)N, 256, 0, inc_s(stream)>>>(w, N, K, Kp, smooth, qw, w_scale, rowsum);
  })
  INC_LAUNCH_RETURN();
}

int inc_sq_quant_act(const void* x, int xdtype, int64_t M, int64_t K, int64_t Kp, const float* in_scale, float sx, float zp,
                     int8_t* out, inc_stream_t stream) {
  INC_CHECK_ARG(x && out && M > 0 && K > 0 && Kp >= K && (Kp % 16) == 0 && sx > 0.f);
  INC_CHECK_ARG((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  const int64_t total = M * (Kp / 16);
  INC_DISPATCH_DTYPE(xdtype, DT, {
    quant_act_kernel<DT><<<(unsigned)ceil_div64(total, 256), 256, 0, inc_s(stream)>>>(x, M, K, Kp, in_scale, sx, zp, out);
  })
  INC_LAUNCH_RETURN();
}

}  // extern "C"
