// quant.hip -- K7: group-wise round-to-nearest quantisation (RTN / AWQ inner op), GPTQ find_params,
// and the squared-error reduction used by the clip / scale searches.
//
// Reference arithmetic (file:line relative to /root/reference/neural_compressor/torch/algorithms/
// weight_only/): quant_tensor utility.py:272-436, qdq_weight_sym :199-244, qdq_weight_asym :162-196,
// search_clip loss :468, Quantizer.find_params gptq.py:1501-1624.
//
// HBM-bound streaming kernels.  A "team" of L lanes (L = 2^j <= 64, L*8 >= group length when the group
// fits) owns one (row, group) pair; every lane moves 8 contiguous elements (16 B bf16 / 32 B fp32) per
// step, min/max are reduced with xor-shuffles inside the team, so a wave streams 64/L consecutive
// groups of one row with full-line transactions.
//
// torch semantics that are reproduced on purpose:
//   * every torch op on a bf16/fp16 tensor rounds its result to that dtype (round_to<DT>);
//   * qdq_weight_asym builds its statistics against an fp32 zeros tensor (utility.py:176-178), so the
//     asym scale / zero-point are fp32 even for a bf16 weight, the sym ones are in the weight dtype;
//   * x / scale is a true IEEE division, rounding is half-to-even (rintf).
#include <math.h>

#include "common.hpp"

namespace {

template <int DT>
__device__ __forceinline__ void load8(const void* p, int64_t idx, int nvalid, bool vec, float (&v)[8]) {
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

template <int DT>
__device__ __forceinline__ void store8(void* p, int64_t idx, int nvalid, bool vec, const float (&v)[8]) {
  if (vec && nvalid >= 8) {
    if constexpr (DT == INC_F32) {
      *reinterpret_cast<float4*>(static_cast<float*>(p) + idx) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(static_cast<float*>(p) + idx + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
      uint32_t r[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint16_t lo, hi;
        if constexpr (DT == INC_BF16) { lo = f32_to_bf16_bits(v[2 * i]); hi = f32_to_bf16_bits(v[2 * i + 1]); }
        else { lo = f32_to_f16_bits(v[2 * i]); hi = f32_to_f16_bits(v[2 * i + 1]); }
        r[i] = (uint32_t)lo | ((uint32_t)hi << 16);
      }
      *reinterpret_cast<uint4*>(static_cast<uint16_t*>(p) + idx) = make_uint4(r[0], r[1], r[2], r[3]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < nvalid) store_from_f32<DT>(p, idx + i, v[i]);
  }
}

__device__ __forceinline__ void store8_i32(int32_t* p, int64_t idx, int nvalid, bool vec, const int (&v)[8]) {
  if (vec && nvalid >= 8) {
    *reinterpret_cast<int4*>(p + idx) = make_int4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<int4*>(p + idx + 4) = make_int4(v[4], v[5], v[6], v[7]);
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < nvalid) p[idx + i] = v[i];
  }
}

struct QParams {
  float scale;  // quantisation step (sign carries the full_range flip)
  float zp;     // asym zero point (0 for sym)
  float qmin, qmax;
};

// ---- group-wise RTN -----------------------------------------------------------------------------
template <int DT, bool SYM>
__global__ __launch_bounds__(256) void groupwise_quant_kernel(
    const void* __restrict__ w, void* qdq, int32_t* __restrict__ iout, float* __restrict__ scale_out,
    float* __restrict__ zp_out, int64_t N, int64_t K, int64_t G, int gs, int L, int bits,
    float quantile, int full_range, int vec_ok) {
  const int lane = threadIdx.x & 63;
  const int tl = lane & (L - 1);
  const int team = lane / L;
  const int teams_per_wave = 64 / L;
  const int64_t wave_global = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int64_t pair = wave_global * teams_per_wave + team;
  const bool active = pair < N * G;
  if (!active) pair = N * G - 1;  // keep the lanes alive for the shuffles, mask the stores
  const int64_t n = pair / G, g = pair - n * G;
  const int64_t kbeg = g * gs;
  const int klen = (int)((K - kbeg) < gs ? (K - kbeg) : gs);
  const int64_t base = n * K + kbeg;
  const bool vec = vec_ok != 0;

  float vmax = -INFINITY, vmin = INFINITY;
  float keep[8];
  for (int c = tl; c * 8 < klen; c += L) {
    float v[8];
    const int nv = klen - c * 8;
    load8<DT>(w, base + c * 8, nv, vec, v);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < nv) { vmax = fmaxf(vmax, v[i]); vmin = fminf(vmin, v[i]); }
    if (c == tl) {
#pragma unroll
      for (int i = 0; i < 8; ++i) keep[i] = v[i];
    }
  }
  for (int o = L >> 1; o > 0; o >>= 1) {
    vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
    vmin = fminf(vmin, __shfl_xor(vmin, o, 64));
  }

  QParams q;
  if constexpr (SYM) {
    // utility.py:216-237 -- statistics stay in the weight dtype
    float maxq = (float)((1 << (bits - 1)) - 1), minq = -(float)(1 << (bits - 1));
    if (bits == 1) { maxq = 1.f; minq = 0.f; }
    const bool flip = fabsf(vmax) > fabsf(vmin);
    float wmax = fmaxf(fabsf(vmax), fabsf(vmin));
    wmax = round_to<DT>(wmax * quantile);
    if (wmax == 0.f) wmax = 1.f;
    float scale;
    if (full_range) {
      scale = round_to<DT>(wmax / (-minq));
      if (flip) scale = -scale;
    } else {
      scale = round_to<DT>(wmax / maxq);
    }
    q.scale = scale; q.zp = 0.f; q.qmin = minq; q.qmax = maxq;
  } else {
    // utility.py:174-188 -- fp32 statistics (torch.zeros(...) is fp32 and promotes)
    const float maxq = (float)((1 << bits) - 1);
    float wmin = fminf(vmin, 0.f) * quantile;
    float wmax = fmaxf(vmax, 0.f) * quantile;
    if (wmin == 0.f && wmax == 0.f) { wmin = -1.f; wmax = 1.f; }
    const float scale = (wmax - wmin) / maxq;
    q.scale = scale; q.zp = rintf(-wmin / scale); q.qmin = 0.f; q.qmax = maxq;
  }
  if (active && tl == 0) {
    if (scale_out) scale_out[pair] = q.scale;
    if (!SYM && zp_out) zp_out[pair] = q.zp;
  }

  for (int c = tl; c * 8 < klen; c += L) {
    float v[8];
    const int nv = klen - c * 8;
    if (c == tl) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = keep[i];
    } else {
      load8<DT>(w, base + c * 8, nv, vec, v);
    }
    int iv[8];
    float dq[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float t = round_to<DT>(v[i] / q.scale);
      t = rintf(t);
      if constexpr (!SYM) t = round_to<DT>(t + q.zp);
      t = fminf(fmaxf(t, q.qmin), q.qmax);
      iv[i] = (int)t;
      if constexpr (!SYM) t = round_to<DT>(t - q.zp);
      dq[i] = round_to<DT>(t * q.scale);
    }
    if (active) {
      if (iout) store8_i32(iout, base + c * 8, nv, vec, iv);
      if (qdq) store8<DT>(qdq, base + c * 8, nv, vec, dq);
    }
  }
}

// ---- group-wise code-book quantisation: NF4 / FP4 (quantize_4bit, utility.py:112-149) ---------------------------
// Per (row, group): scale = max|w| * quantile / max(code book)  (every op rounded to the weight dtype, like the torch ops);
// t = w / scale; the chosen entry is the one whose interval (mid[i-1], mid[i]] holds t -- the reference's where() chain, with
// the Python-double midpoints cast to the tensor dtype at the comparison, the entry value cast to it when accumulated into
// q_tensor.  An all-zero group divides 0 / 0: every comparison with the NaN is false, q = 0 (restated, not "fixed").
struct CodeBook {
  float value[16];  // ascending code-book entries
  float mid[15];    // (value[i] + value[i+1]) / 2, computed in double on the host like the reference's Python list
  int code[16];     // integer written for entry i when the caller wants ints (INT_MAPPING)
  int n;
  float vmax;       // max(code book)
};

template <int DT>
__global__ __launch_bounds__(256) void codebook_quant_kernel(
    const void* __restrict__ w, void* qdq, int32_t* __restrict__ iout, float* __restrict__ scale_out, int64_t N, int64_t K,
    int64_t G, int gs, int L, CodeBook cb, float quantile, int vec_ok, const float* __restrict__ scale_in) {
  const int lane = threadIdx.x & 63;
  const int tl = lane & (L - 1);
  const int team = lane / L;
  const int teams_per_wave = 64 / L;
  const int64_t wave_global = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int64_t pair = wave_global * teams_per_wave + team;
  const bool active = pair < N * G;
  if (!active) pair = N * G - 1;
  const int64_t n = pair / G, g = pair - n * G;
  const int64_t kbeg = g * gs;
  const int klen = (int)((K - kbeg) < gs ? (K - kbeg) : gs);
  const int64_t base = n * K + kbeg;
  const bool vec = vec_ok != 0;

  float scale;
  if (scale_in) {
    // the caller's scale (quantize_4bit(..., scale=...), utility.py:127-128): used as it is -- torch divides / multiplies in the
    // promoted type and rounds the result to the tensor's
    scale = scale_in[pair];
  } else {
    float amax = 0.f;
    for (int c = tl; c * 8 < klen; c += L) {
      float v[8];
      const int nv = klen - c * 8;
      load8<DT>(w, base + c * 8, nv, vec, v);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (i < nv) amax = fmaxf(amax, fabsf(v[i]));
    }
    for (int o = L >> 1; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    scale = round_to<DT>(round_to<DT>(amax * quantile) / cb.vmax);
  }
  if (active && tl == 0 && scale_out) scale_out[pair] = scale;

  for (int c = tl; c * 8 < klen; c += L) {
    float v[8];
    const int nv = klen - c * 8;
    load8<DT>(w, base + c * 8, nv, vec, v);
    int iv[8];
    float dq[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float t = round_to<DT>(v[i] / scale);
      int pick = -1;  // NaN (0 / 0) matches no interval
      if (t <= round_to<DT>(cb.mid[0])) pick = 0;
      else if (t > round_to<DT>(cb.mid[cb.n - 2])) pick = cb.n - 1;
      for (int e = 1; e < cb.n - 1; ++e)
        if (round_to<DT>(cb.mid[e - 1]) < t && t <= round_to<DT>(cb.mid[e])) pick = e;
      iv[i] = pick < 0 ? 0 : cb.code[pick];
      dq[i] = round_to<DT>((pick < 0 ? 0.f : round_to<DT>(cb.value[pick])) * scale);
    }
    if (active) {
      if (iout) store8_i32(iout, base + c * 8, nv, vec, iv);
      if (qdq) store8<DT>(qdq, base + c * 8, nv, vec, dq);
    }
  }
}

// ---- GPTQ Quantizer.find_params (weight=True, perchannel, int, no mse) ---------------------------
__global__ __launch_bounds__(256) void gptq_find_params_kernel(
    const float* __restrict__ w, int64_t N, int64_t K, int64_t col0,
    int gs, int ngroups, int L, int bits, int sym, float* __restrict__ scale, float* __restrict__ zero,
    int64_t G, int64_t g0, int mse_steps, int mse_grid, float mse_norm) {
  const int lane = threadIdx.x & 63;
  const int tl = lane & (L - 1);
  const int team = lane / L;
  const int teams_per_wave = 64 / L;
  const int64_t wave_global = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int64_t pair = wave_global * teams_per_wave + team;
  const int64_t total = N * ngroups;
  const bool active = pair < total;
  if (!active) pair = total - 1;
  const int64_t n = pair / ngroups, g = pair - n * ngroups;
  const int64_t kbeg = col0 + g * gs;
  const int klen = (int)((K - kbeg) < gs ? (K - kbeg) : gs);
  float vmax = -INFINITY, vmin = INFINITY;
  for (int k = tl; k < klen; k += L) {
    const float x = w[n * K + kbeg + k];
    vmax = fmaxf(vmax, x);
    vmin = fminf(vmin, x);
  }
  for (int o = L >> 1; o > 0; o >>= 1) {
    vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
    vmin = fminf(vmin, __shfl_xor(vmin, o, 64));
  }
  // gptq.py:1547-1571
  const float maxq = (float)((1 << bits) - 1);
  float xmin = fminf(vmin, 0.f), xmax = fmaxf(vmax, 0.f);
  if (sym) {
    xmax = fmaxf(fabsf(xmin), xmax);
    if (xmin < 0.f) xmin = -xmax;
  }
  if (xmin == 0.f && xmax == 0.f) { xmin = -1.f; xmax = 1.f; }
  float s = (xmax - xmin) / maxq;
  float z = sym ? (maxq + 1.f) * 0.5f : rintf(-xmin / s);
  if (mse_steps > 0) {
    // shrink-grid search (gptq.py:1567-1584): keep the (scale, zero) with the smallest sum |q(x) - x|^norm; strict '<'
    // like the reference, so the first (largest) range wins ties.  Same un-fused fp32 arithmetic as Quantizer.quantize.
    const float z_sym = z;
    float best = INFINITY;
    for (int i = 0; i < mse_steps; ++i) {
      const float p = (float)(1.0 - (double)i / (double)mse_grid);  // Python double, cast to the tensor dtype
      const float xmin1 = p * xmin, xmax1 = p * xmax;
      const float s1 = (xmax1 - xmin1) / maxq;
      const float z1 = sym ? z_sym : rintf(-xmin1 / s1);
      float err = 0.f;
      for (int k = tl; k < klen; k += L) {
        const float x = w[n * K + kbeg + k];
        float t = rintf(x / s1) + z1;
        t = fminf(fmaxf(t, 0.f), maxq);
        const float q = s1 * (t - z1);
        err += powf(fabsf(q - x), mse_norm);
      }
      for (int o = L >> 1; o > 0; o >>= 1) err += __shfl_xor(err, o, 64);
      if (err < best) {
        best = err;
        s = s1;
        z = z1;
      }
    }
  }
  if (active && tl == 0) {
    scale[n * G + g0 + g] = s;
    zero[n * G + g0 + g] = z;
  }
}

// ---- sum((a-b)^2), fp64, fixed summation order -----------------------------------------------------------
// The AWQ grid searches (awq.py:336-344, 450-458) and search_clip (utility.py:468) take an argmin over such sums: the
// reference accumulates them as Python doubles, and an order-dependent fp32 atomic sum would make the chosen grid point
// vary from run to run.  Pass 1: every workgroup writes ONE fp64 partial (its elements visited in a fixed order, wave
// and workgroup reductions are fixed trees); pass 2: one workgroup adds the partials in a fixed tree and accumulates
// into *out.  Same input -> same bits, on any launch.
constexpr int MSE_MAX_BLOCKS = 2048;

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <int DT>
__global__ __launch_bounds__(256) void mse_partial_kernel(const void* __restrict__ a, const void* __restrict__ b, int64_t n,
                                                          double* __restrict__ partial, int vec_ok) {
  __shared__ double part[4];
  double acc = 0.0;
  const int64_t nchunk = (n + 7) / 8;
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < nchunk;
       c += (int64_t)gridDim.x * blockDim.x) {
    float va[8], vb[8];
    const int64_t left = n - c * 8;
    const int nv = left < 8 ? (int)left : 8;
    load8<DT>(a, c * 8, nv, vec_ok != 0, va);
    load8<DT>(b, c * 8, nv, vec_ok != 0, vb);
    float s8 = 0.f;  // 8 squares in fp32 (what `.float().pow(2)` yields per element), then widened
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float d = round_to<DT>(va[i] - vb[i]);  // the subtraction happens in the tensor dtype
      s8 += d * d;
    }
    acc += (double)s8;
  }
  acc = wave_sum_f64(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

__global__ __launch_bounds__(256) void mse_final_kernel(const double* __restrict__ partial, int nblocks, double* __restrict__ out) {
  __shared__ double part[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) acc += partial[i];
  acc = wave_sum_f64(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) *out += (part[0] + part[1]) + (part[2] + part[3]);
}

inline int team_lanes(int gs) {
  int need = (gs + 7) / 8, L = 1;
  while (L < need && L < 64) L <<= 1;
  return L;
}

}  // namespace

extern "C" {

int inc_groupwise_quant(const void* w, int wdtype, void* qdq_out, int32_t* int_out, float* scale_out,
                        float* zp_out, int64_t N, int64_t K, int group_size, int bits, int scheme,
                        float quantile, int full_range, inc_stream_t stream) {
  INC_CHECK_ARG(w && N > 0 && K > 0 && bits >= 1 && bits <= 8);
  int gs = group_size;
  if (gs <= 0 || gs > K) gs = (int)K;  // utility.py:306-307
  const int64_t G = ceil_div64(K, gs);
  const int L = team_lanes(gs);
  const int64_t waves = ceil_div64(N * G, 64 / L);
  const int64_t blocks = ceil_div64(waves, 4);
  const int esz = wdtype == INC_F32 ? 4 : 2;
  auto al = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const int vec_ok = (K % 8 == 0) && (gs % 8 == 0) && al(w) && al(qdq_out) && al(int_out);
  (void)esz;
  hipStream_t s = inc_s(stream);
  INC_DISPATCH_DTYPE(wdtype, DT, {
    if (scheme == INC_SCHEME_SYM)
      groupwise_quant_kernel<DT, true><<<(unsigned)blocks, 256, 0, s>>>(w, qdq_out, int_out, scale_out, zp_out, N, K, G, gs, L, bits, quantile, full_range, vec_ok);
    else
      groupwise_quant_kernel<DT, false><<<(unsigned)blocks, 256, 0, s>>>(w, qdq_out, int_out, scale_out, zp_out, N, K, G, gs, L, bits, quantile, full_range, vec_ok);
  })
  INC_LAUNCH_RETURN();
}

int inc_gptq_find_params(const float* w, int64_t N, int64_t K, int64_t col0, int group_size,
                         int ngroups, int bits, int sym, float* scale, float* zero, int64_t G,
                         int64_t g0, inc_stream_t stream) {
  INC_CHECK_ARG(w && scale && zero && N > 0 && K > 0 && ngroups > 0 && group_size > 0);
  INC_CHECK_ARG(col0 >= 0 && col0 < K && g0 >= 0 && g0 + ngroups <= G && bits >= 1 && bits <= 8);
  int L = 1;
  while (L < group_size && L < 64) L <<= 1;
  const int64_t waves = ceil_div64(N * ngroups, 64 / L);
  gptq_find_params_kernel<<<(unsigned)ceil_div64(waves, 4), 256, 0, inc_s(stream)>>>(
      w, N, K, col0, group_size, ngroups, L, bits, sym, scale, zero, G, g0, 0, 100, 2.4f);
  INC_LAUNCH_RETURN();
}

int inc_gptq_find_params_mse(const float* w, int64_t N, int64_t K, int64_t col0, int group_size,
                             int ngroups, int bits, int sym, int grid, float maxshrink, float norm,
                             float* scale, float* zero, int64_t G, int64_t g0, inc_stream_t stream) {
  INC_CHECK_ARG(w && scale && zero && N > 0 && K > 0 && ngroups > 0 && group_size > 0 && grid > 0);
  INC_CHECK_ARG(col0 >= 0 && col0 < K && g0 >= 0 && g0 + ngroups <= G && bits >= 1 && bits <= 8);
  INC_CHECK_ARG(maxshrink > 0.f && maxshrink <= 1.f && norm > 0.f);
  int L = 1;
  while (L < group_size && L < 64) L <<= 1;
  const int64_t waves = ceil_div64(N * ngroups, 64 / L);
  const int steps = (int)(maxshrink * (float)grid);  // int(self.maxshrink * self.grid)
  gptq_find_params_kernel<<<(unsigned)ceil_div64(waves, 4), 256, 0, inc_s(stream)>>>(
      w, N, K, col0, group_size, ngroups, L, bits, sym, scale, zero, G, g0, steps, grid, norm);
  INC_LAUNCH_RETURN();
}

int inc_codebook_quant(const void* w, int wdtype, void* qdq_out, int32_t* int_out, float* scale_out, int64_t N, int64_t K,
                       int group_size, const float* values, const int32_t* codes, int n_entries, float quantile,
                       inc_stream_t stream) {
  return inc_codebook_quant_with_scale(w, wdtype, qdq_out, int_out, scale_out, N, K, group_size, values, codes, n_entries, quantile,
                                       nullptr, stream);
}

int inc_codebook_quant_with_scale(const void* w, int wdtype, void* qdq_out, int32_t* int_out, float* scale_out, int64_t N, int64_t K,
                                  int group_size, const float* values, const int32_t* codes, int n_entries, float quantile,
                                  const float* scale_in, inc_stream_t stream) {
  INC_CHECK_ARG(w && values && codes && N > 0 && K > 0 && n_entries >= 2 && n_entries <= 16);
  INC_CHECK_ARG(!scale_in || scale_in != scale_out);
  int gs = group_size;
  if (gs <= 0 || gs > K) gs = (int)K;
  const int64_t G = ceil_div64(K, gs);
  CodeBook cb;
  cb.n = n_entries;
  cb.vmax = values[0];
  for (int i = 0; i < 16; ++i) {
    cb.value[i] = i < n_entries ? values[i] : 0.f;
    cb.code[i] = i < n_entries ? codes[i] : 0;
    if (i < n_entries && values[i] > cb.vmax) cb.vmax = values[i];
  }
  for (int i = 0; i < 15; ++i) cb.mid[i] = i + 1 < n_entries ? (float)(((double)values[i] + (double)values[i + 1]) / 2.0) : 0.f;
  const int L = team_lanes(gs);
  const int64_t waves = ceil_div64(N * G, 64 / L);
  const int elt = wdtype == INC_F32 ? 4 : 2;
  const int vec_ok = (K % 8 == 0) && (gs % 8 == 0) && ((reinterpret_cast<uintptr_t>(w) & 15) == 0) &&
                     (!qdq_out || (reinterpret_cast<uintptr_t>(qdq_out) & 15) == 0) && (!int_out || (reinterpret_cast<uintptr_t>(int_out) & 15) == 0) && elt > 0;
  INC_DISPATCH_DTYPE(wdtype, DT, {
    codebook_quant_kernel<DT><<<(unsigned)ceil_div64(waves, 4), 256, 0, inc_s(stream)>>>(w, qdq_out, int_out, scale_out, N, K, G, gs, L, cb,
                                                                                          quantile, vec_ok, scale_in);
  })
  INC_LAUNCH_RETURN();
}

int64_t inc_mse_accumulate_workspace_bytes(void) { return (int64_t)MSE_MAX_BLOCKS * 8; }

int inc_mse_accumulate(const void* a, const void* b, int dtype, int64_t n, double* out, void* workspace,
                       inc_stream_t stream) {
  INC_CHECK_ARG(a && b && out && workspace && n > 0);
  const int vec_ok = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
  int64_t blocks = ceil_div64(ceil_div64(n, 8), 256);
  if (blocks > MSE_MAX_BLOCKS) blocks = MSE_MAX_BLOCKS;
  double* partial = (double*)workspace;
  INC_DISPATCH_DTYPE(dtype, DT, {
    mse_partial_kernel<DT><<<(unsigned)blocks, 256, 0, inc_s(stream)>>>(a, b, n, partial, vec_ok);
  })
  mse_final_kernel<<<1, 256, 0, inc_s(stream)>>>(partial, (int)blocks, out);
  INC_LAUNCH_RETURN();
}

}  // extern "C"
