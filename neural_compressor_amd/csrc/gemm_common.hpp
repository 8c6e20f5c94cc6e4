// gemm_common.hpp -- helpers shared by the dequant-GEMM translation units (gemm.hip, gemm_d2r.hip): MFMA wrappers,
// the int4 -> bf16 / f16 dequantisation arithmetic and the 256 x 256 x 64 tile constants.
#pragma once
#include "common.hpp"

// gemm_d2r.hip: launcher of the direct-to-register dequant-GEMM, called by inc_woq_gemm (gemm.hip)
int inc_launch_woq_gemm_d2r(const uint16_t* x, const uint32_t* qw, const uint16_t* scales, const uint32_t* qz, const uint16_t* bias,
                            uint16_t* y, int64_t M, int64_t N, int64_t K, int64_t NW, int g_shift, int y_vec_ok, float* part, int steps,
                            int splits, bool bf, int ns, int abl, hipStream_t s);
// the same tile with eight waves, two per SIMD (gemm_d2r8.hip)
int inc_launch_woq_gemm_d2r8(const uint16_t* x, const uint32_t* qw, const uint16_t* scales, const uint32_t* qz, const uint16_t* bias,
                             uint16_t* y, int64_t M, int64_t N, int64_t K, int64_t NW, int g_shift, int y_vec_ok, float* part, int steps,
                             int splits, bool bf, hipStream_t s);

// gemm_strip8.hip: the 128 x 128 mid-M kernel (four waves, K split inside the workgroup) and its split-K plan
int inc_woq_gemm_strip8_splitk(int64_t M, int64_t N, int64_t K);
int inc_launch_woq_gemm_strip8(const uint16_t* x, const uint32_t* qw, const uint16_t* scales, const uint32_t* qz, const uint16_t* bias,
                               uint16_t* y, int64_t M, int64_t N, int64_t K, int64_t NW, int g_shift, float* part, unsigned* counters,
                               int splitk, bool bf, hipStream_t s);

// modules that share x, one launch (inc_woq_gemm_multi): the per-module tensors of the batch, passed to the kernels by value
constexpr int GEMV_MAX_BATCH = 8;
struct GemvBatch {
  const uint32_t* qweight[GEMV_MAX_BATCH];
  const uint16_t* scales[GEMV_MAX_BATCH];
  const uint32_t* qzeros[GEMV_MAX_BATCH];
  const uint16_t* bias[GEMV_MAX_BATCH];
  uint16_t* y[GEMV_MAX_BATCH];
  int64_t N[GEMV_MAX_BATCH];
  int64_t part_off[GEMV_MAX_BATCH];  // first float of the module's split-K slabs in the workspace
  int first[GEMV_MAX_BATCH + 1];     // first strip of every module, then the number of strips
  int n;
};

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <bool IS_BF16>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  if constexpr (IS_BF16) {
    return (uint32_t)f32_to_bf16_bits(lo) | ((uint32_t)f32_to_bf16_bits(hi) << 16);
  } else {
    return (uint32_t)f32_to_f16_bits(lo) | ((uint32_t)f32_to_f16_bits(hi) << 16);
  }
}

template <bool IS_BF16>
__device__ __forceinline__ float cvt16(uint16_t b) {
  if constexpr (IS_BF16) return bf16_bits_to_f32(b);
  else return f16_bits_to_f32(b);
}

// group parameters of one (group, column): fp32 scale and integer zero point
struct GroupQ {
  float s;
  int z;
};

template <int BITS>
__device__ __forceinline__ GroupQ load_group(const uint16_t* __restrict__ scales,
                                             const uint32_t* __restrict__ qzeros, int64_t g, int64_t n,
                                             int64_t N, int64_t NW) {
  constexpr int NP = 32 / BITS;
  constexpr uint32_t MASK = (1u << BITS) - 1u;
  GroupQ r;
  r.s = f16_bits_to_f32(scales[g * N + n]);
  uint32_t zz = ((qzeros[g * NW + n / NP] >> (BITS * (uint32_t)(n % NP))) & MASK) + 1u;  // modules.py:407-410
  r.z = zz > MASK ? 0 : (int)zz;
  return r;
}

// dequantise one packed word (NP consecutive k of one column) into NP/2 dwords of 16-bit pairs
template <int BITS, bool IS_BF16>
__device__ __forceinline__ void dequant_word(uint32_t word, const GroupQ& gq, uint32_t (&out)[16 / BITS]) {
  constexpr int NP = 32 / BITS;
  constexpr uint32_t MASK = (1u << BITS) - 1u;
#pragma unroll
  for (int h = 0; h < NP / 2; ++h) {
    const int q0 = (int)((word >> (BITS * (2 * h))) & MASK);
    const int q1 = (int)((word >> (BITS * (2 * h + 1))) & MASK);
    const float v0 = (float)(int8_t)(q0 - gq.z) * gq.s;
    const float v1 = (float)(int8_t)(q1 - gq.z) * gq.s;
    out[h] = pack2<IS_BF16>(v0, v1);
  }
}

// the same with a per-ELEMENT group (g_idx: GPTQ act_order / HF desc_act, modules.py:427-431): column k of the word takes the
// scale / zero point of group g_idx[k] -- one lookup per element, the general (slow) path of the two small-tile kernels
template <int BITS, bool IS_BF16>
__device__ __forceinline__ void dequant_word_gidx(uint32_t word, const uint16_t* __restrict__ scales, const uint32_t* __restrict__ qzeros,
                                                  const int32_t* __restrict__ g_idx, int64_t kk, int64_t K, int64_t n, int64_t N, int64_t NW,
                                                  uint32_t (&out)[16 / BITS]) {
  constexpr int NP = 32 / BITS;
  constexpr uint32_t MASK = (1u << BITS) - 1u;
#pragma unroll
  for (int h = 0; h < NP / 2; ++h) {
    float v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int j = 2 * h + e;
      const int64_t k = kk + j;
      const GroupQ gq = load_group<BITS>(scales, qzeros, k < K ? (int64_t)g_idx[k] : 0, n, N, NW);
      const int q = (int)((word >> (BITS * j)) & MASK);
      v[e] = (float)(int8_t)(q - gq.z) * gq.s;
    }
    out[h] = pack2<IS_BF16>(v[0], v[1]);
  }
}

template <bool IS_BF16>
__device__ __forceinline__ f32x16 mfma32(const uint4& a, const uint4& b, f32x16 c) {
  if constexpr (IS_BF16) {
    bf16x8 fa, fb;
    __builtin_memcpy(&fa, &a, 16);
    __builtin_memcpy(&fb, &b, 16);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, c, 0, 0, 0);
  } else {
    f16x8 fa, fb;
    __builtin_memcpy(&fa, &a, 16);
    __builtin_memcpy(&fb, &b, 16);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, c, 0, 0, 0);
  }
}
template <bool IS_BF16>
__device__ __forceinline__ f32x4 mfma16(const uint4& a, const uint4& b, f32x4 c) {
  if constexpr (IS_BF16) {
    bf16x8 fa, fb;
    __builtin_memcpy(&fa, &a, 16);
    __builtin_memcpy(&fb, &b, 16);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c, 0, 0, 0);
  } else {
    f16x8 fa, fb;
    __builtin_memcpy(&fa, &a, 16);
    __builtin_memcpy(&fb, &b, 16);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c, 0, 0, 0);
  }
}

// 16-byte write-through store / L2-coherent load of split-K partials (the hand-off of MI355X_MICROARCH.md "Valid forms": sc1 stores ->
// vmcnt(0) -> barrier -> one relaxed agent-scope ticket -> sc1 loads by the last arriver); the caller waits for the loads (vmcnt) itself
__device__ __forceinline__ void splitk_store16_sc1(float* p, f32x4 v) {
  // (s_nop: a VMEM store of more than 64 bits reads its data registers AFTER issue -- the next instruction may not overwrite them for two
  // wait states; the compiler's hazard recogniser inserts that for its own stores but cannot see into an asm statement.  Without it this
  // hand-off returned wrong sums a few times per hundred launches.)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
// pv[sl][i] = 16 bytes at bases[sl] + off[i] (byte offsets < 4 GiB), sl, i in 0..3, sc1 (L2-coherent: the partials were written through by
// other XCDs), and the wait for them -- ONE asm statement: a load whose result the compiler sees before an explicit s_waitcnt may be
// copied or spilled by it before the data has landed (MI355X guide, inline-asm hazards)
__device__ __forceinline__ void splitk_load16x16_sc1(f32x4 (&pv)[4][4], const float* b0, const float* b1, const float* b2, const float* b3,
                                                     uint32_t o0, uint32_t o1, uint32_t o2, uint32_t o3) {
  asm volatile(
      "global_load_dwordx4 %0, %16, %20 sc1\n\tglobal_load_dwordx4 %1, %17, %20 sc1\n\tglobal_load_dwordx4 %2, %18, %20 sc1\n\tglobal_load_dwordx4 %3, %19, %20 sc1\n\t"
      "global_load_dwordx4 %4, %16, %21 sc1\n\tglobal_load_dwordx4 %5, %17, %21 sc1\n\tglobal_load_dwordx4 %6, %18, %21 sc1\n\tglobal_load_dwordx4 %7, %19, %21 sc1\n\t"
      "global_load_dwordx4 %8, %16, %22 sc1\n\tglobal_load_dwordx4 %9, %17, %22 sc1\n\tglobal_load_dwordx4 %10, %18, %22 sc1\n\tglobal_load_dwordx4 %11, %19, %22 sc1\n\t"
      "global_load_dwordx4 %12, %16, %23 sc1\n\tglobal_load_dwordx4 %13, %17, %23 sc1\n\tglobal_load_dwordx4 %14, %18, %23 sc1\n\tglobal_load_dwordx4 %15, %19, %23 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(pv[0][0]), "=&v"(pv[0][1]), "=&v"(pv[0][2]), "=&v"(pv[0][3]), "=&v"(pv[1][0]), "=&v"(pv[1][1]), "=&v"(pv[1][2]), "=&v"(pv[1][3]),
        "=&v"(pv[2][0]), "=&v"(pv[2][1]), "=&v"(pv[2][2]), "=&v"(pv[2][3]), "=&v"(pv[3][0]), "=&v"(pv[3][1]), "=&v"(pv[3][2]), "=&v"(pv[3][3])
      : "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(b0), "s"(b1), "s"(b2), "s"(b3)
      : "memory");
}
// four adjacent outputs of one row: + bias, one rounding to the 16-bit type, one 8-byte store (y + n with n % 4 == 0 and N % 4 == 0)
template <bool IS_BF16>
__device__ __forceinline__ void store_out4(uint16_t* dst, float4 v, const uint16_t* bias4) {
  if (bias4) {
    const uint2 b = *reinterpret_cast<const uint2*>(bias4);
    v.x += cvt16<IS_BF16>((uint16_t)(b.x & 0xffffu)); v.y += cvt16<IS_BF16>((uint16_t)(b.x >> 16));
    v.z += cvt16<IS_BF16>((uint16_t)(b.y & 0xffffu)); v.w += cvt16<IS_BF16>((uint16_t)(b.y >> 16));
  }
  uint2 o;
  o.x = pack2<IS_BF16>(v.x, v.y);
  o.y = pack2<IS_BF16>(v.z, v.w);
  *reinterpret_cast<uint2*>(dst) = o;
}

constexpr int TM = 256, TN = 256, TK = 64;
constexpr int T_ASTAGE = TM * TK * 2;  // bytes
constexpr int T_BSTAGE = TN * TK * 2;

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

template <bool IS_BF16>
__device__ __forceinline__ uint32_t cvt_pair(float a, float b) {
  f32x2 f = {a, b};
  uint32_t r;
  if constexpr (IS_BF16) {
    bf16x2 h = __builtin_convertvector(f, bf16x2);  // v_cvt_pk_bf16_f32 (RNE)
    __builtin_memcpy(&r, &h, 4);
  } else {
    f16x2 h = __builtin_convertvector(f, f16x2);
    __builtin_memcpy(&r, &h, 4);
  }
  return r;
}

// 8 nibbles of `w` -> 8 x rn16((q - z) * s), k-ordered, as 4 dwords.
// int -> float goes through the fp8 converter: an e4m3 byte with value q in 0..15 decodes to q * u (u = 2^-9 for the
// OCP format of gfx950: codes 0..7 are subnormals m * 2^-9, codes 8..15 are (1 + m/8) * 2^-6 = (8 + m) * 2^-9), so ONE
// v_cvt_pk_f32_fp8 turns two nibbles (bytes of `w & 0x0F0F0F0F`) into two floats -- 4 instead of 8 conversions per word.
// `s_over_u` = s / u (exact: a power-of-two rescale); fma(q*u, s/u, -z*s) is exact in fp32 (q, z < 32 and s has 11
// significant bits), so the single rounding is the 16-bit conversion: bit-identical to inc_woq_dequant.
__device__ __forceinline__ float fp8_unit_inverse() {
  return 1.0f / __builtin_amdgcn_cvt_pk_f32_fp8(0x00000001, false)[0];  // measured, not assumed: 2^9 on gfx950
}
// one fp32 FMA that the SLP vectoriser cannot fuse into v_pk_fma_f32 (see dequant8's SFMA form)
__device__ __forceinline__ float fma_single(float a, float b, float c) {
  float d;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// FORM 0 (default): eight v_fma_f32; FORM 1: four v_pk_fma_f32 (the first generation; same values -- each half of the packed op
// is the same fused multiply-add); FORM 2: eight single v_cvt_f32_fp8 (byte select) + eight v_fma_f32.
// MI355X_MICROARCH.md: a packed fp32 VALU op next to MFMAs costs ~22 cycles more than the two scalar ops it replaces -- measured
// here: the producer / consumer GEMM gains 6-10 % from FORM 0 (tools/kbench pcfma), outputs bit-identical.
template <bool IS_BF16, int FORM = 0>
__device__ __forceinline__ uint4 dequant8(uint32_t w, float s_over_u, float nzs) {
  const uint32_t lo = w & 0x0F0F0F0Fu, hi = (w >> 4) & 0x0F0F0F0Fu;  // bytes: k0,k2,k4,k6 / k1,k3,k5,k7
  if constexpr (FORM == 0) {
    const f32x2 c01 = __builtin_amdgcn_cvt_pk_f32_fp8((int)lo, false), c23 = __builtin_amdgcn_cvt_pk_f32_fp8((int)lo, true);
    const f32x2 d01 = __builtin_amdgcn_cvt_pk_f32_fp8((int)hi, false), d23 = __builtin_amdgcn_cvt_pk_f32_fp8((int)hi, true);
    uint4 o;
    o.x = cvt_pair<IS_BF16>(fma_single(c01[0], s_over_u, nzs), fma_single(d01[0], s_over_u, nzs));
    o.y = cvt_pair<IS_BF16>(fma_single(c01[1], s_over_u, nzs), fma_single(d01[1], s_over_u, nzs));
    o.z = cvt_pair<IS_BF16>(fma_single(c23[0], s_over_u, nzs), fma_single(d23[0], s_over_u, nzs));
    o.w = cvt_pair<IS_BF16>(fma_single(c23[1], s_over_u, nzs), fma_single(d23[1], s_over_u, nzs));
    return o;
  }
  if constexpr (FORM == 2) {
    float e[4], f[4];
    e[0] = __builtin_amdgcn_cvt_f32_fp8((int)lo, 0); e[1] = __builtin_amdgcn_cvt_f32_fp8((int)lo, 1);
    e[2] = __builtin_amdgcn_cvt_f32_fp8((int)lo, 2); e[3] = __builtin_amdgcn_cvt_f32_fp8((int)lo, 3);
    f[0] = __builtin_amdgcn_cvt_f32_fp8((int)hi, 0); f[1] = __builtin_amdgcn_cvt_f32_fp8((int)hi, 1);
    f[2] = __builtin_amdgcn_cvt_f32_fp8((int)hi, 2); f[3] = __builtin_amdgcn_cvt_f32_fp8((int)hi, 3);
    uint4 o;
    o.x = cvt_pair<IS_BF16>(fma_single(e[0], s_over_u, nzs), fma_single(f[0], s_over_u, nzs));
    o.y = cvt_pair<IS_BF16>(fma_single(e[1], s_over_u, nzs), fma_single(f[1], s_over_u, nzs));
    o.z = cvt_pair<IS_BF16>(fma_single(e[2], s_over_u, nzs), fma_single(f[2], s_over_u, nzs));
    o.w = cvt_pair<IS_BF16>(fma_single(e[3], s_over_u, nzs), fma_single(f[3], s_over_u, nzs));
    return o;
  }
  const f32x2 sv = {s_over_u, s_over_u}, nv = {nzs, nzs};
  const f32x2 e01 = __builtin_elementwise_fma(__builtin_amdgcn_cvt_pk_f32_fp8((int)lo, false), sv, nv);  // k0, k2
  const f32x2 e23 = __builtin_elementwise_fma(__builtin_amdgcn_cvt_pk_f32_fp8((int)lo, true), sv, nv);   // k4, k6
  const f32x2 o01 = __builtin_elementwise_fma(__builtin_amdgcn_cvt_pk_f32_fp8((int)hi, false), sv, nv);  // k1, k3
  const f32x2 o23 = __builtin_elementwise_fma(__builtin_amdgcn_cvt_pk_f32_fp8((int)hi, true), sv, nv);   // k5, k7
  uint4 o;
  o.x = cvt_pair<IS_BF16>(e01[0], o01[0]);
  o.y = cvt_pair<IS_BF16>(e01[1], o01[1]);
  o.z = cvt_pair<IS_BF16>(e23[0], o23[0]);
  o.w = cvt_pair<IS_BF16>(e23[1], o23[1]);
  return o;
}

}  // namespace
