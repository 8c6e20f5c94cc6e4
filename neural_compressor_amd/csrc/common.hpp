// common.hpp -- shared device helpers for libinc_mi355x.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/inc_mi355x.h"

#define INC_CHECK_ARG(cond) \
  do {                      \
    if (!(cond)) return INC_ERR_BAD_ARG; \
  } while (0)

#define INC_LAUNCH_RETURN()                          \
  do {                                               \
    hipError_t e__ = hipGetLastError();              \
    return e__ == hipSuccess ? INC_OK : INC_ERR_LAUNCH; \
  } while (0)

static inline hipStream_t inc_s(inc_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Timing / A-B switch of tools/kbench: routes the GEMM / Hessian entry points to other generations or to timing-only
// ablations of a kernel.  It exists ONLY in the harness build of the library (tools/Makefile compiles these sources with
// -DINC_KBENCH into tools/libinc_mi355x_kbench.so); in libinc_mi355x.so the flag is the constant 0, the ablation
// instantiations are not compiled and there is no process-global state.
#ifdef INC_KBENCH
int inc_small_tiles_flag(int set_to);  // defined in pack.hip; set_to < 0 -> query only
#else
static inline constexpr int inc_small_tiles_flag(int) { return 0; }
#endif
static inline bool inc_force_small_tiles() { return inc_small_tiles_flag(-1) == 1; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device setting: every launcher keeps a static bitmask of
// the devices it has configured its kernels on.  Thread-safe -- a racing second thread repeats an idempotent call.
#include <atomic>
static inline uint64_t inc_device_bit() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return 1ull << (dev & 63);
}
static inline bool inc_attr_needed(const std::atomic<uint64_t>& mask) { return !(mask.load(std::memory_order_acquire) & inc_device_bit()); }
static inline void inc_attr_done(std::atomic<uint64_t>& mask) { mask.fetch_or(inc_device_bit(), std::memory_order_release); }

// ---- 16-bit float <-> fp32 (bit-exact, round-to-nearest-even) -------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) {
  return __uint_as_float(static_cast<uint32_t>(b) << 16);
}
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return static_cast<uint16_t>((u >> 16) | 0x0040u);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return static_cast<uint16_t>(u >> 16);
}
__device__ __forceinline__ float f16_bits_to_f32(uint16_t b) {
  _Float16 h;
  __builtin_memcpy(&h, &b, 2);
  return static_cast<float>(h);
}
__device__ __forceinline__ uint16_t f32_to_f16_bits(float f) {
  _Float16 h = static_cast<_Float16>(f);  // v_cvt_f16_f32: RNE
  uint16_t b;
  __builtin_memcpy(&b, &h, 2);
  return b;
}

// load/store one element of a dtype-coded tensor as fp32
template <int DT>
__device__ __forceinline__ float load_as_f32(const void* p, int64_t i) {
  if constexpr (DT == INC_F32) return static_cast<const float*>(p)[i];
  else if constexpr (DT == INC_F16) return f16_bits_to_f32(static_cast<const uint16_t*>(p)[i]);
  else return bf16_bits_to_f32(static_cast<const uint16_t*>(p)[i]);
}
template <int DT>
__device__ __forceinline__ void store_from_f32(void* p, int64_t i, float v) {
  if constexpr (DT == INC_F32) static_cast<float*>(p)[i] = v;
  else if constexpr (DT == INC_F16) static_cast<uint16_t*>(p)[i] = f32_to_f16_bits(v);
  else static_cast<uint16_t*>(p)[i] = f32_to_bf16_bits(v);
}
// round an fp32 value to the precision of dtype DT (what a torch op on a DT tensor returns)
template <int DT>
__device__ __forceinline__ float round_to(float v) {
  if constexpr (DT == INC_F32) return v;
  else if constexpr (DT == INC_F16) return f16_bits_to_f32(f32_to_f16_bits(v));
  else return bf16_bits_to_f32(f32_to_bf16_bits(v));
}

// ---- wave64 reductions --------------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Four LDS-DMA instructions (1 KiB each) of one wave: LDS[lds_dst + i*1024 + lane*16] <- base[voff_i].
// Issued from inline asm on purpose: hipcc orders every later ds_read behind a DMA it can see with
// s_waitcnt vmcnt(0) (it cannot prove the two LDS stages disjoint), which would expose the whole HBM
// latency in every K-step.  The DMA has no VGPR destination, so hiding it is register-safe; completion
// is the explicit `s_waitcnt vmcnt(0)` + barrier at the end of the K-step (cdna_hip_programming.md 5.7).
// M0 (LDS base of the DMA) is saved/restored; s_nop 4 covers a freshly written SGPR base, s_nop 0 the
// M0 write -> LDS-DMA hazard.
__device__ __forceinline__ void lds_dma_4x1k(const void* base, uint32_t lds_dst, uint32_t v0, uint32_t v1,
                                             uint32_t v2, uint32_t v3) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_nop 4\n\t"
      "s_mov_b32 m0, %6\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %5\n\t"
      "s_add_u32 m0, %6, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %2, %5\n\t"
      "s_add_u32 m0, %6, 0x800\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %3, %5\n\t"
      "s_add_u32 m0, %6, 0xc00\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %4, %5\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(base), "s"(lds_dst)
      : "memory", "scc");
}

// Eight LDS-DMA instructions of one wave from ONE base: LDS[lds_dst + i*PITCH + lane*16] <- base[v_i], i = 0..7 (the caller folds the
// row stride into the per-lane offsets once per tile).  Same hazards as above; 4 + 3 per piece instructions instead of a 64-bit
// scalar address computation, an M0 save / restore and six wait states per piece.
template <int PITCH>
__device__ __forceinline__ void lds_dma_8x1k(const void* base, uint32_t lds_dst, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3,
                                             uint32_t v4, uint32_t v5, uint32_t v6, uint32_t v7) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_nop 4\n\t"
      "s_mov_b32 m0, %10\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %9\n\t"
      "s_add_u32 m0, %10, %11\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %2, %9\n\t"
      "s_add_u32 m0, %10, %12\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %3, %9\n\t"
      "s_add_u32 m0, %10, %13\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %4, %9\n\t"
      "s_add_u32 m0, %10, %14\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %5, %9\n\t"
      "s_add_u32 m0, %10, %15\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %6, %9\n\t"
      "s_add_u32 m0, %10, %16\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %7, %9\n\t"
      "s_add_u32 m0, %10, %17\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %8, %9\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7), "s"(base), "s"(lds_dst), "n"(PITCH), "n"(2 * PITCH),
        "n"(3 * PITCH), "n"(4 * PITCH), "n"(5 * PITCH), "n"(6 * PITCH), "n"(7 * PITCH)
      : "memory", "scc");
}

// one 1 KiB LDS-DMA of this wave: LDS[lds_dst + lane*16] <- base[voff] (per-lane byte offset); M0 saved / restored
__device__ __forceinline__ void lds_dma_1k(const void* base, uint32_t lds_dst, uint32_t voff) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_nop 4\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(base), "s"(lds_dst)
      : "memory");
}

// logical tile index -> (tm, tn), walking the tile grid in bands of 8 tile-rows, column by column inside a band.  The W8A8
// GEMM first gives every XCD a contiguous range of logical indices (block b runs on XCD b % 8); with this order such a
// range is an 8 x c patch instead of a 2 x 4c strip, so the tiles an XCD runs concurrently share more operand panels in
// its L2 when both operand tiles have the same size (the 4-bit GEMM, whose x tile is 4x its W tile, keeps row-major).
__device__ __forceinline__ void banded_tile_decode(int idx, int tiles_m, int tiles_n, int& tm, int& tn) {
  constexpr int BAND = 8;
  const int per_band = BAND * tiles_n;
  const int band = idx / per_band;
  const int first = band * BAND;
  const int rows = (tiles_m - first) < BAND ? (tiles_m - first) : BAND;
  const int r = idx - band * per_band;
  tm = first + r % rows;
  tn = r / rows;
}

// dispatch a runtime dtype code to a template parameter
#define INC_DISPATCH_DTYPE(code, NAME, ...)                  \
  switch (code) {                                            \
    case INC_F32: {                                          \
      constexpr int NAME = INC_F32;                          \
      __VA_ARGS__;                                           \
    } break;                                                 \
    case INC_F16: {                                          \
      constexpr int NAME = INC_F16;                          \
      __VA_ARGS__;                                           \
    } break;                                                 \
    case INC_BF16: {                                         \
      constexpr int NAME = INC_BF16;                         \
      __VA_ARGS__;                                           \
    } break;                                                 \
    default:                                                 \
      return INC_ERR_UNSUPPORTED;                            \
  }
