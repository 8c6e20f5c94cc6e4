// pack.hip -- K1/K2/K3: bit packing, unpacking and group-wise dequantisation (recover) for every width 1..8 the reference's
// configs tune (torch/quantization/config.py:211; n_pack = compress_bits // bits, modules.py:231 -- 3 / 5 / 6 / 7 bits leave the
// word's high bits unused).
//
// Replaces the Python loops of INCWeightOnlyLinear.{pack,unpack,recover,pack_tensor,unpack_tensor}
// (reference neural_compressor/torch/algorithms/weight_only/modules.py:321-592) and the numba
// packers (neural_compressor/torch/utils/bit_packer.py:35-278).  Integer work: bit-exact.
//
// All kernels are HBM-bound byte shuffles.  The optimum-format kernels transpose through LDS so that
// both the [N,K] side (int32/int16/fp16 rows, contiguous in K) and the [K/n_pack,N] side (packed
// words, contiguous in N) are accessed with full-cache-line wave transactions.
#include "common.hpp"

namespace {

constexpr int TILE = 64;           // 64 rows (n) x 64 packed words (kw) per workgroup
constexpr int TILE_LD = TILE + 1;  // +1 dword pad: conflict-free column reads

// ------------------------------------------------------------------------------------------
// generic row packer / unpacker (any bits 1..8 x container in {8,16,32,64}; n_pack = cbits / bits fields per word)
// ------------------------------------------------------------------------------------------
template <typename UT>
__global__ void pack_rows_kernel(const int32_t* __restrict__ raw, UT* __restrict__ packed,
                                 int64_t rows, int64_t cols, int64_t pcols, int bits, int n_pack) {
  const int64_t total = rows * pcols;
  const uint32_t mask = (1u << bits) - 1u;
  for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < total;
       w += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = w / pcols, j = w - r * pcols;
    const int32_t* src = raw + r * cols + j * n_pack;
    const int64_t left = cols - j * n_pack;
    UT word = 0;
    for (int e = 0; e < n_pack; ++e) {
      if (e < left) word |= static_cast<UT>(static_cast<UT>(static_cast<uint32_t>(src[e]) & mask) << (bits * e));
    }
    packed[w] = word;
  }
}

template <typename ST, typename UT>
__global__ void unpack_rows_kernel(const UT* __restrict__ packed, int16_t* __restrict__ out,
                                   int64_t rows, int64_t pcols, int bits, int n_pack, int cbits,
                                   int mask_sign) {
  const int64_t total = rows * pcols;
  const int64_t ocols = pcols * n_pack;
  const ST mask = static_cast<ST>((1u << bits) - 1u);
  for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < total;
       w += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = w / pcols, j = w - r * pcols;
    const UT word = packed[w];
    int16_t* dst = out + r * ocols + j * n_pack;
    for (int e = 0; e < n_pack; ++e) {
      ST t = static_cast<ST>(static_cast<UT>(word << (cbits - bits * (e + 1))));
      t = static_cast<ST>(t >> (cbits - bits));  // arithmetic shift, like numpy on a signed dtype
      if (mask_sign) t = static_cast<ST>(t & mask);
      dst[e] = static_cast<int16_t>(t);
    }
  }
}

// ------------------------------------------------------------------------------------------
// optimum-format pack: int_weight [N,K] -> qweight [KW,N]
// ------------------------------------------------------------------------------------------
// ONE launch packs the whole module (round 4; three launches before): every workgroup transposes its 64 x 64 tile of packed words and
// then converts its share of the group parameters -- the 64 output columns of its tile x the groups g = blockIdx.x, blockIdx.x +
// gridDim.x, ... : scales [N,G] fp32 -> [G,N] fp16 and zp [N,G] (or the symmetric constant) -> qzeros [G,NW] holding zp - 1.
template <int BITS, typename IN_T, bool VEC>
__global__ __launch_bounds__(256) void woq_pack_qweight_kernel(const IN_T* __restrict__ iw,
                                                               uint32_t* __restrict__ qweight,
                                                               int64_t N, int64_t K, int64_t KW,
                                                               int shift, const float* __restrict__ scales, uint16_t* __restrict__ scales_out,
                                                               const int32_t* __restrict__ zp, uint32_t* __restrict__ qzeros, int64_t G, int64_t NW,
                                                               int zconst) {
  constexpr int NP = 32 / BITS;
  constexpr uint32_t MASK = (1u << BITS) - 1u;
  __shared__ uint32_t tile[TILE * TILE_LD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t kw0 = (int64_t)blockIdx.x * TILE, n0 = (int64_t)blockIdx.y * TILE;

  const int64_t kw = kw0 + lane;
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) {
    const int row = wave * 16 + rr;
    const int64_t n = n0 + row;
    uint32_t word = 0;
    if (n < N && kw < KW) {
      const IN_T* src = iw + n * K + kw * NP;
      if constexpr (VEC) {  // whole word in range, 16-byte aligned int32 rows
        static_assert(sizeof(IN_T) == 4, "vector path is int32 only");
        const int4* v = reinterpret_cast<const int4*>(src);
#pragma unroll
        for (int q = 0; q < NP / 4; ++q) {
          const int4 x = v[q];
          word |= (static_cast<uint32_t>(x.x + shift) & MASK) << (BITS * (4 * q + 0));
          word |= (static_cast<uint32_t>(x.y + shift) & MASK) << (BITS * (4 * q + 1));
          word |= (static_cast<uint32_t>(x.z + shift) & MASK) << (BITS * (4 * q + 2));
          word |= (static_cast<uint32_t>(x.w + shift) & MASK) << (BITS * (4 * q + 3));
        }
      } else {
        const int64_t left = K - kw * NP;
#pragma unroll
        for (int e = 0; e < NP; ++e)
          if (e < left)
            word |= (static_cast<uint32_t>(static_cast<int32_t>(src[e]) + shift) & MASK) << (BITS * e);
      }
    }
    tile[row * TILE_LD + lane] = word;
  }
  __syncthreads();
  const int64_t n = n0 + lane;
#pragma unroll 4
  for (int rr = 0; rr < 16; ++rr) {
    const int kwl = wave * 16 + rr;
    const int64_t kwo = kw0 + kwl;
    if (kwo < KW && n < N) qweight[kwo * N + n] = tile[lane * TILE_LD + kwl];
  }
  // group parameters of this tile's columns
  const int ng = (int)((G - blockIdx.x + gridDim.x - 1) / gridDim.x);  // groups blockIdx.x, + gridDim.x, ... < G  (0 when blockIdx.x >= G)
  if (scales_out) {
    for (int i = threadIdx.x; i < ng * TILE; i += 256) {
      const int64_t g = blockIdx.x + (int64_t)(i / TILE) * gridDim.x, nn = n0 + (i % TILE);
      if (nn < N) scales_out[g * N + nn] = f32_to_f16_bits(scales[nn * G + g]);
    }
  }
  if (qzeros) {
    // words of qzeros per group that START inside this tile's 64 columns (NP need not divide 64: 3 / 5 / 6 / 7 bits)
    const int64_t j0 = (n0 + NP - 1) / NP, j1 = (n0 + TILE + NP - 1) / NP;
    const int WPT = (int)(j1 - j0);
    for (int i = threadIdx.x; i < ng * WPT; i += 256) {
      const int64_t g = blockIdx.x + (int64_t)(i / WPT) * gridDim.x, j = j0 + (i % WPT);
      if (j < NW) {
        uint32_t word = 0;
#pragma unroll
        for (int e = 0; e < NP; ++e) {
          const int64_t nn = j * NP + e;
          if (nn < N) {
            const int32_t z = (zp ? zp[nn * G + g] : zconst) - 1;
            word |= (static_cast<uint32_t>(z) & MASK) << (BITS * e);
          }
        }
        qzeros[g * NW + j] = word;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// optimum-format unpack: qweight [KW,N] -> int_weight [N,K] int16
// ------------------------------------------------------------------------------------------
// (like the pack kernel, ONE launch: after its tile of weights a workgroup unpacks its share of the zero points -- the 64 rows
// of its tile x the groups g = blockIdx.x, blockIdx.x + gridDim.x, ...: stored + 1, values above maxq wrap to 0, modules.py:407-410)
template <int BITS, bool VEC>
__global__ __launch_bounds__(256) void woq_unpack_qweight_kernel(const uint32_t* __restrict__ qweight,
                                                                 int16_t* __restrict__ out, int64_t N,
                                                                 int64_t K, int64_t KW, const uint32_t* __restrict__ qzeros,
                                                                 int16_t* __restrict__ zp, int64_t G, int64_t NW,
                                                                 const uint16_t* __restrict__ scales_gn, uint16_t* __restrict__ scales_ng) {
  constexpr int NP = 32 / BITS;
  constexpr uint32_t MASK = (1u << BITS) - 1u;
  __shared__ uint32_t tile[TILE * TILE_LD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t kw0 = (int64_t)blockIdx.x * TILE, n0 = (int64_t)blockIdx.y * TILE;
  {
    const int64_t n = n0 + lane;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int kwl = wave * 16 + rr;
      const int64_t kw = kw0 + kwl;
      tile[kwl * TILE_LD + lane] = (kw < KW && n < N) ? qweight[kw * N + n] : 0u;
    }
  }
  __syncthreads();
  const int64_t kw = kw0 + lane;
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) {
    const int row = wave * 16 + rr;
    const int64_t n = n0 + row;
    if (n >= N || kw >= KW) continue;
    const uint32_t word = tile[lane * TILE_LD + row];
    int16_t* dst = out + n * K + kw * NP;
    if constexpr (VEC) {
      static_assert(NP % 8 == 0 || NP == 4, "");
      if constexpr (NP == 4) {
        uint2 v;
        v.x = (word & MASK) | (((word >> BITS) & MASK) << 16);
        v.y = ((word >> (2 * BITS)) & MASK) | (((word >> (3 * BITS)) & MASK) << 16);
        *reinterpret_cast<uint2*>(dst) = v;
      } else {
#pragma unroll
        for (int q = 0; q < NP / 8; ++q) {
          uint32_t p[4];
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            const int e = 8 * q + 2 * h;
            p[h] = ((word >> (BITS * e)) & MASK) | (((word >> (BITS * (e + 1))) & MASK) << 16);
          }
          reinterpret_cast<uint4*>(dst)[q] = make_uint4(p[0], p[1], p[2], p[3]);
        }
      }
    } else {
      const int64_t left = K - kw * NP;
#pragma unroll
      for (int e = 0; e < NP; ++e)
        if (e < left) dst[e] = static_cast<int16_t>((word >> (BITS * e)) & MASK);
    }
  }
  if (zp) {
    const int ng = (int)((G - blockIdx.x + gridDim.x - 1) / gridDim.x);
    for (int i = threadIdx.x; i < ng * TILE; i += 256) {
      const int64_t g = blockIdx.x + (int64_t)(i / TILE) * gridDim.x, nn = n0 + (i % TILE);
      if (nn < N) {
        const uint32_t word = qzeros[g * NW + nn / NP];
        uint32_t z = ((word >> (BITS * (uint32_t)(nn % NP))) & MASK) + 1u;
        if (z > MASK) z = 0;
        zp[nn * G + g] = static_cast<int16_t>(z);
      }
    }
  }
  if (scales_ng) {  // scales [G,N] -> [N,G] (modules.py:382: unpack hands the scales back row-major per output channel), same share
    const int ng = (int)((G - blockIdx.x + gridDim.x - 1) / gridDim.x);
    for (int i = threadIdx.x; i < ng * TILE; i += 256) {
      const int64_t g = blockIdx.x + (int64_t)(i / TILE) * gridDim.x, nn = n0 + (i % TILE);
      if (nn < N) scales_ng[nn * G + g] = scales_gn[g * N + nn];
    }
  }
}

// qzeros [G,NW] -> zp [N,G] int16: stored+1, wrap values above maxq to 0 (modules.py:407-410)
__global__ void woq_unpack_qzeros_kernel(const uint32_t* __restrict__ qzeros, int16_t* __restrict__ zp,
                                         int64_t N, int64_t G, int64_t NW, int bits) {
  const int n_pack = 32 / bits;
  const uint32_t mask = (1u << bits) - 1u;
  const int64_t total = N * G;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / G, g = i - n * G;
    const uint32_t word = qzeros[g * NW + n / n_pack];
    uint32_t z = ((word >> (bits * (n % n_pack))) & mask) + 1u;
    if (z > mask) z = 0;
    zp[i] = static_cast<int16_t>(z);
  }
}

// ------------------------------------------------------------------------------------------
// 4-bit words straight from registers (round 6): unpack AND recover without LDS, every access a full 128-byte line
// ------------------------------------------------------------------------------------------
// A packed word holds 8 consecutive k of ONE output row n, and the words of a packed row kw are contiguous along n.  A wave takes a
// patch of 32 n x 8 packed rows (64 k): lane (quad = lane & 7, r = lane >> 3) loads the uint4 of rows n = n0 + 4 quad .. + 3 at packed
// row kw0 + r -- per packed row the eight quads read 128 contiguous bytes -- and writes, for each of its four rows, the word's 8 outputs
// as ONE 16-byte store at out[n][8 kw]: the eight r-lanes of a (quad, row) write 128 contiguous bytes.  No transpose through LDS, no
// barrier, 16 bytes per lane in both directions; the patches of a wave are independent, so all of its loads are issued before the first
// store.  The LDS-transposing kernels below (any width, g_idx, ragged shapes) measured 0.24-0.42 of the HBM peak on cold tensors; they
// stay the general path.  MODE 0: int16 codes (unpack, modules.py:377-411; zero points like woq_unpack_qweight_kernel); MODE 1 / 2:
// fp16 / bf16 weights (recover, modules.py:413-443: int8(q - z) * s exactly in fp32, rounded once -- the same arithmetic as
// woq_dequant_kernel, bit for bit).
typedef uint32_t w4d_u32x4 __attribute__((ext_vector_type(4)));
constexpr int W4D_PATCHES = 2;  // patches (64 k each) per wave: with group_size 128 a wave owns one group of its 32 rows
template <int MODE>
__global__ __launch_bounds__(256) void woq_w4_direct_kernel(const uint32_t* __restrict__ qweight, const uint16_t* __restrict__ scales,
                                                            const uint32_t* __restrict__ qzeros, uint16_t* __restrict__ out, int64_t N, int64_t K,
                                                            int64_t KW, int64_t NW, int group_size, int16_t* __restrict__ zp, int64_t G,
                                                            const uint16_t* __restrict__ scales_gn, uint16_t* __restrict__ scales_ng, int side_blocks) {
  if constexpr (MODE == 0) {
    // the zero points and the [G,N] -> [N,G] scales of a 32-row strip are the work of ONE extra workgroup of that strip (the last index
    // of grid.x when side_blocks = 1): behind a weight patch they would add a dependent load -> store chain to every workgroup's life
    if (side_blocks && blockIdx.x == gridDim.x - 1) {
      const int64_t n0 = (int64_t)blockIdx.y * 32;
      for (int64_t i = threadIdx.x; i < G * 32; i += 256) {
        const int64_t g = i / 32, nn = n0 + (i % 32);
        if (nn >= N) continue;
        if (zp) {
          const uint32_t word = qzeros[g * NW + nn / 8];
          uint32_t z = ((word >> (4 * (uint32_t)(nn % 8))) & 15u) + 1u;
          if (z > 15u) z = 0;
          zp[nn * G + g] = static_cast<int16_t>(z);
        }
        if (scales_ng) scales_ng[nn * G + g] = scales_gn[g * N + nn];
      }
      return;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int quad = lane & 7, r = lane >> 3;
  const int64_t n = (int64_t)blockIdx.y * 32 + 4 * quad;
  const int64_t kwb = ((int64_t)blockIdx.x * 4 + wave) * (8 * W4D_PATCHES);
  uint4 w[W4D_PATCHES];
  uint2 sraw[W4D_PATCHES];
  uint32_t zraw[W4D_PATCHES];
  bool live[W4D_PATCHES];
#pragma unroll
  for (int p = 0; p < W4D_PATCHES; ++p) {
    const int64_t kw = kwb + 8 * p + r;
    live[p] = n < N && kw < KW;
    const int64_t kwc = live[p] ? kw : 0, nc = live[p] ? n : 0;
    const w4d_u32x4 wv = __builtin_nontemporal_load(reinterpret_cast<const w4d_u32x4*>(qweight + kwc * N + nc));  // every word is read once
    w[p] = make_uint4(wv.x, wv.y, wv.z, wv.w);
    if constexpr (MODE != 0) {
      const int64_t g = (kwc * 8) / group_size;
      sraw[p] = *reinterpret_cast<const uint2*>(scales + g * N + nc);
      zraw[p] = qzeros[g * NW + (nc >> 3)];
    }
  }
#pragma unroll
  for (int p = 0; p < W4D_PATCHES; ++p) {
    if (!live[p]) continue;
    const int64_t kw = kwb + 8 * p + r;
    const uint32_t ww[4] = {w[p].x, w[p].y, w[p].z, w[p].w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      uint32_t o[4];
      if constexpr (MODE == 0) {
#pragma unroll
        for (int h = 0; h < 4; ++h) o[h] = ((ww[c] >> (8 * h)) & 15u) | (((ww[c] >> (8 * h + 4)) & 15u) << 16);
      } else {
        const uint32_t sw = c < 2 ? sraw[p].x : sraw[p].y;
        const float sc = f16_bits_to_f32((uint16_t)(sw >> (16 * (c & 1))));
        uint32_t zz = ((zraw[p] >> (4 * (uint32_t)((n + c) & 7))) & 15u) + 1u;
        const int32_t z = zz > 15u ? 0 : (int32_t)zz;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const float v0 = (float)(int8_t)((int32_t)((ww[c] >> (8 * h)) & 15u) - z) * sc;
          const float v1 = (float)(int8_t)((int32_t)((ww[c] >> (8 * h + 4)) & 15u) - z) * sc;
          o[h] = MODE == 1 ? ((uint32_t)f32_to_f16_bits(v0) | ((uint32_t)f32_to_f16_bits(v1) << 16))
                           : ((uint32_t)f32_to_bf16_bits(v0) | ((uint32_t)f32_to_bf16_bits(v1) << 16));
        }
      }
      const w4d_u32x4 ov = {o[0], o[1], o[2], o[3]};
      __builtin_nontemporal_store(ov, reinterpret_cast<w4d_u32x4*>(out + (n + c) * K + kw * 8));  // written once, read by a later kernel
    }
  }
}

// the direct form needs whole words inside one group, 16-byte rows on both sides and four output rows per lane
static bool w4_direct_ok(const void* qweight, const void* out, int64_t N, int64_t K, int group_size, int bits, const void* g_idx) {
  return bits == 4 && !g_idx && (N % 4) == 0 && (K % 8) == 0 && (group_size % 8) == 0 && ((reinterpret_cast<uintptr_t>(qweight) & 15) == 0) &&
         ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
}

// ------------------------------------------------------------------------------------------
// recover: optimum layout -> dense [N,K]
// ------------------------------------------------------------------------------------------
template <int DT>
struct out_elem { using type = uint16_t; };
template <>
struct out_elem<INC_F32> { using type = float; };

template <int DT>
__device__ __forceinline__ typename out_elem<DT>::type enc(float v) {
  if constexpr (DT == INC_F32) return v;
  else if constexpr (DT == INC_F16) return f32_to_f16_bits(v);
  else return f32_to_bf16_bits(v);
}

constexpr int DQ_KT = 128;  // k-extent of one dequant tile

template <int BITS, int DT>
__global__ __launch_bounds__(256) void woq_dequant_kernel(
    const uint32_t* __restrict__ qweight, const uint16_t* __restrict__ scales,
    const uint32_t* __restrict__ qzeros, const int32_t* __restrict__ g_idx,
    typename out_elem<DT>::type* __restrict__ out, int64_t N, int64_t K, int64_t KW, int64_t NW,
    int group_size) {
  using OT = typename out_elem<DT>::type;
  constexpr int NP = 32 / BITS;
  constexpr uint32_t MASK = (1u << BITS) - 1u;
  constexpr int KWT = DQ_KT / NP;                      // packed rows per tile
  constexpr int KT = KWT * NP;                         // k-extent of the tile: 128, or 120 / 126 / 125 / 124 for 3 / 5 / 6 / 7 bits
  constexpr int LD = DQ_KT + (DT == INC_F32 ? 4 : 8);  // padded row pitch (elements): 16-byte aligned rows, 4-bank skew per row
  __shared__ __attribute__((aligned(16))) OT tile[TILE * LD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t k0 = (int64_t)blockIdx.x * KT, n0 = (int64_t)blockIdx.y * TILE;
  const int64_t kw0 = k0 / NP;
  const int64_t n = n0 + lane;
  // phase 1: lane = n (coalesced packed reads), waves stride over the packed rows of the tile
  for (int kwl = wave; kwl < KWT; kwl += 4) {
    const int64_t kw = kw0 + kwl;
    if (n < N && kw < KW) {
      const uint32_t word = qweight[kw * N + n];
      const uint32_t zsh = BITS * (uint32_t)(n % NP);
      int64_t gprev = -1;
      float s = 0.f;
      int32_t z = 0;
      // without g_idx a packed word lies inside ONE group whenever group_size % NP == 0: one division per word instead of a 64-bit
      // division per element (eight per word: they were most of this kernel's time); other group sizes keep the per-element form
      const bool word_in_group = (group_size % NP) == 0;
      const int64_t gword = (kw * NP) / group_size;
      // The word's NP elements are decoded CH at a time (CH = 8 for 1 / 2 / 4 / 8-bit words, the whole word for the odd widths) and leave
      // as 16-byte LDS writes (2-byte writes made this phase LDS-instruction-bound); rows are 16-byte aligned: LD * sizeof(OT) is a
      // multiple of 16 and kwl * NP * sizeof(OT) too.  (A 1-bit word decoded as ONE unrolled run of 32 elements with fp32 output needed
      // 500 registers and 1.8 KiB of scratch: the chunks of a 32-element word are a loop, not an unroll.)
      constexpr int CH = (NP > 8 && NP % 8 == 0) ? 8 : NP;
      constexpr int PER16 = 16 / (int)sizeof(OT);
      auto chunk = [&](int e0) {
        OT vals[CH];
#pragma unroll
        for (int ee = 0; ee < CH; ++ee) {
          const int e = e0 + ee;
          const int64_t k = kw * NP + e;
          vals[ee] = enc<DT>(0.f);
          if (k < K) {
            const int64_t g = g_idx ? (int64_t)g_idx[k] : (word_in_group ? gword : k / group_size);
            if (g != gprev) {
              s = f16_bits_to_f32(scales[g * N + n]);
              uint32_t zz = ((qzeros[g * NW + n / NP] >> zsh) & MASK) + 1u;
              z = (zz > MASK) ? 0 : (int32_t)zz;
              gprev = g;
            }
            const int32_t q = (int32_t)((word >> (BITS * e)) & MASK);
            const float v = (float)(int8_t)(q - z) * s;  // exact in fp32, one rounding below
            vals[ee] = enc<DT>(v);
          }
        }
        if constexpr (CH % PER16 == 0) {
#pragma unroll
          for (int q4 = 0; q4 < CH / PER16; ++q4) {
            uint4 pk;
            __builtin_memcpy(&pk, &vals[q4 * PER16], 16);
            *reinterpret_cast<uint4*>(&tile[lane * LD + kwl * NP + e0 + q4 * PER16]) = pk;
          }
        } else {
#pragma unroll
          for (int ee = 0; ee < CH; ++ee) tile[lane * LD + kwl * NP + e0 + ee] = vals[ee];
        }
      };
      if constexpr (NP / CH > 2) {
#pragma unroll 1
        for (int e0 = 0; e0 < NP; e0 += CH) chunk(e0);
      } else {
#pragma unroll
        for (int e0 = 0; e0 < NP; e0 += CH) chunk(e0);
      }
    }
  }
  __syncthreads();
  // phase 2: rows of 128 contiguous elements, 16 lanes x 8 elements per row
  const int c8 = (threadIdx.x & 15) * 8;
  for (int row = threadIdx.x >> 4; row < TILE; row += 16) {
    const int64_t nn = n0 + row;
    if (nn >= N) continue;
    const int64_t k = k0 + c8;
    OT* dst = out + nn * K + k;
    const OT* src = &tile[row * LD + c8];
    bool done = c8 >= KT;  // (a tile of an odd width is narrower than 128)
    if (done) continue;
    const bool whole = c8 + 8 <= KT && k + 8 <= K && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0);
    if constexpr (DT != INC_F32) {
      if (whole) {
        *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);  // (16-byte aligned: c8 * 2 bytes, LD * 2 = 272)
        done = true;
      }
    } else {
      if (whole) {
        reinterpret_cast<uint4*>(dst)[0] = reinterpret_cast<const uint4*>(src)[0];
        reinterpret_cast<uint4*>(dst)[1] = reinterpret_cast<const uint4*>(src)[1];
        done = true;
      }
    }
    if (!done) {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (c8 + e < KT && k + e < K) dst[e] = src[e];
    }
  }
}
template <int SDT, int DT>
__global__ void dequant_ints_kernel(const int16_t* __restrict__ iw, const void* __restrict__ scales,
                                    const int16_t* __restrict__ zp, const int32_t* __restrict__ g_idx,
                                    typename out_elem<DT>::type* __restrict__ out, int64_t N, int64_t K,
                                    int64_t G, int group_size) {
  const int64_t total = N * K;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / K, k = i - n * K;
    const int64_t g = g_idx ? (int64_t)g_idx[k] : k / group_size;
    const float s = load_as_f32<SDT>(scales, n * G + g);
    const int32_t q = iw[i];
    float v;
    if (zp) v = (float)(int8_t)(q - (int32_t)zp[n * G + g]) * s;
    else v = (float)q * s;
    // torch computes int*scale in the scale dtype: round once to it, then to the output dtype
    out[i] = enc<DT>(round_to<SDT>(v));
  }
}

inline int grid_1d(int64_t total, int block = 256) {
  int64_t g = ceil_div64(total, block);
  const int64_t cap = 256 * 8 * 4;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

#ifdef INC_KBENCH  // harness build only (tools/Makefile): the A/B switch of tools/kbench
int inc_small_tiles_flag(int set_to) {
  static int v = 0;
  if (set_to >= 0) v = set_to;
  return v;
}
extern "C" void inc_debug_set_small_tiles(int on) { (void)inc_small_tiles_flag(on < 0 ? 0 : on); }
#endif

extern "C" {

// 2: + find_params_mse, awq_repack, sq_*, w8a8_* (SmoothQuant), chol_diag_block
// 3: inc_mse_accumulate sums in fp64 in a fixed order (workspace argument); inc_debug_set_small_tiles left the library
// 4: inc_gptq_quantize_layer (the column loop as one call)
// 8: + inc_codebook_quant_with_scale (quantize_4bit with the caller's scale)
int inc_abi_version(void) { return 9; }
const char* inc_target_arch(void) { return "gfx950"; }
const char* inc_error_string(int code) {
  switch (code) {
    case INC_OK: return "ok";
    case INC_ERR_BAD_ARG: return "bad argument (null pointer / non-positive size / inconsistent shape)";
    case INC_ERR_UNSUPPORTED: return "unsupported configuration";
    case INC_ERR_LAUNCH: return "HIP kernel launch failed";
    case INC_ERR_WORKSPACE: return "workspace too small";
    default: return "unknown error";
  }
}

int inc_pack_rows(const int32_t* raw, void* packed, int64_t rows, int64_t cols, int bits, int cbits,
                  inc_stream_t stream) {
  INC_CHECK_ARG(raw && packed && rows > 0 && cols > 0);
  if (bits < 1 || bits > 8) return INC_ERR_UNSUPPORTED;
  if (!(cbits == 8 || cbits == 16 || cbits == 32 || cbits == 64) || cbits < bits) return INC_ERR_UNSUPPORTED;
  const int n_pack = cbits / bits;
  const int64_t pcols = ceil_div64(cols, n_pack);
  const int grid = grid_1d(rows * pcols);
  hipStream_t s = inc_s(stream);
  switch (cbits) {
    case 8: pack_rows_kernel<uint8_t><<<grid, 256, 0, s>>>(raw, (uint8_t*)packed, rows, cols, pcols, bits, n_pack); break;
    case 16: pack_rows_kernel<uint16_t><<<grid, 256, 0, s>>>(raw, (uint16_t*)packed, rows, cols, pcols, bits, n_pack); break;
    case 32: pack_rows_kernel<uint32_t><<<grid, 256, 0, s>>>(raw, (uint32_t*)packed, rows, cols, pcols, bits, n_pack); break;
    default: pack_rows_kernel<uint64_t><<<grid, 256, 0, s>>>(raw, (uint64_t*)packed, rows, cols, pcols, bits, n_pack); break;
  }
  INC_LAUNCH_RETURN();
}

int inc_unpack_rows(const void* packed, int16_t* out, int64_t rows, int64_t packed_cols, int bits,
                    int cbits, int mask_sign, inc_stream_t stream) {
  INC_CHECK_ARG(packed && out && rows > 0 && packed_cols > 0);
  if (bits < 1 || bits > 8) return INC_ERR_UNSUPPORTED;
  if (!(cbits == 8 || cbits == 16 || cbits == 32 || cbits == 64) || cbits < bits) return INC_ERR_UNSUPPORTED;
  const int n_pack = cbits / bits;
  const int grid = grid_1d(rows * packed_cols);
  hipStream_t s = inc_s(stream);
  switch (cbits) {
    case 8: unpack_rows_kernel<int8_t, uint8_t><<<grid, 256, 0, s>>>((const uint8_t*)packed, out, rows, packed_cols, bits, n_pack, cbits, mask_sign); break;
    case 16: unpack_rows_kernel<int16_t, uint16_t><<<grid, 256, 0, s>>>((const uint16_t*)packed, out, rows, packed_cols, bits, n_pack, cbits, mask_sign); break;
    case 32: unpack_rows_kernel<int32_t, uint32_t><<<grid, 256, 0, s>>>((const uint32_t*)packed, out, rows, packed_cols, bits, n_pack, cbits, mask_sign); break;
    default: unpack_rows_kernel<int64_t, uint64_t><<<grid, 256, 0, s>>>((const uint64_t*)packed, out, rows, packed_cols, bits, n_pack, cbits, mask_sign); break;
  }
  INC_LAUNCH_RETURN();
}

int inc_woq_pack(const void* int_weight, int in_bytes, const float* scales, const int32_t* zp,
                 int32_t* qweight, int32_t* qzeros, uint16_t* scales_out, int64_t N, int64_t K,
                 int64_t G, int bits, int shift, inc_stream_t stream) {
  INC_CHECK_ARG(int_weight && qweight && N > 0 && K > 0 && G > 0);
  INC_CHECK_ARG(in_bytes == 4 || in_bytes == 1);
  if (bits < 1 || bits > 8) return INC_ERR_UNSUPPORTED;
  hipStream_t s = inc_s(stream);
  const int n_pack = 32 / bits;
  const int64_t KW = ceil_div64(K, n_pack), NW = ceil_div64(N, n_pack);
  dim3 grid((unsigned)ceil_div64(KW, TILE), (unsigned)ceil_div64(N, TILE));
  const bool vec = in_bytes == 4 && (n_pack % 4 == 0) && (K % n_pack == 0) && (K % 4 == 0) &&
                   ((reinterpret_cast<uintptr_t>(int_weight) & 15) == 0);
  uint32_t* qw = reinterpret_cast<uint32_t*>(qweight);
  const float* sc_in = (scales && scales_out) ? scales : nullptr;
  uint16_t* sc_out = sc_in ? scales_out : nullptr;
  uint32_t* qz = reinterpret_cast<uint32_t*>(qzeros);
  const int zconst = 1 << (bits - 1);
#define INC_PACK_ARGS qw, N, K, KW, shift, sc_in, sc_out, zp, qz, G, NW, zconst
#define INC_PACK_LAUNCH(B)                                                                              \
  if (in_bytes == 4) {                                                                                  \
    if (vec) woq_pack_qweight_kernel<B, int32_t, true><<<grid, 256, 0, s>>>((const int32_t*)int_weight, INC_PACK_ARGS); \
    else woq_pack_qweight_kernel<B, int32_t, false><<<grid, 256, 0, s>>>((const int32_t*)int_weight, INC_PACK_ARGS);    \
  } else {                                                                                              \
    woq_pack_qweight_kernel<B, int8_t, false><<<grid, 256, 0, s>>>((const int8_t*)int_weight, INC_PACK_ARGS);           \
  }
  // (widths whose n_pack is not a multiple of 4 have no 16-byte path: `vec` is false for them)
#define INC_PACK_LAUNCH_ODD(B)                                                                          \
  if (in_bytes == 4) woq_pack_qweight_kernel<B, int32_t, false><<<grid, 256, 0, s>>>((const int32_t*)int_weight, INC_PACK_ARGS); \
  else woq_pack_qweight_kernel<B, int8_t, false><<<grid, 256, 0, s>>>((const int8_t*)int_weight, INC_PACK_ARGS);
  switch (bits) {
    case 4: INC_PACK_LAUNCH(4) break;
    case 8: INC_PACK_LAUNCH(8) break;
    case 2: INC_PACK_LAUNCH(2) break;
    case 1: INC_PACK_LAUNCH(1) break;
    case 3: INC_PACK_LAUNCH_ODD(3) break;
    case 5: INC_PACK_LAUNCH_ODD(5) break;
    case 6: INC_PACK_LAUNCH_ODD(6) break;
    default: INC_PACK_LAUNCH(7) break;
  }
#undef INC_PACK_LAUNCH_ODD
#undef INC_PACK_LAUNCH
#undef INC_PACK_ARGS
  INC_LAUNCH_RETURN();
}

int inc_woq_unpack(const int32_t* qweight, const int32_t* qzeros, int16_t* int_weight, int16_t* zp,
                   int64_t N, int64_t K, int64_t G, int bits, const uint16_t* scales_gn, uint16_t* scales_ng, inc_stream_t stream) {
  INC_CHECK_ARG(N > 0 && K > 0 && G > 0 && (!scales_ng || (scales_gn && int_weight)));
  if (bits < 1 || bits > 8) return INC_ERR_UNSUPPORTED;
  hipStream_t s = inc_s(stream);
  const int n_pack = 32 / bits;
  const int64_t KW = ceil_div64(K, n_pack), NW = ceil_div64(N, n_pack);
  const uint32_t* qzp = reinterpret_cast<const uint32_t*>(qzeros);
  if (zp) INC_CHECK_ARG(qzeros);
  if (int_weight && w4_direct_ok(qweight, int_weight, N, K, 8, bits, nullptr) && ceil_div64(N, 32) <= 65535) {
    INC_CHECK_ARG(qweight);
    const int side = (zp || scales_ng) ? 1 : 0;
    dim3 grid((unsigned)ceil_div64(KW, 4 * 8 * W4D_PATCHES) + side, (unsigned)ceil_div64(N, 32));
    woq_w4_direct_kernel<0><<<grid, 256, 0, s>>>(reinterpret_cast<const uint32_t*>(qweight), nullptr, qzp, reinterpret_cast<uint16_t*>(int_weight), N, K, KW, NW,
                                                8, zp, G, scales_gn, scales_ng, side);
  } else if (int_weight) {
    INC_CHECK_ARG(qweight);
    dim3 grid((unsigned)ceil_div64(KW, TILE), (unsigned)ceil_div64(N, TILE));
    const bool vec = (n_pack % 8 == 0 || n_pack == 4) && (K % n_pack == 0) && (K % 8 == 0) && ((reinterpret_cast<uintptr_t>(int_weight) & 15) == 0);
    const uint32_t* qw = reinterpret_cast<const uint32_t*>(qweight);
#define INC_UNPACK(B, V) woq_unpack_qweight_kernel<B, V><<<grid, 256, 0, s>>>(qw, int_weight, N, K, KW, qzp, zp, G, NW, scales_gn, scales_ng)
    switch (bits) {
      case 4: if (vec) INC_UNPACK(4, true); else INC_UNPACK(4, false); break;
      case 8: if (vec) INC_UNPACK(8, true); else INC_UNPACK(8, false); break;
      case 2: if (vec) INC_UNPACK(2, true); else INC_UNPACK(2, false); break;
      case 1: if (vec) INC_UNPACK(1, true); else INC_UNPACK(1, false); break;
      case 7: if (vec) INC_UNPACK(7, true); else INC_UNPACK(7, false); break;  // (n_pack = 4)
      case 3: INC_UNPACK(3, false); break;
      case 5: INC_UNPACK(5, false); break;
      default: INC_UNPACK(6, false); break;
    }
#undef INC_UNPACK
  } else if (zp) {  // zero points alone
    woq_unpack_qzeros_kernel<<<grid_1d(N * G), 256, 0, s>>>(qzp, zp, N, G, NW, bits);
  }
  INC_LAUNCH_RETURN();
}

int inc_woq_dequant(const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros,
                    const int32_t* g_idx, void* out, int out_dtype, int64_t N, int64_t K, int64_t G,
                    int group_size, int bits, inc_stream_t stream) {
  INC_CHECK_ARG(qweight && scales && qzeros && out && N > 0 && K > 0 && G > 0 && group_size > 0);
  if (bits < 1 || bits > 8) return INC_ERR_UNSUPPORTED;
  hipStream_t s = inc_s(stream);
  const int n_pack = 32 / bits;
  const int64_t KW = ceil_div64(K, n_pack), NW = ceil_div64(N, n_pack);
  dim3 grid((unsigned)ceil_div64(K, (DQ_KT / n_pack) * n_pack), (unsigned)ceil_div64(N, TILE));
  const uint32_t* qw = reinterpret_cast<const uint32_t*>(qweight);
  const uint32_t* qz = reinterpret_cast<const uint32_t*>(qzeros);
  if ((out_dtype == INC_F16 || out_dtype == INC_BF16) && w4_direct_ok(qweight, out, N, K, group_size, bits, g_idx) && (N % 8) == 0 &&
      ((reinterpret_cast<uintptr_t>(scales) & 7) == 0) && ceil_div64(N, 32) <= 65535) {
    dim3 dgrid((unsigned)ceil_div64(KW, 4 * 8 * W4D_PATCHES), (unsigned)ceil_div64(N, 32));
    if (out_dtype == INC_F16) woq_w4_direct_kernel<1><<<dgrid, 256, 0, s>>>(qw, scales, qz, (uint16_t*)out, N, K, KW, NW, group_size, nullptr, G, nullptr, nullptr, 0);
    else woq_w4_direct_kernel<2><<<dgrid, 256, 0, s>>>(qw, scales, qz, (uint16_t*)out, N, K, KW, NW, group_size, nullptr, G, nullptr, nullptr, 0);
    INC_LAUNCH_RETURN();
  }
#define INC_DQ_LAUNCH(B, D) \
  woq_dequant_kernel<B, D><<<grid, 256, 0, s>>>(qw, scales, qz, g_idx, (typename out_elem<D>::type*)out, N, K, KW, NW, group_size)
#define INC_DQ_BITS(D)                                                                                                       \
  switch (bits) {                                                                                                            \
    case 4: INC_DQ_LAUNCH(4, D); break; case 8: INC_DQ_LAUNCH(8, D); break; case 2: INC_DQ_LAUNCH(2, D); break;              \
    case 1: INC_DQ_LAUNCH(1, D); break; case 3: INC_DQ_LAUNCH(3, D); break; case 5: INC_DQ_LAUNCH(5, D); break;              \
    case 6: INC_DQ_LAUNCH(6, D); break; default: INC_DQ_LAUNCH(7, D); break;                                                 \
  }
  if (out_dtype == INC_F32) {
    INC_DQ_BITS(INC_F32)
  } else if (out_dtype == INC_F16) {
    INC_DQ_BITS(INC_F16)
  } else if (out_dtype == INC_BF16) {
    INC_DQ_BITS(INC_BF16)
  } else {
    return INC_ERR_UNSUPPORTED;
  }
#undef INC_DQ_BITS
#undef INC_DQ_LAUNCH
  INC_LAUNCH_RETURN();
}

int inc_dequant_ints(const int16_t* int_weight, const void* scales, int scale_dtype,
                     const int16_t* zp, const int32_t* g_idx, void* out, int out_dtype, int64_t N,
                     int64_t K, int64_t G, int group_size, inc_stream_t stream) {
  INC_CHECK_ARG(int_weight && scales && out && N > 0 && K > 0 && G > 0 && group_size > 0);
  hipStream_t s = inc_s(stream);
  const int grid = grid_1d(N * K);
  INC_DISPATCH_DTYPE(scale_dtype, SDT, {
    INC_DISPATCH_DTYPE(out_dtype, DT, {
      dequant_ints_kernel<SDT, DT><<<grid, 256, 0, s>>>(int_weight, scales, zp, g_idx,
                                                       (typename out_elem<DT>::type*)out, N, K, G, group_size);
    })
  })
  INC_LAUNCH_RETURN();
}

}  // extern "C"


// ---------------------------------------------------------------------------------------------------------------------
// AutoAWQ "GEMM" checkpoint words -> optimum layout (reference weight_only/utility.py:1245-1459:
// unpack_awq + awq_reverse_reorder_int_tensor + pack_from_tensors; caller transformers/quantization/utils.py:655-697).
//   AWQ:     qweight [K, N/8] int32, field i of word (k, c) = code(k, 8c + AWQ_ORDER[i]),  AWQ_ORDER = 0,2,4,6,1,3,5,7
//            qzeros  [G, N/8] int32, same field order, zero points stored as they are
//   optimum: qweight [K/8, N] int32, field r of word (k/8, n) = code(8*(k/8) + r, n)
//            qzeros  [G, N/8] int32, field j of word (g, c) = (zero(g, 8c + j) - 1) & 15
// A pure 4-bit field shuffle, HBM-bound (reads and writes K*N/2 bytes once).  One thread owns an 8(k) x 8(n) tile:
// 8 coalesced word reads (one per k), an 8x8 nibble transpose in registers, one 32-byte store.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t awq_field_of_column(int j) {
  // column 8c + j sits in field AWQ_ORDER^-1[j] = {0,4,1,5,2,6,3,7}[j]
  return (uint32_t)(((j & 1) << 2) | (j >> 1));
}

__global__ __launch_bounds__(256) void awq_repack_weight_kernel(const uint32_t* __restrict__ awq, uint32_t* __restrict__ out,
                                                                int64_t K8, int64_t NW) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // AWQ word column (8 output columns)
  const int64_t kb = blockIdx.y;                                      // block of 8 input rows
  if (c >= NW || kb >= K8) return;
  uint32_t w[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) w[r] = awq[(kb * 8 + r) * NW + c];
  uint32_t o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint32_t sh = 4u * awq_field_of_column(j);
    uint32_t v = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) v |= ((w[r] >> sh) & 15u) << (4 * r);
    o[j] = v;
  }
  uint4* dst = reinterpret_cast<uint4*>(out + kb * NW * 8 + c * 8);
  dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
  dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
}

__global__ __launch_bounds__(256) void awq_repack_zeros_kernel(const uint32_t* __restrict__ awq, uint32_t* __restrict__ out,
                                                               int64_t words) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= words) return;
  const uint32_t w = awq[i];
  uint32_t v = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint32_t z = (w >> (4u * awq_field_of_column(j))) & 15u;
    v |= ((z - 1u) & 15u) << (4 * j);
  }
  out[i] = v;
}

int inc_awq_repack(const int32_t* awq_qweight, const int32_t* awq_qzeros, int64_t K, int64_t N, int64_t G, int bits,
                   int32_t* qweight, int32_t* qzeros, inc_stream_t stream) {
  INC_CHECK_ARG(awq_qweight && awq_qzeros && qweight && qzeros && K > 0 && N > 0 && G > 0);
  if (bits != 4) return INC_ERR_UNSUPPORTED;  // the reference asserts bits == 4 too (utility.py:1252, 1301, 1378)
  INC_CHECK_ARG((K % 8) == 0 && (N % 8) == 0);
  const int64_t NW = N / 8, K8 = K / 8;
  INC_CHECK_ARG(K8 <= 65535);
  hipStream_t s = inc_s(stream);
  awq_repack_weight_kernel<<<dim3((unsigned)ceil_div64(NW, 256), (unsigned)K8), 256, 0, s>>>(
      reinterpret_cast<const uint32_t*>(awq_qweight), reinterpret_cast<uint32_t*>(qweight), K8, NW);
  const int64_t zw = G * NW;
  awq_repack_zeros_kernel<<<(unsigned)ceil_div64(zw, 256), 256, 0, s>>>(
      reinterpret_cast<const uint32_t*>(awq_qzeros), reinterpret_cast<uint32_t*>(qzeros), zw);
  INC_LAUNCH_RETURN();
}
