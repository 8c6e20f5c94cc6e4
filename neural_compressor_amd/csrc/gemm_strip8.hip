// gemm_strip8.hip -- K4c', the mid-M fused INT4 -> bf16 / fp16 dequant-GEMM (128 < M <= 1024, few 256 x 256 tiles), round 6.
//
// The strip kernel of round 2 (gemm.hip, woq_gemm_w4_strip_kernel) gives a workgroup 64 rows x 128 columns: every packed weight is
// dequantised once per 64 rows, and a K-step's dequantisation (8 words x ~19 VALU instructions) is as long as its 32 MFMAs -- PMC:
// VALU 27 %, MFMA 28 % of the wave time, the rest waits (profiles/r2_pmc).  This kernel doubles the rows per dequantised weight
// WITHOUT parking a 256-row tile that such an M would leave half empty:
//   * a workgroup owns 128 rows x 128 columns of y and a K-slice; its FOUR waves (one per SIMD) take a quarter of the slice's K-steps
//     each and hold the whole 128 x 128 fp32 tile: 8 x 8 accumulator fragments = 256 registers per lane;
//   * per K-step (32 k) a wave receives eight x fragments (128 rows) and two packed-weight fragments (128 columns) by ten LDS-DMA
//     requests into its own ring slot (lane-linear image = fragment layout, no VGPR staging), three steps ahead, picks them up with
//     ten conflict-free ds_read_b128, dequantises the 8 words in registers (fp8-decoder trick, bit-identical to inc_woq_dequant) and
//     issues 64 v_mfma_f32_16x16x32: 1024 matrix cycles against ~600 VALU cycles per step -- the wave's own MFMAs cover its VALU work;
//   * the four accumulator tiles meet in LDS (four passes of 64 x 64), split-K slices (<= 4, only to fill the chip) hand over through
//     write-through partials and one relaxed ticket, summed by the last arriver in slice order: deterministic.
// Reference semantics: INCWeightOnlyLinear.forward (modules.py:594-610) = F.linear(x, recover()).
#include "gemm_common.hpp"

namespace {

typedef __attribute__((ext_vector_type(2))) uint32_t s8_u32x2;

constexpr int S8_WAVES = 4;
#ifndef S8_RING_DEPTH
#define S8_RING_DEPTH 3
#endif
#ifndef S8_INTERLEAVE
#define S8_INTERLEAVE 0  // (1: pin "one MFMA, three VALU" with sched_group_barrier -- measured equal to the compiler's own schedule, tools/midm_lab)
#endif
#ifndef S8_ABL
#define S8_ABL 0  // lab builds only (tools/midm_lab): timing-only ablations, WRONG results: 1 no x requests, 2 no W requests, 4 no MFMA / dequantisation, 8 no epilogue
#endif
constexpr int S8_RING = S8_RING_DEPTH;
constexpr int S8_SLOT = 10 * 1024;                                  // 8 x fragments + 2 weight fragments of one K-step
constexpr int S8_SMEM_BYTES = S8_WAVES * S8_RING * S8_SLOT;         // 120 KiB
constexpr int S8_RP = 68;                                           // row pitch (floats) of the reduction buffer
static_assert(S8_SMEM_BYTES >= S8_WAVES * 64 * S8_RP * 4, "the reduction buffer (4 waves x 64 x 68 fp32) aliases the rings");
constexpr int64_t S8_COUNTER_BYTES = 16384;

template <bool IS_BF16>
__global__ __launch_bounds__(64 * S8_WAVES) void woq_gemm_w4_strip8_kernel(
    const uint16_t* __restrict__ x, const uint32_t* __restrict__ qweight, const uint16_t* __restrict__ scales,
    const uint32_t* __restrict__ qzeros, const uint16_t* __restrict__ bias, uint16_t* __restrict__ y,
    float* __restrict__ partial, unsigned* __restrict__ counters, int M, int64_t N, int64_t K, int64_t NW, int g_shift, int splitk) {
  constexpr int MB = 8, NB = 2, WAVES = S8_WAVES, RING = S8_RING, SLOT = S8_SLOT;
  constexpr int ROWS = 16 * MB, COLS = 64 * NB, NT = 64 * WAVES;
  extern __shared__ __attribute__((aligned(16))) char s8_smem[];
  float* const red = reinterpret_cast<float*>(s8_smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int jn = lane & 15, oct = lane >> 4;
  const float inv_u = fp8_unit_inverse();
  // XCD-aware tile order (same form as the strip / d2r kernels): an XCD owns a contiguous range of (row strip, column strip) pairs,
  // row strip major, so the column strips of one 128-row strip share its x rows out of ONE XCD's L2
  int bx = (int)blockIdx.x, by = (int)blockIdx.y;
  {
    const int nx = (int)gridDim.x, nt = nx * (int)gridDim.y, L = by * nx + bx;
    const int q = nt / 8, r = nt % 8, xcd = L % 8, idx = L / 8;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    by = t / nx;
    bx = t - by * nx;
  }
  const int64_t n0 = (int64_t)bx * COLS;
  const int m0 = by * ROWS;
  const int slice = blockIdx.z;

  int64_t ncol[NB];
  int zsh[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    ncol[nb] = n0 + 64 * nb + 4 * jn;
    if (ncol[nb] > N - 4) ncol[nb] = N - 4;  // clamped lanes recompute valid columns; their results are not stored
    zsh[nb] = 4 * (int)(ncol[nb] & 7);
  }
  const int steps_total = (int)(K / 32);
  const int Q = WAVES * splitk, q = slice * WAVES + wave;
  const int lo = (int)((int64_t)steps_total * q / Q), hi = (int)((int64_t)steps_total * (q + 1) / Q);

  struct Step {
    uint4 a[MB];
    uint4 w[NB];
  };
  struct Par {
    s8_u32x2 s[NB];
    uint32_t z[NB];
  };
  const uint32_t ring0 = (uint32_t)(uintptr_t)s8_smem + (uint32_t)wave * (RING * SLOT);
  uint32_t xoff[MB], woff[NB], soff[NB], zoff[NB];
  // x fragments: four adjacent lanes fetch the 64 contiguous bytes (32 k) of one row; lane l lands at byte 16 l of the fragment's KiB
  // and carries row l >> 2, chunk (l & 3) ^ (row >> 2): the pick-up (lane (jn, oct) reads row jn, chunk oct) is conflict-free
  const int xr = lane >> 2, xc = (lane & 3) ^ (xr >> 2);
#pragma unroll
  for (int b = 0; b < MB; ++b) {
    int am = m0 + 16 * b + xr;
    if (am > M - 1) am = M - 1;  // rows past M are computed from a valid row and never stored
    xoff[b] = (uint32_t)(((int64_t)am * K + 8 * xc) * 2);
  }
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    woff[nb] = (uint32_t)(((int64_t)oct * N + ncol[nb]) * 4);
    soff[nb] = (uint32_t)(ncol[nb] * 2);
    zoff[nb] = (uint32_t)((ncol[nb] >> 3) * 4);
  }
  // one K-step = 14 requests in the wave's in-order queue: ten LDS-DMA pieces (8 x, 2 W) and the group's raw parameters (registers)
  auto issue = [&](int slot, Par& p, int st) {
    st = st > hi - 1 ? hi - 1 : st;  // the prefetch past the end re-reads the last step
    const int64_t g = g_shift >= 0 ? (((int64_t)st * 32) >> g_shift) : 0;
    const uint16_t* xb = (S8_ABL & 1) ? x : x + (int64_t)st * 32;             // (ablation: every step re-reads the first tiles: cache hits)
    const uint32_t* wb = (S8_ABL & 2) ? qweight : qweight + (int64_t)st * 4 * N;
    const uint16_t* sb = scales + g * N;
    const uint32_t* zb = qzeros + g * NW;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(ring0 + (uint32_t)slot * SLOT);
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_nop 4\n\t"
        "s_mov_b32 m0, %23\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %5, %19\n\t"
        "s_add_u32 m0, %23, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %6, %19\n\t"
        "s_add_u32 m0, %23, 0x800\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %7, %19\n\t"
        "s_add_u32 m0, %23, 0xc00\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %8, %19\n\t"
        "s_add_u32 m0, %23, 0x1000\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %9, %19\n\t"
        "s_add_u32 m0, %23, 0x1400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %10, %19\n\t"
        "s_add_u32 m0, %23, 0x1800\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %11, %19\n\t"
        "s_add_u32 m0, %23, 0x1c00\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %12, %19\n\t"
        "s_add_u32 m0, %23, 0x2000\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %13, %20\n\t"
        "s_add_u32 m0, %23, 0x2400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %14, %20\n\t"
        "global_load_dwordx2 %1, %15, %21\n\t"
        "global_load_dwordx2 %2, %16, %21\n\t"
        "global_load_dword %3, %17, %22\n\t"
        "global_load_dword %4, %18, %22\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep), "=&v"(p.s[0]), "=&v"(p.s[1]), "=&v"(p.z[0]), "=&v"(p.z[1])
        : "v"(xoff[0]), "v"(xoff[1]), "v"(xoff[2]), "v"(xoff[3]), "v"(xoff[4]), "v"(xoff[5]), "v"(xoff[6]), "v"(xoff[7]), "v"(woff[0]), "v"(woff[1]),
          "v"(soff[0]), "v"(soff[1]), "v"(zoff[0]), "v"(zoff[1]), "s"(xb), "s"(wb), "s"(sb), "s"(zb), "s"(dst)
        : "memory", "scc");
  };
  // the oldest of RING steps in flight has landed in its slot / its parameter registers (the younger ones stay in flight)
  auto landed = [&](Par& p) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(p.s[0]), "+v"(p.s[1]), "+v"(p.z[0]), "+v"(p.z[1]) : "i"(14 * (RING - 1)) : "memory");
  };
  auto fetch = [&](Step& t, int slot) {
    const char* base = s8_smem + wave * (RING * SLOT) + slot * SLOT;
    const int apos = (4 * jn + (oct ^ (jn >> 2))) * 16;  // where row jn, chunk oct of an x fragment landed
#pragma unroll
    for (int b = 0; b < MB; ++b) t.a[b] = *reinterpret_cast<const uint4*>(base + b * 1024 + apos);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) t.w[nb] = *reinterpret_cast<const uint4*>(base + 8192 + nb * 1024 + lane * 16);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slot is free for the next DMA once these have returned
  };

  f32x4 acc[MB][4 * NB];
#pragma unroll
  for (int b = 0; b < MB; ++b)
#pragma unroll
    for (int c = 0; c < 4 * NB; ++c) acc[b][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  float scu[4 * NB], nzs[4 * NB];
  const int gmask = g_shift < 0 ? 0x7fffffff : ((1 << (g_shift - 5)) - 1);
  auto refresh = [&](const Par& p, int st) {
    if (st == lo || (st & gmask) == 0) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float sc = f16_bits_to_f32((uint16_t)(p.s[nb][c >> 1] >> (16 * (c & 1))));
          uint32_t zz = ((p.z[nb] >> (zsh[nb] + 4 * c)) & 15u) + 1u;  // modules.py:407-410 (stored zp - 1; wraps above 15)
          zz = zz > 15u ? 0u : zz;
          scu[4 * nb + c] = sc * inv_u;
          nzs[4 * nb + c] = -(float)zz * sc;
        }
    }
  };
  // One wave per SIMD: nobody else fills the matrix pipe while this wave dequantises, and nobody else feeds the VALU while its MFMAs
  // queue up.  Left to the compiler the step is dequantise(word) -> 8 MFMAs -> dequantise(next word) ...: 64 x 16 matrix cycles PLUS 8 x ~76
  // VALU cycles (tools/midm_lab: the K-loop costs the sum).  Here the NEXT word is dequantised between the MFMAs of the current one -- one
  // MFMA, then up to three VALU instructions, eight times per word (sched_group_barrier: 0x8 = MFMA, 0x2 = VALU).
  auto compute = [&](const Step& t) {
    const uint32_t ww[8] = {t.w[0].x, t.w[0].y, t.w[0].z, t.w[0].w, t.w[1].x, t.w[1].y, t.w[1].z, t.w[1].w};
    uint4 bq = dequant8<IS_BF16>(ww[0], scu[0], nzs[0]);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      uint4 bqn = bq;
      if (c + 1 < 8) bqn = dequant8<IS_BF16>(ww[c + 1], scu[c + 1], nzs[c + 1]);
#pragma unroll
      for (int b = 0; b < MB; ++b) acc[b][c] = mfma16<IS_BF16>(t.a[b], bq, acc[b][c]);
#if S8_INTERLEAVE
#pragma unroll
      for (int b = 0; b < MB; ++b) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
      }
#endif
      bq = bqn;
    }
  };

  static_assert(RING >= 2 && 14 * (RING - 1) < 64, "vmcnt counts 63 requests at most");
  if (lo < hi) {
    Par p[RING];  // (indexed by unrolled constants only: registers)
    Step t;
#pragma unroll
    for (int r = 0; r < RING; ++r) issue(r, p[r], lo + r);
    for (int st = lo; st < hi; st += RING) {
#pragma unroll
      for (int r = 0; r < RING; ++r) {
        if (st + r < hi) {
          landed(p[r]);
          fetch(t, r);
          refresh(p[r], st + r);
          issue(r, p[r], st + r + RING);
          if constexpr ((S8_ABL & 4) == 0) compute(t);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped prefetches past the end
  __syncthreads();                                   // every wave's ring is dead: the reduction buffer takes their place

  if constexpr ((S8_ABL & 8) != 0) {
    if (acc[0][0][0] != 12345.678f) return;
  }
  // ---- the four accumulator tiles meet in LDS, 64 rows x 64 columns per pass (D of an MFMA: column 4 jn + c, row 4 oct + r) ----------
  // Every thread sums and stores FOUR adjacent columns at a time: 16-byte LDS reads, 16-byte write-through partial stores (or one 8-byte
  // store of four 16-bit outputs).  (The first form of this epilogue moved single floats -- 256 ds_read_b32 and 64 four-byte sc1 stores per
  // thread -- and cost 20 us of a 33 us launch at M = 256: tools/midm_lab, timing-only ablation 8.)
  const int64_t slab = (int64_t)M * N;
  constexpr int RP = S8_RP;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (nb + h > 0) __syncthreads();  // the previous pass has been read
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          *reinterpret_cast<float4*>(red + (wave * 64 + 16 * b + 4 * oct + r) * RP + 4 * jn) =
              make_float4(acc[4 * h + b][4 * nb + 0][r], acc[4 * h + b][4 * nb + 1][r], acc[4 * h + b][4 * nb + 2][r], acc[4 * h + b][4 * nb + 3][r]);
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 64 * 16 / NT; ++i) {
        const int idx = tid + NT * i, rr = idx >> 4, c4 = (idx & 15) * 4;
        float4 v = *reinterpret_cast<const float4*>(red + rr * RP + c4);
#pragma unroll
        for (int wv = 1; wv < WAVES; ++wv) {  // fixed order
          const float4 u = *reinterpret_cast<const float4*>(red + (wv * 64 + rr) * RP + c4);
          v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        const int m = m0 + 64 * h + rr;
        const int64_t n = n0 + 64 * nb + c4;
        if (m < M && n < N) {  // N % 4 == 0: the four columns exist together
          if (splitk > 1) {
            splitk_store16_sc1(partial + (int64_t)slice * slab + (int64_t)m * N + n, f32x4{v.x, v.y, v.z, v.w});
          } else {
            store_out4<IS_BF16>(y + (int64_t)m * N + n, v, bias ? bias + n : nullptr);
          }
        }
      }
    }
  }
  if (splitk <= 1) return;
  // publish: every wave drains its write-through stores, then one relaxed agent-scope ticket from thread 0
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  unsigned* const counter = counters + (by * gridDim.x + bx);
  if (tid == 0) {
    const unsigned ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = ticket == (unsigned)(splitk - 1);
    if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm for the next call
    red[0] = last ? 1.f : 0.f;
  }
  __syncthreads();
  if (red[0] == 0.f) return;
  // last arriver: fixed-order sum over the slices, sc1 (L2-coherent) 16-byte loads, four column quads x up to 4 slices in flight per thread
  constexpr int QUADS = ROWS * COLS / 4 / NT;  // 16 per thread
  for (int i0 = 0; i0 < QUADS; i0 += 4) {
    f32x4 pv[4][4];
    int64_t off[4];
    bool ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + NT * (i0 + i), rr = idx / (COLS / 4), c4 = (idx % (COLS / 4)) * 4;
      const int m = m0 + rr;
      const int64_t n = n0 + c4;
      ok[i] = m < M && n < N;
      off[i] = ok[i] ? (int64_t)m * N + n : 0;
    }
    const float* sb[4];
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {  // slab bases are wave-uniform (SGPR pairs); slices past splitk re-read the last one and are not added
      const uint64_t a = (uint64_t)(uintptr_t)(partial + (int64_t)(sl < splitk ? sl : splitk - 1) * slab);
      sb[sl] = reinterpret_cast<const float*>((uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(a >> 32)) << 32) |
                                                          (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)a)));
    }
    splitk_load16x16_sc1(pv, sb[0], sb[1], sb[2], sb[3], (uint32_t)(off[0] * 4), (uint32_t)(off[1] * 4), (uint32_t)(off[2] * 4), (uint32_t)(off[3] * 4));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 v = pv[0][i];
#pragma unroll
      for (int sl = 1; sl < 4; ++sl)
        if (sl < splitk) v += pv[sl][i];
      if (ok[i]) {
        const int64_t n = n0 + ((tid + NT * (i0 + i)) % (COLS / 4)) * 4;
        store_out4<IS_BF16>(y + off[i], make_float4(v[0], v[1], v[2], v[3]), bias ? bias + n : nullptr);
      }
    }
  }
}

}  // namespace

// K-slices for (M, N, K): split-K only to fill the chip (one workgroup per CU), <= 4 slabs, >= 4 K-steps per wave
int inc_woq_gemm_strip8_splitk(int64_t M, int64_t N, int64_t K) {
  const int64_t wgs = ((M + 127) / 128) * ((N + 127) / 128);
  int sk = 1;
  if (wgs < 192) {
    sk = (int)(256 / wgs);
    if (sk > 4) sk = 4;
    while (sk > 1 && (K / 32) / (S8_WAVES * sk) < 4) --sk;
  }
  return sk;
}

// Launcher used by inc_woq_gemm (gemm.hip).  `part` / `counters`: split-K slabs and per-tile arrival counters (zero on first use,
// re-armed by the kernel) or NULL with splitk = 1.
int inc_launch_woq_gemm_strip8(const uint16_t* x, const uint32_t* qw, const uint16_t* scales, const uint32_t* qz, const uint16_t* bias,
                               uint16_t* y, int64_t M, int64_t N, int64_t K, int64_t NW, int g_shift, float* part, unsigned* counters,
                               int splitk, bool bf, hipStream_t s) {
  static std::atomic<uint64_t> attr_set{0};
  if (inc_attr_needed(attr_set)) {
    (void)hipFuncSetAttribute((const void*)woq_gemm_w4_strip8_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, S8_SMEM_BYTES);
    (void)hipFuncSetAttribute((const void*)woq_gemm_w4_strip8_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, S8_SMEM_BYTES);
    inc_attr_done(attr_set);
  }
  dim3 grid((unsigned)((N + 127) / 128), (unsigned)((M + 127) / 128), (unsigned)splitk);
  if (bf) woq_gemm_w4_strip8_kernel<true><<<grid, 64 * S8_WAVES, S8_SMEM_BYTES, s>>>(x, qw, scales, qz, bias, y, part, counters, (int)M, N, K, NW, g_shift, splitk);
  else woq_gemm_w4_strip8_kernel<false><<<grid, 64 * S8_WAVES, S8_SMEM_BYTES, s>>>(x, qw, scales, qz, bias, y, part, counters, (int)M, N, K, NW, g_shift, splitk);
  INC_LAUNCH_RETURN();
}
