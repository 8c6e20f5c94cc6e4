/*
 * inc_mi355x.h -- C-ABI of libinc_mi355x.so: the MI355X (gfx950 / CDNA4) implementation of
 * intel/neural-compressor's weight-only-quant hot path.
 *
 * The reference (INC 3.9) is 100 % Python and has NO FFI for this path; each entry point below
 * replaces the torch/numpy/numba arithmetic of the cited reference function (file:line relative to
 * /root/reference).  INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types.  All pointers are DEVICE pointers (HBM) unless noted.
 *   - the caller owns every buffer (torch-allocated, passed as tensor.data_ptr()); the library
 *     allocates nothing persistent and keeps no state -> thread-safe.
 *   - every call is asynchronous w.r.t. the host and ordered on `stream` (a hipStream_t passed as
 *     void*; NULL = the default stream).
 *   - return value: INC_OK (0) or a negative INC_ERR_* code; the C side never throws.
 *   - tensors are dense row-major; "[N,K]" means N rows of K contiguous elements.
 *   - dtype codes: INC_F32 / INC_F16 / INC_BF16 for floating tensors.
 */
#ifndef INC_MI355X_H_
#define INC_MI355X_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define INC_OK 0
#define INC_ERR_BAD_ARG (-1)     /* null pointer / non-positive size / inconsistent shape          */
#define INC_ERR_UNSUPPORTED (-2) /* valid request this build does not implement (e.g. bits = 9)    */
#define INC_ERR_LAUNCH (-3)      /* hipGetLastError() != hipSuccess after the launch               */
#define INC_ERR_WORKSPACE (-4)   /* workspace too small (see *_workspace_bytes)                    */

#define INC_F32 0
#define INC_F16 1
#define INC_BF16 2

#define INC_SCHEME_ASYM 0
#define INC_SCHEME_SYM 1

typedef void* inc_stream_t; /* hipStream_t */

/* ---- library info ------------------------------------------------------------------------- */
int inc_abi_version(void);                 /* bumps on any signature change                      */
const char* inc_error_string(int code);    /* static string for an INC_ERR_* code                */
const char* inc_target_arch(void);         /* "gfx950"                                           */

/* The library keeps no mutable process-global state: every call is a function of its arguments and the stream.
 * (The A/B switch of tools/kbench, inc_debug_set_small_tiles, exists only in the harness build
 * tools/libinc_mi355x_kbench.so, compiled from the same sources with -DINC_KBENCH.)                              */

/* ---- K1/K2: bit packing ------------------------------------------------------------------- *
 * inc_pack_rows  == INCWeightOnlyLinear.pack_tensor   (weight_only/modules.py:580, :445, :546,
 *                   numba packers torch/utils/bit_packer.py:35-278):
 *     packed[r, j] = OR_e ((raw[r, j*n_pack + e] & (2^bits-1)) << (bits*e)),  n_pack = cbits/bits
 *   raw: int32 [rows, cols]; packed: [rows, ceil(cols/n_pack)] words of `cbits` bits.
 *   bits in 1..8 (every width the reference's configs tune, torch/quantization/config.py:211; 3 / 5 / 6 / 7 leave the
 *   word's high bits unused, modules.py:231); cbits in {8,16,32,64}, cbits >= bits.
 * inc_unpack_rows == INCWeightOnlyLinear.unpack_tensor (modules.py:587, :468, :558):
 *     out[r, j*n_pack+e] = (packed[r,j] << (cbits-bits*(e+1))) >>arith (cbits-bits), then & mask
 *     iff mask_sign != 0 (the reference masks iff the module has `qzeros`); out: int16.
 */
int inc_pack_rows(const int32_t* raw, void* packed, int64_t rows, int64_t cols, int bits, int cbits,
                  inc_stream_t stream);
int inc_unpack_rows(const void* packed, int16_t* out, int64_t rows, int64_t packed_cols, int bits,
                    int cbits, int mask_sign, inc_stream_t stream);

/* ---- K1 fused: pack into the "optimum" (HF/AutoGPTQ) layout ------------------------------- *
 * == INCWeightOnlyLinear.pack with use_optimum_format=True (modules.py:321-375).
 *   int_weight [N,K] (int32 when in_bytes == 4, int8 when in_bytes == 1); `shift` is added to every
 *     value before masking (reference: +2^(bits-1) when zp is None, modules.py:329-334).
 *   scales   [N,G] fp32  -> scales_out [G,N] fp16          (modules.py:346,372)
 *   zp       [N,G] int32 or NULL.  NULL => every zero point is 2^(bits-1) (sym, modules.py:334).
 *            qzeros stores zp-1 (modules.py:364) packed along N: qzeros [G, ceil(N/n_pack)] int32.
 *            (`shift` only moves the weights: pass signed ints with shift=2^(bits-1), or already
 *            offset codes 0..2^bits-1 with shift=0 -- the GPTQ kernel emits the latter.)
 *   qweight  [ceil(K/n_pack), N] int32: nibble e of qweight[r, n] = int_weight[n, r*n_pack+e]+shift.
 */
int inc_woq_pack(const void* int_weight, int in_bytes, const float* scales, const int32_t* zp,
                 int32_t* qweight, int32_t* qzeros, uint16_t* scales_out, int64_t N, int64_t K,
                 int64_t G, int bits, int shift, inc_stream_t stream);

/* ---- K2 fused: unpack the optimum layout -------------------------------------------------- *
 * == INCWeightOnlyLinear.unpack (modules.py:377-411): int_weight [N,K] int16 (0..2^bits-1),
 *   zp [N,G] int16 = stored+1, values > 2^bits-1 wrap to 0 (modules.py:407-410).
 *   Either output pointer may be NULL to skip it.  scales_ng (with int_weight; NULL = off): the module's scales [G,N] fp16
 *   (scales_gn) written as [N,G] in the same launch -- unpack returns `scales.T.contiguous()` (modules.py:382).
 */
int inc_woq_unpack(const int32_t* qweight, const int32_t* qzeros, int16_t* int_weight, int16_t* zp,
                   int64_t N, int64_t K, int64_t G, int bits, const uint16_t* scales_gn, uint16_t* scales_ng,
                   inc_stream_t stream);

/* ---- K3: recover / dequantize ------------------------------------------------------------- *
 * == INCWeightOnlyLinear.recover (modules.py:413-443):
 *     W[n,k] = int8(q[n,k] - zp[n,g(k)]) * scales[n,g(k)],  g(k) = g_idx ? g_idx[k] : k / group_size
 *   computed exactly in fp32 then rounded ONCE to out_dtype (fp16 reproduces the reference bit for
 *   bit).  Optimum layout in: qweight [ceil(K/n_pack),N], scales [G,N] fp16, qzeros [G,ceil(N/np)].
 *   out [N,K] of out_dtype.
 */
int inc_woq_dequant(const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros,
                    const int32_t* g_idx, void* out, int out_dtype, int64_t N, int64_t K, int64_t G,
                    int group_size, int bits, inc_stream_t stream);

/* same arithmetic from already-unpacked integers (non-optimum formats, modules.py:270-314):
 *   int_weight [N,K] int16, scales [N,G] of scale_dtype, zp [N,G] int16 or NULL.               */
int inc_dequant_ints(const int16_t* int_weight, const void* scales, int scale_dtype,
                     const int16_t* zp, const int32_t* g_idx, void* out, int out_dtype, int64_t N,
                     int64_t K, int64_t G, int group_size, inc_stream_t stream);

/* ---- K4: fused INT4/INT8 unpack + group dequant + GEMM ------------------------------------- *
 * == INCWeightOnlyLinear.forward (modules.py:594-610) == F.linear(x, recover(), bias), without
 *   ever materialising the dense weight.  x [M,K] and y [M,N] of dtype `xdtype` (INC_BF16 or
 *   INC_F16), fp32 accumulate; weights dequantised to `xdtype` in registers.
 *   bias [N] of `xdtype` or NULL.  bits in 1..8: 4 and 8 take the fast kernels below; 1 / 2 / 3 / 5 / 6 / 7 (n_pack = 32 / bits
 *   fields per word, modules.py:231) take the 128x128 tile kernel's per-element form (any group_size, any g_idx).
 *   g_idx [K] int32 or NULL: the group of every k
 *   (act_order / HF desc_act checkpoints, modules.py:341-344, 427-431); with a g_idx the general 128x128 tile
 *   kernel (or the M <= 16 split-K kernel) looks scale / zero up per element.  A g_idx that permutes whole groups
 *   is faster through a K-sorted copy of the words and a gather of x, which MI355XWeightOnlyLinear does once per module.
 *   The library picks the kernel itself: 256x256x64 LDS-DMA tile kernel (4-bit, M >= 128, K % 64 == 0,
 *   power-of-two group_size >= 32 or one group), split-K MFMA GEMV (M <= 16), or the generic 128x128
 *   tile kernel for everything else.
 *   `workspace` (inc_woq_gemm_workspace_bytes bytes; may be 0).  Medium M (fewer 256x256 tiles than CUs): fp32
 *   split-K slabs, summed in a fixed order by a second small kernel; without a workspace the call still works,
 *   single pass.  M <= 16:
 *   its first 16 KiB hold the per-strip arrival counters of the in-kernel split-K reduction and MUST BE
 *   ZERO when the workspace is first used (the last-arriving workgroup re-arms them, so a workspace that
 *   is only ever handed to this function stays valid); the fp32 partials follow.  One workspace must not
 *   be shared by calls that may run concurrently (different streams).
 */
int64_t inc_woq_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K);
int inc_woq_gemm(const void* x, int xdtype, const int32_t* qweight, const uint16_t* scales,
                 const int32_t* qzeros, const int32_t* g_idx, const void* bias, void* y, int64_t M,
                 int64_t N, int64_t K, int64_t G, int group_size, int bits, void* workspace,
                 int64_t workspace_bytes, inc_stream_t stream);

/* Several packed modules that multiply the SAME x, in ONE launch -- the decode path of q / k / v (and of gate / up): n calls of
 * INCWeightOnlyLinear.forward (modules.py:594-610) on one activation, which the reference issues one F.linear after the other.
 *   x [M,K] of `xdtype`, M <= 64; module i: qweight[i] [K/8,N[i]], scales[i] [G,N[i]] fp16, qzeros[i] [G,ceil(N[i]/8)], bias[i] [N[i]] of
 *   `xdtype` or NULL (bias itself may be NULL), y[i] [M,N[i]] of `xdtype`; K, group_size and bits (4, or 8 with M <= 16) are common, no g_idx.
 *   The arrays of pointers / sizes live in HOST memory (like inc_gptq_hessian_accum_multi); the state dict is untouched.
 *   Every 64-column strip is computed exactly as inc_woq_gemm's streaming kernel computes it, so the launch is bit-identical to
 *   inc_woq_gemm on the N-concatenated module (deterministic: fixed-order split-K sum).  INC_ERR_UNSUPPORTED (nothing launched)
 *   when the batch is not eligible (n < 2 or > 8, M > 64, bits other than 4 / 8, a group size that is not a power of two >= 32 or one
 *   group, N[i] < 64 or N[i] % 4 != 0, K % 32 != 0, unaligned x, or M > 32 on more than 24 Mi weights, where inc_woq_gemm's
 *   strip kernel is the faster form): call inc_woq_gemm per module then.
 *   `workspace`: inc_woq_gemm_multi_workspace_bytes bytes; its first 16 KiB are arrival counters with the rules of inc_woq_gemm's. */
int64_t inc_woq_gemm_multi_workspace_bytes(int n, int64_t M, const int64_t* N, int64_t K);
int inc_woq_gemm_multi(int n, const void* x, int xdtype, const int32_t* const* qweight, const uint16_t* const* scales,
                       const int32_t* const* qzeros, const void* const* bias, void* const* y, int64_t M, const int64_t* N,
                       int64_t K, int group_size, int bits, void* workspace, int64_t workspace_bytes, inc_stream_t stream);

/* ---- K7: group-wise round-to-nearest quantisation ------------------------------------------ *
 * == quant_tensor / qdq_weight_sym / qdq_weight_asym (weight_only/utility.py:272-436, :199, :162).
 *   w [N,K] of `wdtype`, quantised per row in groups of `group_size` along K (tail group = the
 *   remainder, utility.py:334-376).  scheme INC_SCHEME_SYM / _ASYM, quantile, full_range as the
 *   reference.  Arithmetic is done in `wdtype` precision exactly like the reference's torch ops
 *   (every intermediate rounded to wdtype; fp32 for INC_F32).
 *   Outputs (any may be NULL):
 *     qdq_out   [N,K] wdtype : fake-quantised weight (may alias w -> the reference's in-place mode)
 *     int_out   [N,K] int32  : integer codes (sym: -2^(b-1)..2^(b-1)-1, asym: 0..2^b-1)
 *     scale_out [N,G] fp32, zp_out [N,G] fp32 (asym only; ignored for sym)
 */
int inc_groupwise_quant(const void* w, int wdtype, void* qdq_out, int32_t* int_out, float* scale_out,
                        float* zp_out, int64_t N, int64_t K, int group_size, int bits, int scheme,
                        float quantile, int full_range, inc_stream_t stream);

/* Group-wise code-book quantisation == quantize_4bit (utility.py:112-149; the NF4 / FP4 branch of quant_tensor :246-265):
 *   scale[n,g] = max|w| * quantile / max(values);  w / scale -> nearest entry by the midpoint intervals of the reference.
 *   values [n_entries] ascending fp32 code book and codes [n_entries] (the integers written to int_out, INT_MAPPING) are HOST
 *   arrays read before the call returns.  Outputs (any may be NULL): qdq_out [N,K] wdtype (may alias w), int_out [N,K] int32,
 *   scale_out [N,G] fp32.                                                                                          */
int inc_codebook_quant(const void* w, int wdtype, void* qdq_out, int32_t* int_out, float* scale_out, int64_t N,
                       int64_t K, int group_size, const float* values, const int32_t* codes, int n_entries,
                       float quantile, inc_stream_t stream);

/* The same with the caller's scales: quantize_4bit(tensor, scale=...) (utility.py:127-128: `scale = kwargs["scale"]`, the
 * tensor is divided by it instead of by its own max).  scale_in [N,G] fp32 (device; must not alias scale_out); NULL = compute
 * the scales (== inc_codebook_quant).  `quantile` is ignored when scale_in is given, like in the reference.               */
int inc_codebook_quant_with_scale(const void* w, int wdtype, void* qdq_out, int32_t* int_out, float* scale_out, int64_t N,
                                  int64_t K, int group_size, const float* values, const int32_t* codes, int n_entries,
                                  float quantile, const float* scale_in, inc_stream_t stream);

/* *out += sum((a-b)^2) over n elements, in fp64 with a FIXED summation order (same input -> same bits on every
 * launch; zero *out first).  == the loss of search_clip (utility.py:468) and AWQ's output-MSE (awq.py:336-344,
 * 450-458), which the reference accumulates in Python doubles and takes an argmin over.  `workspace`: at least
 * inc_mse_accumulate_workspace_bytes() bytes of caller-owned device memory (per-workgroup partials).              */
int64_t inc_mse_accumulate_workspace_bytes(void);
int inc_mse_accumulate(const void* a, const void* b, int dtype, int64_t n, double* out, void* workspace,
                       inc_stream_t stream);

/* ---- K5: GPTQ Hessian accumulation ---------------------------------------------------------- *
 * == GPTQ.add_batch (weight_only/gptq.py:1111-1141):
 *     H <- beta*H + alpha * X^T X        (beta = n/(n+b), alpha = 2/(n+b) computed by the caller)
 *   x [T,K] of `xdtype` (row stride ldx elements), H [K,K] fp32.  Only the tiles on/above the
 *   diagonal are touched (syrk); call inc_gptq_hessian_finalize once before factorising.
 *   bf16/fp16 inputs use the bf16/f16 MFMA with fp32 accumulation (products exact);
 *   fp32 inputs use the exact-fp32 MFMA.
 */
int inc_gptq_hessian_accum(const void* x, int xdtype, int64_t T, int64_t K, int64_t ldx, float* H,
                           float beta, float alpha, inc_stream_t stream);

/* The same update for up to 8 Hessians in ONE launch: the distinct layer inputs of one calibration forward of a block
 * (add_batch runs once per hooked layer per forward, gptq.py:670-688) -- xs[i] [T,Ks[i]] 16-bit with row stride
 * ldxs[i], Hs[i] [Ks[i],Ks[i]] fp32, all with the same token count T.  The small Hessians no longer leave half of the
 * CUs idle, and -- given `workspace` (>= inc_gptq_hessian_accum_multi_workspace_bytes(), 16-byte aligned; NULL = off) --
 * the tiles of the launch's last, partly filled round are cut into token ranges computed by otherwise idle CUs and added
 * into H in range order by a second small launch (deterministic).  Every other tile is computed exactly as by
 * inc_gptq_hessian_accum (bit-identical); a tile of the split tail differs from it by the fp32 rounding of adding two to four
 * partial sums instead of one.
 * INC_ERR_UNSUPPORTED (nothing launched) for fp32 inputs, K < 256, unaligned rows or more than 8 problems: call the
 * single-problem entry instead.  The pointer / size arrays are HOST arrays, read before the call returns.             */
int64_t inc_gptq_hessian_accum_multi_workspace_bytes(void);
int inc_gptq_hessian_accum_multi(int n, const void* const* xs, int xdtype, int64_t T, const int64_t* Ks,
                                 const int64_t* ldxs, float* const* Hs, const float* betas, const float* alphas,
                                 void* workspace, int64_t workspace_bytes, inc_stream_t stream);

/* == GPTQ.fasterquant prologue (gptq.py:1186-1189, 1221-1227): mirror the upper triangle to the
 *   lower, dead[i] = (H[i,i]==0) -> H[i,i]=1, damp = percdamp*mean(diag(H)), H[i,i] += damp.
 *   dead: uint8 [K] out.  workspace: >= 16 bytes.                                               */
int inc_gptq_hessian_finalize(float* H, int64_t K, float percdamp, uint8_t* dead, void* workspace,
                              inc_stream_t stream);

/* == `W = W.float(); W[:, dead] = 0` (gptq.py:1176, 1189): weight of `wdtype` -> fp32 working copy
 *   with the columns flagged in dead (uint8 [K], may be NULL) zeroed.                           */
int inc_gptq_prepare_weight(const void* w, int wdtype, float* out, const uint8_t* dead, int64_t N,
                            int64_t K, inc_stream_t stream);

/* == Quantizer.find_params(weight=True) (gptq.py:1501-1624, int dtype, perchannel, no mse):
 *   per row n and per group g over columns [col0 + g*group_size, ...) of w [N,K] fp32:
 *     scale[n, g0+g], zero[n, g0+g]  (fp32 [N,G]) for ngroups groups.
 *   sym: scale = 2*absmax/maxq, zero = (maxq+1)/2; asym: scale=(max-min)/maxq, zero=round(-min/scale)
 */
int inc_gptq_find_params(const float* w, int64_t N, int64_t K, int64_t col0, int group_size,
                         int ngroups, int bits, int sym, float* scale, float* zero, int64_t G,
                         int64_t g0, inc_stream_t stream);

/* == the same with Quantizer's `mse` shrink-grid search (gptq.py:1567-1584, GPTQConfig(use_mse_search=True)):
 *   for i < int(maxshrink*grid): p = 1 - i/grid; range scaled by p; keep the (scale, zero) minimising
 *   sum |quantize(x) - x|^norm over the group (strict '<').  The reference's configure() uses grid=100,
 *   maxshrink=0.8, norm=2.4 (gptq.py:1375-1387).                                                     */
int inc_gptq_find_params_mse(const float* w, int64_t N, int64_t K, int64_t col0, int group_size,
                             int ngroups, int bits, int sym, int grid, float maxshrink, float norm,
                             float* scale, float* zero, int64_t G, int64_t g0, inc_stream_t stream);

/* == the serial column loop of GPTQ.fasterquant for ONE block of columns [i1, i1+count)
 *   (gptq.py:1250-1299), count <= 128, rows independent:
 *     for i: q = scale*(clamp(rint(w/scale)+zero,0,maxq)-zero); err=(w-q)/Hinv[i,i];
 *            W1[:, i:] -= err (x) Hinv[i, i:];
 *   Un-fused fp32 arithmetic (true divisions, mul-then-sub) == the reference's torch ops.
 *   w [N,K] fp32 working copy (read only here), Hinv [K,K] fp32 upper Cholesky factor of H^-1,
 *   scale/zero [N,G] fp32 (group of column c = c / group_size; group_size<=0 -> one group),
 *   outputs: codes uint8 [N,K] (0..maxq; may be NULL), q_out [N,K] of q_dtype (dequantised,
 *            gptq.py:1337; may be NULL), err [N,128] fp32 (Err1, consumed by inc_gptq_lazy_update).
 */
int inc_gptq_quant_block(const float* w, const float* Hinv, const float* scale, const float* zero,
                         uint8_t* codes, void* q_out, int q_dtype, float* err, int64_t N, int64_t K,
                         int64_t G, int64_t i1, int count, int group_size, int bits,
                         inc_stream_t stream);

/* == inc_gptq_quant_block that ALSO computes the (scale, zero) of the block's groups first: Quantizer.find_params
 *   (gptq.py:1501-1571; perchannel, weight=True, no mse search) on the 128 columns as they are when the block starts
 *   (gptq.py:1266-1272), written to scale / zero [N,G].  Only for a full block on a 128-column boundary whose groups lie
 *   inside it (group_size 32 / 64 / 128) and a reference block size of 128: INC_ERR_UNSUPPORTED otherwise (callers then
 *   use inc_gptq_find_params + inc_gptq_quant_block).  Saves one launch per 128 columns of the serial chain.          */
int inc_gptq_quant_block_params(const float* w, const float* Hinv, float* scale, float* zero,
                                uint8_t* codes, void* q_out, int q_dtype, float* err, int64_t N, int64_t K,
                                int64_t G, int64_t i1, int count, int group_size, int bits, int sym,
                                inc_stream_t stream);

/* == W[:, i2:] -= Err1 @ Hinv[i1:i2, i2:]  (gptq.py:1304), fp32 MFMA.  err [N,128].           */
int inc_gptq_lazy_update(float* w, const float* Hinv, const float* err, int64_t N, int64_t K,
                         int64_t i1, int count, inc_stream_t stream);

/* == the same update (gptq.py:1304) for the columns [col_begin, col_end) only; col_begin - (i1 + count) a multiple
 *   of 128, col_end == K or a multiple of 128 columns past col_begin.  Lets the caller apply the next block's 128
 *   columns first and the remainder on a second stream while the next block's column loop runs; W is bit-identical
 *   to the one-call form.  INC_ERR_UNSUPPORTED when count != 128 (callers then use inc_gptq_lazy_update).       */
int inc_gptq_lazy_update_cols(float* w, const float* Hinv, const float* err, int64_t N, int64_t K,
                              int64_t i1, int count, int64_t col_begin, int64_t col_end,
                              inc_stream_t stream);

/* == K6 as ONE call: the blocked column loop of GPTQ.fasterquant (gptq.py:1250-1304) for one (N-stacked) layer, given the
 *   inverse-Cholesky factor: per 128 columns [find_params ->] quantisation chain -> lazy update, issued by the library on
 *   `stream` in the reference's order.  With `aux_stream` != NULL (and K a multiple of 128, K >= 384, block_size a multiple
 *   of 128) the bulk of every lazy update runs on it underneath the next block's chain; results are bit-identical either way.
 *     w [N,K] fp32 working copy (inc_gptq_prepare_weight; column-permuted by the caller for act_order) -- consumed;
 *     Hinv [K,K] fp32; scale / zero [N,G] fp32: written for dynamic groups (find_params on W "as it is now", gptq.py:1266-1272),
 *     read otherwise; loop_scale / loop_zero [N,loop_G] (may be NULL = scale / zero): the table the chain reads, with
 *     kernel_group_size columns per entry (<= 0: one group) -- differs from scale / zero only for act_order + static_groups
 *     (one entry per column); codes uint8 [N,K] and q_out [N,K] of q_dtype (either may be NULL); err_ws fp32 [2, N, 128];
 *     group_size = columns per quantisation group (K for per-channel); block_size = GPTQConfig.block_size (<= 0: K);
 *     flags: INC_GPTQ_DYNAMIC_GROUPS (group_size != -1 and not static_groups), INC_GPTQ_MSE (use_mse_search),
 *            INC_GPTQ_NO_LOOKAHEAD, INC_GPTQ_NO_FUSED_PARAMS (A/B switches; default = fastest bit-identical form).      */
#define INC_GPTQ_DYNAMIC_GROUPS 1
#define INC_GPTQ_MSE 2
#define INC_GPTQ_NO_LOOKAHEAD 4
#define INC_GPTQ_NO_FUSED_PARAMS 8
int inc_gptq_quantize_layer(float* w, const float* Hinv, float* scale, float* zero, int64_t G,
                            const float* loop_scale, const float* loop_zero, int64_t loop_G, uint8_t* codes,
                            void* q_out, int q_dtype, float* err_ws, int64_t N, int64_t K, int group_size,
                            int kernel_group_size, int block_size, int bits, int sym, int flags,
                            inc_stream_t stream, inc_stream_t aux_stream);

/* ---- K6': diagonal block of the blocked inverse-Cholesky factor -------------------------------- *
 * The reference builds Hinv = cholesky(cholesky_inverse(cholesky(H)), upper) (gptq.py:1228-1230) with
 * three LAPACK factorisations.  Here U = J Lr^-1 J with J H J = Lr Lr^T (see gptq.py: inverse_cholesky_upper):
 * a blocked Cholesky + a blocked triangular inverse whose O(K^3) parts are fp32 GEMMs; this entry point is
 * the unblocked kernel for one diagonal block (== LAPACK potf2 + trti2 on it):
 *   A [n,n] fp32 (row stride lda, n <= 128): lower triangle read; on return the lower triangle holds L
 *   (A = L L^T) and the strict upper triangle is zero.  Linv [n,n] (row stride ldi) receives L^-1.
 *   info: device int32, atomically max-ed with `tag` when a pivot is not positive (H not SPD).
 */
int inc_chol_diag_block(float* A, int64_t lda, int n, float* Linv, int64_t ldi, int32_t* info, int tag,
                        inc_stream_t stream);

/* ---- K6': the whole inverse-Cholesky factor (gptq.py:1228-1231) as ONE call ---------------------- *
 * inc_gptq_inverse_factor: U [K,K] fp32 = upper Cholesky factor of H^-1 (H^-1 = U^T U) for the symmetric positive
 *   definite H [K,K] fp32 (dense, row stride K; already damped: inc_gptq_hessian_finalize; H is only read).
 *   == `H = cholesky(H); H = cholesky_inverse(H); Hinv = cholesky(H, upper=True)`, computed as U = J L^-1 J with
 *   J H J = L L^T: one blocked Cholesky + one blocked triangular inverse (128 inside 1024 columns), every product an
 *   exact-fp32 MFMA GEMM of this library (v_mfma_f32_32x32x2_f32, triangular operands skipped by K-range, syrk on the
 *   lower tiles only), the diagonal blocks by the inc_chol_diag_block kernel.  The chain (diagonal blocks, block
 *   inverses, panel solves, the update of the next block's columns) is issued on `stream`; with `aux_stream` (may be
 *   NULL) the rest of every trailing update and the top-level doubling products run there underneath the chain (two
 *   transient HIP events; the call returns with `stream` ordered behind everything).  Deterministic.  With flags = 0 the
 *   two-stream form is bit-identical to the one-stream form; with flags bit 1 it is NOT: the side stream's products stay exact
 *   fp32 (one plane buffer, owned by the main stream), so the factor differs from the one-stream form within the distance
 *   bit 1 is gated by (both <= 1e-6 relative from an fp64 factor).
 *   workspace: >= inc_gptq_inverse_factor_workspace_bytes(K, flags) bytes, 16-byte aligned, contents undefined on return
 *   (3 Kp^2 + 2048 Kp floats, Kp = K rounded up to 128; + 6 (Kp + 256)(Kp + 128) bytes of bf16 planes with flags bit 1:
 *   2.2 GB at K = 11008, 14.8 GB at K = 28672).
 *   info: device int32, written by the call: 0, or the (1-based) index of the last 128-column diagonal block with a
 *   non-positive pivot (H not positive definite -- the reference's torch.linalg.cholesky raises there); U is
 *   undefined in that case.  flags: bit 0 = ignore aux_stream (one stream); bit 1 = the large products with
 *   their fp32 operands split into three bf16 pieces (six bf16 MFMAs per fp32 product: dropped terms <= 2^-24 |a b|, same distance to an
 *   fp64 factor as the exact-fp32 products of flags = 0, 1.3 - 2 x faster; the Python driver's default).                */
int64_t inc_gptq_inverse_factor_workspace_bytes(int64_t K, int flags);
int inc_gptq_inverse_factor(const float* H, int64_t K, float* U, void* workspace, int64_t workspace_bytes, int32_t* info,
                            int flags, inc_stream_t stream, inc_stream_t aux_stream);

/* ---- K8: AWQ statistics ---------------------------------------------------------------------- *
 * inc_awq_act_abs_sum: out[k] += sum_t |x[t,k]|  (fp32 [K], accumulate; caller divides by T)
 *   == _get_act_scale (weight_only/awq.py:151-154).
 * inc_awq_weight_scale: out[k] = sum_n( |w[n,k]| / max_{k' in group(k)} |w[n,k']| )
 *   == _get_weight_scale (awq.py:131-147).  w [N,K] of wdtype, out fp32 [K].
 */
int inc_awq_act_abs_sum(const void* x, int xdtype, int64_t T, int64_t K, float* out,
                        inc_stream_t stream);
/* out[k] += sum_n |w[n,k]| / groupmax (zero `out` first; the caller divides by N).
 * workspace: inc_awq_weight_scale_workspace_bytes (fp32 group maxima [N, K/group_size]).          */
int64_t inc_awq_weight_scale_workspace_bytes(int64_t N, int64_t K, int group_size);
int inc_awq_weight_scale(const void* w, int wdtype, int64_t N, int64_t K, int group_size, float* out,
                         void* workspace, int64_t workspace_bytes, inc_stream_t stream);

/* ---- K9: AutoAWQ checkpoint words -> optimum layout -------------------------------------------- *
 * == repack_awq_to_optimum_format (weight_only/utility.py:1426-1459 = unpack_awq :1273 + awq_reverse_reorder_int_tensor
 *    :1246 + pack_from_tensors :1355), called by repack_awq_and_load_state_dict
 *    (transformers/quantization/utils.py:655-697) when an AutoAWQ checkpoint is loaded.  Integer field shuffle:
 *   awq_qweight [K, N/8] int32 (field i of word (k,c) = code(k, 8c + {0,2,4,6,1,3,5,7}[i]))  ->  qweight [K/8, N] int32
 *   awq_qzeros  [G, N/8] int32 (same field order, plain zero points)                       ->  qzeros  [G, N/8] int32
 *                                                                                            (sequential, (z-1)&15)
 * scales [G, N] fp16 are shared unchanged.  bits must be 4 (as in the reference); K % 8 == 0, N % 8 == 0.      */
int inc_awq_repack(const int32_t* awq_qweight, const int32_t* awq_qzeros, int64_t K, int64_t N, int64_t G, int bits,
                   int32_t* qweight, int32_t* qzeros, inc_stream_t stream);

/* ---- K10-K14: SmoothQuant W8A8 (BASELINE config #4) ------------------------------------------------ *
 * Reference: neural_compressor/torch/algorithms/smooth_quant/utility.py.  The reference executes W8A8 through
 * intel_extension_for_pytorch (smooth_quant.py:105-125; not vendored): the in-tree fake-quant functions are the spec.
 *
 * inc_sq_channel_minmax: mn[k] = min(mn[k], min_t x[t,k]), mx[k] = max(mx[k], max_t x[t,k])          (fp32 [K] each,
 *   initialise to +FLT_MAX / -FLT_MAX) == Calibration._save_input_pc_hook (:858-883), x [T,K] with row stride ld.
 * inc_sq_weight_col_absmax: out[k] = max(out[k], max_n |w[n,k]|) (zero `out` first; call once per Linear that shares
 *   the input) == the `torch.max(torch.abs(torch.cat(weights)), dim=0)` of cal_scale (:617-618).
 * inc_sq_cal_scale: s[k] = clip(amax_x[k]^alpha / clip(amax_w[k], lb)^(1-alpha), 1e-5), s = 1 where amax_x^alpha == 0
 *   == cal_scale (:605-626).
 * inc_sq_quant_weight: per-output-channel symmetric int8 of W * smooth (smooth may be NULL):
 *   scale[n] = clip(max_k |w'| / 127.5, eps), q = clamp(rint(w' / scale), -128, 127) == quant_dequant_w_v1 (:669-690);
 *   qw [N, Kp] int8 (Kp >= K, columns K..Kp-1 are zero), rowsum[n] = sum_k q[n,k] (int32).
 * inc_sq_quant_act: out[m,k] = clamp(rint(x[m,k] * in_scale[k] / sx + zp), 0, 255) - 128 as int8 [M, Kp]
 *   == SQLinearWrapper.forward's mul (:2602) + quant_dequant_x_v1 (:726-755) with the static (sx, zp) of
 *   SQLinearWrapper._calculate_qparams (:2607-2631); in_scale may be NULL (folded smoothing).  Kp % 16 == 0.
 * inc_w8a8_gemm: y[m,n] = alpha[n] * (sum_k xq[m,k] * wq[n,k] + corr[n]) + bias[n]   (v_mfma_i32_32x32x32_i8)
 *   alpha[n] = sx * w_scale[n], corr[n] = (128 - zp) * rowsum[n] (may be NULL), bias in ydtype or NULL,
 *   y bf16 / fp16 [M,N]; K % 128 == 0 (pad with zero weight codes), xq / wq 16-byte aligned.
 *   workspace (inc_w8a8_gemm_workspace_bytes, may be NULL): when the last round of 256x256 tiles would leave most
 *   CUs idle its tiles are split along K into int32 slabs there and a second small kernel finishes them (no
 *   initialisation needed).  Integer sums: the result is bit-identical with or without the workspace.               */
int64_t inc_w8a8_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K);
int inc_sq_channel_minmax(const void* x, int xdtype, int64_t T, int64_t K, int64_t ld, float* mn, float* mx,
                          inc_stream_t stream);
int inc_sq_weight_col_absmax(const void* w, int wdtype, int64_t N, int64_t K, float* out, inc_stream_t stream);
int inc_sq_cal_scale(const float* amax_x, const float* amax_w, int64_t K, float alpha, float weight_max_lb, float* scale,
                     inc_stream_t stream);
int inc_sq_quant_weight(const void* w, int wdtype, int64_t N, int64_t K, int64_t Kp, const float* smooth, int8_t* qw,
                        float* w_scale, int32_t* rowsum, inc_stream_t stream);
int inc_sq_quant_act(const void* x, int xdtype, int64_t M, int64_t K, int64_t Kp, const float* in_scale, float sx, float zp,
                     int8_t* out, inc_stream_t stream);
int inc_w8a8_gemm(const int8_t* xq, const int8_t* wq, const float* alpha, const int32_t* corr, const void* bias, void* y,
                  int ydtype, int64_t M, int64_t N, int64_t K, void* workspace, int64_t workspace_bytes,
                  inc_stream_t stream);

/* ---- measured ceilings (bench.py `ceilings`) -- measurement helpers, not on the hot path ------------- *
 * inc_probe_hbm_triad: a <- b + s*c over n fp32 (n % 4 == 0): 12 n bytes of HBM traffic per call.
 * inc_probe_mfma_bf16: `blocks` workgroups x 4 waves x `iters` x 8 v_mfma_f32_32x32x16_bf16 on operands read from `src`
 *   (>= 64 KiB, any non-zero data); *flops_out (host pointer, may be NULL) receives the flops of the launch.            */
int inc_probe_hbm_triad(float* a, const float* b, const float* c, float s, int64_t n, inc_stream_t stream);
/* dst <- src: `bytes` read + `bytes` written, 16 bytes per lane; `variant` 0..15 picks loads in flight per lane / non-temporal hints /
 * grid shape (probe.hip).  The copy ceiling of the chip (guide: 6.29 TB/s) is what the streaming kernels K1-K3 are priced against. */
int inc_probe_hbm_copy(void* dst, const void* src, int64_t bytes, int variant, inc_stream_t stream);
int inc_probe_mfma_bf16(const void* src, float* sink, int blocks, int iters, double* flops_out, inc_stream_t stream);
/* inc_trace_marker: an EMPTY launch of `id` (1..4096) workgroups of 64 threads on `stream`: a phase boundary that a
 *   `rocprofv3 --kernel-trace` timeline shows as inc_trace_marker_kernel with Grid_Size_X = 64 * id (scripts/step_timeline.py). */
int inc_trace_marker(int id, inc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* INC_MI355X_H_ */
