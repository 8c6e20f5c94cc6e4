"""Tensor-level wrappers over the C-ABI (device memory and streams come from PyTorch-ROCm, arithmetic does not).

Every function takes torch tensors that already live in HBM (`device.type == "cuda"`, i.e. HIP on ROCm),
passes `data_ptr()` + sizes + the current HIP stream to libinc_mi355x.so and returns torch tensors.
Host tensors are rejected: there is no CPU path.
"""

import torch

from ._lib import INC_BF16, INC_F16, INC_F32, INC_SCHEME_ASYM, INC_SCHEME_SYM, check, lib

_DT = {torch.float32: INC_F32, torch.float16: INC_F16, torch.bfloat16: INC_BF16}


def dtype_code(dtype):
    try:
        return _DT[dtype]
    except KeyError as e:
        raise TypeError(f"unsupported floating dtype {dtype}; expected fp32/fp16/bf16") from e


def _dev(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if t.device.type != "cuda":
            raise RuntimeError(
                f"MI355X op got a tensor on {t.device}; tensors must be resident in HBM (device 'cuda' = HIP). "
                "There is no CPU fallback."
            )
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"tensors on different devices: {dev} vs {t.device}")
        if not t.is_contiguous():
            raise RuntimeError("MI355X ops need contiguous tensors")
    return dev


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


# ---------------------------------------------------------------------------------------------------
# K1/K2 generic row packers
# ---------------------------------------------------------------------------------------------------
_CT = {8: torch.int8, 16: torch.int16, 32: torch.int32, 64: torch.int64}


def pack_rows(raw, bits, compress_bits):
    """== INCWeightOnlyLinear.pack_tensor (modules.py:580): [R,C] ints -> [R, ceil(C/n_pack)] words."""
    raw = raw.to(torch.int32).contiguous()
    dev = _dev(raw)
    n_pack = compress_bits // bits
    rows, cols = raw.shape
    out = torch.empty((rows, (cols + n_pack - 1) // n_pack), dtype=_CT[compress_bits], device=dev)
    with torch.cuda.device(dev):
        check(lib.inc_pack_rows(_ptr(raw), _ptr(out), rows, cols, bits, compress_bits, _stream()), "inc_pack_rows")
    return out


def unpack_rows(packed, bits, compress_bits, mask_sign):
    """== INCWeightOnlyLinear.unpack_tensor (modules.py:587): -> int16 [R, C*n_pack]."""
    packed = packed.contiguous()
    dev = _dev(packed)
    n_pack = compress_bits // bits
    rows, pcols = packed.shape
    out = torch.empty((rows, pcols * n_pack), dtype=torch.int16, device=dev)
    with torch.cuda.device(dev):
        check(
            lib.inc_unpack_rows(_ptr(packed), _ptr(out), rows, pcols, bits, compress_bits, int(bool(mask_sign)), _stream()),
            "inc_unpack_rows",
        )
    return out


# ---------------------------------------------------------------------------------------------------
# optimum-format pack / unpack / recover
# ---------------------------------------------------------------------------------------------------
def woq_pack(int_weight, scales, zp, bits, shift, qweight=None, qzeros=None, scales_out=None):
    """== INCWeightOnlyLinear.pack (optimum format, modules.py:321-375).

    int_weight [N,K] int32 or int8/uint8; scales [N,G] fp32; zp [N,G] int32 or None (sym -> `shift`).
    Returns (qweight [K/np, N] int32, qzeros [G, N/np] int32, scales fp16 [G, N]).
    """
    N, K = int_weight.shape
    G = scales.shape[1]
    n_pack = 32 // bits
    if int_weight.dtype in (torch.int8, torch.uint8):
        in_bytes = 1
    else:
        int_weight = int_weight.to(torch.int32)
        in_bytes = 4
    int_weight = int_weight.contiguous()
    scales = scales.to(torch.float32).contiguous()
    zp = None if zp is None else zp.to(torch.int32).contiguous()
    dev = _dev(int_weight, scales, zp)
    if qweight is None:
        qweight = torch.empty(((K + n_pack - 1) // n_pack, N), dtype=torch.int32, device=dev)
    if qzeros is None:
        qzeros = torch.empty((G, (N + n_pack - 1) // n_pack), dtype=torch.int32, device=dev)
    if scales_out is None:
        scales_out = torch.empty((G, N), dtype=torch.float16, device=dev)
    _dev(qweight, qzeros, scales_out)
    with torch.cuda.device(dev):
        check(
            lib.inc_woq_pack(
                _ptr(int_weight), in_bytes, _ptr(scales), _ptr(zp), _ptr(qweight), _ptr(qzeros), _ptr(scales_out),
                N, K, G, bits, shift, _stream(),
            ),
            "inc_woq_pack",
        )
    return qweight, qzeros, scales_out


def woq_unpack(qweight, qzeros, N, K, G, bits, want_weight=True, want_zp=True, scales=None):
    """== INCWeightOnlyLinear.unpack (modules.py:377-411): (int_weight [N,K] int16, zp [N,G] int16).  `scales` [G,N] fp16 (optional,
    with the weights): a third result, the scales as [N,G] (`scales.T.contiguous()`, modules.py:382), written by the same launch."""
    dev = _dev(qweight, qzeros, scales)
    iw = torch.empty((N, K), dtype=torch.int16, device=dev) if want_weight else None
    zp = torch.empty((N, G), dtype=torch.int16, device=dev) if want_zp else None
    sc_t = None
    if scales is not None:
        assert want_weight and scales.dtype == torch.float16 and scales.shape == (G, N) and scales.is_contiguous()
        sc_t = torch.empty((N, G), dtype=torch.float16, device=dev)
    with torch.cuda.device(dev):
        check(
            lib.inc_woq_unpack(_ptr(qweight), _ptr(qzeros), _ptr(iw), _ptr(zp), N, K, G, bits, _ptr(scales), _ptr(sc_t), _stream()),
            "inc_woq_unpack",
        )
    return (iw, zp) if scales is None else (iw, zp, sc_t)


def awq_repack(awq_qweight, awq_qzeros, bits=4):
    """AutoAWQ GEMM-format words -> optimum layout (== repack_awq_to_optimum_format, utility.py:1426-1459).
    awq_qweight [K, N/8] int32, awq_qzeros [G, N/8] int32 -> (qweight [K/8, N], qzeros [G, N/8]); scales are shared."""
    dev = _dev(awq_qweight, awq_qzeros)
    assert awq_qweight.dtype == torch.int32 and awq_qzeros.dtype == torch.int32
    awq_qweight, awq_qzeros = awq_qweight.contiguous(), awq_qzeros.contiguous()
    K, NW = awq_qweight.shape
    G = awq_qzeros.shape[0]
    N = NW * (32 // bits)
    qweight = torch.empty((K // (32 // bits), N), dtype=torch.int32, device=dev)
    qzeros = torch.empty((G, NW), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(lib.inc_awq_repack(_ptr(awq_qweight), _ptr(awq_qzeros), K, N, G, bits, _ptr(qweight), _ptr(qzeros), _stream()),
              "inc_awq_repack")
    return qweight, qzeros


def woq_dequant(qweight, scales, qzeros, g_idx, N, K, group_size, bits, out_dtype=torch.float16):
    """== INCWeightOnlyLinear.recover (modules.py:413-443) from the optimum layout -> dense [N,K]."""
    dev = _dev(qweight, scales, qzeros, g_idx)
    assert scales.dtype == torch.float16, "optimum-format scales are fp16 (modules.py:245)"
    G = scales.shape[0]
    out = torch.empty((N, K), dtype=out_dtype, device=dev)
    with torch.cuda.device(dev):
        check(
            lib.inc_woq_dequant(
                _ptr(qweight), _ptr(scales), _ptr(qzeros), _ptr(g_idx), _ptr(out), dtype_code(out_dtype),
                N, K, G, group_size, bits, _stream(),
            ),
            "inc_woq_dequant",
        )
    return out


def dequant_ints(int_weight, scales, zp, g_idx, group_size, out_dtype):
    """recover() for non-optimum layouts: int16 [N,K], scales [N,G] (any float dtype), zp int16 [N,G] or None."""
    int_weight = int_weight.to(torch.int16).contiguous()
    zp = None if zp is None else zp.to(torch.int16).contiguous()
    scales = scales.contiguous()
    dev = _dev(int_weight, scales, zp, g_idx)
    N, K = int_weight.shape
    G = scales.shape[1]
    out = torch.empty((N, K), dtype=out_dtype, device=dev)
    with torch.cuda.device(dev):
        check(
            lib.inc_dequant_ints(
                _ptr(int_weight), _ptr(scales), dtype_code(scales.dtype), _ptr(zp), _ptr(g_idx), _ptr(out),
                dtype_code(out_dtype), N, K, G, group_size, _stream(),
            ),
            "inc_dequant_ints",
        )
    return out


# ---------------------------------------------------------------------------------------------------
# K4 fused GEMM
# ---------------------------------------------------------------------------------------------------
_ws_cache = {}


def _workspace(dev, nbytes):
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)  # one workspace per (device, stream): see header
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        # zero-filled: the first 16 KiB are the split-K arrival counters of the M <= 16 kernel, which must be zero on
        # first use (the kernel re-arms them); one workspace per (device, stream) so concurrent streams never share them
        buf = torch.zeros(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
        _ws_cache[key] = buf
    return buf


_ws_bytes_cache = {}


def woq_gemm(x2d, qweight, scales, qzeros, bias, N, K, group_size, bits, g_idx=None):
    """y[M,N] = x[M,K] @ dequant(qweight)^T + bias, fused (== INCWeightOnlyLinear.forward, modules.py:594-610).

    This is the decode-path entry (M = 1 runs in ~8 us on the GPU), so the host side is kept lean: argument checks are
    attribute reads, the workspace size is memoised, and the device guard is only entered when the tensor's device is
    not already current."""
    dev = x2d.device
    if dev.type != "cuda" or qweight.device != dev or not x2d.is_contiguous():
        _dev(x2d, qweight, scales, qzeros, bias, g_idx)  # raises the descriptive error
    if x2d.dtype is not torch.bfloat16 and x2d.dtype is not torch.float16:
        raise TypeError("woq_gemm computes in bf16 or fp16")
    if bias is not None and bias.dtype != x2d.dtype:
        bias = bias.to(x2d.dtype)
    M = x2d.shape[0]
    G = scales.shape[0]
    y = torch.empty((M, N), dtype=x2d.dtype, device=dev)
    ws = None
    key = (M, N, K)
    nbytes = _ws_bytes_cache.get(key)
    if nbytes is None:
        nbytes = _ws_bytes_cache[key] = lib.inc_woq_gemm_workspace_bytes(M, N, K)
    if nbytes > 0:  # M <= 16: arrival counters + split-K partials; medium M: split-K slabs of the 256x256 kernel
        ws = _workspace(dev, nbytes)
    stream = torch.cuda.current_stream(dev).cuda_stream
    args = (
        x2d.data_ptr(), INC_BF16 if x2d.dtype is torch.bfloat16 else INC_F16, qweight.data_ptr(), scales.data_ptr(),
        qzeros.data_ptr(), _ptr(g_idx), _ptr(bias), y.data_ptr(), M, N, K, G, group_size, bits, _ptr(ws),
        0 if ws is None else ws.numel(), stream,
    )
    if torch.cuda.current_device() == dev.index:
        rc = lib.inc_woq_gemm(*args)
    else:
        with torch.cuda.device(dev):
            rc = lib.inc_woq_gemm(*args)
    if rc != 0:
        check(rc, "inc_woq_gemm")
    return y


_raw_stream = torch._C._cuda_getCurrentRawStream  # (device index) -> hipStream_t of torch's current stream, no Stream object
_cur_device = torch._C._cuda_getDevice


def _none():
    return None


class WoqGemmCall:
    """inc_woq_gemm with everything that does not change between calls resolved once (the decode path: the kernel is ~6 us,
    so the host side counts).  Built by MI355XWeightOnlyLinear for its packed buffers; per call only the activation pointer,
    the output, the stream and the (device, stream) workspace are looked up.  Holds references to the tensors whose addresses
    it caches; the owner rebuilds it when a buffer is replaced."""

    __slots__ = ("dev", "dev_index", "dtype", "dt", "N", "K", "G", "gs", "bits", "qw", "sc", "qz", "gi", "bi", "keep", "need", "fn",
                 "bias_conv", "versions", "tag")

    def __init__(self, qweight, scales, qzeros, bias, N, K, group_size, bits, dtype, g_idx=None):
        dev = _dev(qweight, scales, qzeros, bias, g_idx)
        if dtype is not torch.bfloat16 and dtype is not torch.float16:
            raise TypeError("woq_gemm computes in bf16 or fp16")
        self.keep = (qweight, scales, qzeros, bias, g_idx)
        self.versions = tuple(None if t is None else t._version for t in self.keep)
        if bias is not None and bias.dtype != dtype:
            bias = bias.to(dtype)  # converted once (the packed module stores fp16, a bf16 model multiplies in bf16)
        self.bias_conv = bias
        self.dev, self.dev_index, self.dtype = dev, dev.index if dev.index is not None else torch.cuda.current_device(), dtype
        self.dt = INC_BF16 if dtype is torch.bfloat16 else INC_F16
        self.N, self.K, self.G, self.gs, self.bits = N, K, scales.shape[0], group_size, bits
        self.qw, self.sc, self.qz, self.gi, self.bi = qweight.data_ptr(), scales.data_ptr(), qzeros.data_ptr(), _ptr(g_idx), _ptr(bias)
        self.need = {}
        self.tag = (None, None)  # (the owner's g_idx buffer, its version) when the owner's plan was chosen
        self.fn = lib.inc_woq_gemm

    # a cache, not state: copies and pickles of the owning module start without it
    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (_none, ())

    def current(self, qweight, scales, qzeros, bias, owner_g_idx=None):
        """Still describes these tensors (same objects, not written to since)?  `owner_g_idx`: the owner's g_idx buffer as it is now --
        the call was built for the owner's plan at that time (`tag`), which a new or rewritten g_idx invalidates."""
        k, v = self.keep, self.versions
        return (k[0] is qweight and k[1] is scales and k[2] is qzeros and k[3] is bias and qweight._version == v[0]
                and scales._version == v[1] and qzeros._version == v[2] and (bias is None or bias._version == v[3])
                and self.tag[0] is owner_g_idx and (owner_g_idx is None or owner_g_idx._version == self.tag[1]))

    def __call__(self, x2d):
        """x2d: contiguous [M, K] of the call's dtype on the call's device (the owner checks)."""
        M = x2d.shape[0]
        y = torch.empty((M, self.N), dtype=self.dtype, device=self.dev)
        need = self.need.get(M)
        if need is None:
            need = self.need[M] = lib.inc_woq_gemm_workspace_bytes(M, self.N, self.K)
        idx = self.dev_index
        stream = _raw_stream(idx)
        wp, wn = None, 0
        if need > 0:
            buf = _ws_cache.get((idx, stream))
            if buf is None or buf.numel() < need:
                buf = _workspace(self.dev, need)
            wp, wn = buf.data_ptr(), buf.numel()
        if _cur_device() == idx:
            rc = self.fn(x2d.data_ptr(), self.dt, self.qw, self.sc, self.qz, self.gi, self.bi, y.data_ptr(), M, self.N, self.K, self.G,
                         self.gs, self.bits, wp, wn, stream)
        else:
            with torch.cuda.device(self.dev):
                rc = self.fn(x2d.data_ptr(), self.dt, self.qw, self.sc, self.qz, self.gi, self.bi, y.data_ptr(), M, self.N, self.K,
                             self.G, self.gs, self.bits, wp, wn, stream)
        if rc != 0:
            check(rc, "inc_woq_gemm")
        return y


class WoqGemmGroupCall:
    """inc_woq_gemm_multi for modules that multiply the SAME activation (q / k / v; gate / up): ONE launch instead of one per module
    (a decode call of one module is ~2 us of weight streaming behind ~5 us of launch boundary and hand-off).  The state dict is
    untouched: the call holds the modules' packed buffers by reference and hands the library host arrays of their addresses.
    `parts` = [(qweight, scales, qzeros, bias or None, N), ...]; K, group_size, bits common.  __call__(x2d) -> [y_i [M, N_i]], or None
    when the library declines the batch (nothing launched: the owner then calls the modules one by one)."""

    def __init__(self, parts, K, group_size, bits, dtype):
        import ctypes

        if dtype is not torch.bfloat16 and dtype is not torch.float16:
            raise TypeError("woq_gemm computes in bf16 or fp16")
        self.n = len(parts)
        self.dev = _dev(*[t for p in parts for t in p[:4]])
        self.dev_index = self.dev.index if self.dev.index is not None else torch.cuda.current_device()
        self.K, self.gs, self.bits, self.dtype = K, group_size, bits, dtype
        self.dt = INC_BF16 if dtype is torch.bfloat16 else INC_F16
        bias = [None if p[3] is None else (p[3] if p[3].dtype == dtype else p[3].to(dtype)) for p in parts]
        self.keep = (parts, bias)
        self.versions = tuple(None if t is None else t._version for p in parts for t in p[:4])
        self.Ns = [int(p[4]) for p in parts]
        n = self.n
        self.qw = (ctypes.c_void_p * n)(*[p[0].data_ptr() for p in parts])
        self.sc = (ctypes.c_void_p * n)(*[p[1].data_ptr() for p in parts])
        self.qz = (ctypes.c_void_p * n)(*[p[2].data_ptr() for p in parts])
        self.bi = (ctypes.c_void_p * n)(*[_ptr(b) for b in bias]) if any(b is not None for b in bias) else None
        self.Narr = (ctypes.c_int64 * n)(*self.Ns)
        self.yarr = (ctypes.c_void_p * n)()
        self.need = {}

    # a cache, not state: copies and pickles of the owning module start without it
    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (_none, ())

    def current(self, parts):
        mine = self.keep[0]
        return (len(parts) == len(mine) and all(a is b for p, q in zip(parts, mine) for a, b in zip(p[:4], q[:4]))
                and self.versions == tuple(None if t is None else t._version for p in parts for t in p[:4]))

    def __call__(self, x2d):
        M = x2d.shape[0]
        need = self.need.get(M)
        if need is None:
            need = self.need[M] = lib.inc_woq_gemm_multi_workspace_bytes(self.n, M, self.Narr, self.K)
        ys = [torch.empty((M, N), dtype=self.dtype, device=self.dev) for N in self.Ns]
        for i, y in enumerate(ys):
            self.yarr[i] = y.data_ptr()
        idx = self.dev_index
        stream = _raw_stream(idx)
        buf = _ws_cache.get((idx, stream))
        if buf is None or buf.numel() < need:
            buf = _workspace(self.dev, need)
        with torch.cuda.device(self.dev):
            rc = lib.inc_woq_gemm_multi(self.n, x2d.data_ptr(), self.dt, self.qw, self.sc, self.qz, self.bi, self.yarr, M, self.Narr, self.K,
                                        self.gs, self.bits, buf.data_ptr(), buf.numel(), stream)
        if rc == -2:  # INC_ERR_UNSUPPORTED: nothing was launched
            return None
        check(rc, "inc_woq_gemm_multi")
        return ys


# ---------------------------------------------------------------------------------------------------
# K7 group-wise RTN
# ---------------------------------------------------------------------------------------------------
def groupwise_quant(w, bits, group_size, scheme, quantile=1.0, full_range=False, return_int=False, inplace=True):
    """== quant_tensor (utility.py:272-436) for dtype "int".

    return_int=False: fake-quantises `w` (in place when `inplace`) and returns it.
    return_int=True : returns (int_weight int32 [N,K], scale fp32 [N,G], zp fp32 [N,G] or None); `w` is untouched.
    """
    dev = _dev(w)
    N, K = w.shape
    gs = K if (group_size == -1 or K < group_size) else group_size
    G = (K + gs - 1) // gs
    sym = scheme == "sym"
    scale = torch.empty((N, G), dtype=torch.float32, device=dev)
    zp = None if sym else torch.empty((N, G), dtype=torch.float32, device=dev)
    if return_int:
        iw = torch.empty((N, K), dtype=torch.int32, device=dev)
        qdq = None
    else:
        iw = None
        qdq = w if inplace else torch.empty_like(w)
    with torch.cuda.device(dev):
        check(
            lib.inc_groupwise_quant(
                _ptr(w), dtype_code(w.dtype), _ptr(qdq), _ptr(iw), _ptr(scale), _ptr(zp), N, K, gs, bits,
                INC_SCHEME_SYM if sym else INC_SCHEME_ASYM, float(quantile), int(bool(full_range)), _stream(),
            ),
            "inc_groupwise_quant",
        )
    if return_int:
        return iw, scale, zp
    return qdq


def codebook_quant(w, values, codes, group_size, quantile=1.0, return_int=False, inplace=True, scale=None):
    """== quantize_4bit through quant_tensor (utility.py:112-149, 246-265): NF4 / FP4 code-book quantisation per row group.

    values / codes: the ascending code book and the integers stored for its entries (FLOAT_MAPPING / INT_MAPPING).
    return_int=False: fake-quantises `w` (in place when `inplace`) and returns it; return_int=True: (int32 codes [N,K],
    scale [N,G] fp32, None) -- there is no zero point in these formats.  `scale` [N,G] (or broadcastable to it): the caller's
    scales, used instead of the rows' own max (quantize_4bit(..., scale=...), utility.py:127-128)."""
    import ctypes

    dev = _dev(w)
    assert w.dim() == 2
    N, K = w.shape
    gs = K if (group_size == -1 or K < group_size) else int(group_size)
    G = -(-K // gs)
    n = len(values)
    vals = (ctypes.c_float * n)(*[float(v) for v in values])
    cds = (ctypes.c_int32 * n)(*[int(c) for c in codes])
    scale_in = None
    if scale is not None:
        # anything the reference's `tensor.div_(scale)` broadcasts against the grouped weight [N * G, group_size] (utility.py:127-128): a
        # per-tensor scalar (0-dim or one element), one value per row [N] / [N,1], or the full [N,G] table.  The kernel divides in fp32
        # (the reference divides in the promoted dtype of weight and scale; equal for fp32 scales, one rounding apart for 16-bit ones)
        sc32 = scale.to(device=dev, dtype=torch.float32)
        if sc32.numel() == 1:
            sc32 = sc32.reshape(1, 1)
        elif sc32.dim() < 2:
            sc32 = sc32.reshape(-1, 1)
        else:
            sc32 = sc32.reshape(sc32.shape[0], -1)
        scale_in = torch.broadcast_to(sc32, (N, G)).contiguous()
    scale = torch.empty((N, G), dtype=torch.float32, device=dev)
    if return_int:
        iout = torch.empty((N, K), dtype=torch.int32, device=dev)
        qdq = None
    else:
        iout = None
        qdq = w if inplace else torch.empty_like(w)
    with torch.cuda.device(dev):
        check(lib.inc_codebook_quant_with_scale(_ptr(w), dtype_code(w.dtype), _ptr(qdq), _ptr(iout), _ptr(scale), N, K, gs, vals, cds, n,
                                                float(quantile), _ptr(scale_in), _stream()), "inc_codebook_quant_with_scale")
    if return_int:
        return iout, scale, None
    return qdq


_MSE_WS = {}


def mse_accumulate(a, b, out=None):
    """out += sum((a-b)^2): fp64 scalar tensor on device, fixed summation order (same input -> same bits)."""
    dev = _dev(a, b)
    assert a.dtype == b.dtype and a.numel() == b.numel()
    if out is None:
        out = torch.zeros(1, dtype=torch.float64, device=dev)
    assert out.dtype == torch.float64
    ws = _MSE_WS.get(dev)
    if ws is None:  # per-workgroup partials; calls on one device are stream-ordered, so one buffer per device
        ws = _MSE_WS[dev] = torch.empty(int(lib.inc_mse_accumulate_workspace_bytes()) // 8, dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        check(lib.inc_mse_accumulate(_ptr(a), _ptr(b), dtype_code(a.dtype), a.numel(), _ptr(out), _ptr(ws), _stream()), "inc_mse_accumulate")
    return out


# ---------------------------------------------------------------------------------------------------
# K5/K6 GPTQ
# ---------------------------------------------------------------------------------------------------
def gptq_hessian_accum(H, x2d, beta, alpha):
    """H <- beta*H + alpha * x^T x on the upper-triangular tiles (== GPTQ.add_batch, gptq.py:1111-1141)."""
    dev = _dev(H)
    if x2d.device.type != "cuda":
        raise RuntimeError("calibration activations must be resident in HBM")
    assert x2d.dim() == 2 and x2d.stride(1) == 1, "x must be [T,K] with unit inner stride"
    T, K = x2d.shape
    assert H.shape == (K, K) and H.dtype == torch.float32
    with torch.cuda.device(dev):
        check(
            lib.inc_gptq_hessian_accum(
                x2d.data_ptr(), dtype_code(x2d.dtype), T, K, x2d.stride(0), _ptr(H), float(beta), float(alpha), _stream()
            ),
            "inc_gptq_hessian_accum",
        )
    return H


# False: every tile of the batched Hessian launch is one workgroup (the last round of a launch then runs on a fraction of the CUs);
# a module attribute for A/B runs, not an environment switch
HESSIAN_TAIL_SPLIT = True
_hessian_ws_cache = {}


def _hessian_workspace(dev):
    """Scratch of the batched Hessian launch's split tail: one per (device, stream), never shared by concurrent launches."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    buf = _hessian_ws_cache.get(key)
    if buf is None:
        buf = _hessian_ws_cache[key] = torch.empty(lib.inc_gptq_hessian_accum_multi_workspace_bytes(), dtype=torch.uint8, device=dev)
    return buf


def gptq_hessian_accum_multi(items):
    """One launch for several Hessians of the same forward: items = [(H [K,K] fp32, x2d [T,K] 16-bit, beta, alpha), ...],
    all x2d with the same dtype and token count.  Returns False (nothing launched) when the library declines the batch
    (fp32 inputs, K < 256, more than 8 problems): the caller then uses gptq_hessian_accum per item."""
    import ctypes

    n = len(items)
    x0 = items[0][1]
    if n > 8 or x0.dtype not in (torch.bfloat16, torch.float16):
        return False
    dev = _dev(*[t for H, x, _, _ in items for t in (H, x)])
    T = x0.shape[0]
    for H, x, _, _ in items:
        assert H.dtype == torch.float32 and x.dim() == 2 and H.shape == (x.shape[1], x.shape[1])
        if x.dtype != x0.dtype or x.shape[0] != T:
            return False
    xs = (ctypes.c_void_p * n)(*[x.data_ptr() for _, x, _, _ in items])
    Hs = (ctypes.c_void_p * n)(*[H.data_ptr() for H, _, _, _ in items])
    Ks = (ctypes.c_int64 * n)(*[x.shape[1] for _, x, _, _ in items])
    ld = (ctypes.c_int64 * n)(*[x.stride(0) for _, x, _, _ in items])
    be = (ctypes.c_float * n)(*[float(b) for _, _, b, _ in items])
    al = (ctypes.c_float * n)(*[float(a) for _, _, _, a in items])
    ws = _hessian_workspace(dev) if HESSIAN_TAIL_SPLIT else None
    with torch.cuda.device(dev):
        rc = lib.inc_gptq_hessian_accum_multi(n, xs, dtype_code(x0.dtype), T, Ks, ld, Hs, be, al, _ptr(ws), 0 if ws is None else ws.numel(),
                                              _stream())
    if rc == -2:  # INC_ERR_UNSUPPORTED: nothing was launched
        return False
    check(rc, "inc_gptq_hessian_accum_multi")
    return True


def gptq_hessian_finalize(H, percdamp):
    """mirror + dead-column fix + damping (gptq.py:1186-1189, 1221-1227). Returns the uint8 dead mask [K]."""
    dev = _dev(H)
    K = H.shape[0]
    dead = torch.empty(K, dtype=torch.uint8, device=dev)
    ws = torch.empty(4, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(lib.inc_gptq_hessian_finalize(_ptr(H), K, float(percdamp), _ptr(dead), _ptr(ws), _stream()), "inc_gptq_hessian_finalize")
    return dead


def gptq_prepare_weight(w, dead=None):
    """W.float() with dead columns zeroed (gptq.py:1176, 1189)."""
    w = w.contiguous()
    dev = _dev(w, dead)
    N, K = w.shape
    out = torch.empty((N, K), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(lib.inc_gptq_prepare_weight(_ptr(w), dtype_code(w.dtype), _ptr(out), _ptr(dead), N, K, _stream()), "inc_gptq_prepare_weight")
    return out


def gptq_find_params(w32, col0, group_size, ngroups, bits, sym, scale, zero, g0, mse=False, grid=100, maxshrink=0.8, norm=2.4):
    """Quantizer.find_params(weight=True) for `ngroups` groups starting at column `col0` (gptq.py:1501-1624); `mse`
    adds the shrink-grid search of :1567-1584."""
    dev = _dev(w32, scale, zero)
    N, K = w32.shape
    G = scale.shape[1]
    with torch.cuda.device(dev):
        if mse:
            check(
                lib.inc_gptq_find_params_mse(_ptr(w32), N, K, col0, group_size, ngroups, bits, int(bool(sym)), int(grid),
                                             float(maxshrink), float(norm), _ptr(scale), _ptr(zero), G, g0, _stream()),
                "inc_gptq_find_params_mse",
            )
        else:
            check(
                lib.inc_gptq_find_params(_ptr(w32), N, K, col0, group_size, ngroups, bits, int(bool(sym)), _ptr(scale), _ptr(zero), G, g0, _stream()),
                "inc_gptq_find_params",
            )


def gptq_quant_block(w32, hinv, scale, zero, codes, q_out, err, i1, count, group_size, bits):
    dev = _dev(w32, hinv, scale, zero, codes, q_out, err)
    N, K = w32.shape
    G = scale.shape[1]
    with torch.cuda.device(dev):
        check(
            lib.inc_gptq_quant_block(
                _ptr(w32), _ptr(hinv), _ptr(scale), _ptr(zero), _ptr(codes), _ptr(q_out),
                dtype_code(q_out.dtype) if q_out is not None else INC_F32, _ptr(err), N, K, G, i1, count,
                group_size, bits, _stream(),
            ),
            "inc_gptq_quant_block",
        )


def gptq_quant_block_params(w32, hinv, scale, zero, codes, q_out, err, i1, count, group_size, bits, sym):
    """`gptq_quant_block` that first computes the block's own group parameters (find_params fused into the launch; see
    include/inc_mi355x.h).  Returns False when the library has no such form for this block."""
    dev = _dev(w32, hinv, scale, zero, codes, q_out, err)
    N, K = w32.shape
    G = scale.shape[1]
    with torch.cuda.device(dev):
        rc = lib.inc_gptq_quant_block_params(
            _ptr(w32), _ptr(hinv), _ptr(scale), _ptr(zero), _ptr(codes), _ptr(q_out),
            dtype_code(q_out.dtype) if q_out is not None else INC_F32, _ptr(err), N, K, G, i1, count, group_size, bits,
            1 if sym else 0, _stream())
    if rc == -2:  # INC_ERR_UNSUPPORTED
        return False
    check(rc, "inc_gptq_quant_block_params")
    return True


def gptq_lazy_update(w32, hinv, err, i1, count):
    dev = _dev(w32, hinv, err)
    N, K = w32.shape
    with torch.cuda.device(dev):
        check(lib.inc_gptq_lazy_update(_ptr(w32), _ptr(hinv), _ptr(err), N, K, i1, count, _stream()), "inc_gptq_lazy_update")


def gptq_lazy_update_cols(w32, hinv, err, i1, count, col_begin, col_end):
    """`gptq_lazy_update` for the trailing columns [col_begin, col_end) only (see include/inc_mi355x.h); launches on the
    CURRENT stream.  Returns False when the library has no column-range form for this block shape."""
    dev = _dev(w32, hinv, err)
    N, K = w32.shape
    with torch.cuda.device(dev):
        rc = lib.inc_gptq_lazy_update_cols(_ptr(w32), _ptr(hinv), _ptr(err), N, K, i1, count, col_begin, col_end, _stream())
    if rc == -2:  # INC_ERR_UNSUPPORTED
        return False
    check(rc, "inc_gptq_lazy_update_cols")
    return True


def probe_hbm_triad(a, b, c, s):
    """a <- b + s * c (fp32, same shape): the stream triad behind bench.py's `ceilings` (include/inc_mi355x.h)."""
    dev = _dev(a, b, c)
    with torch.cuda.device(dev):
        check(lib.inc_probe_hbm_triad(_ptr(a), _ptr(b), _ptr(c), float(s), a.numel(), _stream()), "inc_probe_hbm_triad")


def probe_hbm_copy(dst, src, variant=0):
    """dst <- src (same byte count, 16-byte aligned): the chip's copy ceiling for bench.py; `variant` see inc_probe_hbm_copy."""
    dev = _dev(dst, src)
    nbytes = src.numel() * src.element_size()
    assert nbytes == dst.numel() * dst.element_size() and nbytes % 16 == 0
    with torch.cuda.device(dev):
        check(lib.inc_probe_hbm_copy(_ptr(dst), _ptr(src), nbytes, int(variant), _stream()), "inc_probe_hbm_copy")
    return dst


def probe_mfma_bf16(src, sink, blocks, iters):
    """Bare bf16 MFMA loop (bench.py `ceilings`); returns the flops of the launch."""
    import ctypes

    dev = _dev(src, sink)
    assert src.numel() * src.element_size() >= 65536 and sink.numel() >= blocks * 256
    flops = ctypes.c_double(0.0)
    with torch.cuda.device(dev):
        check(lib.inc_probe_mfma_bf16(_ptr(src), _ptr(sink), int(blocks), int(iters), ctypes.byref(flops), _stream()), "inc_probe_mfma_bf16")
    return flops.value


_HIP_RT = None


def cu_masked_stream(device, mask_words):
    """A HIP stream restricted to the compute units whose bits are set (hipExtStreamCreateWithCUMask), as a torch stream; None when
    the runtime refuses.  On the MI355X the 256 bits are 8 words of 32: word w = CU number w of every shader engine (measured with
    inc_probe_mfma_bf16, scripts/cumask_probe.py: seven words set = 224 CUs, one CU per shader engine left free)."""
    global _HIP_RT
    import ctypes

    try:
        if _HIP_RT is None:
            _HIP_RT = ctypes.CDLL("libamdhip64.so")
        st = ctypes.c_void_p()
        arr = (ctypes.c_uint32 * len(mask_words))(*mask_words)
        with torch.cuda.device(device):
            if _HIP_RT.hipExtStreamCreateWithCUMask(ctypes.byref(st), len(mask_words), arr) != 0 or not st.value:
                return None
            return torch.cuda.ExternalStream(st.value, device=device)
    except Exception:  # pragma: no cover - a runtime without the extension
        return None


def trace_marker(marker_id, device=None):
    """Phase boundary for kernel-trace timelines: an empty launch with Grid_Size_X = 64 * marker_id on the current stream."""
    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        check(lib.inc_trace_marker(int(marker_id), _stream()), "inc_trace_marker")


GPTQ_DYNAMIC_GROUPS, GPTQ_MSE, GPTQ_NO_LOOKAHEAD, GPTQ_NO_FUSED_PARAMS = 1, 2, 4, 8


def gptq_quantize_layer(w32, hinv, scale, zero, loop_scale, loop_zero, codes, q_out, err_ws, group_size, kernel_group_size, block_size,
                        bits, sym, flags, aux_stream=None):
    """The whole blocked column loop of GPTQ.fasterquant (gptq.py:1250-1304) as ONE C-ABI call (include/inc_mi355x.h:
    inc_gptq_quantize_layer): issued on the CURRENT stream, the bulk of the lazy updates on `aux_stream` (a torch stream) when
    given.  `err_ws` is fp32 [2, N, 128]; `loop_scale` / `loop_zero` may be None (= scale / zero)."""
    dev = _dev(w32, hinv, scale, zero, codes, q_out, err_ws)
    N, K = w32.shape
    assert err_ws.dtype == torch.float32 and err_ws.numel() >= 2 * N * 128
    with torch.cuda.device(dev):
        check(
            lib.inc_gptq_quantize_layer(
                _ptr(w32), _ptr(hinv), _ptr(scale), _ptr(zero), scale.shape[1], _ptr(loop_scale), _ptr(loop_zero),
                loop_scale.shape[1] if loop_scale is not None else scale.shape[1], _ptr(codes), _ptr(q_out),
                dtype_code(q_out.dtype) if q_out is not None else INC_F32, _ptr(err_ws), N, K, int(group_size), int(kernel_group_size),
                int(block_size), int(bits), 1 if sym else 0, int(flags), _stream(),
                aux_stream.cuda_stream if aux_stream is not None else None,
            ),
            "inc_gptq_quantize_layer",
        )


def chol_diag_block(A_view, Linv_view, info, tag):
    """In-place Cholesky of one <=128x128 diagonal block (a strided view into a larger fp32 matrix) + inverse of its
    factor into `Linv_view` (also a strided view).  See include/inc_mi355x.h: inc_chol_diag_block."""
    dev = A_view.device
    if dev.type != "cuda" or Linv_view.device != dev:
        raise RuntimeError("chol_diag_block needs HBM-resident views")
    n = A_view.shape[0]
    assert A_view.shape == (n, n) and Linv_view.shape == (n, n) and A_view.stride(1) == 1 and Linv_view.stride(1) == 1
    assert A_view.dtype == torch.float32 and Linv_view.dtype == torch.float32
    with torch.cuda.device(dev):
        check(
            lib.inc_chol_diag_block(A_view.data_ptr(), A_view.stride(0), n, Linv_view.data_ptr(), Linv_view.stride(0),
                                    _ptr(info), int(tag), _stream()),
            "inc_chol_diag_block",
        )


def gptq_inverse_factor(H, aux_stream=None, flags=0):
    """Upper Cholesky factor U of H^-1 for the damped SPD fp32 Hessian H [K,K] (gptq.py:1228-1231) as ONE C-ABI call
    (include/inc_mi355x.h: inc_gptq_inverse_factor).  Returns (U, info): `info` is a device int32 that stays 0 unless H
    is not positive definite (no host synchronisation here)."""
    dev = _dev(H)
    K = H.shape[0]
    assert H.shape == (K, K) and H.dtype == torch.float32 and H.is_contiguous()
    U = torch.empty((K, K), dtype=torch.float32, device=dev)
    info = torch.empty(1, dtype=torch.int32, device=dev)
    wsb = int(lib.inc_gptq_inverse_factor_workspace_bytes(K, int(flags)))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    # (no record_stream for `aux_stream`: the call returns with the current stream ordered behind everything it issued on the second
    # one, so the caching allocator's stream-ordered reuse of H / U / the workspace is already safe -- and record_stream would make
    # every call hipMalloc a fresh workspace: 2.2 GB at K = 11008, 14.8 GB at K = 28672 with the split products' planes; the
    # caching allocator keeps one per stream that factorises, so the driver's concurrent factor streams each hold theirs)
    aux = aux_stream.cuda_stream if aux_stream is not None else None
    with torch.cuda.device(dev):
        check(lib.inc_gptq_inverse_factor(_ptr(H), K, _ptr(U), _ptr(ws), wsb, _ptr(info), int(flags), _stream(), aux), "inc_gptq_inverse_factor")
    return U, info


# ---------------------------------------------------------------------------------------------------
# K8 AWQ statistics
# ---------------------------------------------------------------------------------------------------
def awq_act_abs_sum(x2d, out):
    dev = _dev(x2d, out)
    T, K = x2d.shape
    with torch.cuda.device(dev):
        check(lib.inc_awq_act_abs_sum(_ptr(x2d), dtype_code(x2d.dtype), T, K, _ptr(out), _stream()), "inc_awq_act_abs_sum")
    return out


def awq_weight_scale(w, group_size):
    """== _get_weight_scale (awq.py:131-147): mean over rows of |w|/groupmax|w| -> [K] in w.dtype."""
    w = w.contiguous()
    dev = _dev(w)
    N, K = w.shape
    out = torch.zeros(K, dtype=torch.float32, device=dev)
    nbytes = lib.inc_awq_weight_scale_workspace_bytes(N, K, group_size)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(
            lib.inc_awq_weight_scale(_ptr(w), dtype_code(w.dtype), N, K, group_size, _ptr(out), _ptr(ws), nbytes, _stream()),
            "inc_awq_weight_scale",
        )
    return (out / N).to(w.dtype)


# ---------------------------------------------------------------------------------------------------
# K10-K14 SmoothQuant W8A8 (reference neural_compressor/torch/algorithms/smooth_quant/utility.py)
# ---------------------------------------------------------------------------------------------------
FLT_MAX = 3.4028234663852886e38


def sq_new_minmax(K, device):
    """(min, max) running buffers for sq_channel_minmax."""
    return (torch.full((K,), FLT_MAX, dtype=torch.float32, device=device),
            torch.full((K,), -FLT_MAX, dtype=torch.float32, device=device))


def sq_channel_minmax(x2d, mn, mx):
    """== Calibration._save_input_pc_hook (utility.py:858-883): running per-channel min / max of x [T, K]."""
    dev = _dev(x2d, mn, mx)
    if x2d.stride(-1) != 1:
        x2d = x2d.contiguous()
    T, K = x2d.shape
    with torch.cuda.device(dev):
        check(lib.inc_sq_channel_minmax(_ptr(x2d), dtype_code(x2d.dtype), T, K, x2d.stride(0), _ptr(mn), _ptr(mx), _stream()),
              "inc_sq_channel_minmax")
    return mn, mx


def sq_weight_col_absmax(w, out=None):
    """max_n |w[n,k]| accumulated into `out` (fp32 [K], zero-initialised): the weight side of cal_scale (:617-618)."""
    w = w.contiguous()
    dev = _dev(w, out)
    N, K = w.shape
    if out is None:
        out = torch.zeros(K, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(lib.inc_sq_weight_col_absmax(_ptr(w), dtype_code(w.dtype), N, K, _ptr(out), _stream()), "inc_sq_weight_col_absmax")
    return out


def sq_cal_scale(amax_x, amax_w, alpha, weight_max_lb=1e-5):
    """== cal_scale (utility.py:605-626) from the two abs-max vectors."""
    amax_x = amax_x.to(torch.float32).contiguous()
    amax_w = amax_w.to(torch.float32).contiguous()
    dev = _dev(amax_x, amax_w)
    K = amax_x.numel()
    scale = torch.empty(K, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(lib.inc_sq_cal_scale(_ptr(amax_x), _ptr(amax_w), K, float(alpha), float(weight_max_lb), _ptr(scale), _stream()),
              "inc_sq_cal_scale")
    return scale


def sq_quant_weight(w, smooth=None, Kp=None):
    """== quant_dequant_w_v1 (utility.py:652-695, Linear, sym int8) of w * smooth -> (qw int8 [N,Kp], scale [N], rowsum [N])."""
    w = w.contiguous()
    smooth = None if smooth is None else smooth.to(torch.float32).contiguous()
    dev = _dev(w, smooth)
    N, K = w.shape
    Kp = K if Kp is None else int(Kp)
    qw = torch.empty((N, Kp), dtype=torch.int8, device=dev)
    w_scale = torch.empty(N, dtype=torch.float32, device=dev)
    rowsum = torch.empty(N, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(lib.inc_sq_quant_weight(_ptr(w), dtype_code(w.dtype), N, K, Kp, _ptr(smooth), _ptr(qw), _ptr(w_scale), _ptr(rowsum),
                                      _stream()), "inc_sq_quant_weight")
    return qw, w_scale, rowsum


def sq_quant_act(x2d, in_scale, sx, zp, Kp=None):
    """clamp(rint(x * in_scale / sx + zp), 0, 255) - 128 as int8 [M, Kp] (SQLinearWrapper.forward + quant_dequant_x_v1)."""
    x2d = x2d.contiguous()
    dev = _dev(x2d, in_scale)
    M, K = x2d.shape
    Kp = K if Kp is None else int(Kp)
    out = torch.empty((M, Kp), dtype=torch.int8, device=dev)
    with torch.cuda.device(dev):
        check(lib.inc_sq_quant_act(_ptr(x2d), dtype_code(x2d.dtype), M, K, Kp, _ptr(in_scale), float(sx), float(zp), _ptr(out),
                                   _stream()), "inc_sq_quant_act")
    return out


def w8a8_gemm(xq, wq, alpha, corr, bias, out_dtype):
    """y = alpha[n] * (xq @ wq^T + corr[n]) + bias[n], int32 accumulate on the matrix cores (K14)."""
    dev = _dev(xq, wq, alpha, corr, bias)
    M, K = xq.shape
    N = wq.shape[0]
    assert wq.shape[1] == K and xq.dtype == torch.int8 and wq.dtype == torch.int8
    if bias is not None and bias.dtype != out_dtype:
        bias = bias.to(out_dtype)
    y = torch.empty((M, N), dtype=out_dtype, device=dev)
    nbytes = lib.inc_w8a8_gemm_workspace_bytes(M, N, K)
    ws = _workspace_i8(dev, nbytes) if nbytes > 0 else None
    with torch.cuda.device(dev):
        check(lib.inc_w8a8_gemm(_ptr(xq), _ptr(wq), _ptr(alpha), _ptr(corr), _ptr(bias), _ptr(y), dtype_code(out_dtype), M, N, K,
                                _ptr(ws), 0 if ws is None else ws.numel(), _stream()), "inc_w8a8_gemm")
    return y


_ws_i8_cache = {}


def _workspace_i8(dev, nbytes):
    """Zero-filled on creation (arrival tickets), one per (device, stream); separate from the 4-bit GEMM's workspace
    because both kernels keep their tickets in the first 16 KiB."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    buf = _ws_i8_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.zeros(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
        _ws_i8_cache[key] = buf
    return buf
