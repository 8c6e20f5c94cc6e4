"""The weight-only Linear operator for MI355X.

Drop-in for the reference's operator classes (neural_compressor/torch/algorithms/weight_only/modules.py):
  WeightOnlyLinear (abstract)        :91-154
  INCWeightOnlyLinear                :157-627   -> MI355XWeightOnlyLinear (alias INCWeightOnlyLinear)
  UnpackedWeightOnlyLinearParams     :74-88
  MulLinear                          :907-949

Same constructor, same buffers (`qweight`, `scales`, `qzeros`, `bias`, `g_idx`, `scale_bf16_to_fp8`) with the
same shapes / dtypes, so state_dicts round-trip with the reference and with HF / AutoGPTQ checkpoints.  What
changes is where the arithmetic runs: pack / unpack / recover are HIP kernels over HBM-resident tensors
(no per-nibble device syncs, no Python loop over K), and forward is a fused INT4->bf16 dequant-GEMM on the
matrix cores instead of "recover once, cache the dense weight, F.linear".  There is no CPU path: tensors
must live on a HIP device.
"""

import math
from abc import abstractmethod

import torch

from .... import ops
from ....common.utils import logger


class UnpackedWeightOnlyLinearParams(dict):
    """Unpacked tensors of a packed module: int_weight, scales, scale_bf16_to_fp8, zp, g_idx, bias."""

    def __init__(self, unpack_weight, scales, scale_bf16_to_fp8, unpack_zp, **kwargs):
        super().__init__(int_weight=unpack_weight, scales=scales, scale_bf16_to_fp8=scale_bf16_to_fp8, zp=unpack_zp, **kwargs)

    def to(self, device):
        for key, value in self.items():
            if isinstance(value, torch.Tensor):
                self[key] = value.to(device)
        return self


class WeightOnlyLinear(torch.nn.Module):
    """Abstract operator: a device class implements pack / unpack / forward (reference modules.py:91)."""

    def __init__(self, in_features, out_features, dtype, bits, group_size, device, scale_dtype, **kwargs):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.dtype = dtype
        self.bits = bits
        self.group_size = group_size if group_size != -1 else in_features
        self.device = device
        self.scale_dtype = scale_dtype
        self.kwargs = kwargs

    @abstractmethod
    def pack(self, *args, **kwargs):
        raise NotImplementedError(f"{self.__class__.__name__} doesn't implement `pack` function. ")

    @abstractmethod
    def unpack(self, *args, **kwargs):
        raise NotImplementedError(f"{self.__class__.__name__} doesn't implement `unpack` function. ")

    @abstractmethod
    def forward(self, input):
        raise NotImplementedError(f"{self.__class__.__name__} doesn't implement `forward` function. ")

    def extra_repr(self):
        return "in_features={}, out_features={}, bits={}, group_size={}, bias={}".format(
            self.in_features, self.out_features, self.bits, self.group_size, self.bias is not None
        )


_CBITS = {torch.int8: 8, torch.int16: 16, torch.int32: 32, torch.int64: 64}


def _hip_device(device):
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError(
            f"MI355XWeightOnlyLinear needs a HIP device ('cuda[:N]'), got '{device}'. There is no CPU implementation; "
            "construct the module with device='cuda'."
        )
    return dev


class MI355XWeightOnlyLinear(WeightOnlyLinear):
    """Packed-weight Linear whose pack / unpack / recover / forward are HIP kernels (gfx950)."""

    def __init__(
        self,
        in_features,
        out_features,
        dtype="int",
        bits=4,
        group_size=32,
        zp=False,
        bias=False,
        scale_dtype=torch.float32,
        compression_dtype=torch.int32,
        compression_dim=1,
        g_idx=False,
        device="cuda",
        use_optimum_format=True,
        **kwargs,
    ):
        super().__init__(in_features, out_features, dtype, bits, group_size, device, scale_dtype=scale_dtype, **kwargs)
        if not (isinstance(bits, int) and 1 <= bits <= 8):  # the widths the reference's configs tune (config.py:211)
            raise ValueError(f"bits={bits}: weight-only widths are 1..8 (n_pack = compress_bits // bits, reference modules.py:231)")
        dev = _hip_device(device)
        self.use_optimum_format = use_optimum_format
        self._lut = None
        if "int" not in self.dtype:  # nf4 / fp4 code books (reference modules.py:213-221)
            from .utility import FLOAT_MAPPING, INT_MAPPING

            if self.dtype not in FLOAT_MAPPING:
                raise NotImplementedError(f"dtype={dtype}: integer, NF4 and FP4 weight-only formats are implemented")
            assert bits == 4, "NF4 / FP4 are 4-bit formats"
            self.use_optimum_format = False  # optimum_format doesn't suit symmetric nf4 / fp4 (reference :216)
            # stored integer (-8..7) -> code-book value; unpack() applies it like the reference's int2float_mapping
            lut = torch.zeros(16, dtype=torch.float32)
            lut[torch.tensor(INT_MAPPING[self.dtype]) + 8] = torch.tensor(FLOAT_MAPPING[self.dtype], dtype=torch.float32)
            self._lut = lut.to(dev)
        self.compression_dim = compression_dim
        assert compression_dtype in _CBITS, f"Only support torch.int8|16|32|64 as compressed dtype. but got {compression_dtype}"
        assert compression_dim in (0, 1), "Only support 0 or 1 as compression dimension, 0 is output channel, 1 is input channel."
        K, N, gs = in_features, out_features, self.group_size
        G = math.ceil(K / gs)
        self.register_buffer("scale_bf16_to_fp8", torch.zeros(1, dtype=torch.bfloat16, device=dev))
        if self.use_optimum_format:
            self.float_type = torch.float16
            self.compression_dtype = torch.int32
            self.compress_bits = 32
            self.n_pack = 32 // bits
            self.register_buffer("scales", torch.zeros((G, N), dtype=self.float_type, device=dev))
            self.register_buffer("qweight", torch.zeros((math.ceil(K / self.n_pack), N), dtype=torch.int32, device=dev))
            self.register_buffer("qzeros", torch.zeros((G, math.ceil(N / self.n_pack)), dtype=torch.int32, device=dev))
            self.register_buffer("bias", torch.zeros(N, dtype=self.float_type, device=dev))
        else:
            self.compression_dtype = compression_dtype
            self.compress_bits = _CBITS[compression_dtype]
            self.n_pack = self.compress_bits // bits
            self.float_type = scale_dtype
            self.register_buffer("scales", torch.zeros((N, G), dtype=self.float_type, device=dev))
            if compression_dim == 1:
                self.register_buffer("qweight", torch.zeros((N, math.ceil(K / self.n_pack)), dtype=compression_dtype, device=dev))
                if zp:
                    self.register_buffer("qzeros", torch.zeros((N, math.ceil(G / self.n_pack)), dtype=compression_dtype, device=dev))
            else:
                self.register_buffer("qweight", torch.zeros((math.ceil(N / self.n_pack), K), dtype=compression_dtype, device=dev))
                if zp:
                    self.register_buffer("qzeros", torch.zeros((math.ceil(N / self.n_pack), G), dtype=compression_dtype, device=dev))
            if bias:
                self.register_buffer("bias", torch.zeros(N, dtype=self.float_type, device=dev))
            else:
                self.bias = None
        if g_idx:
            self.register_buffer("g_idx", torch.zeros(K, dtype=torch.int32, device=dev))
        else:
            self.g_idx = None

    # ------------------------------------------------------------------------------------------------
    def pack(self, int_weight, scales, zp, bias, scale_bf16_to_fp8=None, g_idx=None, **kwargs):
        """Pack integer weights (reference modules.py:321-375); arguments are NOT mutated (the reference
        adds the sym shift / subtracts 1 from zp in place)."""
        dev = self.qweight.device
        N, K = self.out_features, self.in_features
        self._plan_key = None  # the kernels write the buffers through raw pointers: drop the cached forward plan
        self.__dict__["_call"] = None  # ... and the prepared call (its converted bias / plan belong to the old contents)
        if bias is not None:
            assert hasattr(self, "bias"), "bias is not set when initializing."
            self.bias = bias.detach().to(dev).type(self.float_type)
        if g_idx is not None:
            assert hasattr(self, "g_idx"), "g_idx is not set when initializing."
            g = g_idx.to(dev).type(torch.int32)
            if self.use_optimum_format:
                g = (torch.argsort(g) // self.group_size).type(torch.int32)  # modules.py:341-344 (index plumbing)
            self.g_idx = g.contiguous()
        if scale_bf16_to_fp8 is not None:
            self.scale_bf16_to_fp8 = scale_bf16_to_fp8.to(dev).type(self.float_type)
        int_weight = int_weight.to(dev)
        scales = scales.to(dev)
        zp = None if zp is None else zp.to(dev)
        if self.use_optimum_format:
            assert tuple(scales.shape) == (N, self.scales.shape[0]), f"{scales.shape} Scale shape is mismatched."
            shift = 2 ** (self.bits - 1) if zp is None else 0
            ops.woq_pack(
                int_weight, scales, zp, self.bits, shift, qweight=self.qweight, qzeros=self.qzeros, scales_out=self.scales
            )
            return
        # non-optimum layouts (modules.py:270-314): generic row packer + layout transposes
        assert scales.shape == self.scales.shape, f"{scales.shape} != {self.scales.shape} Scale shape is mismatched."
        self.scales = scales.type(self.float_type).contiguous()
        iw = int_weight.to(torch.int32)
        if self.compression_dim == 0:
            iw = iw.T.contiguous()
        packed = ops.pack_rows(iw, self.bits, self.compress_bits)
        if self.compression_dim == 0:
            packed = packed.T.contiguous()
        assert packed.shape == self.qweight.shape, "output channels mismatch, please check."
        self.qweight.copy_(packed)
        if zp is not None:
            assert hasattr(self, "qzeros"), "zp is not set when initializing."
            z = zp.to(torch.int32)
            if self.compression_dim == 0:
                z = z.T.contiguous()
            pz = ops.pack_rows(z, self.bits, self.compress_bits)
            if self.compression_dim == 0:
                pz = pz.T.contiguous()
            self.qzeros.copy_(pz)

    def pack_codes(self, codes, scales, zp, bias, g_idx=None):
        """MI355X-native fast path (optimum format only): pack already-offset codes 0..2^bits-1 (uint8 [N,K], what
        the GPTQ column-loop kernel emits) instead of int32 signed ints.  Produces bit-identical buffers to
        `pack(codes - 2^(bits-1), scales, None, ...)` for sym and `pack(codes, scales, zp, ...)` for asym."""
        assert self.use_optimum_format, "pack_codes writes the optimum layout"
        dev = self.qweight.device
        self._plan_key = None
        self.__dict__["_call"] = None
        if bias is not None:
            self.bias = bias.detach().to(dev).type(self.float_type)
        if g_idx is not None:
            assert hasattr(self, "g_idx"), "g_idx is not set when initializing."
            g = g_idx.to(dev).type(torch.int32)
            self.g_idx = (torch.argsort(g) // self.group_size).type(torch.int32).contiguous()
        ops.woq_pack(codes, scales, zp, self.bits, 0, qweight=self.qweight, qzeros=self.qzeros, scales_out=self.scales)

    def unpack(self):
        """Reference modules.py:377-411 -> UnpackedWeightOnlyLinearParams (int16 ints, as the reference)."""
        N, K = self.out_features, self.in_features
        if self.use_optimum_format:
            G = self.scales.shape[0]
            if self.scales.dtype == torch.float16 and self.scales.is_contiguous():
                iw, zp, scales = ops.woq_unpack(self.qweight, self.qzeros, N, K, G, self.bits, scales=self.scales)  # one launch
            else:
                iw, zp = ops.woq_unpack(self.qweight, self.qzeros, N, K, G, self.bits)
                scales = self.scales.T.contiguous()
        else:
            has_zp = hasattr(self, "qzeros")
            qw = self.qweight if self.compression_dim == 1 else self.qweight.T.contiguous()
            iw = ops.unpack_rows(qw.contiguous(), self.bits, self.compress_bits, has_zp)
            if self.compression_dim == 0:
                iw = iw.T.contiguous()
            iw = iw[:N, :K].contiguous()
            if self._lut is not None:  # nf4 / fp4: the unpacked "weight" is the code-book value (reference :391-395)
                iw = self._lut.to(iw.device)[(iw.to(torch.int64) + 8)]
            zp = None
            if has_zp:
                qz = self.qzeros if self.compression_dim == 1 else self.qzeros.T.contiguous()
                zp = ops.unpack_rows(qz.contiguous(), self.bits, self.compress_bits, True)
                if self.compression_dim == 0:
                    zp = zp.T.contiguous()
                zp = zp[: self.scales.shape[0], : self.scales.shape[1]].contiguous()
            scales = self.scales
        return UnpackedWeightOnlyLinearParams(iw, scales, self.scale_bf16_to_fp8, zp, g_idx=self.g_idx, bias=self.bias)

    def recover(self, dtype=None):
        """Dense weight [N,K] (reference modules.py:413-443).  Default dtype = float_type (fp16 in optimum format)."""
        out_dtype = dtype or self.float_type
        if self.use_optimum_format:
            return ops.woq_dequant(
                self.qweight, self.scales, self.qzeros, self.g_idx, self.out_features, self.in_features,
                self.group_size, self.bits, out_dtype=out_dtype,
            )
        p = self.unpack()
        if self._lut is not None:  # code-book value x group scale (reference :438-441), zp is None for these formats
            N, K, gs = self.out_features, self.in_features, self.group_size
            gs = K if gs == -1 or gs > K else gs
            gi = (torch.arange(K, device=p["scales"].device) // gs) if self.g_idx is None else self.g_idx.long()
            return (p["int_weight"].float() * p["scales"].float()[:, gi]).to(out_dtype)
        return ops.dequant_ints(p["int_weight"], p["scales"], p["zp"], self.g_idx, self.group_size, out_dtype)

    def forward(self, input):
        """y = x W^T + b with W dequantised on the fly (reference modules.py:594-610).

        bf16 / fp16 inputs are computed in their own dtype (fp32 accumulate); any other dtype is cast to fp16 for the
        multiplication (what the reference does on an accelerator: `input.type(self.weight.dtype)` with an fp16 weight)
        and an fp32 input gets an fp32 output back, as on the reference's CPU path (:598-600), so an fp32 model keeps
        running after some of its layers are packed (true_sequential re-runs the block mid-way).
        """
        x = input
        # decode path: a prepared call for the plain fused plan (ops.WoqGemmCall); the 6-us kernel makes the host side count
        d = self.__dict__
        call = d.get("_call")
        if call is not None and x.dtype is call.dtype and x.device == call.dev and x.is_contiguous() and x.numel() > 0 and x.shape[-1] == call.K:
            bufs = self._buffers
            if call.current(bufs["qweight"], bufs["scales"], bufs["qzeros"], bufs.get("bias", d.get("bias")), bufs.get("g_idx", d.get("g_idx"))):
                y = call(x if x.dim() == 2 else x.view(-1, call.K))
                return y if x.dim() == 2 else y.view(*x.shape[:-1], call.N)
        if x.dtype not in (torch.bfloat16, torch.float16):
            x = x.to(torch.float16)
        lead = x.shape[:-1]
        x2d = x.reshape(-1, self.in_features)
        if x2d.shape[0] == 0:  # empty batch: nothing to launch (F.linear returns an empty tensor too)
            return x2d.new_empty((*lead, self.out_features), dtype=torch.float32 if input.dtype == torch.float32 else x2d.dtype)
        if not x2d.is_contiguous():
            x2d = x2d.contiguous()
        plan = self._forward_plan()
        d["_call"] = None
        if plan == "fused" and self._fused_max_m is not None and x2d.shape[0] > self._fused_max_m:
            plan = "dense"  # (a group size the fast kernels do not take: see _forward_plan)
        if plan == "fused" and self._fused_max_m is not None:
            y = ops.woq_gemm(x2d, self.qweight, self.scales, self.qzeros, self.bias, self.out_features, self.in_features, self.group_size, self.bits)
        elif plan == "fused":
            call = d["_call"] = ops.WoqGemmCall(self.qweight, self.scales, self.qzeros, self.bias, self.out_features, self.in_features,
                                                self.group_size, self.bits, x2d.dtype)
            gi = self.g_idx
            call.tag = (gi, None if gi is None else gi._version)
            y = call(x2d)
        elif plan == "fused_act_order":
            # act_order: the K axis is sorted by group once (below); per call only the activations are gathered
            y = ops.woq_gemm(
                x2d.index_select(1, self._k_order), self._qweight_sorted, self.scales, self.qzeros, self.bias,
                self.out_features, self.in_features, self.group_size, self.bits,
            )
        elif plan == "fused_g_idx":
            # an irregular g_idx (not a permutation of whole groups): inc_woq_gemm's general tile kernel reads scale / zero of
            # group g_idx[k] per element (reference modules.py:427-431 semantics), still without a dense weight
            y = ops.woq_gemm(x2d, self.qweight, self.scales, self.qzeros, self.bias, self.out_features, self.in_features,
                             self.group_size, self.bits, g_idx=self.g_idx)
        else:
            # non-optimum layouts (compression_dim / int8-16-64 containers, NF4 / FP4 code books): HIP dequant + dense library GEMM
            w = self.recover(dtype=x2d.dtype)
            b = None if self.bias is None else self.bias.to(x2d.dtype)
            y = torch.nn.functional.linear(x2d, w, b)
        if input.dtype == torch.float32:
            y = y.float()
        return y.reshape(*lead, self.out_features)

    # widths other than 4 / 8 bits: True = multiply through inc_woq_gemm's per-element tile form (no dense weight ever exists; 8-10 x
    # slower), False = HIP recover() into a transient dense weight + the library GEMM (what the reference's forward does on its CPU)
    ODD_WIDTH_FUSED = False

    def _forward_plan(self):
        """Pick the forward route once per packed state.  A per-element `g_idx` (GPTQ act_order, HF desc_act
        checkpoints; modules.py:341-344) that is a permutation of whole groups is handled by sorting the K axis by group:
        x W^T = x[:, p] W[:, p]^T, and in the sorted order the groups are contiguous again, so the fused kernel runs on a
        K-sorted copy of the packed words with only an activation gather per call."""
        # in-place re-packing / load_state_dict bump the tensors' version counters -> the plan is rebuilt
        key = (self.qweight.data_ptr(), self.qweight._version,
               None if self.g_idx is None else (self.g_idx.data_ptr(), self.g_idx._version), self.ODD_WIDTH_FUSED)
        if getattr(self, "_plan_key", None) == key:
            return self._plan
        plan = "dense"
        # 4 / 8 bits: whole words per group (the fast kernels); every other width 1..7 runs inc_woq_gemm's per-element tile form,
        # which takes any group size (n_pack = 10 / 6 / 5 for 3 / 5 / 6 bits never divides one)
        # (the per-element form is the memory-saving route, not the fast one: measured 376-480 us against 40-100 us for HIP recover() +
        # the library GEMM at 4096^2 for every M from 1 to 4096, profiles/r6/anyw_route_time.log -- so it is opt-in: ODD_WIDTH_FUSED)
        fusable = self.use_optimum_format and (self.group_size % self.n_pack == 0 if self.bits in (4, 8) else self.ODD_WIDTH_FUSED)
        self._k_order = self._qweight_sorted = None
        # group sizes that are neither a power of two >= 32 nor the whole row (e.g. 96) run inc_woq_gemm's general 128 x 128 tile kernel
        # above 16 rows: 190-245 us at 4096 x 4032 against 34-74 us for HIP recover() + the library GEMM (scripts/route_sweep.py) -- such
        # modules keep the fused form for decode-sized batches only
        gs_eff = self.in_features if (self.group_size == -1 or self.group_size >= self.in_features) else self.group_size
        fast_groups = gs_eff == self.in_features or (gs_eff >= 32 and (gs_eff & (gs_eff - 1)) == 0)
        self._fused_max_m = None if (fast_groups or self.bits not in (4, 8)) else 16
        if fusable:
            K, gs = self.in_features, self.group_size
            if self.g_idx is None:
                plan = "fused"
            else:
                g = self.g_idx.to(torch.int64)
                contiguous = torch.arange(K, device=g.device) // gs
                if torch.equal(g, contiguous):
                    plan = "fused"
                elif K % self.n_pack == 0:
                    order = torch.argsort(g, stable=True)
                    if torch.equal(g[order], contiguous):
                        bits, npk = self.bits, self.n_pack
                        shifts = (torch.arange(npk, device=g.device, dtype=torch.int32) * bits).view(1, npk, 1)
                        mask = (1 << bits) - 1
                        codes = ((self.qweight.unsqueeze(1) >> shifts) & mask).reshape(K, -1)  # [K, N] fields
                        codes = codes.index_select(0, order).reshape(K // npk, npk, -1)
                        words = (codes.to(torch.int64) << shifts.to(torch.int64)).sum(dim=1)  # disjoint fields, < 2^32
                        words = torch.where(words >= 2**31, words - 2**32, words)
                        self._qweight_sorted = words.to(torch.int32).contiguous()
                        self._k_order = order
                        plan = "fused_act_order"
                    else:
                        plan = "fused_g_idx"  # groups of uneven size: the library's general kernel looks the group up per k
                else:
                    plan = "fused_g_idx"
        self._plan_key, self._plan = key, plan
        return plan

    def extra_repr(self):
        s = super().extra_repr()
        if self.use_optimum_format:
            s += ", use_optimum_format=True"
        return s


# the reference's class name for the non-Gaudi device class; code that imports it keeps working
INCWeightOnlyLinear = MI355XWeightOnlyLinear

def woq_linear_group(x, modules):
    """[m(x) for m in modules] for packed modules that multiply the SAME activation -- q / k / v of an attention block, gate / up of
    an MLP -- as ONE launch (inc_woq_gemm_multi) when x is a decode-sized batch (<= 64 rows).  Each module keeps its own buffers and
    state-dict keys (`MI355XWeightOnlyLinear` stays the single-module path); the result of every module is what `inc_woq_gemm`'s
    streaming kernel computes for it.  Anything the batched launch does not cover (prefill-sized x, g_idx plans, other widths,
    non-optimum layouts, a dtype other than bf16 / fp16) is the plain list of single calls: the reference's forward per module
    (modules.py:594-610)."""
    mods = list(modules)
    m0 = mods[0]
    ok = (len(mods) >= 2 and x.dtype in (torch.bfloat16, torch.float16) and x.is_cuda and x.numel() > 0
          and all(isinstance(m, MI355XWeightOnlyLinear) and m.bits in (4, 8) and m.bits == m0.bits and m.in_features == m0.in_features and m.group_size == m0.group_size
                  and m._forward_plan() == "fused" for m in mods))
    if ok:
        K = m0.in_features
        x2d = x.reshape(-1, K)
        if x2d.shape[0] <= 64:
            if not x2d.is_contiguous():
                x2d = x2d.contiguous()
            parts = [(m.qweight, m.scales, m.qzeros, m.bias, m.out_features) for m in mods]
            # the prepared call lives on the group's first module (a cache, not state: it dies with the module, copies and pickles of the
            # module start without it) and is rebuilt when any buffer of the group is replaced or written to
            cache = m0.__dict__.setdefault("_group_calls", {})
            key = tuple(id(m) for m in mods[1:]) + (x.dtype,)
            call = cache.get(key)
            if call is None or not call.current(parts):
                if len(cache) >= 8:
                    cache.clear()
                call = cache[key] = ops.WoqGemmGroupCall(parts, K, m0.group_size, m0.bits, x.dtype)
            ys = call(x2d)
            if ys is not None:
                return [y.view(*x.shape[:-1], m.out_features) for y, m in zip(ys, mods)]
    return [m(x) for m in mods]


class MulLinear(torch.nn.Module):
    """Linear with a per-input-channel multiplier in front (AWQ scale that cannot be folded upstream,
    reference modules.py:907-949)."""

    def __init__(self, module, input_scale=None):
        super().__init__()
        if input_scale is None:
            input_scale = torch.empty(module.in_features)
        self.register_buffer("input_scale", input_scale)
        self.add_module("linear", module)

    @property
    def weight(self):
        return self.linear.weight

    @weight.setter
    def weight(self, weight):
        self.linear.weight = weight

    def forward(self, X):
        return self.linear(torch.mul(X, self.input_scale))

    def _update_linear(self):
        """Fold the multiplier into the weight: y = (x * input_scale) W'^T with W' = W / input_scale (reference :939-943)."""
        with torch.no_grad():
            self.linear.weight.div_(self.input_scale.view(1, -1).to(self.linear.weight.dtype))
        return self.linear

    def _recover_linear(self):
        """Undo `_update_linear` (reference :945-949)."""
        with torch.no_grad():
            self.linear.weight.mul_(self.input_scale.view(1, -1).to(self.linear.weight.dtype))
        return self.linear
