"""SmoothQuant W8A8 on MI355X (SURVEY.md section 8 row f-4 / BASELINE config #4).

Reference: neural_compressor/torch/algorithms/smooth_quant/utility.py
  cal_scale :605-626   quant_dequant_w_v1 :652-723   quant_dequant_x_v1 :726-755   Calibration :840-953
  TorchSmoothQuant :1895 (_scale_layer_weight :1968, _absorb_scales :1994, _cal_scales :2122, _adjust_parameters :2158,
  transform :2289)   SQLinearWrapper :2559-2662

The smoothing itself (calibrate per-channel |x| maxima, s = amax_x^a / amax_w^(1-a), fold s into the weights and 1/s into
the producer or a multiplier) is the reference's algorithm with its own function and class names.  What follows it differs
by necessity: the reference hands the smoothed model to intel_extension_for_pytorch for observers and INT8 kernels
(smooth_quant.py:105-125; IPEX is not in /root/reference), here `W8A8Linear` runs the in-tree fake-quant specification
(`quant_dequant_w_v1` per-channel sym int8 weights, `quant_dequant_x_v1` / `_calculate_qparams` per-tensor asym uint8
activations) exactly, in integers, on the matrix cores (`inc_w8a8_gemm`).
"""

import torch

from .... import ops
from ....common.utils import logger
from ...utils.utility import get_module, set_module

LM_NORM_NAMES = ("LlamaRMSNorm", "T5LayerNorm", "MistralRMSNorm", "Qwen2RMSNorm", "RMSNorm")  # weight-only norms (:2049)


def cal_scale(input_max_abs, weights, alpha, weight_max_lb=1e-5):
    """Smoothing scale of the layers in `weights` (they share the input whose per-channel abs-max is given) (:605-626)."""
    amax_w = None
    for w in weights:
        amax_w = ops.sq_weight_col_absmax(w.detach(), amax_w)
    return ops.sq_cal_scale(input_max_abs.to(amax_w.device), amax_w, alpha, weight_max_lb)


class Calibration:
    """Per-channel running min / max of every hooked layer's input (:840-953), accumulated in HBM by a HIP kernel."""

    def __init__(self, model, dataloder=None, q_func=None, device="cuda"):
        self.model = model
        self.dataloader = dataloder
        self.q_func = q_func
        self.device = device
        self.input_mins, self.input_maxes = {}, {}
        self.hook_handles = []
        self.same_input = {}  # layer name -> name of the first layer that saw the very same input tensor
        self.producer = {}    # layer name -> name of the norm whose output tensor IS this layer's input (folding)
        self.example_call = None  # (args, kwargs) of the first calibration forward: replayed to VERIFY a fold numerically

    def _save_input_pc_hook(self, name):
        def save_input_hook(module, inputs, outputs):
            x = inputs[0]
            x2d = x.reshape(-1, x.shape[-1])
            if name not in self.input_maxes:
                self.input_mins[name], self.input_maxes[name] = ops.sq_new_minmax(x2d.shape[-1], x2d.device)
            ops.sq_channel_minmax(x2d, self.input_mins[name], self.input_maxes[name])
            key = (x.data_ptr(), tuple(x.shape), x.dtype, x._version)
            owner = self._live.setdefault(key, (name, x))[0]  # the tuple keeps x alive: its address cannot be reused
            self.same_input.setdefault(name, owner)
            norm = self._norm_out.get(key)
            if norm is not None and self.producer.setdefault(name, norm[0]) != norm[0]:
                self.producer[name] = None  # fed by different producers on different calls: not foldable

        return save_input_hook

    def _save_norm_output_hook(self, name):
        def hook(module, inputs, output):
            if isinstance(output, torch.Tensor):
                self._norm_out[(output.data_ptr(), tuple(output.shape), output.dtype, output._version)] = (name, output)

        return hook

    def _add_min_max_observer(self, modules):
        self._live, self._norm_out = {}, {}
        self.hook_handles = [m.register_forward_hook(self._save_input_pc_hook(n)) for n, m in modules.items()]
        for n, m in self.model.named_modules():
            if isinstance(m, torch.nn.LayerNorm) or type(m).__name__.endswith(("RMSNorm", "LayerNorm")):
                self.hook_handles.append(m.register_forward_hook(self._save_norm_output_hook(n)))
        root = self.model

        def clear(_m, _i, _o):
            self._live.clear()
            self._norm_out.clear()

        self.hook_handles.append(root.register_forward_hook(clear))

        def remember(_m, args, kwargs):
            if self.example_call is None:
                self.example_call = (args, dict(kwargs))

        self.hook_handles.append(root.register_forward_pre_hook(remember, with_kwargs=True))

    def _remove_observer(self):
        for h in self.hook_handles:
            h.remove()
        self.hook_handles = []
        self._live, self._norm_out = {}, {}

    @torch.no_grad()
    def calibrate(self, calib_iter=100, op_types=(torch.nn.Linear,)):
        hook_modules = {n: m for n, m in self.model.named_modules() if isinstance(m, tuple(op_types))}
        self._add_min_max_observer(hook_modules)
        if self.q_func is not None:
            self.q_func(self.model)
        else:
            assert self.dataloader, "Please set dataloader for calibration."
            for i, batch in enumerate(self.dataloader):
                if i >= calib_iter:
                    break
                self.model(*batch) if isinstance(batch, (tuple, list)) else self.model(batch)
        self._remove_observer()
        return self.input_mins, self.input_maxes


class SQLinearWrapper(torch.nn.Module):
    """`y = sq_linear(x * input_scale)` with `sq_linear.weight = W / input_scale` (:2559-2662): the float form of a
    smoothed layer whose producer could not absorb the scale."""

    def __init__(self, module, input_scale, input_minmax, alpha=0.5, dtype=torch.quint8):
        super().__init__()
        self.register_buffer("input_scale", input_scale)
        self.alpha = alpha
        self.dtype = dtype
        self.scale, self.zero_point = self._calculate_qparams(input_scale, input_minmax, dtype)
        self.add_module("sq_linear", module)
        self._update_sq_linear()

    @property
    def weight(self):
        return self.sq_linear.weight

    def forward(self, X):
        return self.sq_linear(torch.mul(X, self.input_scale.to(X.dtype)))

    @staticmethod
    def _calculate_qparams(input_scale, input_minmax, dtype=torch.quint8):
        """Static per-tensor uint8 parameters of the SMOOTHED input (:2607-2631)."""
        if dtype != torch.quint8:
            raise ValueError(f"Unsupported dtype for quantization parameters: {dtype}")
        quant_min, quant_max = 0, 255
        s = 1.0 if input_scale is None else input_scale.float()
        min_val = torch.min(input_minmax[0].float() * s)
        max_val = torch.max(input_minmax[1].float() * s)
        min_val_neg = torch.min(min_val, torch.zeros_like(min_val))
        max_val_pos = torch.max(max_val, torch.zeros_like(max_val))
        scale = (max_val_pos - min_val_neg) / float(quant_max - quant_min)
        scale = torch.max(scale, torch.tensor(torch.finfo(torch.float32).eps, device=scale.device))
        zero_point = quant_min - torch.round(min_val_neg / scale).to(torch.int)
        zero_point = torch.clamp(zero_point, quant_min, quant_max)
        return scale.reshape(1), zero_point.reshape(1)

    def _update_sq_linear(self):
        with torch.no_grad():
            self.sq_linear.weight /= self.input_scale.view(1, -1).to(self.sq_linear.weight.dtype)

    def _recover_sq_linear(self):
        with torch.no_grad():
            self.sq_linear.weight *= self.input_scale.view(1, -1).to(self.sq_linear.weight.dtype)


class W8A8Linear(torch.nn.Module):
    """INT8 x INT8 Linear: per-output-channel symmetric int8 weights, static per-tensor asymmetric uint8 activations.

    forward = inc_sq_quant_act (mul by input_scale, quantise, shift to signed) + inc_w8a8_gemm (int32 accumulate,
    fp32 epilogue).  Buffers: qweight int8 [N, Kp] (K padded to a multiple of 128 with zeros), w_scale fp32 [N],
    alpha fp32 [N] = act_scale * w_scale, corr int32 [N] = (128 - act_zp) * rowsum(qweight), input_scale fp32 [K]
    (absent when the smoothing was folded into the producer), act_scale / act_zp, bias.
    """

    K_ALIGN = 128

    def __init__(self, in_features, out_features, bias=False, has_input_scale=False, device="cuda", float_type=torch.float16):
        super().__init__()
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError(f"W8A8Linear needs a HIP device ('cuda[:N]'), got '{device}'. There is no CPU implementation.")
        self.in_features, self.out_features = in_features, out_features
        self.kp = -(-in_features // self.K_ALIGN) * self.K_ALIGN
        self.float_type = float_type
        self.register_buffer("qweight", torch.zeros((out_features, self.kp), dtype=torch.int8, device=dev))
        self.register_buffer("w_scale", torch.zeros(out_features, dtype=torch.float32, device=dev))
        self.register_buffer("alpha", torch.zeros(out_features, dtype=torch.float32, device=dev))
        self.register_buffer("corr", torch.zeros(out_features, dtype=torch.int32, device=dev))
        self.register_buffer("act_scale", torch.ones(1, dtype=torch.float32, device=dev))
        self.register_buffer("act_zp", torch.zeros(1, dtype=torch.int32, device=dev))
        self.register_buffer("input_scale", torch.ones(in_features, dtype=torch.float32, device=dev) if has_input_scale else None)
        self.register_buffer("bias", torch.zeros(out_features, dtype=float_type, device=dev) if bias else None)
        self._sx = self._zp = self._ver = None

    @classmethod
    @torch.no_grad()
    def from_float(cls, module, input_min, input_max, device="cuda", stat_scale=None):
        """`module`: nn.Linear (smoothing folded or absent) or SQLinearWrapper; input_min / input_max: the calibrated
        per-channel statistics of the layer's ORIGINAL input; `stat_scale`: the per-channel factor the producer now
        applies to that input when the smoothing was folded (1/s) -- a wrapper brings its own (`input_scale`)."""
        wrapper = module if isinstance(module, SQLinearWrapper) else None
        lin = wrapper.sq_linear if wrapper is not None else module
        in_scale = wrapper.input_scale.float() if wrapper is not None else None
        ft = lin.weight.dtype if lin.weight.dtype in (torch.float16, torch.bfloat16) else torch.float16
        new = cls(lin.in_features, lin.out_features, bias=lin.bias is not None, has_input_scale=in_scale is not None,
                  device=device, float_type=ft)
        dev = new.qweight.device
        w = lin.weight.detach().to(dev)
        qw, w_scale, rowsum = ops.sq_quant_weight(w, None, new.kp)
        stat = in_scale if in_scale is not None else stat_scale
        sx, zp = SQLinearWrapper._calculate_qparams(None if stat is None else stat.to(dev), [input_min.to(dev), input_max.to(dev)])
        new.qweight.copy_(qw)
        new.w_scale.copy_(w_scale)
        new.act_scale.copy_(sx)
        new.act_zp.copy_(zp)
        new.alpha.copy_(w_scale * sx)
        new.corr.copy_(((128 - zp.to(torch.int64)) * rowsum.to(torch.int64)).to(torch.int32))
        if in_scale is not None:
            new.input_scale.copy_(in_scale)
        if lin.bias is not None:
            new.bias.copy_(lin.bias.detach().to(ft))
        return new

    def forward(self, input):
        x = input
        lead = x.shape[:-1]
        x2d = x.reshape(-1, self.in_features)
        ver = (self.act_scale.data_ptr(), self.act_scale._version, self.act_zp._version)
        if self._sx is None or self._ver != ver:  # one host read per (re)load: the static activation parameters are launch arguments
            self._sx, self._zp, self._ver = float(self.act_scale.item()), float(self.act_zp.item()), ver
        out_dtype = x.dtype if x.dtype in (torch.float16, torch.bfloat16) else self.float_type
        if x2d.shape[0] == 0:
            return torch.empty((*lead, self.out_features), dtype=torch.float32 if x.dtype == torch.float32 else out_dtype, device=x.device)
        xq = ops.sq_quant_act(x2d, self.input_scale, self._sx, self._zp, self.kp)
        y = ops.w8a8_gemm(xq, self.qweight, self.alpha, self.corr, self.bias, out_dtype)
        if x.dtype == torch.float32:  # an fp32 model keeps seeing fp32 activations (the kernel's epilogue emits 16-bit)
            y = y.float()
        return y.reshape(*lead, self.out_features)

    def recover(self):
        """Dequantised weight [N, K] fp32 (the `quant_dequant_w_v1` image of the smoothed weight)."""
        return self.qweight[:, : self.in_features].float() * self.w_scale.view(-1, 1)

    def extra_repr(self):
        return f"in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}, w8a8"


class TorchSmoothQuant:
    """Calibrate, compute the smoothing scales and apply them (:1895-2557), for nn.Linear layers.

    folding=False (the reference's default with IPEX >= 2.1): every Linear gets a multiplier (`SQLinearWrapper`); layers
    that receive the very same input tensor share one scale when `scale_sharing`.  folding=True: only layers whose
    producer can absorb 1/s (a norm directly in front: found by the same hook-based discovery the AWQ path uses) are
    smoothed; `absorb_to_layer` may be given explicitly ({absorber name: [layer names]}).
    """

    def __init__(self, model, dataloader=None, example_inputs=None, q_func=None, scale_sharing=True, **kwargs):
        self.model = model
        self.dataloader = dataloader
        self.example_inputs = example_inputs
        self.q_func = q_func
        self.scale_sharing = scale_sharing
        self.input_mins, self.input_maxes = {}, {}
        self.same_input, self.producer = {}, {}
        self.example_call = None
        self.weight_scale_info, self.absorb_scales_info = {}, {}
        self.absorb_to_layer = {}
        self.weight_max_lb = 1e-5
        self.insert_mul, self.allow_absorb = True, False

    # -- pieces with the reference's names -----------------------------------------------------------------------------
    @torch.no_grad()
    def _scale_layer_weight(self, layer_name, scale, alpha=0.5, input_minmax=None):
        layer = get_module(self.model, layer_name)
        if self.insert_mul:
            new_module = SQLinearWrapper(layer, (1.0 / scale).to(layer.weight.device), input_minmax, alpha)
            set_module(self.model, layer_name, new_module)
        elif self.allow_absorb:
            layer.weight.data = (layer.weight.data.float() * scale.view(1, -1).to(layer.weight.device)).to(layer.weight.dtype)
        return scale

    @torch.no_grad()
    def _absorb_scales(self, layer_name, scale):
        """Fold `scale` (= 1/s) into the OUTPUT channels of the producer (:1994-2061)."""
        if self.insert_mul or not self.allow_absorb:
            return
        layer = get_module(self.model, layer_name)
        s = scale.to(layer.weight.device)
        if isinstance(layer, torch.nn.LayerNorm):
            layer.weight.data = (layer.weight.data.float() * s).to(layer.weight.dtype)
            if layer.bias is not None:
                layer.bias.data = (layer.bias.data.float() * s).to(layer.bias.dtype)
        elif isinstance(layer, torch.nn.Linear):
            if layer.bias is not None:
                layer.bias.data = (layer.bias.data.float() * s).to(layer.bias.dtype)
            layer.weight.data = (layer.weight.data.float() * s.view(-1, 1)).to(layer.weight.dtype)
        elif type(layer).__name__ in LM_NORM_NAMES or (hasattr(layer, "weight") and getattr(layer, "bias", None) is None):
            layer.weight.data = (layer.weight.data.float() * s).to(layer.weight.dtype)
        else:
            raise RuntimeError(f"cannot absorb a SmoothQuant scale into {type(layer).__name__} ({layer_name})")

    def _cal_scales(self, absorb_to_layer, input_maxes, alpha=0.5):
        absorb_scales_info, weight_scales_info = {}, {}
        for key, layer_names in absorb_to_layer.items():
            alpha_tmp = alpha[key] if isinstance(alpha, dict) else alpha
            weights = [get_module(self.model, n).weight for n in layer_names]
            scale = cal_scale(input_maxes[layer_names[0]], weights, alpha_tmp, self.weight_max_lb)
            inv = 1.0 / scale
            inv[scale == 0] = 0
            absorb_scales_info[key] = inv
            for n in layer_names:
                weight_scales_info[n] = scale
        return absorb_scales_info, weight_scales_info

    def _adjust_parameters(self, absorb_to_layer, input_maxes, alpha=0.5):
        absorb_scales_info, weight_scales_info = self._cal_scales(absorb_to_layer, input_maxes, alpha)
        for key, layer_names in absorb_to_layer.items():
            alpha_tmp = alpha[key] if isinstance(alpha, dict) else alpha
            self._absorb_scales(key, absorb_scales_info[key])
            for n in layer_names:
                minmax = [self.input_mins[layer_names[0]], self.input_maxes[layer_names[0]]]
                self._scale_layer_weight(n, weight_scales_info[n], alpha_tmp, minmax)
        return weight_scales_info, absorb_scales_info

    def _get_all_layer_names(self, op_types=(torch.nn.Linear,)):
        return {n: [n] for n, m in self.model.named_modules() if isinstance(m, tuple(op_types))}

    def _find_foldable(self):
        """{norm name: [Linear names whose input tensor is that norm's output]} -- observed during calibration, then
        VERIFIED: the hooks only see the selected Linears, so a norm whose output also feeds something else (a Linear
        that is not being smoothed, a residual add, any functional op) would emit x/s to a consumer that keeps unscaled
        weights.  Like the AWQ path's discovery, every candidate fold is therefore tried with a random rescale on the
        first calibration batch and kept only if the model output does not move; a candidate that fails is treated as
        the reference treats a layer it cannot fold (folding=True smooths absorbable layers only)."""
        found = {}
        for layer, norm in self.producer.items():
            if norm is not None:
                found.setdefault(norm, []).append(layer)
        if not found:
            return found
        if self.example_call is None:
            logger.warning("SmoothQuant folding: no calibration forward was recorded, candidate folds cannot be verified and are dropped")
            return {}
        args, kwargs = self.example_call

        def run():
            out = self.model(*args, **kwargs)
            out = out.logits if hasattr(out, "logits") else (out[0] if isinstance(out, (tuple, list)) else out)
            return out.float()

        base = run()
        gen = torch.Generator().manual_seed(0)
        verified = {}
        for norm_name, layers in found.items():
            norm = get_module(self.model, norm_name)
            mods = [get_module(self.model, n) for n in layers]
            K = mods[0].weight.shape[1]
            s = (0.5 + 1.5 * torch.rand(K, generator=gen)).to(base.device)
            saved = [norm.weight.data.clone(), None if getattr(norm, "bias", None) is None else norm.bias.data.clone()] + [m.weight.data.clone() for m in mods]
            try:
                norm.weight.data = (norm.weight.data.float() / s).to(norm.weight.dtype)
                if saved[1] is not None:
                    norm.bias.data = (norm.bias.data.float() / s).to(norm.bias.dtype)
                for m in mods:
                    m.weight.data = (m.weight.data.float() * s.view(1, -1)).to(m.weight.dtype)
                moved = float((run() - base).norm() / base.norm().clamp_min(1e-30))
            finally:
                norm.weight.data = saved[0]
                if saved[1] is not None:
                    norm.bias.data = saved[1]
                for m, w in zip(mods, saved[2:]):
                    m.weight.data = w
            # a correct fold moves the output by rounding noise only (16-bit models: ~1e-2); a missed consumer by O(1)
            if moved <= 5e-2:
                verified[norm_name] = layers
            else:
                logger.warning("SmoothQuant folding: %s also feeds something other than %s (output moved by %.3f under a "
                               "test rescale); these layers are not smoothed", norm_name, layers, moved)
        return verified

    # -- the entry (:2289-2432) ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def transform(self, alpha=0.5, folding=False, calib_iter=100, op_types=(torch.nn.Linear,), scale_sharing=None,
                  absorb_to_layer=None, **kwargs):
        if alpha == "auto":
            raise NotImplementedError("alpha='auto' (the reference's layer-wise alpha tuner) is not implemented on MI355X")
        alpha = max(float(alpha), 0.0) if not isinstance(alpha, dict) else alpha
        self.insert_mul, self.allow_absorb = (False, True) if folding else (True, False)
        if scale_sharing is not None:
            self.scale_sharing = scale_sharing
        if not self.input_maxes:
            calib = Calibration(self.model, self.dataloader, self.q_func)
            self.input_mins, self.input_maxes = calib.calibrate(calib_iter, op_types)
            self.same_input, self.producer = calib.same_input, calib.producer
            self.example_call = calib.example_call
        input_maxes_abs = {k: torch.max(self.input_mins[k].abs(), self.input_maxes[k].abs()) for k in self.input_mins}
        if absorb_to_layer is not None:
            self.absorb_to_layer = {k: list(v) for k, v in absorb_to_layer.items()}
        elif folding:
            self.absorb_to_layer = self._find_foldable()
        else:
            groups = {}
            for name in self._get_all_layer_names(op_types):
                if name not in input_maxes_abs:
                    continue  # never executed during calibration
                owner = self.same_input.get(name, name) if self.scale_sharing else name
                groups.setdefault(owner, []).append(name)
            self.absorb_to_layer = groups
        self.absorb_to_layer = {k: [n for n in v if n in input_maxes_abs] for k, v in self.absorb_to_layer.items()}
        self.absorb_to_layer = {k: v for k, v in self.absorb_to_layer.items() if v}
        if not self.absorb_to_layer:
            logger.warning("empty absorb_to_layer, smoothquant is ignored ")
            return self.model
        self.weight_scale_info, self.absorb_scales_info = self._adjust_parameters(self.absorb_to_layer, input_maxes_abs, alpha)
        self.model._smoothquant_optimized = True
        return self.model
