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
        self.captured_calls = []  # the first `capture_limit` calibration forwards (args, kwargs): the alpha="auto" tuner replays them
        self.capture_limit = 32   # AutoAlpha's default n_samples (reference :1252)

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
            if len(self.captured_calls) < self.capture_limit:
                self.captured_calls.append((args, dict(kwargs)))

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


def quant_dequant_w_v1(m, num_bits=8, scheme="sym"):
    """Per-output-channel int8 fake quantisation of a Linear's weight (reference :652-695).  "sym" (what the W8A8 path uses): the
    codes and scales come from inc_sq_quant_weight (bit-exact against the reference's golden vectors), dequantised in fp32."""
    assert isinstance(m, torch.nn.Linear) and num_bits == 8 and scheme in ("sym", "asym")
    w = m.weight.detach().float().contiguous()
    if scheme == "asym":
        # the reference's per-channel uint8 cell with a zero point (:670-692).  Nothing on its W8A8 path selects it (the tuner calls
        # the default "sym", IPEX's SmoothQuant qconfig is symmetric per-channel); restated in HBM with elementwise ops for the API
        eps = torch.finfo(torch.float32).eps
        zero = torch.zeros(w.shape[0], device=w.device)
        row_min = torch.min(w, dim=1).values
        scale = torch.clip((torch.maximum(torch.max(w, dim=1).values, zero) - torch.minimum(row_min, zero)) / (2**num_bits - 1), min=eps)
        bias = torch.round(0 - row_min / scale).unsqueeze(-1)
        scale = scale.unsqueeze(-1)
        q = torch.round(w / scale + bias).clamp_(0, 2.0**num_bits - 1.0)
        return (q - bias) * scale
    qw, scale, _ = ops.sq_quant_weight(w)
    return qw[:, : w.shape[1]].float() * scale.view(-1, 1)


def quant_dequant_x_v1(x, min_x=None, max_x=None, num_bits=8):
    """Per-tensor asymmetric uint8 fake quantisation of an activation from channel min / max statistics (reference :726-755)."""
    eps = torch.finfo(torch.float32).eps
    q_min, q_max = 0, 2.0**num_bits - 1.0
    if max_x is None or min_x is None:
        max_x, min_x = torch.max(x), torch.min(x)
    else:
        max_x, min_x = torch.max(max_x), torch.min(min_x)
    scale = torch.clip((max_x - min_x) / (2**num_bits - 1), min=eps)
    bias = torch.round((0 - min_x) / scale)
    q_x = torch.round(x / scale + bias)
    q_x.clamp_(q_min, q_max)
    return scale * (q_x - bias)


class WrapperLayer(torch.nn.Module):
    """Fake-quant stand-in of a Linear during alpha tuning (reference :2665-2771): records its (quantised-model) input and its
    output; `q_dq_forward` evaluates the layer for a candidate (input_scale, weight_scale) pair."""

    def __init__(self, layer, input_min, input_max, save_q_input=False):
        super().__init__()
        self.add_module("orig_layer", layer)
        self.quant = False
        self.q_input = None
        self.input_max, self.input_min = input_max, input_min
        self.weight_scale, self.input_scale = None, None
        self.save_q_input = save_q_input
        self.output = None
        self.do_blockwise = False  # block-wise tuning: the weight is already fake-quantised for the candidate alpha (_bw_weight)
        self._bw_weight = None

    def enable_quant(self):
        self.quant = True

    def disable_quant(self):
        self.quant = False

    def update_scale(self, input_scale, weight_scale):
        self.input_scale, self.weight_scale = input_scale, weight_scale

    def q_dq_forward(self, x, input_scale, weight_scale):
        lin = self.orig_layer
        w = lin.weight.detach().float()
        if weight_scale is not None:
            w = w * weight_scale
        tmp = torch.nn.Linear(w.shape[1], w.shape[0], bias=False, device=w.device)
        tmp.weight.data = w
        w_qdq = quant_dequant_w_v1(tmp)
        x = x.float()
        if input_scale is None:
            x = quant_dequant_x_v1(x, self.input_min, self.input_max)
        else:
            x = input_scale * x
            x = quant_dequant_x_v1(x, self.input_min * input_scale, self.input_max * input_scale)
        bias = None if lin.bias is None else lin.bias.float()
        return torch.nn.functional.linear(x, w_qdq, bias)

    def q_dq_forward_blockwise(self, x, input_scale):
        """Reference :2731-2748: the input is fake-quantised as in q_dq_forward, the weight was prepared by the tuner."""
        lin = self.orig_layer
        x = x.float()
        if input_scale is None:
            x = quant_dequant_x_v1(x, self.input_min, self.input_max)
        else:
            x = input_scale * x
            x = quant_dequant_x_v1(x, self.input_min * input_scale, self.input_max * input_scale)
        bias = None if lin.bias is None else lin.bias.float()
        return torch.nn.functional.linear(x, self._bw_weight, bias)

    def prepare_blockwise(self):
        """The tuner's per-alpha preparation of one layer (reference :1676-1681): weight * weight_scale, fake-quantised."""
        w = self.orig_layer.weight.detach().float()
        if self.weight_scale is not None:
            w = w * self.weight_scale
        tmp = torch.nn.Linear(w.shape[1], w.shape[0], bias=False, device=w.device)
        tmp.weight.data = w
        self._bw_weight = quant_dequant_w_v1(tmp)
        self.do_blockwise = True

    def forward(self, x):
        if self.quant:
            if self.save_q_input:
                self.q_input = x
            if self.do_blockwise:
                output = self.q_dq_forward_blockwise(x, self.input_scale).to(x.dtype)
            else:
                output = self.q_dq_forward(x, self.input_scale, self.weight_scale).to(x.dtype)
        else:
            output = self.orig_layer(x)
        self.output = output
        return output


class AutoAlpha:
    """Layer-wise alpha tuner of SmoothQuant (`alpha="auto"`; reference :1232-1892, the model-wise "version1" path).

    For every calibration sample: one float forward and one fake-quant forward (every tuned Linear replaced by a WrapperLayer
    that quantises its smoothed weight per channel and its smoothed input per tensor, and remembers the input it got), then
    every layer is re-evaluated on that remembered input for every alpha of the grid; the loss is
    sum(|y_fp / max|y_fp| - y_q / max|y_fp||^0.5).  Restated as written, including what looks unintended: `loss_alphas` is
    re-initialised for every sample (:1776), so the "accumulated" table only ever holds the CURRENT sample's losses -- the
    alphas are updated every n_samples // 4 samples from that sample alone and the final choice is made on the last sample.
    Everything runs in HBM: forwards are torch, the fake quantisation uses inc_sq_quant_weight + elementwise torch ops.
    `do_blockwise=True` (reference :1821) tunes on BLOCK outputs instead: see _get_one_batch_auto_loss_blockwise."""

    def __init__(self, model, dataloader, absorb_to_layer, op_types, device, q_func, example_inputs, weight_clip=True,
                 alpha_min=0.3, alpha_max=0.7, alpha_step=0.1, shared_criterion="mean", init_alpha=0.5, folding=False,
                 do_blockwise=False, n_samples=32, calibration=None):
        self.model = model
        self.model.eval()
        self.dataloader = dataloader
        self.alpha_min, self.alpha_max, self.alpha_step = alpha_min, alpha_max, alpha_step
        self.shared_criterion = shared_criterion
        self.init_alpha = init_alpha
        self.loss_type = "blockwise" if do_blockwise else "model_wise"
        self.calib_sample_num = n_samples if n_samples else 32
        self.op_types = op_types
        self.absorb_to_layer = absorb_to_layer
        self.q_func = q_func
        self.folding = folding
        self.example_inputs = example_inputs
        self.weight_clip = weight_clip[0] if isinstance(weight_clip, tuple) else weight_clip
        self.input_maxes, self.input_mins, self.input_maxes_abs = {}, {}, {}
        self.device = device
        self.calibration = calibration  # statistics + captured forwards of an earlier calibration pass (prepare/convert flow)
        self.last_loss_alphas = None    # {layer: {str(alpha): loss}} of the final decision (diagnostics / parity tests)

    # -- entry (:1278-1324) ---------------------------------------------------------------------------------------------
    @torch.no_grad()
    def tune(self):
        if self.calibration is not None:
            calib = self.calibration
            self.input_mins, self.input_maxes = calib.input_mins, calib.input_maxes
        else:
            calib = Calibration(self.model, self.dataloader, self.q_func, self.device)
            calib.capture_limit = self.calib_sample_num
            self.input_mins, self.input_maxes = calib.calibrate(100, self.op_types)
        self.calls = list(calib.captured_calls)[: self.calib_sample_num]
        assert self.calls, "alpha='auto' needs calibration forwards to replay"
        for key in self.input_mins:
            self.input_maxes_abs[key] = torch.max(torch.abs(self.input_mins[key]), torch.abs(self.input_maxes[key]))
        if not self.folding:
            for d in set(self.absorb_to_layer.keys()).difference(self.input_mins.keys()):
                del self.absorb_to_layer[d]
        if self.loss_type == "blockwise":  # reference :1304-1322
            # every transformer block tunes ONE alpha for the smoothable layers inside it; a layer outside all blocks is its own "block".
            # Same grouping and same key order as the reference's table (blocks first, in model order, then the stragglers as they are
            # met; a layer whose name contains several block prefixes joins each of them, as there)
            blocks = self.get_blocks()
            members = {blk: [] for blk in blocks}
            stragglers = {}
            for layer in self._get_sq_layer_names():
                homes = [blk for blk in blocks if (blk + ".") in layer]
                for blk in homes:
                    members[blk].append(layer)
                if not homes:
                    stragglers[layer] = [layer]
            self.block_to_module = {**members, **stragglers}
            self.block_names = list(self.block_to_module)
            logger.info("Blockwise auto-tuning: %d blocks found", len(self.block_names))
            return self._auto_tune_alpha_blockwise()
        return self._auto_tune_alpha()

    def get_blocks(self):
        """The children of the model's first ModuleList (reference :1326-1335)."""
        block_names = []
        for n, m in self.model.named_modules():
            if "ModuleList" in type(m).__name__:
                for nn_, _ in m.named_children():
                    block_names.append(n + "." + nn_)
                break
        return block_names

    # -- helpers with the reference's names -------------------------------------------------------------------------------
    def _get_all_hook_module_names(self):
        return [n for n, m in self.model.named_modules() if isinstance(m, tuple(self.op_types))]

    def _get_sq_layer_names(self):
        names = []
        for key in self.absorb_to_layer:
            names += self.absorb_to_layer[key]
        return names

    def _qdq_model_wrapper_for_auto(self, save_q_input=False):
        self.to_unwrap_module_names = self._get_all_hook_module_names()
        for name in self.to_unwrap_module_names:
            if name not in self.input_mins:
                continue
            module = get_module(self.model, name)
            set_module(self.model, name, WrapperLayer(module, self.input_mins[name], self.input_maxes[name], save_q_input=save_q_input))

    def _qdq_model_unwrapper_for_auto(self):
        for name in self.to_unwrap_module_names:
            module = get_module(self.model, name)
            if hasattr(module, "orig_layer"):
                set_module(self.model, name, module.orig_layer)

    def _change_qdq_for_auto(self, enable=True):
        for name in self._get_all_hook_module_names():
            name = name.split(".orig_layer")[0]
            module = get_module(self.model, name)
            if hasattr(module, "orig_layer"):
                module.enable_quant() if enable else module.disable_quant()

    def _cal_scales(self, absorb_to_layer, input_maxes, alpha=0.5):
        absorb_scales_info, weight_scales_info = {}, {}
        for key, layer_names in absorb_to_layer.items():
            alpha_tmp = alpha[key] if isinstance(alpha, dict) else alpha
            if alpha_tmp < 0:
                scale = torch.ones(1, device=self.device)
            else:
                weights = [get_module(self.model, n).orig_layer.weight if hasattr(get_module(self.model, n), "orig_layer")
                           else get_module(self.model, n).weight for n in layer_names]
                scale = cal_scale(input_maxes[layer_names[0]], weights, alpha_tmp)
            inv = 1.0 / scale
            inv[scale == 0] = 0
            absorb_scales_info[key] = inv
            for n in layer_names:
                weight_scales_info[n] = scale
        return absorb_scales_info, weight_scales_info

    def _update_scales_for_auto(self, absorb_scales, weight_scales):
        for key, layer_names in self.absorb_to_layer.items():
            for layer_name in layer_names:
                layer = get_module(self.model, layer_name)
                layer.update_scale(absorb_scales[key].view(1, -1), weight_scales[layer_name].view(1, -1))  # Linear: [1, K]

    @staticmethod
    def _get_auto_loss(output, output_q, loss_type="abs", loss_alpha=1.0):
        output, output_q = output.float(), output_q.float()
        if len(output.shape) <= 2:
            max_value = torch.max(torch.abs(output))
        else:
            output = output.reshape(output.shape[0], -1)
            output_q = output_q.reshape(output_q.shape[0], -1)
            max_value = torch.clip(torch.max(torch.abs(output), dim=-1).values.unsqueeze(-1), 1e-5)
        output = output / max_value
        output_q = output_q / max_value
        if loss_type == "abs":
            return torch.sum(torch.pow(torch.abs(output - output_q), 0.5))
        return torch.sum((output - output_q) ** 2)

    @staticmethod
    def _get_best_alpha(absorb_to_layer, loss_alphas, shared_criterion):
        best_alpha = {}
        for ln_name, layer_names in absorb_to_layer.items():
            cur = "min" if len(layer_names) == 1 else shared_criterion
            if cur == "mean":
                loss_tmp = {}
                for alpha in loss_alphas[layer_names[0]].keys():
                    loss_tmp.setdefault(alpha, 0)
                    for layer_name in layer_names:
                        loss_tmp[alpha] += loss_alphas[layer_name][alpha]
                res = sorted(loss_tmp.items(), key=lambda x: x[1])  # stable, like list.sort on the reference's pairs
                best_alpha[ln_name] = float(res[0][0])
            elif cur in ("min", "max"):
                tmp = []
                for layer_name in layer_names:
                    res = sorted(loss_alphas[layer_name].items(), key=lambda x: x[1])
                    tmp.append(float(res[0][0]))
                best_alpha[ln_name] = min(tmp) if cur == "min" else max(tmp)
            else:
                raise NotImplementedError
        return best_alpha

    def _forward(self, call):
        args, kwargs = call
        return self.model(*args, **kwargs)

    def _get_one_batch_auto_loss(self, call, alpha_space, orig_best_alpha, input_maxes):
        self._change_qdq_for_auto(enable=False)
        module_names = self._get_sq_layer_names()
        self._forward(call)  # quantisation off: float outputs
        fp32_output = {}
        for name in module_names:
            module = get_module(self.model, name)
            fp32_output[name] = module.output
            module.output = None
        self._change_qdq_for_auto(enable=True)
        absorb_input_scales, weight_scales = self._cal_scales(self.absorb_to_layer, input_maxes, orig_best_alpha)
        self._update_scales_for_auto(absorb_input_scales, weight_scales)
        self._forward(call)  # quantisation on at the current alphas: every layer remembers the input it received
        loss_alphas = {}
        for name in module_names:
            module = get_module(self.model, name)
            cur_alpha = orig_best_alpha[name] if isinstance(orig_best_alpha, dict) else orig_best_alpha
            loss_alphas[name] = {str(cur_alpha): self._get_auto_loss(fp32_output[name], module.output)}
        for alpha in alpha_space:
            absorb_input_scales, weight_scales = self._cal_scales(self.absorb_to_layer, input_maxes, alpha)
            self._update_scales_for_auto(absorb_input_scales, weight_scales)
            for name in module_names:
                if str(alpha) in loss_alphas[name]:
                    continue
                module = get_module(self.model, name)
                output = module.q_dq_forward(module.q_input, module.input_scale, module.weight_scale)
                loss_alphas[name][str(alpha)] = self._get_auto_loss(fp32_output[name], output)
        # one device -> host copy per sample (the reference compares 0-d CPU tensors)
        return {n: {a: float(v) for a, v in d.items()} for n, d in loss_alphas.items()}

    def _get_one_batch_auto_loss_blockwise(self, call, alpha_space, orig_best_alpha, input_maxes):
        """Reference :1618-1693.  Per sample: the float outputs of every block, a fake-quant forward at the current alphas (which
        leaves every block's INPUT under quantisation behind), then every block re-run on that input for every alpha of the grid
        with its tuned Linears fake-quantised for that alpha; loss = block output vs float block output.  The reference deep-copies
        the block per (alpha, sample); here the block's own wrappers are switched to the prepared weights and back (same arithmetic).
        Like the reference the replay hands the block ONLY its hidden states (no attention mask: :1685); models whose block cannot
        run like that get the reference's position_ids retry and then the captured call's own arguments."""
        self._change_qdq_for_auto(enable=False)
        block_modules = {key: get_module(self.model, key) for key in self.block_names}
        handles, captured = [], {}

        def save(name):
            def hook(module, args, kwargs, outputs):
                self.block_inputs[name] = args[0]
                self.block_outputs[name] = outputs[0]
                captured[name] = (args, kwargs)
            return hook

        for key, mod in block_modules.items():
            handles.append(mod.register_forward_hook(save(key), with_kwargs=True))
        try:
            self._forward(call)  # quantisation off: float block outputs
            fp32_output = {name: self.block_outputs[name] for name in self.block_names}
            self._change_qdq_for_auto(enable=True)
            absorb_input_scales, weight_scales = self._cal_scales(self.absorb_to_layer, input_maxes, orig_best_alpha)
            self._update_scales_for_auto(absorb_input_scales, weight_scales)
            self._forward(call)  # quantisation on at the current alphas: block inputs / outputs under quantisation
            loss_alphas = {}
            for block_name in self.block_names:
                loss = self._get_auto_loss(fp32_output[block_name], self.block_outputs[block_name])
                cur_alpha = orig_best_alpha
                if isinstance(orig_best_alpha, dict):
                    cur_alpha = orig_best_alpha[self.block_to_module[block_name][0]]
                loss_alphas[block_name] = {str(cur_alpha): loss}
            block_in = dict(self.block_inputs)
            block_call = dict(captured)
            for alpha in alpha_space:
                absorb_input_scales, weight_scales = self._cal_scales(self.absorb_to_layer, input_maxes, alpha)
                self._update_scales_for_auto(absorb_input_scales, weight_scales)
                for block_name in self.block_names:
                    if str(alpha) in loss_alphas[block_name]:
                        continue
                    block = block_modules[block_name]
                    wrappers = [block if (name == block_name and len(self.block_to_module[block_name]) == 1) else get_module(self.model, name)
                                for name in self.block_to_module[block_name]]
                    for w in wrappers:
                        w.prepare_blockwise()
                    try:
                        x = block_in[block_name]
                        try:
                            output = block(x)[0]
                        except Exception:  # the reference's retry (:1687-1689), then the block's own captured arguments
                            try:
                                position_ids = torch.arange(x.size()[1], device=x.device).view(x.size()[0], -1)
                                output = block(x, position_ids=position_ids)[0]
                            except Exception:
                                args, kwargs = block_call[block_name]
                                output = block(*args, **kwargs)[0]
                    finally:
                        for w in wrappers:
                            w.do_blockwise, w._bw_weight = False, None
                    loss_alphas[block_name][str(alpha)] = self._get_auto_loss(fp32_output[block_name], output)
        finally:
            for h in handles:
                h.remove()
        return {n: {a: float(v) for a, v in d.items()} for n, d in loss_alphas.items()}

    def _auto_tune_alpha_blockwise(self):
        """Reference :1821-1892 (with its per-sample reset of the loss table, like the model-wise tuner)."""
        logger.info("Start block-wise alpha tuning")
        self.block_inputs, self.block_outputs = {}, {}
        self.default_tune_setup()
        total_cnt, tmp_cnt, alpha_update_iter, tune_cnt = 0, 0, 0, 4
        multiply_factor = self.calib_sample_num // tune_cnt if self.calib_sample_num >= tune_cnt else self.calib_sample_num
        best_alphas = self.init_alpha
        loss_alphas = {}
        for call in self.calls:
            loss_alphas = {}  # (sic, reference :1846)
            best_alphas_per_module = best_alphas
            if isinstance(best_alphas, dict):
                for key, layer_names in self.absorb_to_layer.items():
                    for layer_name in layer_names:
                        best_alphas_per_module[layer_name] = best_alphas_per_module[key]
            loss_tmp = self._get_one_batch_auto_loss_blockwise(call, self.alpha_space, best_alphas_per_module, self.input_maxes_abs)
            for block_name in self.block_names:  # every Linear of a block carries the block's losses
                for key in self.block_to_module[block_name]:
                    loss_alphas[key] = loss_tmp[block_name]
            total_cnt += 1
            tmp_cnt += 1
            if tmp_cnt // multiply_factor >= 1:
                alpha_update_iter += 1
                tmp_cnt = 0
                best_alphas = self._get_best_alpha(self.absorb_to_layer, loss_alphas, self.shared_criterion)
                for key in best_alphas:
                    logger.info("Auto alpha update iter: %d, %s: %s", alpha_update_iter, key, best_alphas[key])
                absorb_input_scales, weight_scales = self._cal_scales(self.absorb_to_layer, self.input_maxes_abs, best_alphas)
                self._update_scales_for_auto(absorb_input_scales, weight_scales)
            if total_cnt >= self.calib_sample_num:
                break
        best_alphas = self._get_best_alpha(self.absorb_to_layer, loss_alphas, self.shared_criterion)
        self.last_loss_alphas = loss_alphas
        for key in best_alphas:
            logger.info("Final alpha %s:%s", key, best_alphas[key])
        self._qdq_model_unwrapper_for_auto()
        logger.info("block-wise auto tuning done")
        return best_alphas

    def default_tune_setup(self):
        import numpy

        round_num = max(len(str(self.alpha_min).split(".")[1]), len(str(self.alpha_max).split(".")[1]), len(str(self.alpha_step).split(".")[1]))
        self.alpha_space = numpy.round(numpy.arange(self.alpha_min, self.alpha_max + self.alpha_step, self.alpha_step), round_num).tolist()
        self._qdq_model_wrapper_for_auto(save_q_input=True)
        absorb_input_scales, weight_scales = self._cal_scales(self.absorb_to_layer, self.input_maxes_abs, self.init_alpha)
        self._update_scales_for_auto(absorb_input_scales, weight_scales)
        return absorb_input_scales, weight_scales

    def _auto_tune_alpha(self):
        logger.info("Start alpha tuning")
        self.default_tune_setup()
        total_cnt, tmp_cnt, alpha_update_iter, tune_cnt = 0, 0, 0, 4
        multiply_factor = self.calib_sample_num // tune_cnt if self.calib_sample_num >= tune_cnt else self.calib_sample_num
        best_alphas = self.init_alpha
        loss_alphas = {}
        for call in self.calls:
            loss_alphas = {}  # (sic, reference :1776) -- see the class docstring
            best_alphas_per_module = best_alphas
            if isinstance(best_alphas, dict):
                for key, layer_names in self.absorb_to_layer.items():
                    for layer_name in layer_names:
                        best_alphas_per_module[layer_name] = best_alphas_per_module[key]
            loss_tmp = self._get_one_batch_auto_loss(call, self.alpha_space, best_alphas_per_module, self.input_maxes_abs)
            if loss_alphas == {}:
                loss_alphas = loss_tmp
            total_cnt += 1
            tmp_cnt += 1
            if tmp_cnt // multiply_factor >= 1:
                alpha_update_iter += 1
                tmp_cnt = 0
                best_alphas = self._get_best_alpha(self.absorb_to_layer, loss_alphas, self.shared_criterion)
                for key in best_alphas:
                    logger.info("Auto alpha update iter: %d, %s: %s", alpha_update_iter, key, best_alphas[key])
                absorb_input_scales, weight_scales = self._cal_scales(self.absorb_to_layer, self.input_maxes_abs, best_alphas)
                self._update_scales_for_auto(absorb_input_scales, weight_scales)
            if total_cnt >= self.calib_sample_num:
                break
        best_alphas = self._get_best_alpha(self.absorb_to_layer, loss_alphas, self.shared_criterion)
        self.last_loss_alphas = loss_alphas
        for key in best_alphas:
            logger.info("Final alpha %s:%s", key, best_alphas[key])
        self._qdq_model_unwrapper_for_auto()
        logger.info("auto tuning done")
        return best_alphas


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
        self.calibration = None
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
        is_auto = isinstance(alpha, str) and alpha == "auto"
        if not is_auto:
            alpha = max(float(alpha), 0.0) if not isinstance(alpha, dict) else alpha
        self.insert_mul, self.allow_absorb = (False, True) if folding else (True, False)
        if scale_sharing is not None:
            self.scale_sharing = scale_sharing
        if not self.input_maxes:
            calib = Calibration(self.model, self.dataloader, self.q_func)
            if is_auto:
                calib.capture_limit = int((kwargs.get("auto_alpha_args") or {}).get("n_samples", 32) or 32)
            self.input_mins, self.input_maxes = calib.calibrate(calib_iter, op_types)
            self.same_input, self.producer = calib.same_input, calib.producer
            self.example_call = calib.example_call
            self.calibration = calib
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
        if is_auto:  # layer-wise alpha (reference transform :2377-2393)
            args = dict(kwargs.get("auto_alpha_args") or {})
            tuner = AutoAlpha(self.model, self.dataloader, self.absorb_to_layer, op_types=op_types, device=next(self.model.parameters()).device,
                              q_func=self.q_func, folding=folding, example_inputs=self.example_inputs,
                              calibration=getattr(self, "calibration", None), **args)
            alpha = tuner.tune()
            self.auto_alpha_tuner = tuner
        self.alpha = alpha
        self.weight_scale_info, self.absorb_scales_info = self._adjust_parameters(self.absorb_to_layer, input_maxes_abs, alpha)
        self.model._smoothquant_optimized = True
        return self.model
