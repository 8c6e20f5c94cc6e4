"""AWQ (activation-aware weight quantisation) on the MI355X.

Reference: neural_compressor/torch/algorithms/weight_only/awq.py
  _get_absorb_per_block :41   _get_absorb_dict :97    _get_weight_scale :131   _get_act_scale :151
  ActAwareWeightQuant   :157  quantize :201, search_scale :264, apply_scale :364, search_clip :393,
                              apply_quantize_with_clip :472, block_inference :512, module_inference :533
  AWQQuantizer          :552  prepare :567, convert :582

The search itself (20-point alpha grid for the per-input-channel scale, 10-point clip grid, loss = output MSE
over the calibration batches) is the reference's and stays in Python.  What runs on the GPU as hand-written
kernels is every per-element pass of the inner loop:
  * w_max  = mean_n(|W| / groupmax|W|)      -> inc_awq_weight_scale   (reference :131-147)
  * x_max  = mean_tokens(|X|)               -> inc_awq_act_abs_sum    (reference :151-154)
  * quant_tensor(W * s) / s                 -> inc_groupwise_quant    (reference utility.py:272)
  * sum((out_fp - out_q)^2)                 -> inc_mse_accumulate, accumulated on the device: no `.item()`
                                              round trip per batch as in the reference (:341, :454)
Forward passes of the block / module are torch (hipBLASLt) plumbing, as in the reference.

Absorb-layer discovery: the reference traces the whole model with torch.jit on the CPU (utility.py:728-984) and
silently falls back to "no absorption" when tracing fails.  Here the producer of every Linear's input is found
per block from tensor identity under forward hooks (no tracing, no CPU copy of the model) and every candidate
fold is verified numerically before it is trusted (a random rescale of the pair must leave the block output
unchanged); a candidate that fails is treated exactly like the reference treats an untraceable layer (the scale
goes into a MulLinear in front of the layer, modules.py:907).
"""

import copy
from collections import OrderedDict
from functools import partial

import os

import torch

from .... import ops
from ....common.utils import logger
from ...utils.utility import batch_broadcastable, get_accelerator, set_module
from ..base_algorithm import Quantizer
from .modules import MulLinear
from .utility import get_block_prefix, quant_tensor

__all__ = ["AWQQuantizer", "ActAwareWeightQuant"]

import contextlib
import time

# INC_MI355X_AWQ_TIMING=1: wall-clock per phase (device-synchronised) on the converted model as `awq_phase_s` -- diagnostics only
PHASE_TIMING = os.environ.get("INC_MI355X_AWQ_TIMING", "0") == "1"
PREFIX_REPLAY = True  # replay the recorded float outputs of unchanged prefix modules in the search forwards (tests compare with False)


@contextlib.contextmanager
def _phase(log, name):
    if not PHASE_TIMING:
        yield
        return
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    yield
    torch.cuda.synchronize()
    log[name] = log.get(name, 0.0) + time.perf_counter() - t0


# ---------------------------------------------------------------------------------------------------
# model plumbing (reference utility.py:636 fetch_module, :1036 replace_forward, :1079 recover_forward,
# :1098 get_module_input_output)
# ---------------------------------------------------------------------------------------------------
def fetch_module(model, op_name):
    module = model
    for name in op_name.split("."):
        if hasattr(module, name):
            module = getattr(module, name)
    return module


def replace_forward(model):
    """Capture the positional / keyword arguments of the first block for every calibration batch and abort the
    rest of the forward (the reference's ValueError control flow, utility.py:1036-1076)."""
    setattr(model, "total_block_args", [])
    setattr(model, "total_block_kwargs", [])

    def forward(layer, *args, **kwargs):
        model.total_block_args.append(list(args))
        model.total_block_kwargs.append(kwargs)
        raise ValueError

    block_prefix, _ = get_block_prefix(model)
    first_block = fetch_module(model, block_prefix)[0]
    first_block.forward_orig = first_block.forward
    first_block.forward = partial(forward, first_block)
    model.forward_orig = model.forward
    cached = model.forward

    def to_dev(obj, device):
        if isinstance(obj, torch.Tensor):
            return obj.to(device)
        if isinstance(obj, (list, tuple)):
            return type(obj)(to_dev(o, device) for o in obj)
        if isinstance(obj, dict):
            return {k: to_dev(v, device) for k, v in obj.items()}
        return obj

    def model_forward(model, *args, **kwargs):
        device = next(model.parameters()).device  # calibration batches may arrive on the host
        try:
            cached(*to_dev(args, device), **to_dev(kwargs, device))
        except ValueError:
            pass

    model.forward = partial(model_forward, model)
    return model


def recover_forward(model):
    model.forward = model.forward_orig
    block_prefix, _ = get_block_prefix(model)
    first_block = fetch_module(model, block_prefix)[0]
    first_block.forward = first_block.forward_orig
    return model


def get_module_input_output(model, module_hook_config, calib_func):
    """{module name: {"input": [tensor per batch]}} for the modules named in `module_hook_config`."""
    total = {name: {"input": [], "output": []} for name in module_hook_config}
    handles = []

    def make_hook(name, want):
        def hook(_, inputs, output):
            if "input" in want:
                total[name]["input"].append(inputs[0].detach())
            if "output" in want:
                total[name]["output"].append(output.detach() if isinstance(output, torch.Tensor) else output[0].detach())

        return hook

    for name, want in module_hook_config.items():
        handles.append(fetch_module(model, name).register_forward_hook(make_hook(name, want)))
    calib_func(model)
    for h in handles:
        h.remove()
    return total


# ---------------------------------------------------------------------------------------------------
# absorb-layer discovery (hook based, numerically verified)
# ---------------------------------------------------------------------------------------------------
def _is_norm(m):
    n = type(m).__name__
    return isinstance(m, torch.nn.LayerNorm) or (("RMSNorm" in n or "LayerNorm" in n) and getattr(m, "weight", None) is not None and m.weight.dim() == 1)


def _scale_pair(absorber, linears, s, inverse=False):
    """fold `1/s` into `absorber` and `s` into the input channels of `linears` (reference apply_scale :380-391)."""
    if inverse:
        s = 1.0 / s
    if absorber.weight.dim() == 1:
        absorber.weight.div_(s.to(absorber.weight.dtype))
    else:
        absorber.weight.div_(s.view(-1, 1).to(absorber.weight.dtype))
    if getattr(absorber, "bias", None) is not None:
        absorber.bias.div_(s.view(-1).to(absorber.bias.dtype))
    for lin in linears:
        lin.weight.mul_(s.view(1, -1).to(lin.weight.dtype))


@torch.no_grad()
def find_absorb_layers_in_block(block, args, kwargs):
    """-> ({absorber name: [linear names]}, [linear names without an absorber]); names relative to `block`.

    A Linear can hand its input scale to module P when its input tensor IS P's output (same storage), P is a
    Linear or a norm with a per-channel weight, and every module that consumes that tensor is a Linear.  Consumers
    that are not modules (functional ops) are invisible to hooks, so each candidate is verified: rescaling the pair
    by a random positive vector must reproduce the block output.
    """
    produced, consumed, lin_in = {}, {}, OrderedDict()
    handles = []

    def key(t):
        return (t.data_ptr(), tuple(t.shape), t.dtype)

    def post(name):
        def hook(mod, inputs, output):
            out = output if isinstance(output, torch.Tensor) else (output[0] if isinstance(output, (tuple, list)) and len(output) and isinstance(output[0], torch.Tensor) else None)
            if out is not None:
                produced[key(out)] = name

        return hook

    def pre(name):
        def hook(mod, inputs):
            if len(inputs) and isinstance(inputs[0], torch.Tensor):
                consumed.setdefault(key(inputs[0]), []).append(name)
                if isinstance(mod, torch.nn.Linear):
                    lin_in[name] = key(inputs[0])

        return hook

    keep = []  # keep every hooked tensor alive during the forward so that addresses are not recycled
    for name, mod in block.named_modules():
        if name == "" or next(mod.children(), None) is not None:
            continue
        handles.append(mod.register_forward_pre_hook(pre(name)))
        handles.append(mod.register_forward_hook(post(name)))
        handles.append(mod.register_forward_hook(lambda m, i, o: keep.append((i, o))))
    ref_out = block(*args, **kwargs)
    ref_out = ref_out[0] if isinstance(ref_out, (tuple, list)) else ref_out
    for h in handles:
        h.remove()
    keep.clear()

    candidates = OrderedDict()
    no_absorb = []
    for lname, k in lin_in.items():
        pname = produced.get(k)
        pmod = fetch_module(block, pname) if pname is not None else None
        ok = pmod is not None and (isinstance(pmod, torch.nn.Linear) or _is_norm(pmod))
        ok = ok and all(isinstance(fetch_module(block, c), torch.nn.Linear) for c in consumed.get(k, []))
        if ok and isinstance(pmod, torch.nn.Linear) and pmod.out_features != fetch_module(block, lname).in_features:
            ok = False
        if ok:
            candidates.setdefault(pname, []).append(lname)
        else:
            no_absorb.append(lname)
    absorb_to_layer = OrderedDict()
    tol = 2e-2 if ref_out.dtype in (torch.bfloat16, torch.float16) else 1e-3
    for pname, lnames in candidates.items():
        pmod = fetch_module(block, pname)
        lins = [fetch_module(block, n) for n in lnames]
        saved = [p.detach().clone() for p in [pmod.weight] + ([pmod.bias] if getattr(pmod, "bias", None) is not None else []) + [l.weight for l in lins]]
        g = torch.Generator(device="cpu").manual_seed(0)
        s = (0.5 + torch.rand(lins[0].in_features, generator=g)).to(pmod.weight.device, torch.float32)
        _scale_pair(pmod, lins, s)
        out = block(*args, **kwargs)
        out = out[0] if isinstance(out, (tuple, list)) else out
        err = float((out.float() - ref_out.float()).norm() / (ref_out.float().norm() + 1e-20))
        targets = [pmod.weight] + ([pmod.bias] if getattr(pmod, "bias", None) is not None else []) + [l.weight for l in lins]
        for t, sv in zip(targets, saved):
            t.copy_(sv)
        if err <= tol:
            absorb_to_layer[pname] = lnames
        else:
            logger.debug("AWQ: %s -> %s failed the fold check (rel err %.2e); using self-absorption", pname, lnames, err)
            no_absorb.extend(lnames)
    return absorb_to_layer, no_absorb


# ---------------------------------------------------------------------------------------------------
# statistics
# ---------------------------------------------------------------------------------------------------
@torch.no_grad()
def _get_weight_scale(weight, q_group_size=-1):
    """mean over rows of |w| / max|w| of the group (reference :131-147) -> [K] in weight.dtype."""
    K = weight.shape[1]
    gs = q_group_size if q_group_size and q_group_size > 0 else K
    if K % gs != 0:  # the reference's `view(-1, q_group_size)` needs it too
        raise ValueError(f"AWQ weight scale needs in_features ({K}) divisible by group_size ({gs})")
    return ops.awq_weight_scale(weight, gs)


@torch.no_grad()
def _get_act_scale(input_val):
    """mean over all calibration tokens of |x| (reference :151-154) -> [K] in the activation dtype."""
    K = input_val[0].shape[-1]
    acc = torch.zeros(K, dtype=torch.float32, device=input_val[0].device)
    tokens = 0
    for x in input_val:
        x2 = x.reshape(-1, K)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        ops.awq_act_abs_sum(x2, acc)
        tokens += x2.shape[0]
    return (acc / tokens).to(input_val[0].dtype)


class _Loss:
    """sum over batches of mean((a-b)^2), kept on the device (reference: `.item()` per batch, added up as Python
    doubles): fp64, fixed summation order (inc_mse_accumulate), so the argmin over a grid is reproducible."""

    def __init__(self, device):
        self.sum = torch.zeros(1, dtype=torch.float64, device=device)
        self.val = torch.zeros(1, dtype=torch.float64, device=device)

    def add(self, a, b, n=1):
        """`n`: how many equally sized calibration batches `a` / `b` hold (stacked along dim 0 by the batched search):
        n * mean over the stack == the sum of the per-batch means."""
        a = a if a.is_contiguous() else a.contiguous()
        b = b if b.is_contiguous() else b.contiguous()
        self.sum.zero_()
        ops.mse_accumulate(a, b, out=self.sum)
        self.val += self.sum * (n / a.numel())


# ---------------------------------------------------------------------------------------------------
# the algorithm
# ---------------------------------------------------------------------------------------------------
class ActAwareWeightQuant:
    """Activation-aware weight quantisation (interface of reference awq.py:157)."""

    def __init__(self, model, example_inputs=None, data_type="int", bits=4, group_size=32, scheme="asym",
                 use_full_range=False, weight_config={}, total_block_args=[], total_block_kwargs=[], device="auto",
                 absorb_layer_dict={}):
        self.example_inputs = example_inputs
        self.model = model
        self.device = torch.device(get_accelerator(device).current_device_name())
        self.model.to(self.device)
        self.total_block_args = total_block_args
        self.total_block_kwargs = total_block_kwargs
        self.block_prefix, self.block_num = get_block_prefix(model)
        self.block_list = fetch_module(model, self.block_prefix)
        self.data_type = data_type
        self.bits = bits
        self.group_size = group_size
        self.scheme = scheme
        self.use_full_range = use_full_range
        self.weight_config = weight_config
        self.absorb_layer_dict = absorb_layer_dict
        # what the two grid searches saw: {"scale": {"name|name": (loss history, chosen index)}, "clip": {name: (...)}};
        # left on the converted model as `awq_search_log` (diagnostics / parity tests; the reference only logs them)
        self.search_log = {"scale": {}, "clip": {}}

    # -- absorb bookkeeping --------------------------------------------------------------------------
    def _absorb_for_block(self, i, folding):
        """[(tuple of full linear names sharing a scale)], {tuple: full absorber name} for block i."""
        block_name = f"{self.block_prefix}.{i}"
        block = fetch_module(self.model, block_name)
        tuples, inverse = [], {}
        if self.absorb_layer_dict:  # user-provided {absorber: absorbed or [absorbed]} relative to the block (:97-126)
            for k, v in self.absorb_layer_dict.items():
                names = (f"{block_name}.{v}",) if isinstance(v, str) else tuple(f"{block_name}.{vv}" for vv in v)
                tuples.append(names)
                inverse[names] = f"{block_name}.{k}"
            return tuples, inverse
        absorb_to_layer, no_absorb = find_absorb_layers_in_block(block, self.total_block_args[0], self.total_block_kwargs[0])

        def skipped(name):  # layers excluded from AWQ (dtype fp32) neither get nor give a scale (:61-75)
            cfg = self.weight_config.get(name)
            return cfg is not None and cfg.get("dtype") == "fp32"

        for absorber, linears in absorb_to_layer.items():
            names = tuple(f"{block_name}.{n}" for n in linears)
            if any(skipped(n) for n in names):
                continue
            tuples.append(names)
            inverse[names] = f"{block_name}.{absorber}"
        if not folding:
            for n in no_absorb:
                full = f"{block_name}.{n}"
                if skipped(full):
                    continue
                tuples.append((full,))
                inverse[(full,)] = full
        return tuples, inverse

    def _cfg(self, name):
        if name in self.weight_config:
            c = self.weight_config[name]
            return c["dtype"], c["bits"], c["group_size"], c["scheme"]
        return self.data_type, self.bits, self.group_size, self.scheme

    # -- main loop (reference :201-262) ----------------------------------------------------------------
    @torch.no_grad()
    def quantize(self, use_auto_scale=True, use_mse_search=True, folding=False, return_int=False):
        self.absorb_of = {}
        ph = self.phase_s = {}
        for i in range(self.block_num):
            logger.info("Processing block: %d/%d", i + 1, self.block_num)
            with _phase(ph, "absorb_discovery"):
                module_list, inverse = self._absorb_for_block(i, folding if use_auto_scale else False)
            self.absorb_of.update(inverse)
            if len(module_list) == 0:
                logger.info("No need to process this block.")
                out_list = self.block_inference(fetch_module(self.model, f"{self.block_prefix}.{i}"))
                self.update_block_input(out_list)
                continue
            block_name = f"{self.block_prefix}.{i}"
            block = fetch_module(self.model, block_name)
            hook_cfg = {v[0].split(block_name + ".")[1]: ["input"] for v in module_list}

            def block_calibration(m):
                for args, kwargs in zip(self.total_block_args, self.total_block_kwargs):
                    m(*args, **kwargs)

            with _phase(ph, "capture"):
                input_values = get_module_input_output(block, hook_cfg, calib_func=block_calibration)
            with _phase(ph, "search_scale"):
                scale_info = self.search_scale(block, block_name, module_list, input_values) if use_auto_scale else {}
            with _phase(ph, "block_inference"):
                out_list = self.block_inference(block)  # inputs of the next block come from the UNSCALED float block (:246-249)
                self.update_block_input(out_list)
            if use_auto_scale:
                self.apply_scale(scale_info)
            if use_mse_search:
                with _phase(ph, "search_clip"):
                    self.search_clip(block_name, module_list, input_values)
        with _phase(ph, "final_rtn"):
            self.apply_quantize_with_clip(return_int)
        self.model.awq_search_log = self.search_log
        if PHASE_TIMING:
            self.model.awq_phase_s = {k: round(v, 4) for k, v in ph.items()}
        return self.model

    # -- scale search (reference :264-361) ---------------------------------------------------------------
    def search_scale(self, block, block_name, module_list, input_values):
        scale_info = {}
        logger.info("Searching best scales with AWQ algorithm")
        for module_tuple in module_list:
            cur_dtype, cur_bits, cur_group_size, cur_scheme = self._cfg(module_tuple[0])
            if cur_bits < 0:
                continue
            logger.info("[SCALE] Processing module: %s", module_tuple)
            names = [n.split(block_name + ".")[1] for n in module_tuple]
            if PHASE_TIMING:
                torch.cuda.synchronize()
                t_tuple = time.perf_counter()
            mods = OrderedDict((n, fetch_module(block, n)) for n in names)
            weight = torch.cat([m.weight for m in mods.values()], dim=0)
            w_max = _get_weight_scale(weight, q_group_size=cur_group_size)
            del weight
            input_val = input_values[names[0]]["input"]
            x_max = _get_act_scale(input_val)
            org_w = {n: m.weight.detach().clone() for n, m in mods.items()}
            multi = len(module_tuple) > 1
            evaluate = (lambda: self._search_block_outputs(block)) if multi else (lambda: self._search_module_outputs(mods[names[0]], input_val))
            # (all tuples of a block are searched on the same float block: its outputs are computed once, :304-310 / :246-249)
            org_out = self._float_block_outputs(block) if multi else evaluate()
            replay = self._prefix_replay(block, list(mods.values())) if multi else None
            n_grid = 20
            losses, cand = [], []
            for step in range(n_grid):
                ratio = step * 1 / n_grid
                scales = (x_max.pow(ratio) / w_max.pow(1 - ratio)).clamp(min=1e-4).view(-1)
                scales = scales / (scales.max() * scales.min()).sqrt()
                for n, m in mods.items():
                    wq = m.weight.data.mul(scales.view(1, -1)).contiguous()
                    # the reference passes `data_type=` / `num_bits=`, which quant_tensor swallows: the search always
                    # quantises as 4-bit integers whatever the configured width (awq.py:328-335)
                    wq = quant_tensor(wq, data_type=cur_dtype, num_bits=cur_bits, group_size=cur_group_size, scheme=cur_scheme,
                                      full_range=self.use_full_range)
                    m.weight.data = wq / scales.view(1, -1)
                loss = _Loss(self.device)
                outs = replay.run(evaluate) if replay is not None else evaluate()
                for (o1, n1), (o2, _) in zip(org_out, outs):
                    loss.add(o1, o2, n1)
                del outs
                losses.append(loss.val)
                cand.append(scales)
                for n, m in mods.items():
                    m.weight.data = org_w[n].clone()
            if replay is not None:
                replay.close()
            hist = torch.cat(losses).tolist()  # one device->host copy per module tuple
            if PHASE_TIMING:
                self.phase_s["scale:" + "|".join(names)] = time.perf_counter() - t_tuple
            best, best_i = float("inf"), None
            for i_, v in enumerate(hist):
                if v < best:  # first strict minimum, like the reference's scan
                    best, best_i = v, i_
            self.search_log["scale"]["|".join(module_tuple)] = (hist, best_i)
            assert best_i is not None, "Loss is infinity! Cannot find the correct scale."
            best_scales = cand[best_i].view(-1)
            assert torch.isnan(best_scales).sum() == 0, best_scales
            scale_info[module_tuple] = best_scales.detach()
            logger.debug("The loss history of different scale: %s", hist)
            logger.info("The best scale alpha of %s: %s", module_tuple, best_i / n_grid)
        return scale_info

    # -- apply (reference :364-391) -----------------------------------------------------------------------
    @torch.no_grad()
    def apply_scale(self, scale_info):
        for module_tuple, scale in scale_info.items():
            assert module_tuple in self.absorb_of, "cannot find the absorb module."
            absorb_name = self.absorb_of[module_tuple]
            absorb_module = fetch_module(self.model, absorb_name)
            if absorb_name == module_tuple[0]:  # self-absorption: a multiplier in front of the layer
                new_module = MulLinear(absorb_module, (1.0 / scale).to(absorb_module.weight.dtype))
                new_module._update_linear()
                set_module(self.model, absorb_name, new_module)
            else:
                _scale_pair(absorb_module, [fetch_module(self.model, n) for n in module_tuple], scale)

    # -- clip search (reference :393-470) -------------------------------------------------------------------
    def search_clip(self, block_name, module_list, input_values):
        logger.info("Searching the best clip range with AWQ algorithm")
        for module_tuple in module_list:
            input_val = input_values[module_tuple[0].split(block_name + ".")[1]]["input"]
            for module_name in module_tuple:
                cur_dtype, cur_bits, cur_group_size, cur_scheme = self._cfg(module_name)
                if cur_bits < 0:
                    continue
                logger.info("[CLIP] Processing module: %s", module_name)
                module = fetch_module(self.model, module_name)
                org_w = module.weight.detach().clone()
                org_out = self._search_module_outputs(module, input_val)
                n_grid, max_shrink = 100, 0.1
                losses, ratios = [], []
                for i_s in range(int(max_shrink * n_grid)):
                    ratio = 1 - i_s / n_grid
                    wq = module.weight.data.contiguous()
                    wq = quant_tensor(wq, data_type=cur_dtype, num_bits=cur_bits, group_size=cur_group_size, scheme=cur_scheme,
                                      full_range=self.use_full_range, quantile=ratio)
                    if isinstance(module, MulLinear):
                        module.linear.weight.data = wq
                    else:
                        module.weight.data = wq
                    loss = _Loss(self.device)
                    for (o1, n1), (o2, _) in zip(org_out, self._search_module_outputs(module, input_val)):
                        loss.add(o1, o2, n1)
                    losses.append(loss.val)
                    ratios.append(ratio)
                    if isinstance(module, MulLinear):
                        module.linear.weight.data = org_w.clone()
                    else:
                        module.weight.data = org_w.clone()
                hist = torch.cat(losses).tolist()
                best, best_ratio = float("inf"), None
                for r_, v in zip(ratios, hist):
                    if v < best:
                        best, best_ratio = v, r_
                self.search_log["clip"][module_name] = (hist, ratios.index(best_ratio))
                logger.debug("The loss history of different clip range: %s", hist)
                if module_name not in self.weight_config:
                    self.weight_config[module_name] = {"bits": cur_bits, "group_size": cur_group_size, "scheme": cur_scheme}
                self.weight_config[module_name]["quantile"] = best_ratio
                if isinstance(module, MulLinear):
                    self.weight_config[module_name + ".linear"] = self.weight_config[module_name]
                    self.weight_config.pop(module_name)
                logger.debug("The best clip ratio for %s: %s", module_name, best_ratio)

    # -- final RTN with the searched clip (reference :472-493) ---------------------------------------------
    def apply_quantize_with_clip(self, return_int=False):
        logger.info("Quantizing the AWQ optimized fp32 model")
        from .rtn import RTNQuantizer

        # layers wrapped in MulLinear without a clip search keep their config under the inner Linear's name
        for name, mod in list(self.model.named_modules()):
            if isinstance(mod, MulLinear) and name in self.weight_config:
                self.weight_config[name + ".linear"] = self.weight_config.pop(name)
        for cfg in self.weight_config.values():
            cfg.setdefault("group_dim", 1)
            cfg.setdefault("use_full_range", self.use_full_range)
            cfg.setdefault("use_mse_search", False)
            cfg["use_mse_search"] = False  # the clip ratio is already in cfg["quantile"]
        rtn = RTNQuantizer(quant_config=self.weight_config)
        self.model = rtn.quantize(self.model, bits=self.bits, group_size=self.group_size, scheme=self.scheme,
                                  return_int=return_int, use_full_range=self.use_full_range)
        logger.info("AWQ quantization is done.")

    # -- forwards ------------------------------------------------------------------------------------------
    def update_block_input(self, input_list):
        self._block_chunks = None  # the stacked copies used by the batched search belong to the previous block
        self._float_block = None
        for i, inp in enumerate(input_list):
            if len(self.total_block_args[i]) > 0:
                self.total_block_args[i][0] = inp
            elif "hidden_states" in self.total_block_kwargs[i]:
                self.total_block_kwargs[i]["hidden_states"] = inp
            else:  # pragma: no cover
                assert False, "cannot find hidden_states position for next block"

    def block_inference(self, model):
        total_out = []
        for args, kwargs in zip(self.total_block_args, self.total_block_kwargs):
            if kwargs.get("layer_past", None) is not None:
                kwargs["layer_past"] = None
            out = model(*args, **kwargs)
            if isinstance(out, tuple):
                out = out[0]
            total_out.append(out)
        return total_out

    # -- batched evaluation for the grid searches ----------------------------------------------------------------------
    # The searches only compare losses, and a Linear (or a whole decoder block whose keyword arguments are the same for
    # every calibration batch) treats the rows of a stacked input independently, so the ~20 forwards per grid are run on
    # stacks of `search_batch` calibration batches (INC_MI355X_AWQ_SEARCH_BATCH, default 32; 1 = the reference's
    # one-batch-at-a-time loop): larger GEMMs, 32x fewer launches, same sums.  The activations handed to the next block
    # still come from the per-batch `block_inference`.
    @property
    def search_batch(self):
        return max(1, int(os.environ.get("INC_MI355X_AWQ_SEARCH_BATCH", "32")))

    def _stack(self, tensors):
        B = self.search_batch
        if B <= 1 or len(tensors) < 2 or any(t.shape != tensors[0].shape for t in tensors):
            return [(t, 1) for t in tensors]
        return [(torch.cat(tensors[i : i + B], dim=0), len(tensors[i : i + B])) for i in range(0, len(tensors), B)]

    def _search_module_outputs(self, module, inputs):
        cache = self.__dict__.setdefault("_module_chunks", {})
        key = id(inputs)
        if key not in cache or cache[key][0] is not inputs:
            cache.clear()  # one input list at a time: the stacks of the previous module tuple are dropped
            cache[key] = (inputs, self._stack(inputs))
        outs = []
        for x, n in cache[key][1]:
            out = module(x)
            outs.append((out[0] if isinstance(out, tuple) else out, n))
        return outs

    @staticmethod
    def _same_kwargs(a, b):
        if a.keys() != b.keys():
            return False
        for k in a:
            va, vb = a[k], b[k]
            if va is vb:
                continue
            if isinstance(va, torch.Tensor) and isinstance(vb, torch.Tensor):
                if va.shape != vb.shape or not torch.equal(va, vb):
                    return False
            elif isinstance(va, (tuple, list)) and isinstance(vb, (tuple, list)) and len(va) == len(vb):
                for xa, xb in zip(va, vb):
                    if isinstance(xa, torch.Tensor) and isinstance(xb, torch.Tensor):
                        if xa.shape != xb.shape or not torch.equal(xa, xb):
                            return False
                    elif xa != xb:
                        return False
            elif va != vb:
                return False
        return True

    def _block_chunk_list(self, block):
        chunks = getattr(self, "_block_chunks", None)
        if chunks is None:
            args0, kw0 = self.total_block_args[0], self.total_block_kwargs[0]
            hidden_in_args = len(args0) > 0
            get = (lambda a, k: a[0]) if hidden_in_args else (lambda a, k: k["hidden_states"])
            stackable = self.search_batch > 1 and len(self.total_block_args) > 1 and all(
                len(a) == len(args0) and get(a, k).shape == get(args0, kw0).shape and get(a, k).shape[0] == 1
                and all(x is y or (isinstance(x, torch.Tensor) and isinstance(y, torch.Tensor) and x.shape == y.shape and torch.equal(x, y))
                        for x, y in zip(a[1:], args0[1:]))
                and self._same_kwargs({kk: v for kk, v in k.items() if kk != "hidden_states"},
                                      {kk: v for kk, v in kw0.items() if kk != "hidden_states"})
                for a, k in zip(self.total_block_args, self.total_block_kwargs))
            # every other argument must also broadcast over the batch (leading dimension 1): batch-folded arguments such
            # as Bloom / Falcon / MPT `alibi` [batch*heads, 1, T] do not, and such models keep the reference's
            # one-batch-per-forward loop
            stackable = stackable and batch_broadcastable(list(args0[1:])) and batch_broadcastable(
                {kk: v for kk, v in kw0.items() if kk != "hidden_states"})
            if stackable:
                hs = [get(a, k) for a, k in zip(self.total_block_args, self.total_block_kwargs)]
                chunks = [("stack", x, n) for x, n in self._stack(hs)]
                if not self._stacked_block_is_faithful(block, chunks[0], hidden_in_args):
                    logger.warning("AWQ: a stacked forward of this block does not reproduce its per-batch outputs; "
                                   "searching with one calibration batch per forward")
                    stackable = False
            if not stackable:
                chunks = [("single", i, 1) for i in range(len(self.total_block_args))]
            self._block_chunks = chunks
        return chunks

    _chunk_cursor = -1  # index of the chunk whose forward is running (read by the hooks of _float_block_outputs / _PrefixReplay)

    def _search_block_outputs(self, block):
        chunks = self._block_chunk_list(block)
        outs = []
        for ci, (kind, x, n) in enumerate(chunks):
            self._chunk_cursor = ci
            if kind == "single":
                args, kwargs = self.total_block_args[x], self.total_block_kwargs[x]
            else:
                args0, kw0 = self.total_block_args[0], self.total_block_kwargs[0]
                if len(args0) > 0:
                    args, kwargs = [x] + list(args0[1:]), kw0
                else:
                    args, kwargs = args0, dict(kw0, hidden_states=x)
            if kwargs.get("layer_past", None) is not None:
                kwargs["layer_past"] = None
            out = block(*args, **kwargs)
            outs.append((out[0] if isinstance(out, tuple) else out, n))
        self._chunk_cursor = -1
        return outs

    # -- float outputs of the block, once per block; replay of the part of the block in front of the searched Linears ------
    # Reference: every module tuple's search first runs the float block (`org_out`, :304-310) and then the whole block once
    # per grid point (:337).  The float block is the same for every tuple of a block (scales are applied after all searches,
    # :236-252), so its outputs are computed once; and the direct children of the block that finish BEFORE the first searched
    # Linear starts (for gate/up: input_layernorm, self_attn, post_attention_layernorm) see the same inputs and weights at
    # every grid point, so their recorded float outputs are replayed instead of being recomputed -- the same kernels on the
    # same data would reproduce them bit for bit (checked once per set of replayed children against a full forward).
    # (PREFIX_REPLAY = False runs every forward in full: the tests' A/B partner.)
    def _float_block_outputs(self, block):
        fb = getattr(self, "_float_block", None)
        if fb is not None and fb["block"] is block:
            return fb["outs"]
        self._block_chunk_list(block)  # (its one-off stacking check runs forwards of its own: before the hooks go on)
        events, rec, handles = [], {}, []
        children = dict(block.named_children())
        leaves = {n: m for n, m in block.named_modules() if n and len(list(m.children())) == 0}

        def child_pre(name):
            def hook(mod, args, kwargs):
                if self._chunk_cursor == 0:
                    events.append(("start", name))
            return hook

        def child_post(name):
            def hook(mod, args, kwargs, output):
                if self._chunk_cursor == 0:
                    events.append(("end", name))
                rec.setdefault(name, []).append((output, _versions(output)))
            return hook

        def leaf_pre(name):
            def hook(mod, args):
                if self._chunk_cursor == 0:
                    events.append(("leaf", name))
            return hook

        for n, m in children.items():
            handles.append(m.register_forward_pre_hook(child_pre(n), with_kwargs=True))
            handles.append(m.register_forward_hook(child_post(n), with_kwargs=True))
        for n, m in leaves.items():
            if n not in children:
                handles.append(m.register_forward_pre_hook(leaf_pre(n)))
        try:
            outs = self._search_block_outputs(block)
        finally:
            for h in handles:
                h.remove()
        # a child whose output was modified in place later in the forward cannot be replayed
        rec = {n: [o for o, _ in lst] for n, lst in rec.items() if all(_versions(o) == v for o, v in lst)}
        # ... and a child behind which no leaf module starts can never be part of a replayed prefix (the block's last child, e.g. the
        # MLP): its recorded outputs -- one multi-GB tensor per stacked chunk at the BASELINE calibration size -- are dropped here
        last_leaf = max((i for i, (kind, _) in enumerate(events) if kind == "leaf"), default=-1)
        ends = {n: i for i, (kind, n) in enumerate(events) if kind == "end"}
        rec = {n: lst for n, lst in rec.items() if ends.get(n, len(events)) < last_leaf}
        self._float_block = dict(block=block, outs=outs, events=events, rec=rec, leaves={id(m): n for n, m in leaves.items()})
        return outs

    def _prefix_replay(self, block, changed):
        fb = getattr(self, "_float_block", None)
        if not PREFIX_REPLAY or fb is None or fb["block"] is not block:
            return None
        changed_names = {fb["leaves"].get(id(m)) for m in changed}
        if None in changed_names:
            return None
        first = next((i for i, (kind, n) in enumerate(fb["events"])
                      if (kind == "leaf" and n in changed_names) or (kind == "start" and n in changed_names)), None)
        if first is None:
            return None
        n_chunks = len(self._block_chunks)
        started = [n for kind, n in fb["events"][:first] if kind == "start"]
        ended = [n for kind, n in fb["events"][:first] if kind == "end"]
        names = [n for n in ended if started.count(n) == 1 and ended.count(n) == 1 and len(fb["rec"].get(n, ())) == n_chunks
                 and sum(1 for k, nn in fb["events"] if k == "start" and nn == n) == 1]
        # replaying a child makes the children nested in front of it irrelevant; keep only children with real work
        names = [n for n in names if any(True for _ in getattr(block, n).parameters())]
        if not names:
            return None
        return _PrefixReplay(self, block, names, fb["rec"])

    def _stacked_block_is_faithful(self, block, chunk, hidden_in_args):
        """The first stacked forward against the per-batch forwards of the same batches (guards against blocks that are
        not row-independent; compared up to GEMM-shape rounding)."""
        _, x, n = chunk
        if n <= 1:
            return True
        args0, kw0 = self.total_block_args[0], self.total_block_kwargs[0]

        def run(h):
            if hidden_in_args:
                out = block(*([h] + list(args0[1:])), **kw0)
            else:
                out = block(*args0, **dict(kw0, hidden_states=h))
            return (out[0] if isinstance(out, tuple) else out).float()

        try:
            got = run(x)
        except Exception as e:
            logger.warning("AWQ: stacked block forward failed (%s)", e)
            return False
        ref = torch.cat([run(x[i : i + 1]) for i in range(n)], dim=0)
        if got.shape != ref.shape:
            return False
        tol = 1e-4 if x.dtype == torch.float32 else 3e-2
        return bool((got - ref).norm() <= tol * ref.norm().clamp_min(1e-30))

    def module_inference(self, model, inputs):
        total_out = []
        for inp in inputs:
            out = model(inp)
            if isinstance(out, tuple):
                out = out[0]
            total_out.append(out)
        return total_out


def _tensors_of(obj):
    if isinstance(obj, torch.Tensor):
        yield obj
    elif isinstance(obj, (tuple, list)):
        for o in obj:
            yield from _tensors_of(o)
    elif isinstance(obj, dict):
        for o in obj.values():
            yield from _tensors_of(o)


def _versions(obj):
    return tuple(t._version for t in _tensors_of(obj))


class _PrefixReplay:
    """Replays the recorded float outputs of `names` (direct children of `block`) during the grid forwards of one module tuple."""

    def __init__(self, owner, block, names, rec):
        self.owner, self.block, self.names, self.rec = owner, block, names, rec
        # checked once per quantiser run (= per model) and set of replayed children against full forwards: another model with the same
        # block class gets its own check
        self.verified = owner.__dict__.setdefault("_replay_verified", set())
        self.key = (type(block).__qualname__, tuple(names))
        self.versions = {n: [_versions(o) for o in rec[n]] for n in names}
        self.saved = None
        self.broken = False

    def _patch(self):
        self.saved = {}
        for n in self.names:
            mod = getattr(self.block, n)
            self.saved[n] = mod.__dict__.get("forward")
            mod.forward = (lambda lst: (lambda *a, **k: lst[self.owner._chunk_cursor]))(self.rec[n])

    def _unpatch(self):
        if self.saved is None:
            return
        for n, old in self.saved.items():
            mod = getattr(self.block, n)
            if old is None:
                mod.__dict__.pop("forward", None)
            else:
                mod.forward = old
        self.saved = None

    def _intact(self):
        return all(_versions(o) == v for n in self.names for o, v in zip(self.rec[n], self.versions[n]))

    def run(self, evaluate):
        if self.broken:
            return evaluate()
        self._patch()
        try:
            outs = evaluate()
        finally:
            self._unpatch()
        ok = self._intact()
        if ok and self.key not in self.verified:
            full = evaluate()
            ok = len(full) == len(outs) and all(torch.equal(a, b) for (a, _), (b, _) in zip(full, outs))
            if ok:
                self.verified.add(self.key)
        if not ok:
            logger.warning("AWQ: replaying %s does not reproduce the full forward of %s; running the grid forwards in full",
                           self.names, self.key[0])
            self.broken = True
            return evaluate()
        return outs

    def close(self):
        self._unpatch()


class AWQQuantizer(Quantizer):
    """Algorithm plug-in (reference awq.py:552)."""

    def __init__(self, quant_config: OrderedDict = {}, absorb_layer_dict: dict = {}):
        super().__init__(quant_config)
        self.absorb_layer_dict = absorb_layer_dict

    @torch.no_grad()
    def prepare(self, model, *args, **kwargs):
        assert isinstance(model, torch.nn.Module), "AWQ algorithm only supports torch module"
        device = torch.device(get_accelerator(kwargs.get("device", "auto")).current_device_name())
        model.to(device)  # the calibration forwards run on the GPU: the captured block inputs are born in HBM
        return replace_forward(model)

    @torch.no_grad()
    def convert(self, model, bits=4, group_size=32, scheme="asym", example_inputs=None, use_auto_scale=True,
                use_mse_search=True, folding=False, return_int=False, use_full_range=False, data_type="int", *args, **kwargs):
        model = recover_forward(model)
        total_block_args = getattr(model, "total_block_args", [])
        total_block_kwargs = getattr(model, "total_block_kwargs", [])
        delattr(model, "total_block_args")
        delattr(model, "total_block_kwargs")
        assert len(total_block_args) > 0, "AWQ needs calibration: run the model on calibration data between prepare() and convert()"
        awq = ActAwareWeightQuant(
            model, example_inputs=example_inputs, data_type=data_type, bits=bits, group_size=group_size, scheme=scheme,
            use_full_range=use_full_range, weight_config=self.quant_config, total_block_args=total_block_args,
            total_block_kwargs=total_block_kwargs, absorb_layer_dict=self.absorb_layer_dict,
        )
        return awq.quantize(use_auto_scale=use_auto_scale, use_mse_search=use_mse_search, folding=folding, return_int=return_int)
