"""On-disk formats of a weight-only-quantised model.

Reference: neural_compressor/torch/algorithms/weight_only/save_load.py
  save :56-108   load :111-143   WOQModelLoader :146 (load_inc_format_woq_model :207, load_hf_format_woq_model :329,
  _build_woq_model :420, _replace_woqlinear_modules :476, _load_data_to_new_module :527)   change_config_to_hf_format :1094
  neural_compressor/common/utils/save_load.py:23-60 (save_config_mapping / load_config_mapping)

Two formats, byte-compatible with the reference's because the packed module keeps the reference's buffer names, shapes
and dtypes (`qweight [K/8,N] int32`, `scales [K/g,N] fp16`, `qzeros [K/g,N/8] int32`, optional `g_idx`, `bias`):
  "default"      quantized_weight.pt (torch.save of the state_dict) + qconfig.json
                 ({"('op_name', 'op_type')": {algo: config.to_dict()}})
  "huggingface"  save_pretrained() safetensors + quantize_config.json in the AutoGPTQ vocabulary
                 (bits, group_size, damp_percent, desc_act, sym, true_sequential, quant_method="gptq")
Loading rebuilds `MI355XWeightOnlyLinear` modules (wrapped in `MulLinear` where the checkpoint carries an AWQ
`input_scale`) directly in HBM; no dense weight is ever materialised and nothing is re-packed: the buffers are the kernel
format.  Only local directories are supported (the reference also downloads from the hub; there is no network here).
"""

import json
import os
import re

import torch

from .... import ops
from ....common.utils import logger
from ...utils.utility import set_module
from .modules import MI355XWeightOnlyLinear, MulLinear

WEIGHT_NAME = "quantized_weight.pt"  # reference torch/utils/utility.py:56
QCONFIG_NAME = "qconfig.json"        # reference torch/utils/utility.py:60
HF_QCONFIG_NAME = "quantize_config.json"
LAYOUT_NAME = "mi355x_woq_layout.json"  # layout of modules saved in the non-optimum format (NF4 / FP4, use_optimum_format=False)
LM_HEAD_NAMES = [".*lm_head", ".*output_layer", ".*embed_out"]  # reference torch/utils/constants.py:69

# device string -> packed Linear class (the reference's plug-in seam, save_load.py:50)
device_woqlinear_mapping = {"cuda": MI355XWeightOnlyLinear}


def save_config_mapping(config_mapping, qconfig_file_path):
    """{"('name', 'type')": {algo_name: config.to_dict()}} (reference common/utils/save_load.py:23-36)."""
    per_op = {}
    for (op_name, op_type), op_config in config_mapping.items():
        per_op[str((op_name, op_type))] = {op_config.name: op_config.to_dict()}
    with open(qconfig_file_path, "w") as f:
        json.dump(per_op, f, indent=4)


def change_config_to_hf_format(config_mappings):
    """One model-level AutoGPTQ-style dict; every quantised module must share the settings (reference :1094-1156)."""
    out = {
        "bits": 4, "group_size": 128, "damp_percent": 0.01, "desc_act": True, "sym": True, "true_sequential": True,
        "model_name_or_path": None, "model_file_base_name": "model", "quant_method": "gptq",
    }
    first = None
    for (name, _type), cfg in config_mappings.items():
        if any(re.match(p, name) for p in LM_HEAD_NAMES):
            if cfg.dtype != "fp32":
                raise ValueError(f"{name} should not be quantized if you want to save in huggingface format.")
            continue
        cur = dict(bits=cfg.bits, group_size=cfg.group_size, sym=cfg.use_sym, damp_percent=getattr(cfg, "percdamp", 0),
                   desc_act=getattr(cfg, "act_order", False), true_sequential=getattr(cfg, "true_sequential", False))
        if first is None:
            first = cur
        else:
            for k in ("bits", "group_size", "sym"):
                assert first[k] == cur[k], f"{k} should be the same for all modules, got {first[k]} and {cur[k]}."
    if first:
        out.update(first)
    return out


def save(model, output_dir="./saved_results", format="default", **kwargs):
    """Save the quantised model and its configuration (reference save_load.py:56-108)."""
    fmt = getattr(format, "value", format)
    # modules in the reference's NON-optimum layout (NF4 / FP4 always, integer formats with use_optimum_format=False): the default
    # format keeps their layout in a side file the loader reads (the reference's loader infers nothing either: it rebuilds with the
    # flags of its own process); the huggingface format IS the optimum (AutoGPTQ) layout and cannot hold them
    layout = {}
    for name, mod in model.named_modules():
        if isinstance(mod, MI355XWeightOnlyLinear) and not getattr(mod, "use_optimum_format", True):
            if fmt == "huggingface":
                raise ValueError(f"{name} was packed with use_optimum_format=False; the huggingface format is the optimum layout")
            layout[name] = dict(use_optimum_format=False, compression_dim=mod.compression_dim, compression_dtype=str(mod.compression_dtype).replace("torch.", ""),
                                scale_dtype=str(mod.float_type).replace("torch.", ""), dtype=mod.dtype, bits=mod.bits, group_size=mod.group_size)
    os.makedirs(output_dir, exist_ok=True)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if fmt == "huggingface":
        if not hasattr(model.config, "quantization_config") or model.config.quantization_config is None:
            qc = change_config_to_hf_format(model.qconfig)
            model.config.quantization_config = qc
        else:
            qc = model.config.quantization_config
        model.save_pretrained(output_dir, max_shard_size=kwargs.get("max_shard_size", "5GB"),
                              safe_serialization=kwargs.get("safe_serialization", True))
        with open(os.path.join(output_dir, HF_QCONFIG_NAME), "w", encoding="utf-8") as f:
            json.dump(qc if isinstance(qc, dict) else qc.to_dict(), f, indent=2)
        if getattr(model, "generation_config", None) is not None:
            model.generation_config.save_pretrained(output_dir)
        if kwargs.get("tokenizer") is not None:
            kwargs["tokenizer"].save_pretrained(output_dir)
        return
    if fmt != "default":
        raise ValueError(f"unknown save format {format!r}: expected 'default' or 'huggingface'")
    folder = os.path.abspath(os.path.expanduser(output_dir))
    save_config_mapping(model.qconfig, os.path.join(folder, QCONFIG_NAME))
    if layout:
        with open(os.path.join(folder, LAYOUT_NAME), "w", encoding="utf-8") as f:
            json.dump(layout, f, indent=1)
    state = {k: v.detach().cpu() for k, v in model.state_dict().items()}  # device-neutral file, like the reference's CPU model
    torch.save(state, os.path.join(folder, WEIGHT_NAME))
    logger.info("Save quantized model weight to %s.", os.path.join(folder, WEIGHT_NAME))
    logger.info("Save configuration of quantized model to %s.", os.path.join(folder, QCONFIG_NAME))


# ---------------------------------------------------------------------------------------------------------------------
# loading
# ---------------------------------------------------------------------------------------------------------------------
def _module_config(quantization_config, name, module):
    """Per-module dict from an INC qconfig.json ({"('name', 'Linear')": {algo: {...}}}) or the HF model-level dict."""
    pattern = rf"(\(.*{re.escape(name)}.*{re.escape(type(module).__name__)}.*\))"
    for key, value in quantization_config.items():
        if isinstance(value, dict) and re.search(pattern, key):
            return next(iter(value.values()))
    return quantization_config


def _build(original_model, state, quantization_config, device, layout=None):
    keys = set(state.keys())
    layout = layout or {}
    try:
        from transformers import Conv1D
    except Exception:  # pragma: no cover
        Conv1D = ()
    for name, module in list(original_model.named_modules()):
        if not isinstance(module, (torch.nn.Linear, Conv1D) if Conv1D else torch.nn.Linear):
            continue
        if name + ".qweight" not in keys and name + ".linear.qweight" not in keys:
            continue  # not quantised
        if isinstance(module, torch.nn.Linear):
            in_features, out_features = module.in_features, module.out_features
        else:  # transformers.Conv1D (GPT-2) stores [in, out]; RTN / GPTQ pack it like the transposed Linear (rtn.py)
            in_features, out_features = module.weight.shape[0], module.weight.shape[1]
        cfg = _module_config(quantization_config, name, module)
        target = name
        if name + ".linear.qweight" in keys:  # AWQ / TEQ: a multiplier in front of the packed layer (reference :479-482)
            set_module(original_model, name, MulLinear(module))
            target = name + ".linear"
        lay = layout.get(target)
        if lay is not None:  # saved from a non-optimum module: rebuild exactly that layout
            new = MI355XWeightOnlyLinear(
                in_features, out_features, dtype=lay["dtype"], bits=lay["bits"], group_size=lay["group_size"],
                zp=(target + ".qzeros") in keys, bias=(target + ".bias") in keys, g_idx=(target + ".g_idx") in keys,
                use_optimum_format=False, compression_dim=lay["compression_dim"], compression_dtype=getattr(torch, lay["compression_dtype"]),
                scale_dtype=getattr(torch, lay["scale_dtype"]), device=device,
            )
        else:
            new = MI355XWeightOnlyLinear(
                in_features, out_features, dtype=cfg.get("dtype", "int"), bits=cfg.get("bits", 4),
                group_size=cfg.get("group_size", 32), zp=(target + ".qzeros") in keys, bias=module.bias is not None,
                g_idx=(target + ".g_idx") in keys, use_optimum_format=True, device=device,
            )
        own = {}
        for k in ("qweight", "scales", "scale_bf16_to_fp8", "qzeros", "bias", "g_idx"):
            full = f"{target}.{k}"
            if full in state:
                own[k] = state.pop(full)
        own = {k: v.to(device) for k, v in own.items()}
        n_pack = 32 // new.bits
        awq_shape = (in_features, out_features // n_pack)
        if "qweight" in own and tuple(own["qweight"].shape) == awq_shape and awq_shape != tuple(new.qweight.shape):
            # an AutoAWQ "GEMM" checkpoint ([K, N/8] words, interleaved fields): shuffle to the optimum layout on the
            # device (reference repack_awq_and_load_state_dict, transformers/quantization/utils.py:655-697)
            own["qweight"], own["qzeros"] = ops.awq_repack(own["qweight"], own["qzeros"], new.bits)
            own["scales"] = own["scales"].to(torch.float16)
        missing = new.load_state_dict(own, strict=False)
        if "qweight" in missing.missing_keys or "scales" in missing.missing_keys:
            raise RuntimeError(f"checkpoint is missing packed buffers of {target}")
        set_module(original_model, target, new)
    left = sorted(k for k in state if k.endswith(".qweight"))
    if left:  # e.g. a module type this loader does not rebuild: never hand back a silently-float model
        raise RuntimeError(f"checkpoint holds packed weights that no module of the model consumed: {left[:4]}{' ...' if len(left) > 4 else ''}")
    return original_model


def load(model_name_or_path, original_model=None, format="default", device="cuda", **kwargs):
    """Load a weight-only-quantised checkpoint into packed MI355X modules (reference save_load.py:111-143).

    format="default": `original_model` (the float architecture; its weights may live on any device, "meta" included)
    plus quantized_weight.pt / qconfig.json.  format="huggingface": a local save_pretrained() directory.
    """
    fmt = getattr(format, "value", format)
    dev = torch.device(device if device not in (None, "auto", "cpu") else "cuda")
    folder = os.path.abspath(os.path.expanduser(str(model_name_or_path)))
    if fmt == "default":
        assert original_model is not None, "format='default' needs original_model (the float architecture)"
        wpath, cpath = os.path.join(folder, WEIGHT_NAME), os.path.join(folder, QCONFIG_NAME)
        assert os.path.exists(wpath), f"Cannot load model weight from path {wpath}."
        assert os.path.exists(cpath), f"Cannot load model quantization config from path {cpath}."
        state = torch.load(wpath, map_location="cpu", weights_only=True)  # safe defaults, reference :287-305
        with open(cpath) as f:
            qcfg = json.load(f)
        layout = None
        if os.path.exists(os.path.join(folder, LAYOUT_NAME)):
            with open(os.path.join(folder, LAYOUT_NAME)) as f:
                layout = json.load(f)
        model = _build(original_model, state, qcfg, dev, layout)
        # what is left are float tensors AWQ/TEQ touched (folded norms, MulLinear.input_scale) and untouched parameters
        model.load_state_dict(state, strict=False, assign=True)
        model.to(dev)
        model.eval()
        return model
    if fmt == "huggingface":
        from transformers import AutoConfig, AutoModelForCausalLM

        assert os.path.isdir(folder), "only local HuggingFace-format directories are supported (no network)"
        config = AutoConfig.from_pretrained(folder, **{k: v for k, v in kwargs.items() if k in ("trust_remote_code", "revision")})
        qpath = os.path.join(folder, HF_QCONFIG_NAME)
        if os.path.exists(qpath):
            with open(qpath) as f:
                qcfg = json.load(f)
        else:
            qcfg = getattr(config, "quantization_config", None)
            qcfg = qcfg if isinstance(qcfg, dict) else qcfg.to_dict()
        assert qcfg.get("quant_method", "gptq") in ("gptq", "awq", "rtn", "intel/auto-round", "autoround"), qcfg.get("quant_method")
        if hasattr(config, "quantization_config"):
            delattr(config, "quantization_config")  # keep HF from looking for its own GPTQ kernels
        model_class = kwargs.get("model_class") or AutoModelForCausalLM
        with torch.device("meta"):
            model = model_class.from_config(config)
        state = {}
        from safetensors.torch import load_file

        files = sorted(f for f in os.listdir(folder) if f.endswith(".safetensors"))
        if files:
            for fn in files:
                state.update(load_file(os.path.join(folder, fn), device="cpu"))
        else:
            for fn in sorted(f for f in os.listdir(folder) if f.endswith(".bin")):
                state.update(torch.load(os.path.join(folder, fn), map_location="cpu", weights_only=True))
        model = _build(model, state, qcfg, dev)
        missing = model.load_state_dict(state, strict=False, assign=True)
        still_meta = [n for n, p in list(model.named_parameters()) + list(model.named_buffers()) if p.device.type == "meta"]
        tied = getattr(config, "tie_word_embeddings", False)
        if still_meta and tied and hasattr(model, "tie_weights"):
            model.tie_weights()
            still_meta = [n for n, p in list(model.named_parameters()) + list(model.named_buffers()) if p.device.type == "meta"]
        # non-persistent buffers (rotary inv_freq) are rebuilt by re-instantiating them on the device
        for n in still_meta:
            if n.endswith("inv_freq"):
                continue
            raise RuntimeError(f"checkpoint in {folder} has no tensor for {n} (missing: {missing.missing_keys[:5]})")
        model = _materialise_buffers(model, config, dev)
        model.to(dev)
        model.eval()
        return model
    raise ValueError(f"unknown load format {format!r}: expected 'default' or 'huggingface'")


def _materialise_buffers(model, config, dev):
    """Recreate non-persistent buffers that were built on the meta device (rotary embedding tables)."""
    for name, mod in model.named_modules():
        for bname, buf in list(mod.named_buffers(recurse=False)):
            if buf.device.type != "meta":
                continue
            if bname == "inv_freq" and hasattr(mod, "rope_init_fn"):
                inv, _ = mod.rope_init_fn(getattr(mod, "config", config), dev)
                mod.register_buffer("inv_freq", inv, persistent=False)
                if hasattr(mod, "original_inv_freq"):
                    mod.original_inv_freq = inv
            else:
                mod.register_buffer(bname, torch.zeros(buf.shape, dtype=buf.dtype, device=dev), persistent=False)
    return model
