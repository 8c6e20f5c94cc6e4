"""save / load of a SmoothQuant W8A8 model (interface of the reference's smooth_quant/save_load.py:24 -> static_quant
save / load: `model.save(output_dir)`, `load(output_dir, original_model)`).

The reference stores a TorchScript of the IPEX model (`quantized_model.pt`) plus the IPEX qconfig JSON; neither exists here.
The files keep the reference's names where they mean the same thing:
  quantized_weight.pt   state_dict: int8 `qweight`, `w_scale`, `alpha`, `corr`, `act_scale`, `act_zp`, `input_scale`, bias of every
                        W8A8Linear, the (possibly folded) float parameters of everything else
  qconfig.json          {"smooth_quant": {alpha, folding, absorb_to_layer}, "w8a8_modules": [names]}
"""

import json
import os

import torch

from ...utils.utility import get_module, set_module
from .utility import W8A8Linear

WEIGHT_NAME = "quantized_weight.pt"
QCONFIG_NAME = "qconfig.json"


def save(model, output_dir="./saved_results"):
    os.makedirs(output_dir, exist_ok=True)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    names = [n for n, m in model.named_modules() if isinstance(m, W8A8Linear)]
    info = dict(getattr(model, "sq_info", {}))
    info["absorb_to_layer"] = {k: list(v) for k, v in (info.get("absorb_to_layer") or {}).items()}
    with open(os.path.join(output_dir, QCONFIG_NAME), "w") as f:
        json.dump({"smooth_quant": info, "w8a8_modules": names}, f, indent=2)
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, os.path.join(output_dir, WEIGHT_NAME))


def load(output_dir, original_model, device="cuda"):
    """Rebuild the W8A8 model from the float architecture `original_model` (weights may be on any device, "meta" included)."""
    with open(os.path.join(output_dir, QCONFIG_NAME)) as f:
        cfg = json.load(f)
    state = torch.load(os.path.join(output_dir, WEIGHT_NAME), map_location="cpu", weights_only=True)
    dev = torch.device(device)
    for name in cfg["w8a8_modules"]:
        lin = get_module(original_model, name)
        assert isinstance(lin, torch.nn.Linear), f"{name}: expected nn.Linear in the float architecture, got {type(lin).__name__}"
        ft = state[name + ".bias"].dtype if (name + ".bias") in state else (
            lin.weight.dtype if lin.weight.dtype in (torch.float16, torch.bfloat16) else torch.float16)
        new = W8A8Linear(lin.in_features, lin.out_features, bias=(name + ".bias") in state,
                         has_input_scale=(name + ".input_scale") in state, device=dev, float_type=ft)
        own = {k[len(name) + 1:]: v for k, v in state.items() if k.startswith(name + ".")}
        new.load_state_dict({k: v.to(dev) for k, v in own.items()}, strict=True)
        for k in own:
            state.pop(name + "." + k)
        set_module(original_model, name, new)
    original_model.load_state_dict(state, strict=False, assign=True)
    original_model.to(dev)
    original_model.eval()
    original_model.sq_info = cfg.get("smooth_quant", {})
    original_model._smoothquant_optimized = True
    return original_model
