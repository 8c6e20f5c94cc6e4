"""2.x-named configuration shim: `PostTrainingQuantConfig(approach="weight_only", ...)`.

BASELINE.json's north_star names the INC 2.x surface (`quantization.fit()` / `PostTrainingQuantConfig` / the PyTorch
adaptor's `WeightOnlyLinear`).  The reference snapshot is INC 3.9, where that surface no longer exists (SURVEY.md
section 0.1: no `neural_compressor/quantization.py`, no `adaptor/`); the 3.x code under `neural_compressor/torch/` is its
successor and is what parity is pinned against.  This module only translates the 2.x vocabulary
(docs of INC 2.x, "Weight Only Quantization": op_type_dict / op_name_dict entries of the form
{"weight": {"bits", "group_size", "scheme", "algorithm"}} plus `recipes` {"rtn_args", "gptq_args", "awq_args"})
into the 3.x config objects of `neural_compressor_amd.torch.quantization`.
"""

from .torch.quantization.config import AWQConfig, GPTQConfig, RTNConfig

__all__ = ["PostTrainingQuantConfig"]


class PostTrainingQuantConfig:
    def __init__(self, approach="weight_only", op_type_dict=None, op_name_dict=None, recipes=None, **kwargs):
        recipes = recipes or {}
        # 2.x SmoothQuant: approach "static" (the default there) + recipes={"smooth_quant": True, "smooth_quant_args": {...}}
        # (reference docs/source/smooth_quant.md); plain static INT8 without smoothing is IPEX / PT2E territory
        self.smooth_quant = bool(recipes.get("smooth_quant", False))
        if approach != "weight_only" and not (approach in ("static", "post_training_static_quant") and self.smooth_quant):
            raise NotImplementedError(f"approach={approach!r}: only 'weight_only' and 'static' with recipes['smooth_quant'] are in the "
                                      "MI355X hot-path scope (SURVEY.md section 8)")
        self.approach = approach
        self.op_type_dict = op_type_dict or {".*": {"weight": {"bits": 4, "group_size": 32, "scheme": "sym", "algorithm": "RTN"}}}
        self.op_name_dict = op_name_dict or {}
        self.recipes = recipes or {}
        self.extra = kwargs

    @staticmethod
    def _algo_of(entry):
        return str(entry.get("weight", {}).get("algorithm", "RTN")).upper()

    def _one(self, entry):
        """3.x config object for one 2.x {"weight": {...}} entry."""
        w = entry.get("weight", {})
        dtype = w.get("dtype", "int")
        if dtype == "fp32":
            return RTNConfig(dtype="fp32")
        bits, gs = int(w.get("bits", 4)), int(w.get("group_size", 32))
        sym = w.get("scheme", "sym") == "sym"
        algo = self._algo_of(entry)
        if algo == "RTN":
            a = self.recipes.get("rtn_args", {})
            return RTNConfig(bits=bits, group_size=gs, use_sym=sym, use_full_range=a.get("enable_full_range", False),
                             use_mse_search=a.get("enable_mse_search", False), use_layer_wise=False)
        if algo == "GPTQ":
            a = self.recipes.get("gptq_args", {})
            return GPTQConfig(bits=bits, group_size=gs, use_sym=sym, percdamp=a.get("percdamp", 0.01),
                              act_order=a.get("act_order", False), block_size=a.get("block_size", 128),
                              static_groups=a.get("static_groups", False), true_sequential=a.get("true_sequential", False))
        if algo == "AWQ":
            a = self.recipes.get("awq_args", {})
            return AWQConfig(bits=bits, group_size=gs, use_sym=sym, use_auto_scale=a.get("enable_auto_scale", True),
                             use_auto_clip=a.get("enable_mse_search", True), folding=a.get("folding", False))
        raise NotImplementedError(f"weight-only algorithm {algo!r} is outside the hot-path scope (RTN / GPTQ / AWQ)")

    def to_3x(self):
        """Global config from the '.*' (or first) op_type entry; op_name_dict entries become name-local configs."""
        if self.smooth_quant and self.approach != "weight_only":
            from .torch.quantization import SmoothQuantConfig

            a = self.recipes.get("smooth_quant_args", {})
            cfg = SmoothQuantConfig(alpha=a.get("alpha", 0.5), folding=bool(a.get("folding", False)),
                                    scale_sharing=bool(a.get("scale_sharing", False)))
            for pattern, entry in self.op_name_dict.items():
                if entry.get("weight", {}).get("dtype") == "fp32" or entry.get("activation", {}).get("dtype") == "fp32":
                    cfg.set_local(pattern, SmoothQuantConfig(w_dtype="fp32"))
            return cfg
        entries = dict(self.op_type_dict)
        base_entry = entries.pop(".*", None) or next(iter(entries.values()))
        cfg = self._one(base_entry)
        algo = self._algo_of(base_entry)
        for pattern, entry in list(entries.items()) + list(self.op_name_dict.items()):
            if self._algo_of(entry) != algo and entry.get("weight", {}).get("dtype") != "fp32":
                raise NotImplementedError("mixing algorithms inside one PostTrainingQuantConfig is not supported by this shim")
            local = self._one(entry)
            if local.name != cfg.name:  # an fp32 exclusion expressed with the global algorithm's config class
                local = type(cfg)(dtype="fp32")
            cfg.set_local(pattern, local)
        return cfg
