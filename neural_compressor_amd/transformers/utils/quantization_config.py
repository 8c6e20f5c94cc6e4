"""Front-end quantisation configs (reference neural_compressor/transformers/utils/quantization_config.py).

  QuantizationMethod :38   INCQuantizationConfigMixin :45   RtnConfig :242   GPTQConfig :296   AwqConfig :387
  TeqConfig :456

Same constructor arguments, defaults and serialised keys (`quantize_config.json`), because those keys are what a saved
low-bit checkpoint is re-opened with.  Differences, all forced by the target: `compute_dtype` / `scale_dtype` default to
"fp16" (the packed module's scales are fp16 and the fused kernel computes in fp16 / bf16), `device` defaults to "cuda",
and `dataset` may also be an iterable of token tensors (there is no network to fetch NeelNanda/pile-10k from).
"""

import copy
import json
import os
from enum import Enum
from typing import Any, Dict, Union

QUANT_CONFIG = "quantize_config.json"  # reference :25

_NOT_CONVERTED = ["lm_head", "transformer.output_layer", "embed_out"]  # reference :261-263
_RUNTIME_ONLY = [  # reference remove_redundant_parameters :147-185: never serialised
    "calib_dataloader", "dataset", "calib_func", "calib_iters", "calib_len", "mse_range", "scheme", "tokenizer",
    "use_layer_wise", "blocksize", "nsamples", "max_input_length", "static_groups", "lr", "minmax_lr", "iters",
    "use_quant_input", "device", "calib_dataset", "calib_pad_val", "calib_shuffle", "calib_padding", "example_inputs",
    "excluded_precisions", "op_name_dict", "op_type_dict", "train_dataloader", "train_func", "train_iters", "train_len",
    "train_padding", "train_dataset", "train_pad_val", "train_shuffle", "train_batch_size",
]


class QuantizationMethod(str, Enum):
    GPTQ = "gptq"
    RTN = "rtn"
    AWQ = "awq"
    TEQ = "teq"


class INCQuantizationConfigMixin:
    """Serialisation + validation shared by the four configs (reference :45-239)."""

    quant_method: QuantizationMethod

    def update(self, **kwargs):
        """Set the attributes that exist, return the rest (reference :48-67)."""
        unused = {}
        for key, value in kwargs.items():
            if hasattr(self, key):
                setattr(self, key, value)
            else:
                unused[key] = value
        return unused

    def post_init(self):
        """The checks of the reference's post_init_cpu (:69-96), with the MI355X value sets."""
        if self.compute_dtype is None:
            self.compute_dtype = "fp16"
        elif self.compute_dtype not in ("fp16", "bf16"):
            raise ValueError("compute_dtype must be 'fp16' or 'bf16' (the fused dequant-GEMM computes in the activation dtype).")
        if self.bits is None:
            self.bits = 4
        elif self.bits not in (4, 8):
            raise ValueError(f"Only support quantization to [4, 8] bits but found {self.bits}")
        if self.scale_dtype is None:
            self.scale_dtype = "fp16"
        elif self.scale_dtype not in ("fp32", "bf16", "fp16"):
            raise ValueError("scale_dtype must be a string in 'fp32', 'bf16', 'fp16'")
        if not isinstance(self.group_size, int):
            raise ValueError("group_size must be a int")
        if not isinstance(self.scheme, str):
            raise ValueError("scheme must be a string")

    # -- dict / json -------------------------------------------------------------------------------------------------
    def to_dict(self) -> Dict[str, Any]:
        out = {}
        for k, v in copy.copy(self.__dict__).items():
            if k in ("tokenizer", "calib_dataloader", "dataset") and not isinstance(v, (str, type(None))):
                continue  # live objects are not serialisable (reference to_json_file :139-143 nulls them)
            out[k] = v.value if isinstance(v, Enum) else copy.deepcopy(v)
        return out

    def to_diff_dict(self) -> Dict[str, Any]:
        """Only what differs from the class defaults (reference :273-293)."""
        default = type(self)().to_dict()
        return {k: v for k, v in self.to_dict().items() if k not in default or v != default[k]}

    @classmethod
    def from_dict(cls, config_dict, return_unused_kwargs=False, **kwargs):
        cfg = dict(config_dict)
        cfg.pop("quant_method", None)
        obj = cls(**{k: v for k, v in cfg.items() if k in cls.__init__.__code__.co_varnames})
        for k, v in cfg.items():  # keys saved by other tools (AutoAWQ's "version", "backend", ...) ride along
            if not hasattr(obj, k):
                setattr(obj, k, v)
        unused = obj.update(**kwargs)
        return (obj, unused) if return_unused_kwargs else obj

    def to_json_string(self, use_diff: bool = True) -> str:
        d = self.to_diff_dict() if use_diff else self.to_dict()
        return json.dumps(d, indent=2, sort_keys=True) + "\n"

    def to_json_file(self, json_file_path: Union[str, os.PathLike], use_diff: bool = True):
        with open(json_file_path, "w", encoding="utf-8") as writer:
            writer.write(self.to_json_string(use_diff=use_diff))

    def remove_redundant_parameters(self):
        for name in _RUNTIME_ONLY:
            if hasattr(self, name):
                delattr(self, name)

    def save_pretrained(self, save_directory: Union[str, os.PathLike], **kwargs):
        """<dir>/quantize_config.json (reference :187-233; hub upload is out of scope: no network)."""
        if os.path.isfile(save_directory):
            raise AssertionError(f"Provided path ({save_directory}) should be a directory, not a file")
        os.makedirs(save_directory, exist_ok=True)
        self.to_json_file(os.path.join(save_directory, QUANT_CONFIG), use_diff=False)

    @classmethod
    def from_pretrained(cls, save_directory, **kwargs):
        with open(os.path.join(save_directory, QUANT_CONFIG), encoding="utf-8") as f:
            return cls.from_dict(json.load(f), **kwargs)

    def __repr__(self):
        return f"{type(self).__name__} {self.to_json_string(use_diff=False)}"


def _common(self, bits, group_size, compute_dtype, scale_dtype, use_layer_wise, quant_lm_head, kwargs):
    self.bits = bits
    self.compute_dtype = compute_dtype
    self.weight_dtype = "int4" if bits == 4 else "int8"
    self.scale_dtype = scale_dtype
    self.group_size = group_size
    self.use_layer_wise = use_layer_wise
    self.quant_lm_head = quant_lm_head
    self.modules_to_not_convert = list(kwargs.get("modules_to_not_convert", _NOT_CONVERTED))
    if quant_lm_head:
        self.modules_to_not_convert = []
    self.device = kwargs.get("device", "cuda")


class RtnConfig(INCQuantizationConfigMixin):
    """Reference :242-293."""

    def __init__(self, bits: int = 4, group_size: int = 32, compute_dtype: Any = None, scale_dtype: Any = None,
                 sym: bool = True, use_layer_wise: bool = None, quant_lm_head: bool = False, **kwargs):
        self.quant_method = QuantizationMethod.RTN
        _common(self, bits, group_size, compute_dtype, scale_dtype, use_layer_wise, quant_lm_head, kwargs)
        self.model_path = kwargs.get("model_path", "")
        self.sym = sym
        self.scheme = "sym" if sym else "asym"


class GPTQConfig(INCQuantizationConfigMixin):
    """Reference :296-384."""

    def __init__(self, bits: int = 4, tokenizer: Any = None, dataset: Any = "NeelNanda/pile-10k", batch_size: int = 8,
                 group_size: int = 32, compute_dtype: Any = None, scale_dtype: Any = None, sym: bool = True,
                 blocksize: int = 128, damp_percent: float = 0.1, desc_act: bool = False, n_samples: int = 128,
                 seq_len: int = 2048, static_groups: bool = False, use_mse_search: bool = False,
                 true_sequential: bool = False, use_layer_wise: bool = None, quant_lm_head: bool = False, **kwargs):
        self.quant_method = QuantizationMethod.GPTQ
        _common(self, bits, group_size, compute_dtype, scale_dtype, use_layer_wise, quant_lm_head, kwargs)
        self.tokenizer = tokenizer
        self.dataset = dataset
        self.batch_size = batch_size
        self.sym = sym
        self.scheme = "sym" if sym else "asym"
        self.blocksize = blocksize
        self.n_samples = n_samples
        self.damp_percent = damp_percent
        self.desc_act = desc_act
        self.static_groups = static_groups
        self.use_mse_search = use_mse_search
        self.true_sequential = true_sequential
        self.model_path = kwargs.get("model_path", "")
        self.seq_len = seq_len
        self.post_init_gptq()

    def post_init_gptq(self):
        if self.bits not in [4, 8]:
            raise ValueError(f"Only support quantization to [4, 8] bits but found {self.bits}")
        if not (0 < self.damp_percent < 1):
            raise ValueError("damp_percent must between 0 and 1.")


class AwqConfig(INCQuantizationConfigMixin):
    """Reference :387-453."""

    def __init__(self, bits: int = 4, tokenizer: Any = None, dataset: Any = "NeelNanda/pile-10k", group_size: int = 32,
                 compute_dtype: Any = None, weight_dtype: Any = None, scale_dtype: Any = None,
                 use_layer_wise: bool = None, n_samples: int = 128, seq_len: int = 2048, auto_scale: bool = True,
                 auto_clip: bool = True, zero_point: bool = True, absorb_layer_dict: dict = {},
                 quant_lm_head: bool = False, backend: str = None, **kwargs):
        self.quant_method = QuantizationMethod.AWQ
        _common(self, bits, group_size, compute_dtype, scale_dtype, use_layer_wise, quant_lm_head, kwargs)
        self.tokenizer = tokenizer
        self.dataset = dataset
        self.zero_point = zero_point
        self.auto_scale = auto_scale
        self.auto_clip = auto_clip
        self.n_samples = n_samples
        self.seq_len = seq_len
        self.absorb_layer_dict = absorb_layer_dict
        self.backend = backend
        self.scheme = "asym" if zero_point else "sym"
        self.sym = not zero_point
        self.batch_size = kwargs.pop("batch_size", 8)


class TeqConfig(INCQuantizationConfigMixin):
    """Reference :456-519.  TEQ trains its scales with autograd; SURVEY.md section 8 marks it out of the hot path, so
    the config exists for API compatibility and conversion raises."""

    def __init__(self, bits: int = 4, tokenizer: Any = None, dataset: Any = "NeelNanda/pile-10k", group_size: int = 32,
                 compute_dtype: Any = None, weight_dtype: Any = None, scale_dtype: Any = None,
                 use_layer_wise: bool = None, n_samples: int = 128, seq_len: int = 2048, sym: bool = True,
                 folding: bool = False, absorb_layer_dict: dict = {}, quant_lm_head: bool = False, **kwargs):
        self.quant_method = QuantizationMethod.TEQ
        _common(self, bits, group_size, compute_dtype, scale_dtype, use_layer_wise, quant_lm_head, kwargs)
        self.tokenizer = tokenizer
        self.dataset = dataset
        self.sym = sym
        self.scheme = "sym" if sym else "asym"
        self.n_samples = n_samples
        self.seq_len = seq_len
        self.folding = folding
        self.absorb_layer_dict = absorb_layer_dict
        self.batch_size = kwargs.pop("batch_size", 8)
