"""Front-end -> hot path glue (reference neural_compressor/transformers/quantization/utils.py).

  default_run_fn :258-334   convert_to_quantized_model :337-487   save_low_bit :576-652
  repack_awq_and_load_state_dict :655-697 (here: the loader in torch/algorithms/weight_only/save_load.py calls the
  inc_awq_repack kernel when it meets AutoAWQ-shaped words)

The reference finishes with `replace_linear` (:99-255), which swaps INCWeightOnlyLinear for IPEX CPU/XPU kernels; on
MI355X the packed module produced by convert() already IS the inference module (fused dequant-GEMM), so there is no
second module family and `for_inference` only controls nothing but the returned object's `save_pretrained` flavour.
"""

import json
import os
import types

import torch

from ...common.utils import logger
from ...torch.quantization import AWQConfig, GPTQConfig, RTNConfig, convert, prepare


def _as_batches(dataset, tokenizer, max_length, n_samples, batch_size):
    """Yield int64 [b, max_length] token batches.  `dataset` is a dataset name / path (datasets.load_dataset, as in the
    reference :261-262), or -- the offline route -- any iterable of token tensors / lists / {"input_ids": ...} / str."""
    rows = []

    def push(ids):
        ids = torch.as_tensor(ids, dtype=torch.long).reshape(-1)
        if ids.numel() >= max_length:  # reference :296 drops short rows, :300-303 truncates long ones
            rows.append(ids[:max_length])

    if isinstance(dataset, (str, bytes, os.PathLike)):
        from datasets import load_dataset

        if tokenizer is None:
            raise ValueError("Please provide the tokenizer in quantization_config.")
        data = load_dataset(dataset, split="train").shuffle(seed=42)
        for ex in data:
            key = next((k for k in ("prompt", "code", "text") if k in ex), None)
            if key is None:
                raise ValueError("Please check dataset prompt identifier, NeelNanda/pile-10k is default used calibration dataset.")
            push(tokenizer(ex[key])["input_ids"])
            if len(rows) >= n_samples:
                break
    else:
        for ex in dataset:
            if isinstance(ex, dict):
                ex = ex["input_ids"]
            if isinstance(ex, str):
                if tokenizer is None:
                    raise ValueError("Please provide the tokenizer in quantization_config.")
                ex = tokenizer(ex)["input_ids"]
            t = torch.as_tensor(ex)
            for row in (t.reshape(-1, t.shape[-1]) if t.dim() > 1 else [t]):
                push(row)
            if len(rows) >= n_samples:
                break
    if not rows:
        raise AssertionError("The dataset does not have data that meets the required input length. Please reduce seq_len.")
    rows = rows[:n_samples]
    for i in range(0, len(rows), batch_size):
        yield torch.stack(rows[i : i + batch_size])


def default_run_fn(model, tokenizer, dataset, max_length=512, n_samples=100, batch_size=8, algo="rtn"):
    """Calibration loop (reference :258-334): feed `n_samples` rows of `max_length` tokens through the prepared model."""
    device = next((p.device for p in model.parameters() if p.device.type != "meta"), torch.device("cuda"))
    for input_ids in _as_batches(dataset, tokenizer, max_length, n_samples, batch_size):
        try:
            model(input_ids=input_ids.to(device))
        except ValueError:  # the GPTQ capture aborts the forward after block 0 (reference :331-334)
            pass


def _exclude(quant_config, cfg_cls, modules_to_not_convert):
    for module in modules_to_not_convert:  # reference :366-369
        quant_config.set_local(".*" + module, cfg_cls(dtype="fp32"))


def convert_to_quantized_model(model, config, device="cuda", for_inference=True):
    """Map the front-end config onto RTNConfig / GPTQConfig / AWQConfig and run prepare -> calibrate -> convert
    (reference :337-487)."""
    dtype = "int4" if config.weight_dtype == "int4_fullrange" else config.weight_dtype
    dtype = "int" if dtype in ("int4", "int8") else dtype
    method = getattr(config.quant_method, "value", config.quant_method)
    run_args = None
    if method == "rtn":
        quant_config = RTNConfig(dtype=dtype, bits=config.bits, use_sym=config.sym, group_size=config.group_size,
                                 use_layer_wise=False, model_path=config.model_path, quant_lm_head=config.quant_lm_head)
        _exclude(quant_config, RTNConfig, config.modules_to_not_convert)
        logger.info("Do RTN algorithm with config %s", quant_config)
        model = convert(prepare(model, quant_config))
    elif method == "gptq":
        model.seqlen = config.seq_len
        quant_config = GPTQConfig(
            dtype=dtype, bits=config.bits, use_sym=config.sym, group_size=config.group_size, use_layer_wise=False,
            model_path=config.model_path, act_order=config.desc_act, percdamp=config.damp_percent,
            block_size=config.blocksize, static_groups=config.static_groups, use_mse_search=config.use_mse_search,
            true_sequential=config.true_sequential, quant_lm_head=config.quant_lm_head,
        )
        _exclude(quant_config, GPTQConfig, config.modules_to_not_convert)
        logger.info("Do GPTQ algorithm with config %s", quant_config)
        run_args = (config.tokenizer, config.dataset, config.seq_len, config.n_samples, config.batch_size, method)
        model = prepare(model=model, quant_config=quant_config)
        default_run_fn(model, *run_args)
        model = convert(model)
    elif method == "awq":
        quant_config = AWQConfig(
            dtype=dtype, bits=config.bits, use_sym=config.sym, group_size=config.group_size, use_layer_wise=False,
            use_auto_scale=config.auto_scale, use_auto_clip=config.auto_clip, folding=True,
            absorb_layer_dict=config.absorb_layer_dict, quant_lm_head=config.quant_lm_head,
        )
        _exclude(quant_config, AWQConfig, config.modules_to_not_convert)
        logger.info("Do AWQ algorithm with config %s", quant_config)
        run_args = (config.tokenizer, config.dataset, config.seq_len, config.n_samples, config.batch_size, method)
        example_inputs = torch.ones([1, 512], dtype=torch.long, device=device)  # reference :432
        model = prepare(model=model, quant_config=quant_config, example_inputs=example_inputs)
        default_run_fn(model, *run_args)
        model = convert(model)
    elif method == "teq":
        raise NotImplementedError("TEQ trains its scales with autograd and is outside the MI355X weight-only hot path")
    else:
        raise AssertionError("The Supported algorithm are RTN, AWQ, GPTQ")
    model.eval()
    return model.to(device)


def make_contiguous(model):
    for param in model.parameters():
        if param.data.ndimension() > 1:
            param.data = param.data.contiguous()


def save_low_bit(self, save_directory, push_to_hub=False, **kwargs):
    """Bound to the quantised model as `save_pretrained` (reference :576-652): HF safetensors with the packed buffers
    under the reference's names + quantize_config.json + all_checkpoint_keys.json."""
    assert hasattr(self, "quantization_config"), "Detected this model is not a low-bit model."
    if push_to_hub:
        raise NotImplementedError("push_to_hub needs network access")
    if os.path.isfile(save_directory):
        logger.error("Provided path (%s) should be a directory, not a file", save_directory)
        return
    os.makedirs(save_directory, exist_ok=True)
    del self.save_pretrained  # fall back to the class's own save_pretrained for the call below
    try:
        make_contiguous(self)
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self.save_pretrained(save_directory=save_directory, **kwargs)
    finally:
        self.save_pretrained = types.MethodType(save_low_bit, self)
    with open(os.path.join(save_directory, "all_checkpoint_keys.json"), "w") as f:  # reference :627-633
        json.dump({"all_checkpoint_keys": list(self.state_dict().keys())}, f)
    self.quantization_config.save_pretrained(save_directory, **kwargs)
