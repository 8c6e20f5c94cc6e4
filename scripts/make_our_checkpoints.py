"""Checkpoint fixtures SAVED BY THIS repository (run on the GPU box:  python scripts/make_our_checkpoints.py <out_dir>):

  <out_dir>/ours_rtn_default/   quantized_weight.pt + qconfig.json      (save(), format "default")
  <out_dir>/ours_gptq_hf/       safetensors + config.json + quantize_config.json   (format "huggingface")
  <out_dir>/ours_logits.npz     logits of both quantised models on calib_ids()[0] (GPU, fp16 compute, stored fp32)

Copied to tests/golden/ckpt/ and committed; tests/test_interop_reference_cpu.py feeds them to the UNMODIFIED reference's
load() (neural_compressor/torch/algorithms/weight_only/save_load.py:111-143) in the build container."""

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests.model_zoo import calib_ids, tiny_llama  # noqa: E402


def to_half(model):
    for mod in model.modules():
        for p in mod.parameters(recurse=False):
            if p.is_floating_point():
                p.data = p.data.half()
    return model


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "ckpt_ours")
    os.makedirs(out, exist_ok=True)
    from neural_compressor_amd.torch.quantization import GPTQConfig, RTNConfig, convert, prepare, quantize

    ids = calib_ids()
    logits = {}
    q = quantize(tiny_llama(), RTNConfig(bits=4, group_size=32, use_sym=False, use_layer_wise=False))
    q.save(os.path.join(out, "ours_rtn_default"))
    with torch.no_grad():
        logits["rtn_default"] = to_half(q)(ids[0].to("cuda")).logits.float().cpu().numpy()
    model = prepare(tiny_llama(), GPTQConfig(bits=4, group_size=32, use_sym=True, block_size=128))
    for x in ids:
        model(x)
    q = convert(model)
    q.save(os.path.join(out, "ours_gptq_hf"), format="huggingface")
    with torch.no_grad():
        logits["gptq_hf"] = to_half(q)(ids[0].to("cuda")).logits.float().cpu().numpy()
    np.savez_compressed(os.path.join(out, "ours_logits.npz"), **logits)
    for root, _, files in os.walk(out):
        for f in files:
            p = os.path.join(root, f)
            print(f"{os.path.getsize(p):9d}  {os.path.relpath(p, out)}")


if __name__ == "__main__":
    main()
