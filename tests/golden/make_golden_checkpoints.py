"""Checkpoint fixtures SAVED BY the unmodified reference (build container only):

    python tests/golden/make_golden_checkpoints.py

  tests/golden/ckpt/ref_rtn_default/      quantized_weight.pt + qconfig.json   (reference save(), format "default", save_load.py:56-108)
  tests/golden/ckpt/ref_gptq_hf/          save_pretrained() safetensors + config.json + quantize_config.json (format "huggingface")
  tests/golden/ckpt/ref_logits.npz        logits of both quantised reference models on calib_ids()[0] (fp32, CPU)

tests/model_zoo.tiny_llama (hidden 64, 2 blocks, vocab 128: the whole checkpoint is ~150 KB) quantised with RTN asym g32 /
GPTQ sym g32 by the reference on CPU.  tests/test_gpu_models.py loads both directories with THIS repository's load() and must
rebuild the same packed buffers and logits; tests/test_interop_reference_cpu.py feeds directories saved HERE to the
reference's load().
"""

import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden import REF, _install_stubs  # noqa: E402


def main():
    sys.dont_write_bytecode = True
    _install_stubs()
    sys.path.insert(0, REF)
    import torch
    import transformers  # noqa: F401
    from neural_compressor.torch.quantization import GPTQConfig, RTNConfig, convert, prepare, quantize

    from tests.model_zoo import calib_ids, tiny_llama

    ids = calib_ids()
    out = os.path.join(HERE, "ckpt")
    logits = {}

    d = os.path.join(out, "ref_rtn_default")
    shutil.rmtree(d, ignore_errors=True)
    q = quantize(tiny_llama(), RTNConfig(bits=4, group_size=32, use_sym=False, use_layer_wise=False))
    q.save(d)  # format "default"
    with torch.no_grad():
        logits["rtn_default"] = q(ids[0]).logits.float().numpy()

    d = os.path.join(out, "ref_gptq_hf")
    shutil.rmtree(d, ignore_errors=True)
    tmp = tempfile.mkdtemp()
    model = prepare(tiny_llama(), GPTQConfig(model_path=tmp, bits=4, group_size=32, use_sym=True, block_size=128))
    for x in ids:
        model(x)
    q = convert(model)
    q.save(d, format="huggingface")
    with torch.no_grad():
        logits["gptq_hf"] = q(ids[0]).logits.float().numpy()
    np.savez_compressed(os.path.join(out, "ref_logits.npz"), **logits)
    for root, _, files in os.walk(out):
        for f in files:
            p = os.path.join(root, f)
            print(f"{os.path.getsize(p):9d}  {os.path.relpath(p, out)}")


if __name__ == "__main__":
    main()
