"""Model-level golden fixtures at widths other than 4 / 8 bits from the UNMODIFIED reference (build container only):

    python tests/golden/make_golden_bits_models.py

  gptq_tiny_llama_b3_sym_g32.npz : prepare -> run_fn -> convert with GPTQConfig(bits=3)  (n_pack = 10: two unused high bits per word)
  rtn_tiny_llama_b{3,6}_asym_g32.npz : quantize(model, RTNConfig(bits=3 / 6))
-> per-module qweight / scales / qzeros + logits of the quantised model (the reference's config tunes bits = [4, 1, 2, 3, 5, 6, 7, 8],
torch/quantization/config.py:211).
"""

import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden import REF, _install_stubs  # noqa: E402
from make_golden_models import dump_modules  # noqa: E402


def main():
    sys.dont_write_bytecode = True
    _install_stubs()
    sys.path.insert(0, REF)
    import torch
    import transformers  # noqa: F401  (before the reference: see make_golden_models.py)
    from neural_compressor.torch.quantization import GPTQConfig, RTNConfig, convert, prepare, quantize

    from tests.model_zoo import calib_ids, tiny_llama

    ids = calib_ids()
    tmp = tempfile.mkdtemp()
    model = prepare(tiny_llama(), GPTQConfig(model_path=tmp, bits=3, group_size=32, use_sym=True, block_size=128))
    for x in ids:
        model(x)
    q = convert(model)
    out = {}
    dump_modules(q, out)
    with torch.no_grad():
        out["logits"] = q(ids[0]).logits.float().numpy()
    np.savez_compressed(os.path.join(HERE, "gptq_tiny_llama_b3_sym_g32.npz"), **out)
    print("gptq b3 modules:", int(out["n_modules"]))
    for bits in (3, 6):
        q = quantize(tiny_llama(), RTNConfig(bits=bits, group_size=32, use_sym=False))
        out = {}
        dump_modules(q, out)
        with torch.no_grad():
            out["logits"] = q(ids[0]).logits.float().numpy()
        np.savez_compressed(os.path.join(HERE, f"rtn_tiny_llama_b{bits}_asym_g32.npz"), **out)
        print(f"rtn b{bits} modules:", int(out["n_modules"]))


if __name__ == "__main__":
    main()
