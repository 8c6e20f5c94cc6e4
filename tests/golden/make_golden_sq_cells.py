"""The non-default fake-quant cells the reference DEFINES in-tree (build container only; unmodified reference):

    python tests/golden/make_golden_sq_cells.py   ->  tests/golden/sq_cells_golden.npz

  w [48, 96], qdq_w_asym            quant_dequant_w_v1(Linear, scheme="asym")   (smooth_quant/utility.py:652-695: per-output-channel
                                    uint8 with a zero point; the W8A8 EXECUTION of the reference (IPEX) never uses it -- the tuner and
                                    IPEX's SmoothQuant qconfig are symmetric per-channel -- but the function is part of the path's API)
  x [4, 40, 96], qdq_x_dynamic      quant_dequant_x_v1(x) with min / max taken from x itself (:745-746: dynamic per-tensor)
"""

import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden import REF, _install_stubs  # noqa: E402


def main():
    sys.dont_write_bytecode = True
    _install_stubs()
    peft = types.ModuleType("peft")
    peft.PeftModel = type("PeftModel", (), {})
    sys.modules["peft"] = peft
    sys.path.insert(0, REF)
    import torch
    from neural_compressor.torch.algorithms.smooth_quant import utility as U

    g = torch.Generator().manual_seed(11)
    w = torch.randn(48, 96, generator=g) * 0.1
    w[5] = w[5].abs()        # a row without negative values: the zero point clamps at 0
    w[7] = 0.0               # an all-zero row: scale = eps
    lin = torch.nn.Linear(96, 48, bias=False)
    lin.weight.data.copy_(w)
    x = torch.randn(4, 40, 96, generator=g) * 3 + 0.5
    out = dict(w=w.numpy(), qdq_w_asym=U.quant_dequant_w_v1(lin, num_bits=8, scheme="asym").detach().numpy(), x=x.numpy(),
               qdq_x_dynamic=U.quant_dequant_x_v1(x.clone()).numpy())
    np.savez_compressed(os.path.join(HERE, "sq_cells_golden.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
