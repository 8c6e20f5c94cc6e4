"""quantize_4bit with the caller's scale and double_quant_return_int, from the UNMODIFIED reference (build container only):

    python tests/golden/make_golden_q4scale.py      ->  tests/golden/q4scale_golden.npz

  q4s_<dtype>_<wd>_{w, scale, qdq, int}   quantize_4bit(tensor [rows, group], scale=scale [rows, 1], dtype=...) (utility.py:112-149,
        the `scale` kwarg :127-128) on seeded weights, fp32 and bf16 storage; the scales are NOT the rows' own max (0.7 .. 1.3 x it),
        so entries saturate at both ends of the code book
  dqri_error   what quant_tensor(..., double_quant=True, double_quant_return_int=True) raises in the reference (utility.py:383-405)
"""

import os
import sys

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
    from neural_compressor.torch.algorithms.weight_only.utility import quant_tensor, quantize_4bit

    out = {}
    g = torch.Generator().manual_seed(4242)
    for dtype in ("nf4", "fp4", "fp4_e2m1"):
        for wd, td in (("f32", torch.float32), ("bf16", torch.bfloat16)):
            w = (torch.randn(48, 32, generator=g) * 0.05).to(td)
            own = w.float().abs().max(1)[0]
            scale = (own * (0.7 + 0.6 * torch.rand(48, generator=g))).to(td).unsqueeze(-1)
            tag = f"q4s_{dtype}_{wd}"
            out[f"{tag}_w"] = w.float().numpy().copy()
            out[f"{tag}_scale"] = scale.float().numpy().copy()
            out[f"{tag}_qdq"] = quantize_4bit(w.clone(), dtype=dtype, scale=scale.clone()).float().numpy()
            res = quantize_4bit(w.clone(), dtype=dtype, return_int=True, scale=scale.clone())
            out[f"{tag}_int"] = res[0].float().numpy()
            assert torch.equal(res[1], scale) and res[2] is None
    # double_quant_return_int (utility.py:383-384, a TODO): the reference drops the inner call's result and fails on the unpack --
    # recorded so that the tests pin the failure, not a guess at what it would return
    try:
        quant_tensor(torch.randn(64, 128, generator=g) * 0.05, return_int=True, dtype="int", bits=4, group_size=32, scheme="asym",
                     double_quant=True, double_quant_return_int=True)
        out["dqri_error"] = np.array("")
    except Exception as e:  # noqa: BLE001
        out["dqri_error"] = np.array(f"{type(e).__name__}: {e}")
    np.savez_compressed(os.path.join(HERE, "q4scale_golden.npz"), **out)
    print("entries:", len(out), "double_quant_return_int ->", str(out["dqri_error"]))


if __name__ == "__main__":
    main()
