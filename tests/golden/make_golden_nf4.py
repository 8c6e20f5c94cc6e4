"""NF4 / FP4 / double-quant golden vectors from the UNMODIFIED reference (build container only):

    python tests/golden/make_golden_nf4.py      ->  tests/golden/nf4_golden.npz

  <tag>_w, _qdq, _int, _scale(, _zp)   quant_tensor(...) on seeded fp32 weights (utility.py:272-436; quantize_4bit :112-149):
        nf4_g32, fp4_g32, fp4e2m1_g32, nf4_tail (K = 80, group 32: a 16-wide tail group), nf4_pc (group_size -1),
        nf4_zero (one all-zero group: the reference's 0/0 path), nf4_q09 (quantile 0.9)
        dq_int4 (int4 asym g32 + double_quant int8 asym group 256), dq_nf4 (nf4 g32 + the same double quant)
        dq_bf16 / dq_bf16_sym (bf16 weights: the asym actor hands fp32 scales on, the sym / code-book actors bf16 ones -- double-quantised in THAT dtype), dq_ragged (K = 80: no double quant, utility.py:334-376)
  rtn_nf4_<module>.qweight / .scales + rtn_nf4_logits   RTNConfig(dtype="nf4", group_size=32) on tests/model_zoo.tiny_llama
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
    import transformers  # noqa: F401
    from neural_compressor.torch.algorithms.weight_only.utility import quant_tensor
    from neural_compressor.torch.quantization import RTNConfig, quantize

    from tests.model_zoo import calib_ids, tiny_llama

    out = {}
    g = torch.Generator().manual_seed(77)
    DQ = dict(double_quant=True, double_quant_dtype="int", double_quant_bits=8, double_quant_scheme="asym", double_quant_group_size=256)
    cases = {
        "nf4_g32": (24, 96, dict(dtype="nf4", group_size=32)),
        "fp4_g32": (24, 96, dict(dtype="fp4", group_size=32)),
        "fp4e2m1_g32": (24, 96, dict(dtype="fp4_e2m1", group_size=32)),
        "nf4_tail": (20, 80, dict(dtype="nf4", group_size=32)),
        "nf4_pc": (16, 64, dict(dtype="nf4", group_size=-1)),
        "nf4_zero": (8, 64, dict(dtype="nf4", group_size=32)),
        "nf4_q09": (16, 64, dict(dtype="nf4", group_size=32, quantile=0.9)),
        "dq_int4": (64, 128, dict(dtype="int", bits=4, group_size=32, scheme="asym", **DQ)),
        "dq_nf4": (64, 128, dict(dtype="nf4", group_size=32, **DQ)),
    }
    for tag, (N, K, kw) in cases.items():
        w = torch.randn(N, K, generator=g) * 0.05
        if tag == "nf4_zero":
            w[3, 32:] = 0
        out[f"{tag}_w"] = w.numpy().copy()
        out[f"{tag}_qdq"] = quant_tensor(w.clone(), **kw).numpy()
        res = quant_tensor(w.clone(), return_int=True, **kw)
        out[f"{tag}_int"] = res[0].numpy()
        out[f"{tag}_scale"] = res[1].numpy()
        if res[2] is not None:
            out[f"{tag}_zp"] = res[2].numpy()

    # 16-bit weights: the reference double-quantises the scales IN THE WEIGHT DTYPE (mean / sub / quant / add on a bf16 tensor),
    # and a ragged K (K % group_size != 0) returns from its "case 3" branch BEFORE the double-quant step (utility.py:334-376)
    w16 = (torch.randn(64, 128, generator=g) * 0.05).to(torch.bfloat16)
    out["dq_bf16_w"] = w16.float().numpy().copy()
    res = quant_tensor(w16.clone(), return_int=True, dtype="int", bits=4, group_size=32, scheme="asym", **DQ)
    out["dq_bf16_int"], out["dq_bf16_scale"], out["dq_bf16_zp"] = res[0].float().numpy(), res[1].float().numpy(), res[2].float().numpy()
    out["dq_bf16_qdq"] = quant_tensor(w16.clone(), dtype="int", bits=4, group_size=32, scheme="asym", **DQ).float().numpy()
    res = quant_tensor(w16.clone(), return_int=True, dtype="int", bits=4, group_size=32, scheme="sym", **DQ)
    out["dq_bf16_sym_int"], out["dq_bf16_sym_scale"] = res[0].float().numpy(), res[1].float().numpy()
    out["dq_bf16_sym_scale_is_bf16"] = np.bool_(res[1].dtype == torch.bfloat16)
    out["dq_bf16_sym_qdq"] = quant_tensor(w16.clone(), dtype="int", bits=4, group_size=32, scheme="sym", **DQ).float().numpy()
    wr = torch.randn(16, 80, generator=g) * 0.05
    out["dq_ragged_w"] = wr.numpy().copy()
    res = quant_tensor(wr.clone(), return_int=True, dtype="int", bits=4, group_size=32, scheme="asym", **DQ)
    out["dq_ragged_int"], out["dq_ragged_scale"], out["dq_ragged_zp"] = res[0].numpy(), res[1].numpy(), res[2].numpy()

    q = quantize(tiny_llama(), RTNConfig(dtype="nf4", group_size=32, use_layer_wise=False))
    n = 0
    for name, mod in q.named_modules():
        if type(mod).__name__ == "INCWeightOnlyLinear":
            out[f"rtn_nf4_{name}.qweight"] = mod.qweight.numpy()
            out[f"rtn_nf4_{name}.scales"] = mod.scales.numpy()
            n += 1
    out["rtn_nf4_n_modules"] = np.int64(n)
    with torch.no_grad():
        out["rtn_nf4_logits"] = q(calib_ids()[0]).logits.float().numpy()
    np.savez_compressed(os.path.join(HERE, "nf4_golden.npz"), **out)
    print("entries:", len(out), "modules:", n, "qweight shape", out["rtn_nf4_model.layers.0.self_attn.q_proj.qweight"].shape,
          out["rtn_nf4_model.layers.0.self_attn.q_proj.qweight"].dtype)


if __name__ == "__main__":
    main()
