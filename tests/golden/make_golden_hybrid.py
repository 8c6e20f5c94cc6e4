"""GPTQ hybrid_order (columns rearranged by diag(H) inside their groups, groups by their largest diag(H); gptq.py:1203-1209,
1320-1328, 1389-1461) from the UNMODIFIED reference (build container only):

    python tests/golden/make_golden_hybrid.py      ->  tests/golden/gptq_hybrid_golden.npz

  <tag>_{W, X, H, scale, zero, Q, ints}   GPTQ.add_batch + fasterquant(hybrid_order=True) on seeded layers:
        hyb_sym_g32 (16 x 128, one block), hyb_asym_g32, hyb_sym_g64_2blk (24 x 256, two 128-column blocks), hyb_sym_g32_mse
  and gptq_tiny_llama_hybrid.npz: prepare / calibrate / convert of tests/model_zoo.tiny_llama with GPTQConfig(hybrid_order=True)
        (packed modules + logits, like tests/golden/make_golden_models.py)
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
    import transformers  # noqa: F401  -- BEFORE the reference (its gptq.py binds the name only when transformers is already imported)
    from neural_compressor.torch.algorithms.weight_only.gptq import GPTQ
    from neural_compressor.torch.algorithms.weight_only.utility import quant_weight_w_scale

    out = {}
    g = torch.Generator().manual_seed(2024)

    def run(tag, N, K, nb, seq, cfg, blocksize, groupsize):
        layer = torch.nn.Linear(K, N, bias=False)
        W = torch.randn(N, K, generator=g) * 0.05
        layer.weight.data.copy_(W)
        gq = GPTQ(layer, W.clone(), "cpu")
        full = dict(dtype="int", bits=4, sym=True, group_size=groupsize, mse=False, perchannel=True, use_double_quant=False,
                    double_quant_dtype="int", double_quant_bits=4, double_quant_sym=False, double_quant_group_size=128)
        full.update(cfg)
        gq.quantizer.configure(full)
        xs = []
        for _ in range(nb):
            x = torch.randn(1, seq, K, generator=g)
            x[..., ::17] *= 8.0  # a few outlier channels: the permutation is far from the identity
            x[..., 5::29] *= 3.0
            xs.append(x)
            gq.add_batch(x, None)
        H = gq.H.clone()
        scale, _, zero, Q = gq.fasterquant(W.clone(), blocksize=blocksize, percdamp=0.01, groupsize=groupsize, hybrid_order=True)
        out[f"{tag}_W"] = W.numpy()
        out[f"{tag}_X"] = torch.cat(xs, 0).numpy()
        out[f"{tag}_H"] = H.numpy()
        out[f"{tag}_scale"] = scale.numpy()
        out[f"{tag}_zero"] = zero.numpy()
        out[f"{tag}_Q"] = Q.numpy()
        out[f"{tag}_ints"] = quant_weight_w_scale(Q.clone(), scale, None, None if full["sym"] else zero, groupsize, dtype="int").numpy()

    run("hyb_sym_g32", 16, 128, 4, 48, dict(bits=4, sym=True), 128, 32)
    run("hyb_asym_g32", 16, 128, 4, 48, dict(bits=4, sym=False), 128, 32)
    run("hyb_sym_g64_2blk", 24, 256, 4, 80, dict(bits=4, sym=True), 128, 64)
    run("hyb_sym_g32_mse", 16, 128, 4, 48, dict(bits=4, sym=True, mse=True), 128, 32)
    np.savez_compressed(os.path.join(HERE, "gptq_hybrid_golden.npz"), **out)
    print("entries:", len(out))

    import tempfile

    from make_golden_models import dump_modules
    from neural_compressor.torch.quantization import GPTQConfig, convert, prepare

    from tests.model_zoo import calib_ids, tiny_llama

    ids = calib_ids()
    model = prepare(tiny_llama(), GPTQConfig(model_path=tempfile.mkdtemp(), bits=4, group_size=32, use_sym=True, block_size=128, hybrid_order=True))
    for x in ids:
        model(x)
    q = convert(model)
    mo = {}
    dump_modules(q, mo)
    with torch.no_grad():
        mo["logits"] = q(ids[0]).logits.float().numpy()
    assert not any(k.endswith(".g_idx") for k in mo), "hybrid order keeps the groups contiguous: no g_idx expected"
    np.savez_compressed(os.path.join(HERE, "gptq_tiny_llama_hybrid.npz"), **mo)
    print("tiny llama hybrid: modules", int(mo["n_modules"]))


if __name__ == "__main__":
    main()
