"""Golden vectors for GPTQ layouts whose groups or reference blocks are WIDER than 128 columns, from the UNMODIFIED reference
(intel/neural-compressor v3.9 at /root/reference):  python tests/golden/make_golden_gptq_wide.py  ->  gptq_wide_golden.npz

These are the layouts in which `find_params` of a 128-column step reads columns that lie outside it (gptq.py:1266-1272 reads the
global W "as it is now"): group_size 256 with block_size 128, block_size 256 with group sizes 128 and 64.  Stubs as in make_golden.py.
"""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, _install_stubs  # noqa: E402

CASES = {
    "gqw_sym_g256_bs128": dict(N=16, K=512, nb=2, seq=48, sym=True, blocksize=128, groupsize=256),
    "gqw_asym_g128_bs256": dict(N=16, K=512, nb=2, seq=48, sym=False, blocksize=256, groupsize=128),
    "gqw_sym_g64_bs256": dict(N=12, K=512, nb=2, seq=40, sym=True, blocksize=256, groupsize=64),
    "gqw_asym_g256_bs384": dict(N=8, K=768, nb=2, seq=40, sym=False, blocksize=384, groupsize=256),
}


def main():
    sys.dont_write_bytecode = True
    _install_stubs()
    sys.path.insert(0, REF)
    import torch
    from neural_compressor.torch.algorithms.weight_only.gptq import GPTQ
    from neural_compressor.torch.algorithms.weight_only.utility import quant_weight_w_scale

    g = torch.Generator().manual_seed(20260922)
    out = {}
    for tag, c in CASES.items():
        N, K = c["N"], c["K"]
        layer = torch.nn.Linear(K, N, bias=False)
        W = torch.randn(N, K, generator=g) * 0.05
        layer.weight.data.copy_(W)
        gq = GPTQ(layer, W.clone(), "cpu")
        gq.quantizer.configure(dict(dtype="int", bits=4, sym=c["sym"], group_size=c["groupsize"], mse=False, perchannel=True,
                                    use_double_quant=False, double_quant_dtype="int", double_quant_bits=4, double_quant_sym=False,
                                    double_quant_group_size=128))
        xs = []
        for _ in range(c["nb"]):
            x = torch.randn(1, c["seq"], K, generator=g)
            x[..., ::17] *= 8.0  # a few outlier channels
            xs.append(x)
            gq.add_batch(x, None)
        scale, _, zero, Q = gq.fasterquant(W.clone(), blocksize=c["blocksize"], percdamp=0.01, groupsize=c["groupsize"])
        ints = quant_weight_w_scale(Q.clone(), scale, None, None if c["sym"] else zero, c["groupsize"], dtype="int")
        out[f"{tag}_W"] = W.numpy()
        out[f"{tag}_X"] = torch.cat(xs, 0).numpy()
        out[f"{tag}_scale"] = scale.numpy()
        out[f"{tag}_zero"] = zero.numpy()
        out[f"{tag}_Q"] = Q.numpy()
        out[f"{tag}_ints"] = ints.numpy()
    path = os.path.join(HERE, "gptq_wide_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items() if k.endswith("_Q")})


if __name__ == "__main__":
    main()
