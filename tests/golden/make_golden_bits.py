"""Generate tests/golden/woq_bits_golden.npz from the UNMODIFIED reference: INCWeightOnlyLinear at the widths other than 2 / 4 / 8.

The reference packs ANY `bits` (modules.py:231 `n_pack = compress_bits // bits`; the numpy fallback :520-536 for widths the
numba packers do not cover) and its configs tune bits = [4, 1, 2, 3, 5, 6, 7, 8] (torch/quantization/config.py:211): 3 / 5 / 6 / 7
bits leave the high 2 / 2 / 2 / 4 bits of an int32 word unused.  Cases: optimum layout (pack / unpack / recover, sym and asym, K
and N that are NOT multiples of n_pack, a group size no n_pack divides), the generic row packer for every container, and a
non-optimum module (compression_dim 0 and 1).

Run in the build container only (the reference does not travel to the GPU box):
    python tests/golden/make_golden_bits.py
"""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, _install_stubs  # noqa: E402


def main():
    sys.dont_write_bytecode = True
    _install_stubs()
    sys.path.insert(0, REF)
    import torch
    from neural_compressor.torch.algorithms.weight_only.modules import INCWeightOnlyLinear
    from neural_compressor.torch.algorithms.weight_only.utility import quant_tensor

    out = {}
    g = torch.Generator().manual_seed(20240905)
    # ---- 1. optimum layout: quant_tensor -> pack -> unpack -> recover at every width the configs tune ------------------
    for bits in (1, 2, 3, 5, 6, 7):
        for scheme in ("sym", "asym"):
            if bits == 1 and scheme == "sym":
                continue  # (sym 1-bit: the reference's own quant range is degenerate, utility.py:186-189; asym covers the packer)
            N, K, gs = 21, 150, 32  # N and K are multiples of no n_pack; 150 = 4 groups of 32 + a tail of 22
            tag = f"b{bits}{scheme}"
            w = torch.randn(N, K, generator=g)
            iw, sc, zp = quant_tensor(w.clone(), bits=bits, group_size=gs, scheme=scheme, return_int=True)
            out[f"{tag}_w"] = w.numpy()
            out[f"{tag}_int"] = iw.numpy().copy()
            out[f"{tag}_scale"] = sc.numpy().copy()
            if zp is not None:
                out[f"{tag}_zp"] = zp.numpy().copy()
            mod = INCWeightOnlyLinear(K, N, bits=bits, group_size=gs, zp=zp is not None, device="cpu")
            mod.pack(iw.clone().int(), sc.clone(), None if zp is None else zp.clone(), None)
            out[f"{tag}_qweight"] = mod.qweight.numpy()
            out[f"{tag}_qzeros"] = mod.qzeros.numpy()
            out[f"{tag}_scales16"] = mod.scales.numpy()
            up = mod.unpack()
            out[f"{tag}_unpack_int"] = up["int_weight"].numpy()
            out[f"{tag}_unpack_zp"] = up["zp"].numpy()
            out[f"{tag}_recover"] = mod.recover().numpy()
            # the forward the reference runs on CPU: fp32 F.linear on the cached recovered weight (modules.py:594-610)
            x = torch.randn(5, K, generator=g)
            out[f"{tag}_x"] = x.numpy()
            mod2 = INCWeightOnlyLinear(K, N, bits=bits, group_size=gs, zp=zp is not None, device="cpu")
            mod2.pack(iw.clone().int(), sc.clone(), None if zp is None else zp.clone(), None)
            mod2.bias = None
            out[f"{tag}_y"] = mod2(x).detach().float().numpy()

    # ---- 2. generic row packer: every container at the odd widths (pack_tensor / unpack_tensor, modules.py:445-592) -----
    raw = torch.randint(-128, 128, (5, 43), generator=g, dtype=torch.int32)
    out["rows_raw"] = raw.numpy()
    for bits in (1, 3, 5, 6, 7):
        for cdt, cb in ((torch.int8, 8), (torch.int16, 16), (torch.int32, 32), (torch.int64, 64)):
            for zp in (False, True):  # with qzeros the unpacked fields are masked (unsigned), without they are sign-extended
                mod = INCWeightOnlyLinear(43, 5, bits=bits, group_size=-1, zp=zp, compression_dtype=cdt, use_optimum_format=False, device="cpu")
                packed = mod.pack_tensor(raw.clone())
                if not zp:
                    out[f"rows_b{bits}_c{cb}"] = packed.numpy()
                out[f"rows_b{bits}_c{cb}_unpack_{'masked' if zp else 'signed'}"] = mod.unpack_tensor(packed.clone()).numpy()

    # ---- 3. a non-optimum module (fp32 scales, compression_dim 0 / 1) at 3 and 6 bits ----------------------------------
    for bits in (3, 6):
        for cd in (0, 1):
            N, K, gs = 12, 70, 32
            tag = f"raw_b{bits}_cd{cd}"
            w = torch.randn(N, K, generator=g)
            iw, sc, zp = quant_tensor(w.clone(), bits=bits, group_size=gs, scheme="asym", return_int=True)
            mod = INCWeightOnlyLinear(K, N, bits=bits, group_size=gs, zp=True, compression_dim=cd, use_optimum_format=False, device="cpu")
            mod.pack(iw.clone().int(), sc.clone(), zp.clone(), None)
            out[f"{tag}_int"] = iw.numpy().copy()
            out[f"{tag}_scale"] = sc.numpy().copy()
            out[f"{tag}_zp"] = zp.numpy().copy()
            out[f"{tag}_qweight"] = mod.qweight.numpy()
            out[f"{tag}_qzeros"] = mod.qzeros.numpy()
            out[f"{tag}_recover"] = mod.recover().numpy()

    path = os.path.join(HERE, "woq_bits_golden.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
