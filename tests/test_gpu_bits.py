"""GPU: every weight-only width the reference's configs tune (bits = 1..8, torch/quantization/config.py:211), not only 2 / 4 / 8.

The reference's module packs any width (modules.py:231 `n_pack = compress_bits // bits`: 3 / 5 / 6 / 7 bits leave high bits of
the word unused; the numpy fallback :520-536 when numba has no packer).  Fixtures: tests/golden/woq_bits_golden.npz, written by
the UNMODIFIED reference (tests/golden/make_golden_bits.py).  Bars: integer work bit-exact; fp16 recover() bit for bit; the
fused forward against F.linear on the oracle's weight to output rounding.
"""

import numpy as np
import pytest
import torch

from oracle import woq_oracle as O

pytestmark = pytest.mark.gpu

CASES = [(b, sch) for b in (1, 2, 3, 5, 6, 7) for sch in ("sym", "asym") if not (b == 1 and sch == "sym")]


def _t(a, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(dev)


def rel_fro(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("bits,scheme", CASES)
def test_module_every_width_against_reference_golden(hip, golden_bits, bits, scheme):
    """MI355XWeightOnlyLinear.pack / unpack / recover / forward == the reference's INCWeightOnlyLinear at this width."""
    from neural_compressor_amd.torch.algorithms.weight_only.modules import MI355XWeightOnlyLinear

    g, tag, N, K, gs = golden_bits, f"b{bits}{scheme}", 21, 150, 32
    zp = _t(g[f"{tag}_zp"], hip) if f"{tag}_zp" in g.files else None
    iw, sc = _t(g[f"{tag}_int"], hip, torch.int32), _t(g[f"{tag}_scale"], hip)
    m = MI355XWeightOnlyLinear(K, N, bits=bits, group_size=gs, zp=zp is not None, device=hip)
    m.pack(iw, sc, zp, None)
    assert np.array_equal(m.qweight.cpu().numpy(), g[f"{tag}_qweight"])
    assert np.array_equal(m.qzeros.cpu().numpy(), g[f"{tag}_qzeros"])
    assert np.array_equal(m.scales.cpu().numpy().view(np.uint16), g[f"{tag}_scales16"].view(np.uint16))
    up = m.unpack()
    assert np.array_equal(up["int_weight"].cpu().numpy(), g[f"{tag}_unpack_int"])
    assert np.array_equal(up["zp"].cpu().numpy(), g[f"{tag}_unpack_zp"])
    assert np.array_equal(m.recover().cpu().numpy(), g[f"{tag}_recover"])  # fp16, bit for bit
    for dt in (torch.float32, torch.bfloat16):  # the other output types: the exact product, rounded once
        got = m.recover(dtype=dt).float().cpu()
        want = O.woq_dense_weight(g[f"{tag}_qweight"], g[f"{tag}_scales16"], g[f"{tag}_qzeros"], N, K, bits, gs, compute_dtype=dt)
        assert torch.equal(got, want)
    # forward: the reference's CPU module multiplies in fp32 on the fp16 weight; ours in fp16 / bf16 with fp32 accumulation
    m.bias = None
    x = _t(g[f"{tag}_x"], hip)
    y = m(x)
    assert m._plan == "dense" and y.dtype == torch.float32  # default route of the odd widths: HIP recover() + library GEMM
    assert rel_fro(y.cpu(), torch.from_numpy(g[f"{tag}_y"])) <= 2e-3
    xb = x.to(torch.bfloat16)
    ref = O.woq_linear(xb.cpu(), g[f"{tag}_qweight"], g[f"{tag}_scales16"], g[f"{tag}_qzeros"], None, N, K, bits, gs, compute_dtype=torch.bfloat16)
    assert rel_fro(m(xb).float().cpu(), ref.to(torch.bfloat16).float()) <= 2e-3
    m.ODD_WIDTH_FUSED = True  # opt-in: inc_woq_gemm's per-element tile form (no dense weight at all)
    assert rel_fro(m(xb).float().cpu(), ref.to(torch.bfloat16).float()) <= 2e-3 and m._plan == ("fused" if bits not in (4, 8) else m._plan)


@pytest.mark.parametrize("bits", [1, 3, 5, 6, 7])
@pytest.mark.parametrize("cbits", [8, 16, 32, 64])
def test_pack_rows_odd_widths_all_containers(hip, golden_bits, bits, cbits):
    from neural_compressor_amd import ops

    g = golden_bits
    raw = _t(g["rows_raw"], hip)
    packed = ops.pack_rows(raw, bits, cbits)
    assert np.array_equal(packed.cpu().numpy(), g[f"rows_b{bits}_c{cbits}"])
    assert np.array_equal(ops.unpack_rows(packed, bits, cbits, False).cpu().numpy(), g[f"rows_b{bits}_c{cbits}_unpack_signed"])
    assert np.array_equal(ops.unpack_rows(packed, bits, cbits, True).cpu().numpy(), g[f"rows_b{bits}_c{cbits}_unpack_masked"])


@pytest.mark.parametrize("bits", [3, 6])
@pytest.mark.parametrize("cd", [0, 1])
def test_non_optimum_module_odd_widths(hip, golden_bits, bits, cd):
    """fp32 scales, compression_dim 0 / 1 (modules.py:270-314) at 3 and 6 bits."""
    from neural_compressor_amd.torch.algorithms.weight_only.modules import MI355XWeightOnlyLinear

    g, tag, N, K, gs = golden_bits, f"raw_b{bits}_cd{cd}", 12, 70, 32
    m = MI355XWeightOnlyLinear(K, N, bits=bits, group_size=gs, zp=True, compression_dim=cd, use_optimum_format=False, device=hip)
    m.pack(_t(g[f"{tag}_int"], hip, torch.int32), _t(g[f"{tag}_scale"], hip), _t(g[f"{tag}_zp"], hip), None)
    assert np.array_equal(m.qweight.cpu().numpy(), g[f"{tag}_qweight"])
    assert np.array_equal(m.qzeros.cpu().numpy(), g[f"{tag}_qzeros"])
    assert np.array_equal(m.recover().cpu().numpy(), g[f"{tag}_recover"])


@pytest.mark.parametrize("bits,N,K,gs,M", [(3, 1000, 1576, 128, 300), (5, 257, 520, 64, 33), (6, 512, 4096, 128, 1), (7, 384, 768, 32, 130),
                                            (2, 640, 1024, 128, 17), (1, 128, 256, 32, 5), (3, 4096, 4096, 128, 64)])
def test_every_width_larger_layers_vs_oracle(hip, bits, N, K, gs, M):
    """Ragged and headline-sized layers: pack -> unpack is the identity, the packed words / recover() equal the numpy oracle on a
    slice of rows, and the fused forward (inc_woq_gemm's per-element tile form: no dense weight, no library GEMM) equals
    F.linear on the oracle's weight to output rounding, is bit-reproducible and exactly linear in x."""
    from neural_compressor_amd import ops
    from neural_compressor_amd.torch.algorithms.weight_only.modules import MI355XWeightOnlyLinear

    g = torch.Generator().manual_seed(bits * 1000 + N)
    iw = torch.randint(0, 2**bits, (N, K), generator=g, dtype=torch.int32)
    G = -(-K // gs)
    sc = torch.rand(N, G, generator=g) * 0.05 + 0.005
    zp = torch.randint(0, 2**bits, (N, G), generator=g, dtype=torch.int32)
    m = MI355XWeightOnlyLinear(K, N, bits=bits, group_size=gs, zp=True, device=hip)
    m.pack(iw.to(hip), sc.to(hip), zp.to(hip), None)
    m.bias = None
    ui, uz = ops.woq_unpack(m.qweight, m.qzeros, N, K, G, bits)
    assert torch.equal(ui.cpu().to(torch.int32), iw)
    rows = min(N, 64)
    oqw, oqz, osc = O.woq_pack_optimum(iw[:rows].numpy(), sc[:rows].numpy(), zp[:rows].numpy(), bits)
    assert np.array_equal(m.qweight[:, :rows].cpu().numpy(), oqw)
    assert np.array_equal(m.recover()[:rows].cpu().numpy(), O.woq_recover(oqw, osc, oqz, rows, K, bits, gs))
    x = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(hip)
    w = m.recover(dtype=torch.bfloat16).float()
    ref = x.float() @ w.t()
    y0 = m(x)
    assert m._plan == "dense"  # the default route of these widths: HIP recover() + the library GEMM (8-10 x faster than the fused form)
    assert rel_fro(y0.float(), ref.to(torch.bfloat16).float()) <= 2e-3
    m.ODD_WIDTH_FUSED = True   # opt-in: the memory-saving fused form
    y = m(x)
    assert m._plan == "fused"
    assert rel_fro(y.float(), ref.to(torch.bfloat16).float()) <= 2e-3
    assert torch.equal(m(x), y)
    assert torch.equal(m(x * 2), y * 2)


def test_rtn_config_at_3_and_6_bits_end_to_end(hip):
    """RTNConfig(bits=3 / 6) through prepare / convert: codes and scales equal the oracle's quant_tensor, the packed module
    round-trips them, and its forward follows the fake-quantised weight."""
    from neural_compressor_amd.torch.quantization import RTNConfig, convert, prepare

    for bits in (3, 6):
        torch.manual_seed(bits)
        model = torch.nn.Sequential(torch.nn.Linear(160, 96, bias=True), torch.nn.ReLU(), torch.nn.Linear(96, 40, bias=False)).to(hip)
        ws = [model[0].weight.detach().cpu().clone(), model[2].weight.detach().cpu().clone()]
        q = convert(prepare(model, RTNConfig(bits=bits, group_size=32, use_sym=False)))
        for idx, w in zip((0, 2), ws):
            mod = q[idx]
            assert type(mod).__name__ == "MI355XWeightOnlyLinear" and mod.bits == bits
            oi, os_, oz = O.quant_tensor(w.clone(), bits=bits, group_size=32, scheme="asym", return_int=True)
            up = mod.unpack()
            assert torch.equal(up["int_weight"].cpu().to(torch.int32), oi.to(torch.int32))
            assert torch.equal(up["zp"].cpu().to(torch.int32), oz.to(torch.int32))
            assert torch.equal(up["scales"].cpu(), os_.to(torch.float16))
        x = torch.randn(7, 160, device=hip)
        y = q(x)
        assert y.shape == (7, 40) and torch.isfinite(y).all()


# ---------------------------------------------------------------------------------------------------
# model level: the unmodified reference's quantised tiny Llama at 3 / 6 bits (tests/golden/make_golden_bits_models.py)
# ---------------------------------------------------------------------------------------------------
def _field_match(a, b, bits):
    """Fraction of identical `bits`-wide fields of two packed int32 arrays (n_pack = 32 // bits fields per word)."""
    a, b = a.astype(np.uint32).reshape(-1), b.astype(np.uint32).reshape(-1)
    npk, mask = 32 // bits, np.uint32(2**bits - 1)
    same = sum(int((((a >> np.uint32(bits * e)) & mask) == ((b >> np.uint32(bits * e)) & mask)).sum()) for e in range(npk))
    return same / (npk * a.size)


def _woq(model):
    from neural_compressor_amd.torch.algorithms.weight_only.modules import MI355XWeightOnlyLinear

    return {n: m for n, m in model.named_modules() if isinstance(m, MI355XWeightOnlyLinear)}


@pytest.mark.parametrize("bits", [3, 6])
def test_rtn_tiny_llama_odd_widths_bit_exact(hip, bits):
    import os

    from neural_compressor_amd.torch.quantization import RTNConfig, quantize
    from tests.model_zoo import calib_ids, tiny_llama

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"rtn_tiny_llama_b{bits}_asym_g32.npz"))
    q = quantize(tiny_llama(), RTNConfig(bits=bits, group_size=32, use_sym=False, use_layer_wise=False))
    mods = _woq(q)
    assert len(mods) == int(g["n_modules"]) == 14
    for name, m in mods.items():
        assert m.bits == bits
        assert np.array_equal(m.qweight.cpu().numpy(), g[f"{name}.qweight"]), name
        assert np.array_equal(m.qzeros.cpu().numpy(), g[f"{name}.qzeros"]), name
        assert np.array_equal(m.scales.cpu().numpy().view(np.uint16), g[f"{name}.scales"].view(np.uint16)), name
    with torch.no_grad():
        y = q(calib_ids()[0].to("cuda")).logits.float().cpu()
    ref = torch.from_numpy(g["logits"])
    assert float((y - ref).norm() / ref.norm()) <= 1e-2  # fp16 fused forward vs the reference's fp32 F.linear


def test_gptq_tiny_llama_3bit_vs_reference(hip):
    """GPTQConfig(bits=3) end to end: Hessians, factor, column loop at maxq = 7, 10-field words.  Block 0 sees the reference's inputs:
    its codes must agree except at rounding ties; block 1 inherits what block 0 flipped."""
    import os

    from neural_compressor_amd.torch.quantization import GPTQConfig, convert, prepare
    from tests.model_zoo import calib_ids, tiny_llama

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gptq_tiny_llama_b3_sym_g32.npz"))
    model = prepare(tiny_llama(), GPTQConfig(bits=3, group_size=32, use_sym=True, block_size=128))
    for x in calib_ids():
        model(x)
    q = convert(model)
    mods = _woq(q)
    assert len(mods) == int(g["n_modules"]) == 14
    per_block = {0: [], 1: []}
    for name, m in mods.items():
        assert m.bits == 3 and m.qweight.shape == g[f"{name}.qweight"].shape
        per_block[0 if ".layers.0." in name else 1].append(_field_match(m.qweight.cpu().numpy(), g[f"{name}.qweight"], 3))
        assert np.array_equal(m.qzeros.cpu().numpy(), g[f"{name}.qzeros"]), name  # sym: the constant zero point, 10 fields per word
    print(f"\n[gptq tiny_llama 3-bit] codes identical: block 0 {min(per_block[0]):.5f}, block 1 {min(per_block[1]):.5f}")
    assert min(per_block[0]) >= 0.999 and min(per_block[1]) >= 0.99, per_block
    with torch.no_grad():
        y = q(calib_ids()[0].to("cuda")).logits.float().cpu()
    ref = torch.from_numpy(g["logits"])
    assert float((y - ref).norm() / ref.norm()) <= 3e-2
