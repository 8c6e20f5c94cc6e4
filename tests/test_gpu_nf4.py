"""-m gpu: quant_tensor's non-integer branches on the MI355X -- NF4 / FP4 code-book quantisation (inc_codebook_quant ==
quantize_4bit, reference utility.py:112-149), double quantisation of the scales (:378-436) and RTN with dtype="nf4" through
the packed module -- against outputs of the UNMODIFIED reference (tests/golden/nf4_golden.npz) and the oracle."""

import os

import numpy as np
import pytest
import torch

from oracle import woq_oracle as O
from tests.model_zoo import calib_ids, tiny_llama

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = {
    "nf4_g32": dict(dtype="nf4", group_size=32), "fp4_g32": dict(dtype="fp4", group_size=32),
    "fp4e2m1_g32": dict(dtype="fp4_e2m1", group_size=32), "nf4_tail": dict(dtype="nf4", group_size=32),
    "nf4_pc": dict(dtype="nf4", group_size=-1), "nf4_zero": dict(dtype="nf4", group_size=32),
    "nf4_q09": dict(dtype="nf4", group_size=32, quantile=0.9),
}
DQ = dict(double_quant=True, double_quant_dtype="int", double_quant_bits=8, double_quant_scheme="asym", double_quant_group_size=256)


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(ROOT, "tests", "golden", "nf4_golden.npz"))


@pytest.mark.parametrize("tag", list(CASES))
def test_codebook_quant_vs_reference_golden(hip, g, tag):
    """Bit-exact: fake-quantised weights, stored integers and scales (tail group, per-channel, quantile, an all-zero group)."""
    from neural_compressor_amd.torch.algorithms.weight_only.utility import quant_tensor

    w = torch.from_numpy(g[f"{tag}_w"]).to(hip)
    qdq = quant_tensor(w.clone(), **CASES[tag])
    assert np.array_equal(qdq.cpu().numpy(), g[f"{tag}_qdq"])
    iw, sc, zp = quant_tensor(w.clone(), return_int=True, **CASES[tag])
    assert zp is None
    assert np.array_equal(iw.cpu().numpy().astype(np.float32), g[f"{tag}_int"])
    assert np.array_equal(sc.cpu().numpy(), g[f"{tag}_scale"])


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16, torch.float32])
def test_codebook_quant_4096_vs_oracle(hip, dt):
    """BASELINE-size layer, every dtype: the kernel rounds each step to the weight dtype exactly where the torch ops do."""
    from neural_compressor_amd.torch.algorithms.weight_only.utility import quant_tensor

    torch.manual_seed(3)
    w = (torch.randn(1024, 4096) * 0.02).to(dt)
    ref = O.quant_tensor(w, dtype="nf4", group_size=128)
    got = quant_tensor(w.clone().to(hip), dtype="nf4", group_size=128)
    assert torch.equal(got.cpu(), ref)
    ri, rs, _ = O.quant_tensor(w, dtype="nf4", group_size=128, return_int=True)
    gi, gs, _ = quant_tensor(w.clone().to(hip), dtype="nf4", group_size=128, return_int=True)
    assert torch.equal(gi.cpu().float(), ri.float()) and torch.equal(gs.cpu(), rs.float())


@pytest.mark.parametrize("tag,kw", [("dq_int4", dict(dtype="int", bits=4, group_size=32, scheme="asym")), ("dq_nf4", dict(dtype="nf4", group_size=32))])
def test_double_quant_vs_reference_golden(hip, g, tag, kw):
    """The [N, G] scales quantised as ONE row (int8, mean-centred "asym", groups of 256): integers / zero points bit-exact, the
    double-quantised scales and the dequantised weight to fp32 rounding (the mean of the scales is a device reduction)."""
    from neural_compressor_amd.torch.algorithms.weight_only.utility import quant_tensor

    w = torch.from_numpy(g[f"{tag}_w"]).to(hip)
    iw, sc, zp = quant_tensor(w.clone(), return_int=True, **kw, **DQ)
    assert np.array_equal(iw.cpu().numpy().astype(np.float32), g[f"{tag}_int"])
    np.testing.assert_allclose(sc.cpu().numpy(), g[f"{tag}_scale"], rtol=2e-6, atol=0)
    if f"{tag}_zp" in g.files:
        assert np.array_equal(zp.cpu().numpy(), g[f"{tag}_zp"])
    qdq = quant_tensor(w.clone(), **kw, **DQ)
    np.testing.assert_allclose(qdq.cpu().numpy(), g[f"{tag}_qdq"], rtol=3e-6, atol=1e-9)


def test_double_quant_16bit_and_ragged_vs_reference_golden(hip, g):
    """bf16 weights: the scales are double-quantised in the dtype the reference's actor returns them in (fp32 from the asym actor,
    bf16 from the sym one: mean / sub / quant / add round to it); a ragged K returns (ints, scale, zp) WITHOUT double
    quantisation, the reference's "case 3" (utility.py:334-376)."""
    from neural_compressor_amd.torch.algorithms.weight_only.utility import quant_tensor

    w = torch.from_numpy(g["dq_bf16_w"]).to(torch.bfloat16).to(hip)
    for tag, scheme, is16 in (("dq_bf16", "asym", False), ("dq_bf16_sym", "sym", True)):
        kw = dict(dtype="int", bits=4, group_size=32, scheme=scheme)
        iw, sc, zp = quant_tensor(w.clone(), return_int=True, **kw, **DQ)
        assert np.array_equal(iw.cpu().float().numpy(), g[f"{tag}_int"]), tag
        assert (sc.dtype == torch.bfloat16) == is16 == bool(g["dq_bf16_sym_scale_is_bf16"] if is16 else False), tag
        # the mean of the scales is a device reduction: allow the last bit of the working dtype on a few entries
        ulp = 2.0 ** -7 if is16 else 4e-6
        d = np.abs(sc.cpu().float().numpy() - g[f"{tag}_scale"]) / np.abs(g[f"{tag}_scale"])
        assert float(d.max()) <= ulp and float((d > 0).mean()) <= (0.05 if is16 else 1.0), (tag, float(d.max()), float((d > 0).mean()))
        qdq = quant_tensor(w.clone(), **kw, **DQ)
        d = np.abs(qdq.cpu().float().numpy() - g[f"{tag}_qdq"]) / np.maximum(np.abs(g[f"{tag}_qdq"]), 1e-6)
        assert float(d.max()) <= 2.0 ** -7 and float((d > 0).mean()) <= 0.05, (tag, float(d.max()), float((d > 0).mean()))
    kw = dict(dtype="int", bits=4, group_size=32, scheme="asym")
    wr = torch.from_numpy(g["dq_ragged_w"]).to(hip)
    res = quant_tensor(wr.clone(), **kw, **DQ)  # (the reference returns the tuple here even without return_int)
    assert isinstance(res, tuple)
    assert np.array_equal(res[0].cpu().numpy().astype(np.float32), g["dq_ragged_int"])
    assert np.array_equal(res[1].cpu().numpy(), g["dq_ragged_scale"]) and np.array_equal(res[2].cpu().numpy(), g["dq_ragged_zp"])


def test_rtn_nf4_tiny_llama_vs_reference(hip, g):
    """RTNConfig(dtype="nf4"): the reference packs the stored integers in its NON-optimum layout (qweight [N, K/8] int32, scales
    [N, G], no zero points, modules.py:213-221) -- every packed buffer bit-identical, recover() = code-book value x scale, and
    the model computes the reference's logits."""
    from neural_compressor_amd.torch.algorithms.weight_only.modules import MI355XWeightOnlyLinear
    from neural_compressor_amd.torch.quantization import RTNConfig, quantize

    q = quantize(tiny_llama(), RTNConfig(dtype="nf4", group_size=32, use_layer_wise=False))
    mods = {n: m for n, m in q.named_modules() if isinstance(m, MI355XWeightOnlyLinear)}
    assert len(mods) == int(g["rtn_nf4_n_modules"]) == 14
    for n, m in mods.items():
        assert not m.use_optimum_format and not hasattr(m, "qzeros")
        assert np.array_equal(m.qweight.cpu().numpy(), g[f"rtn_nf4_{n}.qweight"]), n
        assert np.array_equal(m.scales.cpu().numpy(), g[f"rtn_nf4_{n}.scales"]), n
    with torch.no_grad():
        for mod in q.modules():  # fp16 compute like the reference's accelerator semantics (see test_gpu_models._to_half)
            for p in mod.parameters(recurse=False):
                if p.is_floating_point():
                    p.data = p.data.half()
        y = q(calib_ids()[0].to(hip)).logits.float().cpu().numpy()
    ref = g["rtn_nf4_logits"]
    assert np.linalg.norm(y - ref) / np.linalg.norm(ref) <= 2e-2


def test_rtn_nf4_save_load_round_trip(hip, g, tmp_path):
    """NF4 modules live in the reference's non-optimum layout; save() keeps that layout in a side file and load() rebuilds it:
    every packed buffer and the logits survive the round trip."""
    from neural_compressor_amd.torch.algorithms.weight_only.modules import MI355XWeightOnlyLinear
    from neural_compressor_amd.torch.algorithms.weight_only.save_load import load, save
    from neural_compressor_amd.torch.quantization import RTNConfig, quantize

    q = quantize(tiny_llama(), RTNConfig(dtype="nf4", group_size=32, use_layer_wise=False))
    ids = calib_ids()[0].to(hip)
    with torch.no_grad():
        y0 = q(ids).logits.float().cpu()
    save(q, str(tmp_path))
    back = load(str(tmp_path), original_model=tiny_llama(), device=hip)
    a = {n: m for n, m in q.named_modules() if isinstance(m, MI355XWeightOnlyLinear)}
    b = {n: m for n, m in back.named_modules() if isinstance(m, MI355XWeightOnlyLinear)}
    assert a.keys() == b.keys() and len(a) == 14
    for n in a:
        assert not b[n].use_optimum_format and b[n].dtype == "nf4"
        assert torch.equal(a[n].qweight, b[n].qweight) and torch.equal(a[n].scales, b[n].scales)
    with torch.no_grad():
        y1 = back(ids).logits.float().cpu()
    assert torch.equal(y0, y1)


@pytest.mark.parametrize("dtype", ["nf4", "fp4", "fp4_e2m1"])
@pytest.mark.parametrize("wd", ["f32", "bf16"])
def test_quantize_4bit_with_given_scale_vs_reference_golden(hip, dtype, wd):
    """quantize_4bit(tensor, scale=...) (reference utility.py:127-128) -> inc_codebook_quant_with_scale: fake-quantised values and
    stored integers bit-exact against the unmodified reference (scales 0.7 .. 1.3 x the rows' own max: both ends of the book
    saturate); double_quant_return_int raises what the reference raises (it has never worked there, :383-405)."""
    from neural_compressor_amd.torch.algorithms.weight_only.utility import quant_tensor, quantize_4bit

    g = np.load(os.path.join(ROOT, "tests", "golden", "q4scale_golden.npz"))
    td = torch.float32 if wd == "f32" else torch.bfloat16
    tag = f"q4s_{dtype}_{wd}"
    w = torch.from_numpy(g[f"{tag}_w"]).to(td).to(hip)
    sc = torch.from_numpy(g[f"{tag}_scale"]).to(td).to(hip)
    t = w.clone()
    out = quantize_4bit(t, dtype=dtype, scale=sc)
    assert out.data_ptr() == t.data_ptr()  # in place, like the reference
    assert np.array_equal(out.float().cpu().numpy(), g[f"{tag}_qdq"])
    ints, s2, zp = quantize_4bit(w.clone(), dtype=dtype, return_int=True, scale=sc)
    assert zp is None and s2 is sc
    assert np.array_equal(ints.cpu().numpy().astype(np.float32), g[f"{tag}_int"])
    # without a scale the same call computes its own (the pre-existing path through the same kernel)
    own = quantize_4bit(w.clone(), dtype=dtype)
    ref = O.quantize_4bit(w.cpu().clone(), dtype=dtype)
    assert torch.equal(own.cpu(), ref)
    kind, msg = str(g["dqri_error"]).split(": ", 1)
    with pytest.raises(ValueError, match=msg.split("(")[0].strip()):
        quant_tensor(w.float().clone(), bits=4, group_size=32, scheme="asym", return_int=True, double_quant=True, double_quant_return_int=True)
