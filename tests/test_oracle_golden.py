"""CPU: the oracle restatement must reproduce the outputs of the unmodified reference stored in tests/golden/."""

import os

import numpy as np
import pytest
import torch

from oracle import woq_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kat_pack(golden):
    qw, qz, sc = O.woq_pack_optimum(golden["kat_ints"], np.full((8, 2), 0.5, np.float32), None, 4)
    assert np.array_equal(qw, golden["kat_qweight"])
    assert np.array_equal(qz, golden["kat_qzeros"])
    assert np.array_equal(sc.view(np.uint16), golden["kat_scales"].view(np.uint16))
    # SURVEY 8(c) literal KAT: row0 = -8..7 -> 0x76543210 / 0xFEDCBA98, sym qzeros = 0x77777777
    u = qw.view(np.uint32)
    assert u[0, 0] == 0x76543210 and u[1, 0] == 0xFEDCBA98
    assert np.all(qz.view(np.uint32) == 0x77777777)
    rec = O.woq_recover(qw, sc, qz, 8, 16, 4, 8)
    assert np.array_equal(rec, golden["kat_recover"])


@pytest.mark.parametrize("tag,N,K,gs,bits", [("m4sym", 24, 64, 32, 4), ("m4asym", 20, 96, 32, 4), ("m8sym", 16, 64, 64, 8), ("m8asym", 16, 64, 32, 8)])
def test_module_pack_unpack_recover(golden, tag, N, K, gs, bits):
    zp = golden[f"{tag}_zp"] if f"{tag}_zp" in golden.files else None
    qw, qz, sc = O.woq_pack_optimum(golden[f"{tag}_int"], golden[f"{tag}_scale"], zp, bits)
    assert np.array_equal(qw, golden[f"{tag}_qweight"])
    assert np.array_equal(qz, golden[f"{tag}_qzeros"])
    assert np.array_equal(sc.view(np.uint16), golden[f"{tag}_scales16"].view(np.uint16))
    G = sc.shape[0]
    iw, z = O.woq_unpack_optimum(qw, qz, N, K, G, bits)
    assert np.array_equal(iw, golden[f"{tag}_unpack_int"])
    assert np.array_equal(z, golden[f"{tag}_unpack_zp"])
    assert np.array_equal(O.woq_recover(qw, sc, qz, N, K, bits, gs), golden[f"{tag}_recover"])


@pytest.mark.parametrize("bits", [2, 4, 8])
@pytest.mark.parametrize("cbits", [8, 16, 32, 64])
def test_pack_rows_all_containers(golden, bits, cbits):
    packed = O.pack_rows(golden["rows_raw"], bits, cbits)
    assert np.array_equal(packed, golden[f"rows_b{bits}_c{cbits}"])
    assert np.array_equal(O.unpack_rows(packed, bits, cbits, mask_sign=False), golden[f"rows_b{bits}_c{cbits}_unpack_signed"])


BITS_CASES = [(b, sch) for b in (1, 2, 3, 5, 6, 7) for sch in ("sym", "asym") if not (b == 1 and sch == "sym")]


@pytest.mark.parametrize("bits,scheme", BITS_CASES)
def test_module_pack_unpack_recover_every_width(golden_bits, bits, scheme):
    """Widths other than 2 / 4 / 8 (reference modules.py:231 n_pack = 32 // bits: 3 / 5 / 6 / 7 bits leave high bits unused)."""
    g, tag, N, K, gs = golden_bits, f"b{bits}{scheme}", 21, 150, 32
    zp = g[f"{tag}_zp"] if f"{tag}_zp" in g.files else None
    qw, qz, sc = O.woq_pack_optimum(g[f"{tag}_int"], g[f"{tag}_scale"], zp, bits)
    assert np.array_equal(qw, g[f"{tag}_qweight"])
    assert np.array_equal(qz, g[f"{tag}_qzeros"])
    assert np.array_equal(sc.view(np.uint16), g[f"{tag}_scales16"].view(np.uint16))
    iw, z = O.woq_unpack_optimum(qw, qz, N, K, sc.shape[0], bits)
    assert np.array_equal(iw, g[f"{tag}_unpack_int"])
    assert np.array_equal(z, g[f"{tag}_unpack_zp"])
    assert np.array_equal(O.woq_recover(qw, sc, qz, N, K, bits, gs), g[f"{tag}_recover"])
    # the reference's CPU forward: fp32 F.linear on the fp16 recovered weight (modules.py:594-610)
    y = O.woq_linear(torch.from_numpy(g[f"{tag}_x"]), qw, sc, qz, None, N, K, bits, gs, compute_dtype=torch.float32,
                     dense=torch.from_numpy(g[f"{tag}_recover"]).float())
    assert float((y - torch.from_numpy(g[f"{tag}_y"])).norm() / torch.from_numpy(g[f"{tag}_y"]).norm()) <= 1e-6


@pytest.mark.parametrize("bits", [1, 3, 5, 6, 7])
@pytest.mark.parametrize("cbits", [8, 16, 32, 64])
def test_pack_rows_odd_widths_all_containers(golden_bits, bits, cbits):
    g = golden_bits
    packed = O.pack_rows(g["rows_raw"], bits, cbits)
    assert np.array_equal(packed, g[f"rows_b{bits}_c{cbits}"])
    assert np.array_equal(O.unpack_rows(packed, bits, cbits, mask_sign=False), g[f"rows_b{bits}_c{cbits}_unpack_signed"])
    assert np.array_equal(O.unpack_rows(packed, bits, cbits, mask_sign=True), g[f"rows_b{bits}_c{cbits}_unpack_masked"])


QT_CASES = {
    "qt_sym4_g32": dict(bits=4, group_size=32, scheme="sym"),
    "qt_asym4_g32": dict(bits=4, group_size=32, scheme="asym"),
    "qt_sym4_g128_tail": dict(bits=4, group_size=128, scheme="sym"),
    "qt_asym4_g128_tail": dict(bits=4, group_size=128, scheme="asym"),
    "qt_sym8_pc": dict(bits=8, group_size=-1, scheme="sym"),
    "qt_asym8_pc": dict(bits=8, group_size=-1, scheme="asym"),
    "qt_sym4_full": dict(bits=4, group_size=32, scheme="sym", full_range=True),
    "qt_sym4_q09": dict(bits=4, group_size=32, scheme="sym", quantile=0.9),
    "qt_asym4_q085": dict(bits=4, group_size=32, scheme="asym", quantile=0.85),
    "qt_sym3_g32": dict(bits=3, group_size=32, scheme="sym"),
}


@pytest.mark.parametrize("tag", list(QT_CASES))
def test_quant_tensor(golden, tag):
    w = torch.from_numpy(golden["qt_w"])
    kw = QT_CASES[tag]
    assert np.array_equal(O.quant_tensor(w, **kw).numpy(), golden[f"{tag}_qdq"])
    iw, sc, zp = O.quant_tensor(w, return_int=True, **kw)
    assert np.array_equal(iw.numpy(), golden[f"{tag}_int"])
    assert np.array_equal(sc.numpy(), golden[f"{tag}_scale"])
    if zp is not None:
        assert np.array_equal(zp.numpy(), golden[f"{tag}_zp"])


@pytest.mark.parametrize("tag,scheme", [("qtbf16_sym", "sym"), ("qtbf16_asym", "asym")])
def test_quant_tensor_bf16(golden, tag, scheme):
    w = torch.from_numpy(golden["qtbf16_w"]).to(torch.bfloat16)
    assert np.array_equal(O.quant_tensor(w, bits=4, group_size=128, scheme=scheme).float().numpy(), golden[f"{tag}_qdq"])
    iw, sc, zp = O.quant_tensor(w, bits=4, group_size=128, scheme=scheme, return_int=True)
    assert np.array_equal(iw.float().numpy(), golden[f"{tag}_int"])
    assert np.array_equal(sc.float().numpy(), golden[f"{tag}_scale"])


def test_search_clip(golden):
    w = torch.from_numpy(golden["qt_w"])
    assert O.search_clip(w, bits=4, group_size=32, scheme="sym") == float(golden["clip_sym4_g32"])
    assert O.search_clip(w, bits=4, group_size=128, scheme="asym") == float(golden["clip_asym4_g128"])


GQ_CASES = {
    "gq_sym_g32": dict(bits=4, sym=True, blocksize=128, groupsize=32),
    "gq_asym_g32": dict(bits=4, sym=False, blocksize=128, groupsize=32),
    "gq_sym_pc": dict(bits=4, sym=True, blocksize=128, groupsize=-1),
    "gq_sym_g128_2blk": dict(bits=4, sym=True, blocksize=128, groupsize=128),
    "gq_sym_g32_blk2048": dict(bits=4, sym=True, blocksize=2048, groupsize=32),
    "gq_sym_act": dict(bits=4, sym=True, blocksize=128, groupsize=32, act_order=True),
    "gq_sym8_g64": dict(bits=8, sym=True, blocksize=128, groupsize=64),
    "gq_sym_g32_mse": dict(bits=4, sym=True, blocksize=128, groupsize=32, mse=True),
    "gq_asym_g64_mse": dict(bits=4, sym=False, blocksize=128, groupsize=64, mse=True),
}


@pytest.mark.parametrize("tag", list(GQ_CASES))
def test_gptq_layer(golden, tag):
    kw = GQ_CASES[tag]
    W = torch.from_numpy(golden[f"{tag}_W"])
    X = torch.from_numpy(golden[f"{tag}_X"])
    H, n = torch.zeros(W.shape[1], W.shape[1]), 0
    for j in range(X.shape[0]):
        H, n = O.gptq_add_batch(H, n, X[j : j + 1])
    assert np.array_equal(H.numpy(), golden[f"{tag}_H"])
    r = O.gptq_fasterquant(W, H, **kw)
    assert np.array_equal(r["scale"].numpy(), golden[f"{tag}_scale"])
    assert np.array_equal(r["zero"].numpy(), golden[f"{tag}_zero"])
    assert np.array_equal(r["Q"].numpy(), golden[f"{tag}_Q"])
    ints = O.gptq_export_ints(r["Q"], r["scale"], r["zero"], kw["sym"], kw["groupsize"], r["perm"])
    assert np.array_equal(ints.numpy(), golden[f"{tag}_ints"].astype(np.int32))


@pytest.mark.parametrize("tag,kw", [
    ("gq_sym_static", dict(bits=4, sym=True, blocksize=128, groupsize=32, static_groups=True)),
    ("gq_asym_act_static", dict(bits=4, sym=False, blocksize=128, groupsize=32, act_order=True, static_groups=True)),
])
def test_gptq_static_groups(golden, tag, kw):
    """static_groups: the reference returns only the last group's (scale, zero) (gptq.py:1341-1345); Q is the output."""
    W = torch.from_numpy(golden[f"{tag}_W"])
    r = O.gptq_fasterquant(W, torch.from_numpy(golden[f"{tag}_H"]), **kw)
    assert np.array_equal(r["Q"].numpy(), golden[f"{tag}_Q"])
    assert np.array_equal(r["scale"].numpy(), golden[f"{tag}_scale"])
    if kw.get("act_order"):
        assert np.array_equal(r["perm"].numpy(), golden[f"{tag}_perm"])


def test_awq_checkpoint_repack(golden):
    """Integer restatement of repack_awq_to_optimum_format == the reference's dequantise-and-re-round route."""
    qw, qz = O.awq_repack_to_optimum(golden["awqpack_qweight_in"], golden["awqpack_qzeros_in"], 4)
    assert np.array_equal(qw, golden["awqpack_qweight"])
    assert np.array_equal(qz, golden["awqpack_qzeros"])


def test_smooth_quant_functions(golden):
    """SmoothQuant restatements == the reference's in-tree functions (cal_scale, quant_dequant_w_v1, quant_dequant_x_v1,
    SQLinearWrapper) on the same inputs."""
    amax_x = torch.from_numpy(golden["sq_amax_x"])
    w1, w2 = torch.from_numpy(golden["sq_w1"]), torch.from_numpy(golden["sq_w2"])
    for a in (0.5, 0.8):
        s = O.sq_cal_scale(amax_x.clone(), [w1, w2], a)
        assert np.array_equal(s.numpy(), golden[f"sq_scale_a{int(a * 10)}"])
    assert float(O.sq_cal_scale(amax_x.clone(), [w1, w2], 0.5)[5]) == 1.0
    q, sw, qdq = O.sq_quant_w(w1)
    assert np.array_equal(qdq.numpy(), golden["sq_qdq_w_sym"])
    assert int(q.min()) >= -128 and int(q.max()) <= 127
    x = torch.from_numpy(golden["sq_x"])
    assert np.array_equal(O.sq_quant_dequant_x(x.clone(), x.min(dim=0)[0], x.max(dim=0)[0]).numpy(), golden["sq_qdq_x"])
    in_scale = torch.from_numpy(golden["sq_wrap_in_scale"])
    sx, zp = O.sq_act_qparams(in_scale, x.min(dim=0)[0], x.max(dim=0)[0])
    assert np.float32(sx) == golden["sq_wrap_scale"].reshape(-1)[0] and zp == int(golden["sq_wrap_zp"].reshape(-1)[0])
    # the wrapper holds W / input_scale and multiplies the activations by input_scale: same function as the float layer
    assert np.allclose((w1 / in_scale.view(1, -1)).numpy(), golden["sq_wrap_weight"], rtol=0, atol=0)
    y = torch.nn.functional.linear(x * in_scale, torch.from_numpy(golden["sq_wrap_weight"]))
    assert np.allclose(y.numpy(), golden["sq_wrap_out"], rtol=1e-6, atol=1e-6)
    # and the integer W8A8 form stays within the quantisation error of that float output
    y8 = O.sq_w8a8_linear(x, torch.from_numpy(golden["sq_wrap_weight"]), in_scale, sx, zp)
    rel = float((y8 - y).norm() / y.norm())
    assert rel < 0.05, rel


def test_awq_stats(golden):
    w = torch.from_numpy(golden["awq_w"])
    assert np.array_equal(O.awq_weight_scale(w, 32).numpy(), golden["awq_wscale_g32"])
    assert np.array_equal(O.awq_weight_scale(w, -1).numpy(), golden["awq_wscale_pc"])
    x = torch.from_numpy(golden["awq_x"])
    assert np.array_equal(O.awq_act_scale([x[i : i + 1] for i in range(x.shape[0])]).numpy(), golden["awq_xscale"])


def test_oracle_awq_searches_match_the_reference_trace():
    """oracle.awq_search_scale_module / awq_search_clip_module (awq.py:264-361, 393-470) against what the UNMODIFIED
    reference's two grid searches saw on tiny_llama with every Linear self-absorbed (tests/golden/awq_trace_*.npz,
    made by make_golden_awq_trace.py): all 20 + 10 losses of every module, and the chosen grid points."""
    from tests.model_zoo import calib_ids, tiny_llama

    tr = np.load(os.path.join(ROOT, "tests", "golden", "awq_trace_tiny_llama_self.npz"))
    model = tiny_llama()
    model.config.use_cache = False
    ids = calib_ids()
    inputs = {}
    hooks = []
    for name, mod in model.named_modules():
        if isinstance(mod, torch.nn.Linear) and ".layers." in name:
            hooks.append(mod.register_forward_hook(lambda m, inp, out, n=name: inputs.setdefault(n, []).append(inp[0].detach())))
    with torch.no_grad():
        for x in ids:
            model(x)
    for h in hooks:
        h.remove()
    mods = dict(model.named_modules())
    clip_names = list(tr["clip_names"])
    assert len(tr["scale_names"]) == 14 and len(clip_names) == 14
    for i, name in enumerate(tr["scale_names"]):
        name = str(name)
        W = mods[name].weight.detach().clone()
        r = O.awq_search_scale_module(W, None, inputs[name], group_size=32, scheme="asym")
        np.testing.assert_allclose(np.array(r["history"]), tr["scale_hist"][i], rtol=2e-5, err_msg=name)
        assert r["best_index"] == int(tr["scale_best"][i]), name
        # apply_scale, self-absorption (awq.py:375-379): MulLinear(input_scale = 1/s), linear.weight /= input_scale
        isc = 1.0 / r["best_scales"]
        W2 = W / isc.view(1, -1)
        j = clip_names.index(name)
        c = O.awq_search_clip_module(W2, None, inputs[name], group_size=32, scheme="asym", input_scale=isc)
        np.testing.assert_allclose(np.array(c["history"]), tr["clip_hist"][j], rtol=2e-5, err_msg=name)
        assert c["best_index"] == int(tr["clip_best"][j]), name


NF4_CASES = {
    "nf4_g32": dict(dtype="nf4", group_size=32), "fp4_g32": dict(dtype="fp4", group_size=32),
    "fp4e2m1_g32": dict(dtype="fp4_e2m1", group_size=32), "nf4_tail": dict(dtype="nf4", group_size=32),
    "nf4_pc": dict(dtype="nf4", group_size=-1), "nf4_zero": dict(dtype="nf4", group_size=32),
    "nf4_q09": dict(dtype="nf4", group_size=32, quantile=0.9),
    "dq_int4": dict(dtype="int", bits=4, group_size=32, scheme="asym", double_quant=True),
    "dq_nf4": dict(dtype="nf4", group_size=32, double_quant=True),
}


@pytest.mark.parametrize("tag", list(NF4_CASES))
def test_oracle_nf4_fp4_double_quant_vs_reference_golden(tag):
    """quant_tensor's non-integer branches (quantize_4bit utility.py:112-149; double quantisation of the scales :378-436)
    restated in oracle/ against outputs of the unmodified reference (tests/golden/make_golden_nf4.py): bit-exact."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "nf4_golden.npz"))
    w = torch.from_numpy(g[f"{tag}_w"])
    kw = NF4_CASES[tag]
    assert np.array_equal(O.quant_tensor(w, **kw).numpy(), g[f"{tag}_qdq"])
    iw, sc, zp = O.quant_tensor(w, return_int=True, **kw)
    assert np.array_equal(iw.numpy(), g[f"{tag}_int"]) and np.array_equal(sc.numpy(), g[f"{tag}_scale"])
    assert (zp is None) == (f"{tag}_zp" not in g.files)
    if zp is not None:
        assert np.array_equal(zp.numpy(), g[f"{tag}_zp"])


# ---------------------------------------------------------------------------------------------------
# GPTQ layouts wider than the 128-column step (tests/golden/make_golden_gptq_wide.py)
# ---------------------------------------------------------------------------------------------------
GQW_CASES = {
    "gqw_sym_g256_bs128": dict(bits=4, sym=True, blocksize=128, groupsize=256),
    "gqw_asym_g128_bs256": dict(bits=4, sym=False, blocksize=256, groupsize=128),
    "gqw_sym_g64_bs256": dict(bits=4, sym=True, blocksize=256, groupsize=64),
    "gqw_asym_g256_bs384": dict(bits=4, sym=False, blocksize=384, groupsize=256),
}


@pytest.fixture(scope="module")
def golden_wide():
    return np.load(os.path.join(ROOT, "tests", "golden", "gptq_wide_golden.npz"))


@pytest.mark.parametrize("tag", list(GQW_CASES))
def test_gptq_layer_wide_groups_and_blocks(golden_wide, tag):
    """The oracle's fasterquant against the unmodified reference where find_params reads past the 128-column step."""
    kw = GQW_CASES[tag]
    W = torch.from_numpy(golden_wide[f"{tag}_W"])
    X = torch.from_numpy(golden_wide[f"{tag}_X"])
    H, n = torch.zeros(W.shape[1], W.shape[1]), 0
    for j in range(X.shape[0]):
        H, n = O.gptq_add_batch(H, n, X[j : j + 1])
    r = O.gptq_fasterquant(W, H, **kw)
    assert np.array_equal(r["scale"].numpy(), golden_wide[f"{tag}_scale"])
    assert np.array_equal(r["zero"].numpy(), golden_wide[f"{tag}_zero"])
    assert np.array_equal(r["Q"].numpy(), golden_wide[f"{tag}_Q"])
    ints = O.gptq_export_ints(r["Q"], r["scale"], r["zero"], kw["sym"], kw["groupsize"], r["perm"])
    assert np.array_equal(ints.numpy(), golden_wide[f"{tag}_ints"].astype(np.int32))


def test_smoothquant_non_default_cells_vs_reference():
    """quant_dequant_w_v1(scheme="asym") and the dynamic form of quant_dequant_x_v1 (min / max from the tensor itself): the
    restatements against outputs of the unmodified reference (tests/golden/make_golden_sq_cells.py)."""
    import os

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "sq_cells_golden.npz"))
    w = torch.from_numpy(g["w"])
    assert np.array_equal(O.sq_quant_w_asym(w).numpy(), g["qdq_w_asym"])
    x = torch.from_numpy(g["x"])
    assert np.array_equal(O.sq_quant_dequant_x(x, None, None).numpy(), g["qdq_x_dynamic"])


@pytest.mark.parametrize("dtype", ["nf4", "fp4", "fp4_e2m1"])
@pytest.mark.parametrize("wd", ["f32", "bf16"])
def test_oracle_quantize_4bit_with_given_scale_vs_reference_golden(dtype, wd):
    """quantize_4bit(tensor, scale=...) (utility.py:127-128) and the failure of double_quant_return_int (:383-405), restated in
    oracle/ against the unmodified reference (tests/golden/make_golden_q4scale.py): bit-exact / the same exception."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "q4scale_golden.npz"))
    td = torch.float32 if wd == "f32" else torch.bfloat16
    tag = f"q4s_{dtype}_{wd}"
    w = torch.from_numpy(g[f"{tag}_w"]).to(td)
    sc = torch.from_numpy(g[f"{tag}_scale"]).to(td)
    assert np.array_equal(O.quantize_4bit(w.clone(), dtype=dtype, scale=sc.clone()).float().numpy(), g[f"{tag}_qdq"])
    ints, s2, zp = O.quantize_4bit(w.clone(), dtype=dtype, return_int=True, scale=sc.clone())
    assert zp is None and torch.equal(s2, sc) and np.array_equal(ints.float().numpy(), g[f"{tag}_int"])
    kind, msg = str(g["dqri_error"]).split(": ", 1)
    assert kind == "ValueError"
    with pytest.raises(ValueError, match=msg.split("(")[0].strip()):
        O.quant_tensor(w.float(), bits=4, group_size=32, scheme="asym", return_int=True, double_quant=True, double_quant_return_int=True)


HYB_CASES = {
    "hyb_sym_g32": dict(bits=4, sym=True, blocksize=128, groupsize=32),
    "hyb_asym_g32": dict(bits=4, sym=False, blocksize=128, groupsize=32),
    "hyb_sym_g64_2blk": dict(bits=4, sym=True, blocksize=128, groupsize=64),
    "hyb_sym_g32_mse": dict(bits=4, sym=True, blocksize=128, groupsize=32, mse=True),
}


@pytest.mark.parametrize("tag", list(HYB_CASES))
def test_gptq_layer_hybrid_order(tag):
    """GPTQ.fasterquant(hybrid_order=True) (gptq.py:1203-1209, 1320-1328: columns rearranged by diag(H) inside their groups, groups by
    their largest diag(H); parameters returned in the groups' ORIGINAL order, so the export needs no g_idx) restated in oracle/ against
    the unmodified reference (tests/golden/make_golden_hybrid.py): bit-exact."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "gptq_hybrid_golden.npz"))
    kw = HYB_CASES[tag]
    W = torch.from_numpy(g[f"{tag}_W"])
    X = torch.from_numpy(g[f"{tag}_X"])
    H, n = torch.zeros(W.shape[1], W.shape[1]), 0
    for j in range(X.shape[0]):
        H, n = O.gptq_add_batch(H, n, X[j : j + 1])
    assert np.array_equal(H.numpy(), g[f"{tag}_H"])
    r = O.gptq_fasterquant(W, H, hybrid_order=True, **kw)
    assert not torch.equal(r["final_perm"], torch.arange(W.shape[1]))  # (the fixture's permutation is not the identity)
    assert np.array_equal(r["scale"].numpy(), g[f"{tag}_scale"])
    assert np.array_equal(r["zero"].numpy(), g[f"{tag}_zero"])
    assert np.array_equal(r["Q"].numpy(), g[f"{tag}_Q"])
    ints = O.gptq_export_ints(r["Q"], r["scale"], r["zero"], kw["sym"], kw["groupsize"], None)
    assert np.array_equal(ints.numpy(), g[f"{tag}_ints"].astype(np.int32))
