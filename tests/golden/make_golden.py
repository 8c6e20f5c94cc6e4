"""Generate tests/golden/*.npz from the UNMODIFIED reference (intel/neural-compressor v3.9 at /root/reference).

Run in the build container only (the reference does not travel to the GPU box):
    python tests/golden/make_golden.py
The two stub modules below stand in for `py-cpuinfo` and `prettytable`, which the reference imports at module
load (neural_compressor/common/utils/utility.py:26,28) and which are not installable here (no network).
numba is absent too, so packing takes the reference's torch/numpy fallback (modules.py:511-512), which the
reference's own test pins as bit-identical to the numba path (test/torch/algorithms/weight_only/test_woq_module.py).
"""

import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _install_stubs():
    cpuinfo = types.ModuleType("cpuinfo")
    cpuinfo.get_cpu_info = lambda: {"brand_raw": "stub", "flags": [], "count": os.cpu_count()}
    sys.modules["cpuinfo"] = cpuinfo
    pt = types.ModuleType("prettytable")

    class PrettyTable:
        def __init__(self, *a, **k):
            self.rows, self.field_names = [], []

        def add_row(self, r):
            self.rows.append(r)

        def get_string(self, *a, **k):
            return "\n".join(str(r) for r in self.rows)

        __str__ = get_string

    pt.PrettyTable = PrettyTable
    sys.modules["prettytable"] = pt
    # smooth_quant/utility.py imports intel_extension_for_pytorch at module level (:22); only its pure-torch functions
    # (cal_scale, quant_dequant_w_v1, quant_dequant_x_v1, SQLinearWrapper) are called here, so an empty module suffices
    import importlib.machinery

    ipex = types.ModuleType("intel_extension_for_pytorch")
    ipex.__version__ = "2.1.100+stub"
    ipex.__spec__ = importlib.machinery.ModuleSpec("intel_extension_for_pytorch", loader=None)
    sys.modules["intel_extension_for_pytorch"] = ipex


def main():
    sys.dont_write_bytecode = True
    _install_stubs()
    sys.path.insert(0, REF)
    import torch
    from neural_compressor.torch.algorithms.weight_only.gptq import GPTQ
    from neural_compressor.torch.algorithms.weight_only.modules import INCWeightOnlyLinear
    from neural_compressor.torch.algorithms.weight_only.utility import quant_tensor, quant_weight_w_scale, search_clip
    from neural_compressor.torch.algorithms.weight_only import awq as ref_awq

    out = {}

    # ---- 1. pack / unpack / recover known-answer vectors -------------------------------
    ints = torch.tensor([list(range(-8, 8)), list(range(7, -9, -1))] * 4, dtype=torch.int32)  # [8,16]
    scales = torch.full((8, 2), 0.5)
    m = INCWeightOnlyLinear(16, 8, bits=4, group_size=8, device="cpu")
    m.pack(ints.clone(), scales.clone(), None, None)
    out["kat_ints"] = ints.numpy()
    out["kat_qweight"] = m.qweight.numpy()
    out["kat_qzeros"] = m.qzeros.numpy()
    out["kat_scales"] = m.scales.numpy()
    out["kat_recover"] = m.recover().numpy()

    # ---- 2. module pack/unpack/recover on random data, sym + asym, bits 4/8, with a ragged N ------------
    g = torch.Generator().manual_seed(1234)
    for tag, (N, K, gs, bits, scheme) in {
        "m4sym": (24, 64, 32, 4, "sym"),
        "m4asym": (20, 96, 32, 4, "asym"),
        "m8sym": (16, 64, -1, 8, "sym"),
        "m8asym": (16, 64, 32, 8, "asym"),
    }.items():
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

    # ---- 3. generic packer: every (bits, container) pair of the reference's 12-way test -----------------
    raw = torch.randint(-128, 128, (6, 40), generator=g, dtype=torch.int32)
    out["rows_raw"] = raw.numpy()
    for bits in (2, 4, 8):
        for cdt, cb in ((torch.int8, 8), (torch.int16, 16), (torch.int32, 32), (torch.int64, 64)):
            mod = INCWeightOnlyLinear(40, 6, bits=bits, group_size=-1, compression_dtype=cdt, use_optimum_format=False, device="cpu")
            packed = mod.pack_tensor(raw.clone())
            out[f"rows_b{bits}_c{cb}"] = packed.numpy()
            out[f"rows_b{bits}_c{cb}_unpack_signed"] = mod.unpack_tensor(packed.clone()).numpy()

    # ---- 4. quant_tensor: sym / asym / full_range / quantile / tail group / bf16 --------------------------
    w = torch.randn(12, 300, generator=g) * 0.05
    out["qt_w"] = w.numpy()
    cases = {
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
    for tag, kw in cases.items():
        out[f"{tag}_qdq"] = quant_tensor(w.clone(), **kw).numpy()
        iw, sc, zp = quant_tensor(w.clone(), return_int=True, **kw)
        out[f"{tag}_int"] = iw.numpy().copy()
        out[f"{tag}_scale"] = sc.numpy().copy()
        if zp is not None:
            out[f"{tag}_zp"] = zp.numpy().copy()
    wb = (torch.randn(8, 256, generator=g) * 0.05).to(torch.bfloat16)
    out["qtbf16_w"] = wb.float().numpy()
    for tag, kw in {"qtbf16_sym": dict(bits=4, group_size=128, scheme="sym"), "qtbf16_asym": dict(bits=4, group_size=128, scheme="asym")}.items():
        out[f"{tag}_qdq"] = quant_tensor(wb.clone(), **kw).float().numpy()
        iw, sc, zp = quant_tensor(wb.clone(), return_int=True, **kw)
        out[f"{tag}_int"] = iw.float().numpy().copy()
        out[f"{tag}_scale"] = sc.float().numpy().copy()
        if zp is not None:
            out[f"{tag}_zp"] = zp.float().numpy().copy()

    class _M:  # search_clip wants a module with .weight.data
        pass

    lin = torch.nn.Linear(300, 12, bias=False)
    lin.weight.data.copy_(w)
    out["clip_sym4_g32"] = np.float64(search_clip(lin, bits=4, group_size=32, scheme="sym"))
    out["clip_asym4_g128"] = np.float64(search_clip(lin, bits=4, group_size=128, scheme="asym"))

    # ---- 5. GPTQ: add_batch + fasterquant on small layers ---------------------------------------------------
    def run_gptq(tag, N, K, nb, seq, cfg, blocksize, groupsize, act_order=False, static_groups=False):
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
            x[..., ::17] *= 8.0  # a few outlier channels
            xs.append(x)
            gq.add_batch(x, None)
        H = gq.H.clone()
        scale, _, zero, Q = gq.fasterquant(W.clone(), blocksize=blocksize, percdamp=0.01, groupsize=groupsize, act_order=act_order,
                                           static_groups=static_groups)
        out[f"{tag}_W"] = W.numpy()
        out[f"{tag}_X"] = torch.cat(xs, 0).numpy()
        out[f"{tag}_H"] = H.numpy()
        out[f"{tag}_scale"] = scale.numpy()
        out[f"{tag}_zero"] = zero.numpy()
        out[f"{tag}_Q"] = Q.numpy()
        if act_order:
            out[f"{tag}_perm"] = gq.perm.numpy()
        if static_groups:
            return  # the reference returns only the last group's scale here (gptq.py:1341-1345): Q is the checked quantity
        ints = quant_weight_w_scale(
            (Q[:, gq.perm] if act_order else Q).clone(), scale, None, None if full["sym"] else zero, groupsize, dtype="int"
        )
        if act_order:
            ints = ints[:, torch.argsort(gq.perm)]
        out[f"{tag}_ints"] = ints.numpy()

    run_gptq("gq_sym_g32", 16, 128, 4, 48, dict(bits=4, sym=True), 128, 32)
    run_gptq("gq_asym_g32", 16, 128, 4, 48, dict(bits=4, sym=False), 128, 32)
    run_gptq("gq_sym_pc", 12, 96, 3, 64, dict(bits=4, sym=True), 128, -1)
    run_gptq("gq_sym_g128_2blk", 24, 256, 4, 80, dict(bits=4, sym=True), 128, 128)
    run_gptq("gq_sym_g32_blk2048", 16, 256, 4, 80, dict(bits=4, sym=True), 2048, 32)
    run_gptq("gq_sym_act", 16, 128, 4, 48, dict(bits=4, sym=True), 128, 32, act_order=True)
    run_gptq("gq_sym8_g64", 8, 128, 3, 64, dict(bits=8, sym=True), 128, 64)
    # use_mse_search: Quantizer.find_params shrink grid (gptq.py:1567-1584; reference test test_gptq.py:137)
    run_gptq("gq_sym_g32_mse", 16, 128, 4, 48, dict(bits=4, sym=True, mse=True), 128, 32)
    run_gptq("gq_asym_g64_mse", 12, 128, 4, 48, dict(bits=4, sym=False, mse=True), 128, 64)

    # static_groups (no reference test; Q of fasterquant is the well-defined output)
    run_gptq("gq_sym_static", 16, 128, 4, 48, dict(bits=4, sym=True), 128, 32, static_groups=True)
    run_gptq("gq_asym_act_static", 16, 256, 4, 80, dict(bits=4, sym=False), 128, 32, act_order=True, static_groups=True)

    # ---- 6. AWQ statistics ---------------------------------------------------------------------------------------
    w = torch.randn(24, 128, generator=g)
    out["awq_w"] = w.numpy()
    out["awq_wscale_g32"] = ref_awq._get_weight_scale(w.clone(), 32).numpy()
    out["awq_wscale_pc"] = ref_awq._get_weight_scale(w.clone(), -1).numpy()
    xs = [torch.randn(1, 20, 128, generator=g) for _ in range(3)]
    out["awq_x"] = torch.cat(xs, 0).numpy()
    out["awq_xscale"] = ref_awq._get_act_scale(xs).numpy()

    # ---- 6b. AutoAWQ checkpoint repack (utility.py:1426-1459) ---------------------------------------------------
    from neural_compressor.torch.algorithms.weight_only.utility import repack_awq_to_optimum_format

    Ka, Na, gsa = 256, 128, 128  # unpack_awq reshapes through [-1, group_size, K]: N must be a multiple of group_size
    aq = torch.randint(-(2**31), 2**31 - 1, (Ka, Na // 8), generator=g, dtype=torch.int64).to(torch.int32)
    az = torch.randint(-(2**31), 2**31 - 1, (Ka // gsa, Na // 8), generator=g, dtype=torch.int64).to(torch.int32)
    asc = (torch.rand(Ka // gsa, Na, generator=g) * 0.02 + 0.004).half()
    rq, rz, rs = repack_awq_to_optimum_format(aq.clone(), az.clone(), asc.clone(), 4, gsa)
    out["awqpack_qweight_in"], out["awqpack_qzeros_in"], out["awqpack_scales"] = aq.numpy(), az.numpy(), asc.numpy()
    out["awqpack_qweight"], out["awqpack_qzeros"] = rq.numpy(), rz.numpy()
    assert torch.equal(rs, asc)

    # ---- 6c. SmoothQuant (BASELINE config #4): the in-tree pure-torch functions --------------------------------
    from neural_compressor.torch.algorithms.smooth_quant import utility as sq

    Ks, N1, N2 = 96, 24, 40
    amax_x = torch.rand(Ks, generator=g) * 6.0
    amax_x[5] = 0.0  # a dead channel: scale must come out as 1
    w1 = torch.randn(N1, Ks, generator=g) * 0.1
    w2 = torch.randn(N2, Ks, generator=g) * 0.1
    w2[:, 7] = 0.0  # a column below weight_max_lb
    out["sq_amax_x"], out["sq_w1"], out["sq_w2"] = amax_x.numpy(), w1.numpy(), w2.numpy()
    for a in (0.5, 0.8):
        out[f"sq_scale_a{int(a * 10)}"] = sq.cal_scale(amax_x.clone(), [w1.clone(), w2.clone()], a).numpy()
    lin = torch.nn.Linear(Ks, N1, bias=False)
    lin.weight.data.copy_(w1)
    out["sq_qdq_w_sym"] = sq.quant_dequant_w_v1(lin, num_bits=8, scheme="sym").detach().numpy()
    xs = torch.randn(20, Ks, generator=g) * 2.0 + 0.3
    out["sq_x"] = xs.numpy()
    mnx, mxx = xs.min(dim=0)[0], xs.max(dim=0)[0]
    out["sq_qdq_x"] = sq.quant_dequant_x_v1(xs.clone(), mnx, mxx, num_bits=8).numpy()
    in_scale = 1.0 / sq.cal_scale(torch.max(mnx.abs(), mxx.abs()), [w1.clone()], 0.5)
    wrap = sq.SQLinearWrapper(lin, in_scale.clone(), [mnx.clone(), mxx.clone()], alpha=0.5)
    out["sq_wrap_in_scale"] = in_scale.numpy()
    out["sq_wrap_scale"] = wrap.scale.numpy()
    out["sq_wrap_zp"] = wrap.zero_point.numpy()
    out["sq_wrap_weight"] = wrap.sq_linear.weight.detach().numpy()  # W / input_scale (:2646-2653)
    with torch.no_grad():
        out["sq_wrap_out"] = wrap(xs).numpy()

    np.savez_compressed(os.path.join(HERE, "woq_golden.npz"), **out)
    print(f"wrote {len(out)} arrays to tests/golden/woq_golden.npz")

    # ---- 7. config #1: OPT-125M-shaped RTN INT8 via the reference's CPU path -> per-layer digests ----------------
    make_rtn_model_golden(torch)


def opt125m_like(torch, layers=12, hidden=768, ffn=3072, vocab=512, seed=0):
    """OPT-125M-shaped decoder stack built from plain nn modules (12 x {q,k,v,out_proj,fc1,fc2}), fp32, seeded.
    (transformers' OPT class is avoided so that the GPU box regenerates bit-identical weights from the seed
    without depending on HF init order.)"""
    g = torch.Generator().manual_seed(seed)

    class Block(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.q_proj = torch.nn.Linear(hidden, hidden)
            self.k_proj = torch.nn.Linear(hidden, hidden)
            self.v_proj = torch.nn.Linear(hidden, hidden)
            self.out_proj = torch.nn.Linear(hidden, hidden)
            self.fc1 = torch.nn.Linear(hidden, ffn)
            self.fc2 = torch.nn.Linear(ffn, hidden)

        def forward(self, x):
            a = torch.tanh(self.q_proj(x)) * torch.sigmoid(self.k_proj(x)) + self.v_proj(x)
            x = x + self.out_proj(a)
            return x + self.fc2(torch.relu(self.fc1(x)))

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.embed = torch.nn.Embedding(vocab, hidden)
            self.layers = torch.nn.ModuleList([Block() for _ in range(layers)])
            self.lm_head = torch.nn.Linear(hidden, vocab, bias=False)

        def forward(self, ids):
            x = self.embed(ids)
            for layer in self.layers:
                x = layer(x)
            return self.lm_head(x)

    model = Model()
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    return model


def digest(t):
    """Order-sensitive 64-bit digest of an integer tensor (numpy, no torch dependency on the reading side)."""
    a = np.ascontiguousarray(t).reshape(-1).view(np.uint8).astype(np.uint64)
    idx = (np.arange(a.size, dtype=np.uint64) % np.uint64(65521)) + np.uint64(1)
    return np.uint64((a * idx).sum() % np.uint64(2**61 - 1))


def make_rtn_model_golden(torch):
    from neural_compressor.torch.quantization import RTNConfig, quantize

    model = opt125m_like(torch)
    q = quantize(model, RTNConfig(bits=8, group_size=-1, use_layer_wise=False))
    out = {}
    for name, mod in q.named_modules():
        if type(mod).__name__ == "INCWeightOnlyLinear":
            out[f"{name}.qweight"] = digest(mod.qweight.numpy())
            out[f"{name}.qzeros"] = digest(mod.qzeros.numpy())
            out[f"{name}.scales"] = digest(mod.scales.numpy().view(np.uint16))
    torch.manual_seed(0)
    ids = torch.randint(0, 512, (2, 16))
    with torch.no_grad():
        y = q(ids)
    out["logits"] = y.float().numpy()
    out["n_modules"] = np.int64(sum(1 for _, m in q.named_modules() if type(m).__name__ == "INCWeightOnlyLinear"))
    np.savez_compressed(os.path.join(HERE, "rtn_opt125m_like.npz"), **out)
    print(f"wrote {len(out)} entries to tests/golden/rtn_opt125m_like.npz")


if __name__ == "__main__":
    main()
