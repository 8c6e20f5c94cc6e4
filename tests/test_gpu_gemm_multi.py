"""-m gpu: inc_woq_gemm_multi -- the decode forward of several packed modules that share x (q / k / v; gate / up) in ONE launch.

Reference semantics: n independent INCWeightOnlyLinear.forward calls on the same activation (modules.py:594-610).  Checked here:
  * against the oracle's forward per module (O.woq_linear, <= 1e-3 like every fused-GEMM test);
  * bit-identical to inc_woq_gemm on the N-CONCATENATED module (what a fused qkv module would compute): every 64-column strip runs
    the same streaming body with the same K-slices, so the batched launch is the single launch of the wide module, bit for bit;
  * against the n single calls: equal up to fp32 summation order only (a single 4096 x 4096 call at M <= 4 takes the no-split
    kernel, which sums K in another order) -- the distance is printed and gated at float noise;
  * batches the library declines fall back to the single calls and return exactly their results.
"""

import numpy as np
import pytest
import torch

from oracle import woq_oracle as O

pytestmark = pytest.mark.gpu

GS = 128


def _packed(hip, N, K, seed, bias=False, sym=True, dtype=torch.bfloat16):
    from neural_compressor_amd.torch.algorithms.weight_only.modules import MI355XWeightOnlyLinear
    from neural_compressor_amd.torch.algorithms.weight_only.utility import quant_tensor

    g = torch.Generator().manual_seed(seed)
    w = (torch.randn(N, K, generator=g) * 0.02).to(hip)
    iw, sc, zp = quant_tensor(w, bits=4, group_size=GS, scheme="sym" if sym else "asym", return_int=True)
    m = MI355XWeightOnlyLinear(K, N, bits=4, group_size=GS, zp=not sym, bias=bias, device=hip)
    b = (torch.randn(N, generator=g) * 0.1).to(hip) if bias else None
    m.pack(iw, sc, zp if not sym else None, b)
    if not bias:
        m.bias = None
    return m


def _concat(hip, mods):
    from neural_compressor_amd.torch.algorithms.weight_only.modules import MI355XWeightOnlyLinear

    K, N = mods[0].in_features, sum(m.out_features for m in mods)
    c = MI355XWeightOnlyLinear(K, N, bits=4, group_size=GS, zp=True, bias=mods[0].bias is not None, device=hip)
    c.qweight = torch.cat([m.qweight for m in mods], dim=1).contiguous()
    c.scales = torch.cat([m.scales for m in mods], dim=1).contiguous()
    c.qzeros = torch.cat([m.qzeros for m in mods], dim=1).contiguous()
    c.bias = None if mods[0].bias is None else torch.cat([m.bias for m in mods]).contiguous()
    return c


@pytest.mark.parametrize("M", [1, 4, 16, 33, 64])
@pytest.mark.parametrize("group", ["qkv 3 x 4096x4096", "gate+up 2 x 11008x4096", "down-like 2 x 4096x11008"])
def test_gemm_multi_llama_groups_vs_oracle_concat_and_singles(hip, group, M):
    from neural_compressor_amd import ops
    from neural_compressor_amd.torch.algorithms.weight_only.modules import woq_linear_group

    n, N, K = {"qkv 3 x 4096x4096": (3, 4096, 4096), "gate+up 2 x 11008x4096": (2, 11008, 4096), "down-like 2 x 4096x11008": (2, 4096, 11008)}[group]
    mods = [_packed(hip, N, K, 10 * i + K % 97) for i in range(n)]
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(hip)
    ys = woq_linear_group(x, mods)
    call = ops.WoqGemmGroupCall([(m.qweight, m.scales, m.qzeros, m.bias, m.out_features) for m in mods], K, GS, 4, torch.bfloat16)
    ys2 = call(x)
    if M > 32:
        # more than 32 rows on these sizes: the library declines (inc_woq_gemm's strip kernel is the faster form) -> the single calls
        assert ys2 is None
        assert all(torch.equal(y, m(x)) for y, m in zip(ys, mods))
        return
    # the batched entry point really ran (rc 0), not the fallback
    assert ys2 is not None, "the library declined an eligible batch"
    assert all(torch.equal(a, b) for a, b in zip(ys, ys2))
    # (1) bit-identical to the single launch of the N-concatenated module
    yc = _concat(hip, mods)(x)
    off = 0
    for y in ys:
        assert y.shape == (M, N) and y.dtype == torch.bfloat16
        assert torch.equal(y, yc[:, off:off + N]), "batched launch != single launch of the concatenated module"
        off += N
    # (2) oracle per module, (3) distance to the single calls
    worst_o = worst_s = 0.0
    for m, y in zip(mods, ys):
        qw, sc, qz = m.qweight.cpu().numpy(), m.scales.cpu().numpy(), m.qzeros.cpu().numpy()
        rows = torch.arange(M)[: 8]
        ref = O.woq_linear(x[rows].cpu(), qw, sc, qz, None, N, K, 4, GS, compute_dtype=torch.bfloat16)
        e = float((y[rows].float().cpu() - ref.to(torch.bfloat16).float()).norm() / ref.norm())
        worst_o = max(worst_o, e)
        ys_single = m(x)
        worst_s = max(worst_s, float((y.float() - ys_single.float()).norm() / ys_single.float().norm()))
    print(f"\n[gemm multi {group} M={M}] vs oracle (bf16-rounded) {worst_o:.2e}; vs the single calls {worst_s:.2e}; == concatenated single launch")
    assert worst_o <= 1e-3
    assert worst_s <= 2e-3  # bf16 output: one ulp = 3.9e-3 relative on an element, a handful of elements move by one


def test_gemm_multi_bias_asym_fp16_and_ragged_columns(hip):
    from neural_compressor_amd.torch.algorithms.weight_only.modules import woq_linear_group

    K = 1024
    mods = [_packed(hip, N, K, 3 + i, bias=True, sym=False) for i, N in enumerate((256, 1000, 64, 4100))]
    from neural_compressor_amd import ops

    for dtype, rows in ((torch.float16, 5), (torch.bfloat16, 5), (torch.bfloat16, 20), (torch.float16, 40)):  # 1 / 2 / 4 row blocks of 16
        x = torch.randn(rows, K, generator=torch.Generator().manual_seed(2)).to(dtype).to(hip)
        call = ops.WoqGemmGroupCall([(m.qweight, m.scales, m.qzeros, m.bias, m.out_features) for m in mods], K, GS, 4, dtype)
        assert call(x) is not None, "this batch is eligible for the one-launch form"
        ys = woq_linear_group(x, mods)
        for m, y in zip(mods, ys):
            qw, sc, qz = m.qweight.cpu().numpy(), m.scales.cpu().numpy(), m.qzeros.cpu().numpy()
            ref = O.woq_linear(x.cpu(), qw, sc, qz, m.bias.float().cpu(), m.out_features, K, 4, GS, compute_dtype=dtype)
            e = float((y.float().cpu() - ref.to(dtype).float()).norm() / ref.norm())
            assert y.shape == (rows, m.out_features) and e <= 2e-3, (m.out_features, dtype, rows, e)
    x = x[:5].contiguous()
    # leading dimensions are kept
    y3 = woq_linear_group(x.view(1, 5, K), mods)
    assert all(a.shape == (1, 5, m.out_features) for a, m in zip(y3, mods))


def test_gemm_multi_falls_back_to_single_calls_when_not_eligible(hip):
    from neural_compressor_amd import ops
    from neural_compressor_amd.torch.algorithms.weight_only.modules import woq_linear_group

    K = 512
    mods = [_packed(hip, 256, K, 7), _packed(hip, 128, K, 8)]
    x_big = torch.randn(65, K).to(torch.bfloat16).to(hip)      # prefill-sized: not the decode launch
    ys = woq_linear_group(x_big, mods)
    assert all(torch.equal(y, m(x_big)) for y, m in zip(ys, mods))
    x32 = torch.randn(3, K).to(hip)                              # fp32 activations: the module's own cast-and-return path
    ys = woq_linear_group(x32, mods)
    assert all(y.dtype == torch.float32 and torch.equal(y, m(x32)) for y, m in zip(ys, mods))
    # the library itself declines (nothing launched) what it cannot batch: one module, M > 64
    call = ops.WoqGemmGroupCall([(m.qweight, m.scales, m.qzeros, None, m.out_features) for m in mods[:1]], K, GS, 4, torch.bfloat16)
    assert call(x_big[:4].contiguous()) is None
    call = ops.WoqGemmGroupCall([(m.qweight, m.scales, m.qzeros, None, m.out_features) for m in mods], K, GS, 4, torch.bfloat16)
    assert call(x_big) is None
    # deterministic: two launches, same bits
    x = torch.randn(2, K).to(torch.bfloat16).to(hip)
    a, b = call(x), call(x)
    assert all(torch.equal(p, q) for p, q in zip(a, b))


def test_gemm_multi_cache_is_not_module_state(hip):
    """The prepared group call is a cache on the group's first module: the state dict, copies and pickles of the module do not carry it,
    and replacing a buffer of any module of the group rebuilds it."""
    import copy
    import pickle

    from neural_compressor_amd.torch.algorithms.weight_only.modules import woq_linear_group

    K = 512
    mods = [_packed(hip, 256, K, 21), _packed(hip, 128, K, 22)]
    keys = set(mods[0].state_dict().keys())
    x = torch.randn(2, K).to(torch.bfloat16).to(hip)
    y0 = woq_linear_group(x, mods)
    assert "_group_calls" in mods[0].__dict__ and set(mods[0].state_dict().keys()) == keys
    clone = copy.deepcopy(mods[0])
    assert not any(clone.__dict__.get("_group_calls", {}).values())
    again = pickle.loads(pickle.dumps(mods[0]))
    assert torch.equal(again.qweight.cpu(), mods[0].qweight.cpu())
    assert all(torch.equal(a, b) for a, b in zip(woq_linear_group(x, [clone, mods[1]]), y0))
    # a replaced buffer (re-packing, load_state_dict): the cached addresses must not be used again
    other = _packed(hip, 128, K, 23)
    mods[1].qweight = other.qweight.clone()
    mods[1].scales = other.scales.clone()
    mods[1].qzeros = other.qzeros.clone()
    y1 = woq_linear_group(x, mods)
    assert torch.equal(y1[0], y0[0]) and torch.equal(y1[1], other(x)) and not torch.equal(y1[1], y0[1])


def _packed8(hip, N, K, gs, seed, sym=True, bias=False):
    from neural_compressor_amd.torch.algorithms.weight_only.modules import MI355XWeightOnlyLinear
    from neural_compressor_amd.torch.algorithms.weight_only.utility import quant_tensor

    g = torch.Generator().manual_seed(seed)
    w = (torch.randn(N, K, generator=g) * 0.02).to(hip)
    iw, sc, zp = quant_tensor(w, bits=8, group_size=gs, scheme="sym" if sym else "asym", return_int=True)
    m = MI355XWeightOnlyLinear(K, N, bits=8, group_size=gs, zp=not sym, bias=bias, device=hip)
    b = (torch.randn(N, generator=g) * 0.1).to(hip) if bias else None
    m.pack(iw, sc, zp if not sym else None, b)
    if not bias:
        m.bias = None
    return m


@pytest.mark.parametrize("N,K,gs,sym", [(4096, 4096, -1, True), (11008, 4096, 128, True), (1000, 1024, 32, False), (4096, 11008, -1, False)])
@pytest.mark.parametrize("M", [1, 5, 16, 17, 33, 64])
def test_int8_decode_streaming_kernel_vs_oracle(hip, N, K, gs, sym, M):
    """Weight-only INT8 (BASELINE config #1's format) at decode sizes: inc_woq_gemm's 8-bit streaming form (two packed rows per lane and
    step, int8 wrap of q - z, one rounding to the 16-bit type) against the oracle's forward, against HIP recover() + fp32 matmul, and
    bit-reproducible; per-channel, g128, g32 asym with ragged N, K = 11008.  M = 17 / 33 / 64 are the 2- and 4-row-block forms (and, at
    11008 x 4096 above 32 rows, the 8-bit tile kernel the route falls back to)."""
    m = _packed8(hip, N, K, gs, 31 + N % 7, sym=sym, bias=not sym)
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(hip)
    y = m(x)
    assert y.shape == (M, N) and y.dtype == torch.bfloat16
    w = m.recover(dtype=torch.bfloat16).float()
    ref = x.float() @ w.t() + (m.bias.float() if m.bias is not None else 0.0)
    e_dev = float((y.float() - ref.to(torch.bfloat16).float()).norm() / ref.norm())
    qw, sc, qz = m.qweight.cpu().numpy(), m.scales.cpu().numpy(), m.qzeros.cpu().numpy()
    gsz = K if gs == -1 else gs
    rows = torch.arange(M)[:4]
    oref = O.woq_linear(x[rows].cpu(), qw, sc, qz, None if m.bias is None else m.bias.float().cpu(), N, K, 8, gsz, compute_dtype=torch.bfloat16)
    e_or = float((y[rows].float().cpu() - oref.to(torch.bfloat16).float()).norm() / oref.norm())
    print(f"\n[int8 decode {N}x{K} gs={gs} {'sym' if sym else 'asym'} M={M}] vs HIP recover + fp32 matmul {e_dev:.2e}, vs oracle {e_or:.2e}")
    tol = 1e-3 if m.bias is None else 2e-3  # (the module's fp16 bias is added as a bf16 value: its own rounding on top of the output's)
    assert e_dev <= tol and e_or <= tol
    assert torch.equal(m(x), y)
    if m.bias is None:
        assert torch.equal(m(x * 2), y * 2)  # linear in x: power-of-two scaling is exact


def test_int8_group_is_the_concatenated_single_launch(hip):
    from neural_compressor_amd import ops
    from neural_compressor_amd.torch.algorithms.weight_only.modules import MI355XWeightOnlyLinear, woq_linear_group

    K = 4096
    mods = [_packed8(hip, 4096, K, 128, 41 + i) for i in range(3)]
    for M in (1, 16):
        x = torch.randn(M, K, generator=torch.Generator().manual_seed(M)).to(torch.bfloat16).to(hip)
        call = ops.WoqGemmGroupCall([(m.qweight, m.scales, m.qzeros, m.bias, m.out_features) for m in mods], K, 128, 8, torch.bfloat16)
        ys = call(x)
        assert ys is not None, "an INT8 decode group is eligible for the one-launch form"
        assert all(torch.equal(a, b) for a, b in zip(ys, woq_linear_group(x, mods)))
        c = MI355XWeightOnlyLinear(K, 3 * 4096, bits=8, group_size=128, zp=True, device=hip)
        c.qweight = torch.cat([m.qweight for m in mods], dim=1).contiguous()
        c.scales = torch.cat([m.scales for m in mods], dim=1).contiguous()
        c.qzeros = torch.cat([m.qzeros for m in mods], dim=1).contiguous()
        c.bias = None
        yc = c(x)
        for i, y in enumerate(ys):
            assert torch.equal(y, yc[:, 4096 * i:4096 * (i + 1)])
            assert torch.equal(y, mods[i](x))  # 8-bit: the single call takes the same kernel with the same slices
    assert ops.WoqGemmGroupCall([(m.qweight, m.scales, m.qzeros, m.bias, m.out_features) for m in mods], K, 128, 8, torch.bfloat16)(
        torch.randn(17, K).to(torch.bfloat16).to(hip)) is None  # more than 16 rows: declined (the tile kernels are the 8-bit route there)


def test_group_size_that_is_no_power_of_two_takes_the_dense_route_above_16_rows(hip):
    """group_size 96 (neither a power of two nor the whole row): decode-sized batches stay fused (the M <= 16 split-K kernel), larger ones
    go through HIP recover() + the library GEMM -- the general tile kernel they used to take is 3-7 x slower (scripts/route_sweep.py).
    Both routes agree with the dense reference."""
    from neural_compressor_amd.torch.algorithms.weight_only.modules import MI355XWeightOnlyLinear
    from neural_compressor_amd.torch.algorithms.weight_only.utility import quant_tensor

    N, K, gs = 512, 960, 96
    w = (torch.randn(N, K, generator=torch.Generator().manual_seed(3)) * 0.02).to(hip)
    iw, sc, zp = quant_tensor(w, bits=4, group_size=gs, scheme="asym", return_int=True)
    m = MI355XWeightOnlyLinear(K, N, bits=4, group_size=gs, zp=True, device=hip)
    m.pack(iw, sc, zp, None)
    m.bias = None
    wd = m.recover(dtype=torch.bfloat16).float()
    for M in (3, 16, 17, 200):
        x = torch.randn(M, K, generator=torch.Generator().manual_seed(M)).to(torch.bfloat16).to(hip)
        y = m(x)
        ref = (x.float() @ wd.t()).to(torch.bfloat16).float()
        assert m._plan == "fused" and m._fused_max_m == 16
        assert float((y.float() - ref).norm() / ref.norm()) <= 2e-3, M
        assert torch.equal(m(x), y)
