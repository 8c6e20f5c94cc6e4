"""-m gpu: BASELINE config #5 shapes -- Llama-2-70B layers (hidden 8192, ffn 28672): Hessians of 256 MiB / 3.06 GiB, a
224-block inverse-Cholesky factorisation, a 224-block column loop, and the fused GEMM on [28672, 8192] / [8192, 28672].
Nothing here is new arithmetic; the point is that index math, workspaces and the Python-driven blocked factorisation survive
the sizes SURVEY.md 8(e) lists.  Referees: fp64 on the device (the CPU oracle's LAPACK trio at K = 28672 would take minutes)
plus the oracle's column loop on sampled rows (row-independent given Hinv)."""

import math

import numpy as np
import pytest
import torch

from oracle import woq_oracle as O

pytestmark = pytest.mark.gpu
GS = 128
_C = {}


def rel_fro(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def _x(hip, K, T):
    key = ("x", K, T)
    if key not in _C:
        g = torch.Generator(device="cpu").manual_seed(K)
        x = torch.randn(T, K, generator=g)
        x[:, ::113] *= 15.0
        _C[key] = x.to(torch.bfloat16).to(hip)
    return _C[key]


def _hessian(hip, K, T):
    """H = (2/n) X^T X from the HIP kernel in 16384-token launches (n = launches x 8 'batches'), mirrored + damped."""
    from neural_compressor_amd import ops

    key = ("H", K, T)
    if key not in _C:
        x = _x(hip, K, T)
        H = torch.zeros(K, K, device=hip)
        n = 0
        for t0 in range(0, T, 16384):
            xb = x[t0:t0 + 16384]
            b = 8
            ops.gptq_hessian_accum(H, xb, n / (n + b), 2.0 / (n + b))
            n += b
        _C[key] = (H, n)
    return _C[key]


@pytest.mark.parametrize("K,T", [(8192, 16384), (28672, 32768)])
def test_hessian_70b_shapes_vs_fp64(hip, K, T):
    H, n = _hessian(hip, K, T)
    x = _x(hip, K, T)
    H64 = torch.zeros(K, K, dtype=torch.float64, device=hip)
    for t0 in range(0, T, 8192):  # fp64 referee in slabs (exact products, fp64 accumulation)
        xs = x[t0:t0 + 8192].double()
        H64.addmm_(xs.t(), xs)
        del xs
    H64.mul_(2.0 / n)
    worst, where = 0.0, None
    tile = 2048  # coarse tiles here (the 256-tile sweep is done at 7B size); every entry of the upper triangle is covered
    nt = -(-K // tile)
    num = den = 0.0
    for i in range(nt):
        for j in range(i, nt):
            a = H[i * tile:(i + 1) * tile, j * tile:(j + 1) * tile].double()
            b = H64[i * tile:(i + 1) * tile, j * tile:(j + 1) * tile]
            if i == j:
                a, b = torch.triu(a), torch.triu(b)
            d2, b2 = float((a - b).pow(2).sum()), float(b.pow(2).sum())
            num, den = num + d2, den + b2
            r = math.sqrt(d2 / max(b2, 1e-300))
            if r > worst:
                worst, where = r, (i, j)
    total = math.sqrt(num / den)
    print(f"\n[hessian K={K} T={T}] upper triangle vs fp64: rel-Frobenius {total:.2e}, worst 2048-tile {worst:.2e} at {where}")
    assert total <= 2e-6 and worst <= 6e-6
    del H64


def _factor(hip, K, T):
    from neural_compressor_amd import ops
    from neural_compressor_amd.torch.algorithms.weight_only.gptq import inverse_cholesky_upper

    key = ("U", K, T)
    if key not in _C:
        H, _ = _hessian(hip, K, T)
        Hd = H.clone()
        dead = ops.gptq_hessian_finalize(Hd, 0.01)
        U = inverse_cholesky_upper(Hd)
        _C[key] = (Hd, U, dead)
    return _C[key]


def test_inverse_cholesky_upper_8192_vs_fp64_trio(hip):
    Hd, U, _ = _factor(hip, 8192, 16384)
    H64 = Hd.double()
    U64 = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(H64)), upper=True)
    e = rel_fro(U, U64)
    d = float(((torch.diagonal(U).double() - torch.diagonal(U64)) / torch.diagonal(U64)).abs().max())
    print(f"\n[inverse_cholesky_upper K=8192] vs fp64 trio: rel-Frobenius {e:.2e}, worst relative diagonal error {d:.2e}")
    assert torch.equal(torch.triu(U), U)
    assert e <= 1e-5 and d <= 1e-5


def test_inverse_cholesky_upper_28672_residual(hip):
    """224 diagonal blocks, 3 GiB operands: U must satisfy  U (H + damp I) U^T = I  (U^T U = H^-1), checked in fp64."""
    K = 28672
    Hd, U, dead = _factor(hip, K, 32768)
    assert int(dead.sum()) == 0 and torch.equal(torch.triu(U), U) and bool(torch.isfinite(U).all())
    U64 = U.double()
    R = U64 @ Hd.double()
    R = R @ U64.t()
    del U64
    R.diagonal().sub_(1.0)
    res = float(R.norm()) / math.sqrt(K)
    worst = float(R.abs().max())
    print(f"\n[inverse_cholesky_upper K=28672] ||U H U^T - I||_F / sqrt(K) = {res:.2e}, max |entry| = {worst:.2e}")
    assert res <= 2e-4 and worst <= 5e-3
    del R


@pytest.mark.parametrize("name,N,K,T", [("70B down_proj 8192x28672", 8192, 28672, 32768), ("70B gate+up stacked 57344x8192", 57344, 8192, 16384)])
def test_column_loop_70b_shapes_vs_oracle(hip, name, N, K, T):
    """The blocked column loop at 224 / 64 blocks with the factor computed above injected on BOTH sides: the oracle's loop on 48
    sampled rows vs the HIP kernels on all rows -- identical codes up to a row's first rounding tie, as at 7B size."""
    from neural_compressor_amd.torch.algorithms.weight_only.gptq import GPTQ, HessianAccumulator

    Hd, U, dead = _factor(hip, K, T)
    g = torch.Generator().manual_seed(N + K)
    W = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16)
    rows = torch.sort(torch.randperm(N, generator=g)[:48])[0]
    rows[0], rows[-1] = 0, N - 1
    Ucpu = U.cpu()
    Hcpu_diag_ok = torch.zeros(K, K)  # only diag(H) == 0 matters to the oracle (dead columns): none here
    Hcpu_diag_ok.diagonal().fill_(1.0)
    ref = O.gptq_fasterquant(W[rows].float(), Hcpu_diag_ok, bits=4, sym=True, blocksize=128, percdamp=0.01, groupsize=GS, Hinv=Ucpu, trace=True)
    del Hcpu_diag_ok
    layer = torch.nn.Linear(K, N, bias=False, device=hip, dtype=torch.bfloat16)
    layer.weight.data.copy_(W.to(hip))
    acc = HessianAccumulator(K, hip)
    acc._n = 8
    acc.finalized = ((0.01, False), U, dead, None)
    gq = GPTQ(layer, device=hip, accumulator=acc)
    gq.configure(dict(bits=4, sym=True, dtype="int", mse=False))
    scale, _, zero, Q = gq.fasterquant(layer.weight.data, blocksize=128, percdamp=0.01, groupsize=GS)
    sc = ref["scale"].repeat_interleave(GS, 1)[:, :K]
    ref_codes = torch.round(ref["Q"] / sc + 8).to(torch.int32)
    got = gq.codes[rows.to(hip)].cpu().to(torch.int32)
    neq = got != ref_codes
    bad = torch.nonzero(neq.any(1)).flatten().tolist()
    tie = 0.0
    for r in bad:
        c = int(torch.nonzero(neq[r]).flatten()[0])
        u = float(ref["Win"][r, c] / sc[r, c])
        tie = max(tie, abs((u - math.floor(u)) - 0.5))
    clean = ~neq.any(1)
    s_rel = float(((scale[rows.to(hip)].cpu()[clean] - ref["scale"][clean]).abs() / ref["scale"][clean]).max()) if bool(clean.any()) else 0.0
    print(f"\n[column loop {name}] 48 sampled rows x {K} columns: {int(neq.sum())} codes differ in {len(bad)} rows, largest first-difference "
          f"tie distance {tie:.2e} steps, max scale rel diff on identical rows {s_rel:.2e}")
    assert tie <= 1e-3 and len(bad) <= 12 and s_rel <= 1e-3
    assert int(gq.codes.max()) <= 15
    grid = (gq.codes.float() - 8.0) * scale.repeat_interleave(GS, dim=1)
    assert torch.equal(grid.to(torch.bfloat16), Q)


@pytest.mark.parametrize("M", [1, 64, 4096])
@pytest.mark.parametrize("N,K", [(28672, 8192), (8192, 28672)])
def test_fused_gemm_70b_shapes_vs_oracle(hip, M, N, K):
    from neural_compressor_amd.torch.algorithms.weight_only.modules import MI355XWeightOnlyLinear
    from neural_compressor_amd.torch.algorithms.weight_only.utility import quant_tensor

    key = ("packed", N, K)
    if key not in _C:
        g = torch.Generator().manual_seed(N + 5 * K)
        w = (torch.randn(N, K, generator=g) * 0.02).to(hip)
        iw, sc, _ = quant_tensor(w, bits=4, group_size=GS, scheme="sym", return_int=True)
        m = MI355XWeightOnlyLinear(K, N, bits=4, group_size=GS, device=hip)
        m.pack(iw, sc, None, None)
        m.bias = None
        cols = torch.sort(torch.randperm(N // 8, generator=g)[:64])[0]  # 64 qzeros words = 512 output columns
        cidx = (cols.view(-1, 1) * 8 + torch.arange(8).view(1, -1)).reshape(-1)
        qw = m.qweight.cpu()[:, cidx].numpy()
        scs = m.scales.cpu()[:, cidx].numpy()
        qz = m.qzeros.cpu()[:, cols].numpy()
        _C[key] = (m, cidx, O.woq_dense_weight(qw, scs, qz, len(cidx), K, 4, GS, torch.bfloat16))
        del w, iw
    m, cidx, dense = _C[key]
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16)
    y = m(x.to(hip))
    sel = torch.arange(M) if M <= 64 else torch.sort(torch.randperm(M, generator=g)[:32])[0]
    ref = torch.nn.functional.linear(x[sel].float(), dense)
    got = y[sel.to(hip)][:, cidx.to(hip)].float().cpu()
    e = rel_fro(got, ref.to(torch.bfloat16).float())
    print(f"\n[fused gemm 70B M={M} N={N} K={K}] 512 sampled columns vs oracle rounded to bf16: {e:.2e}")
    assert e <= 1e-3
