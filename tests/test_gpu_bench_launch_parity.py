"""-m gpu: parity at EXACTLY what bench.py times (VERDICT r5 "next round" item 1).

The bench's Hessian call is `inc_gptq_hessian_accum_multi` over the four distinct inputs of a Llama-2-7B block
(K = 4096, 4096, 4096, 11008) at T = 65 536 tokens per launch (32 stacked samples of 2048), bf16, with the launch's last
partial round split over token ranges (`ops.HESSIAN_TAIL_SPLIT`).  tests/test_gpu_baseline_parity.py compares the kernels
at T = 16 384 and starts every solve from the ORACLE'S H.  Here:

  (a) that launch, through the driver's own objects (HessianAccumulator.defer + flush_many, zero-copy input), against
      `O.gptq_add_batch` (gptq.py:1111-1141) on the whole matrix and against an fp64 referee on sampled 256 x 256 tiles
      that include EVERY tile of the split tail round;
  (b) one un-injected chain per headline shape: that HIP Hessian -> inc_gptq_hessian_finalize -> inc_gptq_inverse_factor ->
      inc_gptq_quantize_layer, against `O.gptq_fasterquant` (gptq.py:1143-1351: fp32 LAPACK trio + column loop) started from
      the oracle's own H of the same activations, on sampled rows;
  (c) what must hold on every sampled row, flipped ties or not: the row's GPTQ objective (w - q)^T H (w - q) against the
      oracle's, and the fraction of groups whose scale differs by more than north_star's 1e-3.

The oracle's CPU work is ~1 minute on the GPU box's host (a [65536, 11008] fp32 GEMM and three row-sampled column loops).
"""

import math

import pytest
import torch

from oracle import woq_oracle as O

pytestmark = pytest.mark.gpu

GS = 128
T_LAUNCH = 65536          # HessianAccumulator.STAGE_TOKENS: one launch of the driver
SAMPLES = 32              # 32 stacked calibration samples of 2048 tokens
KS = (4096, 4096, 4096, 11008)   # q/k/v input, o_proj input, gate/up input, down_proj input
N_PRIOR = (1, 1, 0, 0)    # problems 0 / 1 start from one earlier sample (beta != 0 in the launch), 2 / 3 from zero

_C = {}


def rel_fro(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def _acts(p):
    """Synthetic calibration activations of problem p: N(0,1), ~1 % outlier channels x20, bf16 [32, 2048, K]."""
    K = KS[p]
    g = torch.Generator().manual_seed(100 + p)
    x = torch.randn(SAMPLES, T_LAUNCH // SAMPLES, K, generator=g)
    x[..., (7 * p) % 97:: 97] *= 20.0
    return x.to(torch.bfloat16)


def _prior(p):
    """One earlier 2048-token sample folded in by the oracle: the H (and n = 1) the launch starts from when N_PRIOR[p] = 1."""
    K = KS[p]
    g = torch.Generator().manual_seed(200 + p)
    x0 = torch.randn(1, 2048, K, generator=g).to(torch.bfloat16)
    return O.gptq_add_batch(torch.zeros(K, K), 0, x0.float())


def _launch(hip):
    """The bench's launch, once per session: (HIP H list [device], oracle H list [CPU], activations [device], tail-tile masks)."""
    if "launch" in _C:
        return _C["launch"]
    from neural_compressor_amd import ops
    from neural_compressor_amd.torch.algorithms.weight_only.gptq import HessianAccumulator

    assert HessianAccumulator.STAGE_TOKENS == T_LAUNCH and ops.HESSIAN_TAIL_SPLIT, "the test must run the bench's configuration"
    xs_cpu = [_acts(p) for p in range(4)]
    xs = [x.to(hip) for x in xs_cpu]
    priors = [_prior(p) if N_PRIOR[p] else None for p in range(4)]

    def run(split):
        ops.HESSIAN_TAIL_SPLIT = split
        try:
            accs = []
            for p, K in enumerate(KS):
                acc = HessianAccumulator(K, hip)
                acc.defer = True     # RAWGPTQuantizer._run_block: the full stage waits for flush_many
                acc._zc_ok = True    # steady state of the driver: stacked inputs are read in place (no staging copy)
                if priors[p] is not None:
                    acc.H, acc._n = priors[p][0].to(hip).clone(), priors[p][1]
                accs.append(acc)
            calls = []
            orig = ops.gptq_hessian_accum_multi
            ops.gptq_hessian_accum_multi = lambda items: (calls.append(len(items)), orig(items))[1]
            try:
                for acc, x in zip(accs, xs):
                    acc.add_batch(x)
                HessianAccumulator.flush_many(accs)
            finally:
                ops.gptq_hessian_accum_multi = orig
            assert calls == [4], "the four Hessians must have gone out as ONE inc_gptq_hessian_accum_multi launch"
            assert all(a._n == SAMPLES + n0 and a._pending == 0 for a, n0 in zip(accs, N_PRIOR))
            return [a.H for a in accs]
        finally:
            ops.HESSIAN_TAIL_SPLIT = True

    H_split = run(True)      # what the bench launches
    H_whole = run(False)     # every tile one workgroup: differs from H_split exactly on the split tail's tiles
    tail = []
    for a, b, K in zip(H_split, H_whole, KS):
        nt = K // 256
        d = (a != b).reshape(nt, 256, nt, 256).any(dim=3).any(dim=1)
        tail.append(torch.triu(d))
    Ho = []
    for p in range(4):
        H0, n0 = priors[p] if priors[p] is not None else (torch.zeros(KS[p], KS[p]), 0)
        H, n = O.gptq_add_batch(H0, n0, xs_cpu[p].float())
        assert n == SAMPLES + N_PRIOR[p]
        Ho.append(H)
    _C["launch"] = (H_split, Ho, xs, tail, priors)
    return _C["launch"]


def test_hessian_multi_at_the_bench_launch_T65536_vs_oracle_and_fp64(hip):
    H_hip, H_or, xs, tail, priors = _launch(hip)
    n_tail = sum(int(t.sum()) for t in tail)
    # 3 x 136 + 946 = 1354 tiles on 256 CUs: 5 full rounds, 74 tiles left for the split round
    print(f"\n[hessian multi T={T_LAUNCH}, K={'+'.join(map(str, KS))}, tail split on] tiles whose bits differ from the unsplit launch "
          f"(= the split round): {n_tail} of {sum((K // 256) * (K // 256 + 1) // 2 for K in KS)}")
    assert 0 < n_tail <= 74
    g = torch.Generator().manual_seed(3)
    for p, K in enumerate(KS):
        nt = K // 256
        Hd, Hod = H_hip[p], H_or[p].to(hip)
        iu = torch.triu_indices(K, K, device=hip)
        e_pair = rel_fro(Hd[iu[0], iu[1]], Hod[iu[0], iu[1]])
        # every upper tile against the oracle
        A = torch.triu(Hd).double().reshape(nt, 256, nt, 256)
        B = torch.triu(Hod).double().reshape(nt, 256, nt, 256)
        num = (A - B).pow(2).sum(dim=(1, 3)).sqrt()
        den = B.pow(2).sum(dim=(1, 3)).sqrt().clamp_min(1e-30)
        upper = torch.triu(torch.ones(nt, nt, dtype=torch.bool, device=hip))
        t_pair = float((num / den)[upper].max())
        del A, B
        # fp64 referee on sampled tiles: all tiles of the split round + 24 others (diagonal and off-diagonal)
        picks = {(int(i), int(j)) for i, j in torch.nonzero(tail[p]).tolist()}
        cand = [(i, j) for i in range(nt) for j in range(i, nt)]
        for idx in torch.randperm(len(cand), generator=g)[:24].tolist():
            picks.add(cand[idx])
        picks |= {(0, 0), (nt - 1, nt - 1), (0, nt - 1)}
        n0 = N_PRIOR[p]
        beta, alpha = n0 / (n0 + SAMPLES), 2.0 / (n0 + SAMPLES)
        x2 = xs[p].reshape(-1, K)
        worst_hip = worst_or = worst_tail = 0.0
        for (i, j) in sorted(picks):
            xi = x2[:, i * 256:(i + 1) * 256].double()
            xj = x2[:, j * 256:(j + 1) * 256].double()
            ref = alpha * (xi.t() @ xj)
            if n0:
                ref += beta * priors[p][0][i * 256:(i + 1) * 256, j * 256:(j + 1) * 256].to(hip).double()
            a = Hd[i * 256:(i + 1) * 256, j * 256:(j + 1) * 256].double()
            b = Hod[i * 256:(i + 1) * 256, j * 256:(j + 1) * 256].double()
            if i == j:
                ref, a, b = torch.triu(ref), torch.triu(a), torch.triu(b)
            eh, eo = float((a - ref).norm() / ref.norm()), float((b - ref).norm() / ref.norm())
            worst_hip, worst_or = max(worst_hip, eh), max(worst_or, eo)
            if bool(tail[p][i, j]):
                worst_tail = max(worst_tail, eh)
        print(f"[hessian multi T={T_LAUNCH} problem {p}: K={K}, beta={beta:.4f}] HIP vs oracle: upper triangle {e_pair:.2e}, worst 256-tile {t_pair:.2e}; "
              f"vs fp64 on {len(picks)} sampled tiles ({int(tail[p].sum())} of them in the split round): HIP worst {worst_hip:.2e} "
              f"(split-round tiles {worst_tail:.2e}), oracle worst {worst_or:.2e}")
        # same gates as the T = 16384 test: float noise from the oracle, and no farther from the exact product than the oracle is (x2)
        assert e_pair <= 2e-6
        assert t_pair <= 4e-6
        assert worst_hip <= max(4e-6, 2 * worst_or)


# (name, Hessian problem, N, K, sampled rows); budgets of the un-injected chain = 2 x measured (profiles/r6/parity_report.txt)
CHAIN = [("o_proj 4096x4096", 1, 4096, 4096, 384), ("gate+up stacked 22016x4096", 2, 22016, 4096, 384), ("down_proj 4096x11008", 3, 4096, 11008, 256)]
# (rows with any differing code, fraction of differing codes); measured 3 rows / 3.75e-5, 1 / 1.78e-5, 17 / 6.14e-4, ties <= 9.5e-7 steps
# (a flipped row re-rolls the rest of ITS row, so small counts are quantised: floors of 4 rows / 1e-4)
CHAIN_BUDGET = {"o_proj 4096x4096": (6, 1e-4), "gate+up stacked 22016x4096": (4, 1e-4), "down_proj 4096x11008": (34, 1.3e-3)}


def _chain(hip, name):
    """HIP chain and oracle chain of one headline shape on the same activations and weights (cached per session)."""
    key = ("chain", name)
    if key in _C:
        return _C[key]
    from neural_compressor_amd.torch.algorithms.weight_only.gptq import GPTQ, HessianAccumulator

    _, p, N, K, nsample = next(c for c in CHAIN if c[0] == name)
    H_hip, H_or, _, _, _ = _launch(hip)
    g = torch.Generator().manual_seed(N + K + 1)
    W = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16)
    rows = torch.sort(torch.randperm(N, generator=g)[:nsample])[0]
    rows[0], rows[-1] = 0, N - 1
    ref = O.gptq_fasterquant(W[rows].float(), H_or[p], bits=4, sym=True, blocksize=128, percdamp=0.01, groupsize=GS, trace=True)

    layer = torch.nn.Linear(K, N, bias=False, device=hip, dtype=torch.bfloat16)
    layer.weight.data.copy_(W.to(hip))
    acc = HessianAccumulator(K, hip)
    acc.H, acc._n = H_hip[p].clone(), SAMPLES + N_PRIOR[p]   # the HIP launch's H (upper triangle; finalize mirrors it)
    gq = GPTQ(layer, device=hip, accumulator=acc)
    gq.configure(dict(bits=4, sym=True, dtype="int", mse=False))
    scale, _, zero, Q = gq.fasterquant(layer.weight.data.clone(), blocksize=128, percdamp=0.01, groupsize=GS)
    out = dict(W=W, rows=rows, ref=ref, codes=gq.codes[rows.to(hip)].cpu().to(torch.int32), scale=scale[rows.to(hip)].cpu(),
               Q=Q[rows.to(hip)].float().cpu(), p=p, K=K, N=N)
    _C[key] = out
    return out


@pytest.mark.parametrize("name", [c[0] for c in CHAIN])
def test_uninjected_chain_hip_hessian_to_codes_vs_oracle_chain(hip, name):
    c = _chain(hip, name)
    ref, K = c["ref"], c["K"]
    sc = ref["scale"].repeat_interleave(GS, 1)[:, :K]
    ref_codes = torch.round(ref["Q"] / sc + 8.0).to(torch.int32)
    neq = c["codes"] != ref_codes
    bad = torch.nonzero(neq.any(1)).flatten()
    ties = []
    for r in bad.tolist():
        col = int(torch.nonzero(neq[r]).flatten()[0])
        u = float(ref["Win"][r, col] / sc[r, col])
        ties.append(abs((u - math.floor(u)) - 0.5))
        assert abs(int(c["codes"][r, col]) - int(ref_codes[r, col])) == 1
    frac = float(neq.float().mean())
    clean = ~neq.any(1)
    s_rel = float(((c["scale"][clean] - ref["scale"][clean]).abs() / ref["scale"][clean]).max()) if bool(clean.any()) else 0.0
    print(f"\n[un-injected chain {name}: HIP Hessian (T={T_LAUNCH}, multi launch) -> own factor -> column loop, vs oracle H -> LAPACK trio -> loop] "
          f"{len(c['rows'])} sampled rows x {K} columns: {int(neq.sum())} codes differ ({frac:.2e}) in {len(bad)} rows; every first difference a +-1 "
          f"flip, largest distance of the oracle's own value from the rounding boundary there {max(ties) if ties else 0.0:.2e} steps; "
          f"max scale rel diff on the {int(clean.sum())} identical rows {s_rel:.2e}")
    max_rows, max_frac = CHAIN_BUDGET[name]
    # both the Hessian (different fp32 summation tree) and the factor (one factorisation instead of three) are float noise away
    # from the oracle's: a first difference may only sit at a rounding tie
    assert (max(ties) if ties else 0.0) <= 2e-5, "a code differs where the oracle's value was not at a rounding tie"
    assert len(bad) <= max_rows and frac <= max_frac, (len(bad), frac)
    assert s_rel <= 1e-5
    # the first block has seen no lazy update: only H's diagonal block and the factor's first rows act on it
    first = (c["codes"][:, :128] != ref_codes[:, :128]).any(1)
    assert int(first.sum()) <= max(2, max_rows // 4)


# budgets of (c) = 2 x measured (profiles/r6/parity_report.txt): (largest |objective ratio - 1| of a sampled row, fraction of groups whose scale
# is > 1e-3 off).  Measured: flipped rows land within +0.56 % / -0.46 % (4096 columns) and +0.98 % / -1.04 % (11008 columns) of the oracle's
# objective -- either sign: past a flipped tie the row follows another, equally good trajectory -- with the MEAN over the sampled rows at
# 1.00001-1.00004; 1.6e-3 / 6.5e-4 / 1.44e-2 of the sampled (row, group) scales move by more than 1e-3 (about a fifth of a flipped row's groups)
OBJ_BUDGET = {"o_proj 4096x4096": (1.2e-2, 3.3e-3), "gate+up stacked 22016x4096": (1.2e-2, 1.6e-3), "down_proj 4096x11008": (2.1e-2, 2.9e-2)}


@pytest.mark.parametrize("name", [c[0] for c in CHAIN])
def test_uninjected_chain_row_objective_and_scale_gates(hip, name):
    """North_star's tolerance is on per-group scales and dequantised weights; GPTQ is chaotic past a flipped tie (the rest of that
    row follows another trajectory), so on EVERY sampled row -- identical or not -- gate what must survive a flip: the row's GPTQ
    objective (w - q)^T H (w - q) (H = the oracle's fp32 Hessian, fp64 arithmetic) may not exceed the oracle's beyond noise, and the
    fraction of (row, group) scales that moved by more than 1e-3 is printed and budgeted."""
    c = _chain(hip, name)
    _, H_or, _, _, _ = _launch(hip)
    ref, K = c["ref"], c["K"]
    H64 = H_or[c["p"]].to(hip).double()
    W = c["W"][c["rows"]].float().to(hip).double()
    e_hip = W - c["Q"].to(hip).double()
    e_ref = W - ref["Q"].to(torch.bfloat16).float().to(hip).double()   # the reference stores Q in the weight dtype too (gptq.py:1330)
    obj_hip = ((e_hip @ H64) * e_hip).sum(1)
    obj_ref = ((e_ref @ H64) * e_ref).sum(1)
    ratio = (obj_hip / obj_ref).cpu()
    sc = ref["scale"].repeat_interleave(GS, 1)[:, :K]
    ref_codes = torch.round(ref["Q"] / sc + 8.0).to(torch.int32)
    flipped = (c["codes"] != ref_codes).any(1)
    s_dev = ((c["scale"] - ref["scale"]).abs() / ref["scale"])
    frac_groups = float((s_dev > 1e-3).float().mean())
    frac_groups_flipped = float((s_dev[flipped] > 1e-3).float().mean()) if bool(flipped.any()) else 0.0
    # dequantised weights: relative Frobenius distance of the two quantised rows (bf16 Q), per row
    q_rel = ((c["Q"] - ref["Q"].to(torch.bfloat16).float()).norm(dim=1) / ref["Q"].norm(dim=1))
    worst = float(ratio.max())
    print(f"\n[row objective {name}] (w-q)^T H (w-q), HIP / oracle over {len(ratio)} sampled rows: max {worst:.6f}, min {float(ratio.min()):.6f}, "
          f"mean {float(ratio.mean()):.6f}; on the {int(flipped.sum())} rows with a flipped tie: max {float(ratio[flipped].max()) if bool(flipped.any()) else 1.0:.6f}; "
          f"on the identical rows: max |ratio - 1| {float((ratio[~flipped] - 1).abs().max()) if bool((~flipped).any()) else 0.0:.2e}")
    print(f"[scales {name}] (row, group) scales deviating > 1e-3: {frac_groups:.2e} of all sampled ({frac_groups_flipped:.2e} of the flipped rows' groups; "
          f"largest deviation {float(s_dev.max()):.2e}); dequantised rows: max rel-Frobenius distance to the oracle's row {float(q_rel.max()):.2e}, "
          f"rows farther than 1e-3: {int((q_rel > 1e-3).sum())} of {len(q_rel)}")
    max_obj, max_groups = OBJ_BUDGET[name]
    # rows without a flip: same codes, scales a float-noise away (H and the factor differ by 1e-6) -> the same objective to 1e-4
    if bool((~flipped).any()):
        assert float((ratio[~flipped] - 1).abs().max()) <= 1e-4
    # rows with a flip: no row may be WORSE than the oracle's beyond the measured spread, and there is no systematic loss -- the mean
    # objective over all sampled rows equals the oracle's to 2e-4 (a per-row gate of 1.001 does not hold for GPTQ past a flipped tie:
    # the oracle itself, run with another BLAS summation order, moves its own rows by the same +-1 %)
    assert float((ratio - 1).abs().max()) <= max_obj, f"a row's GPTQ objective is {worst:.5f} x the oracle's"
    assert abs(float(ratio.mean()) - 1.0) <= 2e-4
    assert frac_groups <= max_groups
    # rows that did not flip are within north_star's 1e-3 on scales and dequantised weights (in fact identical)
    assert float(s_dev[~flipped].max()) <= 1e-3 and float(q_rel[~flipped].max()) <= 1e-3
