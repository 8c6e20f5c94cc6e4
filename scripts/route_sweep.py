"""Sweep of inc_woq_gemm's routes: bits x group size x M x layer shape (incl. ragged N, act_order g_idx), fused forward against HIP
recover() + library GEMM (cache-resident graph replay) -- rows where the fused route is the slower one are flagged.
usage: python scripts/route_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_compressor_amd.torch.algorithms.weight_only.modules import MI355XWeightOnlyLinear  # noqa: E402
from neural_compressor_amd.torch.algorithms.weight_only.utility import quant_tensor  # noqa: E402

dev = torch.device("cuda:0")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def graph_time(fn, calls=10, reps=3):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(calls):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * calls) * 1e3


cases = []
for bits in (4, 8):
    for gs in (32, 128, -1):
        for (N, K) in ((4096, 4096), (11008, 4096), (4096, 11008), (1000, 1024)):
            cases.append((bits, gs, N, K, "sym", False))
cases += [(4, 128, 4096, 4096, "asym", False), (4, 128, 4096, 4096, "sym", True), (4, 64, 4096, 4096, "sym", False), (4, 96, 4096, 4032, "sym", False)]
for bits, gs, N, K, scheme, act_order in cases:
    w = torch.randn(N, K, device=dev) * 0.02
    iw, sc, zp = quant_tensor(w, bits=bits, group_size=gs, scheme=scheme, return_int=True)
    gsz = K if gs == -1 else gs
    g_idx = None
    m = MI355XWeightOnlyLinear(K, N, bits=bits, group_size=gs, zp=scheme == "asym", g_idx=act_order, device=dev)
    if act_order:
        perm = torch.randperm(K // gsz, device=dev).repeat_interleave(gsz)  # whole groups permuted (GPTQ act_order after sorting)
        g_idx = perm[torch.randperm(K, device=dev)].to(torch.int32)
    m.pack(iw, sc, zp if scheme == "asym" else None, None, g_idx=g_idx)
    m.bias = None
    for M in (1, 8, 16, 17, 32, 64, 65, 128, 512, 2048):
        x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        t = graph_time(lambda: m(x))
        td = graph_time(lambda: torch.nn.functional.linear(x, m.recover(dtype=torch.bfloat16)))
        flag = "  <-- fused slower" if t > 1.15 * td else ""
        print(f"bits={bits} gs={gs} {scheme}{' act_order' if act_order else ''} {N}x{K} M={M}: plan={m._plan} fused {t:8.1f} us, recover + library GEMM {td:8.1f} us, ratio {t / td:5.2f}{flag}")
