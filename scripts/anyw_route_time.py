"""Odd-width modules (1 / 2 / 3 / 5 / 6 / 7 bits): inc_woq_gemm's per-element tile form against recover() + the library GEMM, per M
(the crossover sets modules.ANYW_FUSED_MAX_M).  usage: python scripts/anyw_route_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_compressor_amd import ops  # noqa: E402
from neural_compressor_amd.torch.algorithms.weight_only.modules import MI355XWeightOnlyLinear  # noqa: E402
from neural_compressor_amd.torch.algorithms.weight_only.utility import quant_tensor  # noqa: E402

dev = torch.device("cuda:0")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for bits in (2, 3):
    for N, K in ((4096, 4096), (11008, 4096)):
        w = torch.randn(N, K, device=dev) * 0.02
        iw, sc, zp = quant_tensor(w, bits=bits, group_size=128, scheme="asym", return_int=True)
        m = MI355XWeightOnlyLinear(K, N, bits=bits, group_size=128, zp=True, device=dev)
        m.pack(iw, sc, zp, None)
        m.bias = None
        for M in (1, 16, 64, 128, 256, 512, 1024, 4096):
            x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
            fused = timed(lambda: ops.woq_gemm(x, m.qweight, m.scales, m.qzeros, None, N, K, 128, bits))
            dense = timed(lambda: torch.nn.functional.linear(x, m.recover(dtype=torch.bfloat16)))
            print(f"bits={bits} {N}x{K} M={M}: fused tile form {fused:8.1f} us, recover + library GEMM {dense:8.1f} us -> {'fused' if fused < dense else 'dense'}")
