"""Weight-only INT8 (BASELINE config #1's format: bits = 8, per-channel or g128) forward through inc_woq_gemm, per M, next to HIP recover +
library GEMM and to the INT4 module of the same shape.  usage: python scripts/w8_gemm_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_compressor_amd.torch.algorithms.weight_only.modules import MI355XWeightOnlyLinear  # noqa: E402
from neural_compressor_amd.torch.algorithms.weight_only.utility import quant_tensor  # noqa: E402

dev = torch.device("cuda:0")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def graph_time(fn, calls=20, reps=5):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
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


for N, K in ((4096, 4096), (11008, 4096)):
    for bits, gs in ((8, -1), (8, 128), (4, 128)):
        w = torch.randn(N, K, device=dev) * 0.02
        iw, sc, _ = quant_tensor(w, bits=bits, group_size=gs, scheme="sym", return_int=True)
        m = MI355XWeightOnlyLinear(K, N, bits=bits, group_size=gs, device=dev)
        m.pack(iw, sc, None, None)
        m.bias = None
        byts = m.qweight.numel() * 4 + m.scales.numel() * 2 + m.qzeros.numel() * 4
        for M in (1, 16, 17, 32, 48, 64, 65, 256, 4096):
            x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
            y = m(x)
            ref = x.float() @ m.recover(dtype=torch.bfloat16).float().t()
            err = float((y.float() - ref).norm() / ref.norm())
            t = graph_time(lambda: m(x))
            td = graph_time(lambda: torch.nn.functional.linear(x, m.recover(dtype=torch.bfloat16)))
            print(f"bits={bits} gs={gs} {N}x{K} M={M}: fused {t:8.1f} us ({byts / t / 1e3 / 8000:.3f} of HBM on the packed bytes, {2.0 * M * N * K / t / 1e6:7.1f} TFLOP/s), "
                  f"recover + library GEMM {td:8.1f} us, rel err {err:.1e} (cache-resident graph replay)")
