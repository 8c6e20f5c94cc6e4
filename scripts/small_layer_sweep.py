"""Large M on small layers (the shapes of BASELINE config #1's OPT-125M-like stack and a few between them and 4096^2): fused forward
against HIP recover() + library GEMM, cache-resident graph replay.  Both a cache-resident graph replay and a back-to-back eager loop (profiles/NOTES.md round 6: no routing rule came out of it).
usage: python scripts/small_layer_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_compressor_amd.torch.algorithms.weight_only.modules import MI355XWeightOnlyLinear  # noqa: E402
from neural_compressor_amd.torch.algorithms.weight_only.utility import quant_tensor  # noqa: E402

e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def graph_time(fn, calls=10, reps=3):  # (scripts/route_sweep.py's timer)
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

dev = torch.device("cuda:0")


def eager_time(fn, calls=60):  # back-to-back eager calls (host launch cost and the allocator included), as bench.py's prefill rows do
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(calls):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / calls * 1e3


if __name__ == "__main__":
    for bits in (4, 8):
        for gs in (128, -1):
            for (N, K) in ((768, 768), (3072, 768), (768, 3072), (1000, 1024), (2048, 2048), (4096, 1024), (2560, 2560), (4096, 4096)):
                w = torch.randn(N, K, device=dev) * 0.02
                iw, sc, zp = quant_tensor(w, bits=bits, group_size=gs, scheme="sym", return_int=True)
                m = MI355XWeightOnlyLinear(K, N, bits=bits, group_size=gs, zp=False, device=dev)
                m.pack(iw, sc, None, None)
                m.bias = None
                for M in (128, 256, 512, 1024, 2048, 4096, 8192):
                    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
                    t = graph_time(lambda: m(x))
                    td = graph_time(lambda: torch.nn.functional.linear(x, m.recover(dtype=torch.bfloat16)))
                    te = eager_time(lambda: m(x))
                    tde = eager_time(lambda: torch.nn.functional.linear(x, m.recover(dtype=torch.bfloat16)))
                    flag = "  <-- fused slower in both" if (t > 1.1 * td and te > 1.1 * tde) else ""
                    print(f"bits={bits} gs={gs} {N}x{K} ({N * K / 2**20:.2f} Mi) M={M}: fused {t:8.1f} us, recover + library GEMM {td:8.1f} us, ratio {t / td:5.2f}; "
                          f"eager loop {te:8.1f} / {tde:8.1f} us, ratio {te / tde:5.2f}{flag}", flush=True)
