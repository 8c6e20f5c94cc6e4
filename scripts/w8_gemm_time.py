import sys, torch
sys.path.insert(0, "/root/repo")
from tests.test_gpu_parity import _packed_layer
dev = torch.device("cuda:0")
for (M, N, K, gs) in [(4096, 4096, 4096, 128), (4096, 3072, 768, -1), (4096, 11008, 4096, 128), (64, 4096, 4096, 128)]:
    m = _packed_layer(dev, N, K, gs, 8, True, seed=1, bias=False); m.bias = None
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    for _ in range(5): m(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): m(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"W8 GEMM M={M} N={N} K={K} gs={gs}: {ms:.4f} ms  {2.0*M*N*K/ms/1e9:.1f} TFLOP/s")
