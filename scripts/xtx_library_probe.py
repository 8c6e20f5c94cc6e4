"""Probe (round 5): the library's GEMM on the OFF-DIAGONAL blocks of X^T X -- (K1 x T) . (T x K2) with X [T, K] row-major, i.e. both operands
token-major -- against this repository's syrk tile on the same shapes.  Timing only."""
import torch, time
dev = "cuda"
T = 65536
def t(fn, reps=6):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
x = torch.randn(T, 11008, device=dev, dtype=torch.bfloat16)
for K1, K2 in ((5504, 5504), (2752, 2752), (4096, 4096), (2048, 2048), (11008, 11008)):
    a, b = x[:, :K1], x[:, 11008 - K2:]
    ms = t(lambda: torch.mm(a.t(), b))
    print(f"torch.mm(X1^T [{K1} x {T}], X2 [{T} x {K2}]) strided views: {ms:.3f} ms = {2.0 * T * K1 * K2 / ms / 1e9:.0f} TFLOP/s", flush=True)
    ac, bc = a.contiguous(), b.contiguous()
    ms = t(lambda: torch.mm(ac.t(), bc))
    print(f"  contiguous panels: {ms:.3f} ms = {2.0 * T * K1 * K2 / ms / 1e9:.0f} TFLOP/s", flush=True)
    try:
        ms = t(lambda: torch.mm(ac.t(), bc, out_dtype=torch.float32))
        print(f"  out_dtype=float32: {ms:.3f} ms = {2.0 * T * K1 * K2 / ms / 1e9:.0f} TFLOP/s", flush=True)
    except Exception as e:
        print("  out_dtype=float32 unsupported:", type(e).__name__, str(e)[:100])
