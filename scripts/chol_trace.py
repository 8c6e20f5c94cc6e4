"""inverse_cholesky_upper at one K, a few times (for `rocprofv3 --kernel-trace`; scripts/gpu_r5.sh choltrace summarises the last run).
usage: python scripts/chol_trace.py [K]"""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import neural_compressor_amd.torch.algorithms.weight_only.gptq as G  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 11008
G.CHOL_LOOKAHEAD = True
dev = torch.device("cuda")
torch.manual_seed(K)
X = torch.randn(2 * K, K, device=dev)
H = (X.t() @ X) / X.shape[0]
H.diagonal().add_(0.01 * H.diagonal().mean())
del X
for i in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    U = G.inverse_cholesky_upper(H, check=False)[0]
    torch.cuda.synchronize()
    print(f"run {i}: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
G.CHOL_LOOKAHEAD = False
torch.cuda.synchronize()
t0 = time.perf_counter()
U1 = G.inverse_cholesky_upper(H, check=False)[0]
torch.cuda.synchronize()
print(f"one stream: {(time.perf_counter() - t0) * 1e3:.2f} ms; identical to the look-ahead result: {bool(torch.equal(U, U1))}", flush=True)
R = U @ H @ U.t()
print("residual", float((R - torch.eye(K, device=dev)).norm() / K ** 0.5))
