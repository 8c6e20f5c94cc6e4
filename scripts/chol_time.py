"""Time the inverse-Cholesky factor of GPTQ (Hinv) alone on the GPU: the one-call C-ABI form (inc_gptq_inverse_factor) and the
Python + torch.mm form it replaced.  usage: python scripts/chol_time.py [K ...]"""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import neural_compressor_amd.torch.algorithms.weight_only.gptq as G  # noqa: E402
from tests.ab_partners import inverse_cholesky_upper_python  # noqa: E402  (the Python + torch.mm A/B partner)


def main():
    forms = ("cabi1", "x3", "python")  # cabi1: one stream; x3: large products as three-way bf16 splits (flags bit 1, the driver's default)
    args = sys.argv[1:]
    if "--form" in args:
        i = args.index("--form")
        forms = (args[i + 1],)
        del args[i:i + 2]
    Ks = [int(a) for a in args] or [4096, 11008]
    dev = torch.device("cuda")
    for K in Ks:
        torch.manual_seed(K)
        X = torch.randn(4 * K if K <= 4096 else 2 * K, K, device=dev)
        H = (X.t() @ X) / X.shape[0]
        H.diagonal().add_(0.01 * H.diagonal().mean())
        H64 = H.double()
        ref = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(H64)), upper=True) if K <= 11008 else None
        first = None
        for form in forms:
            G.CHOL_LOOKAHEAD = form == "cabi"
            G.CHOL_BF16X3 = form == "x3"
            factor = (lambda h, check=True: inverse_cholesky_upper_python(h, check=check)) if form == "python" else G.inverse_cholesky_upper
            U = factor(H)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                U = factor(H, check=False)[0]
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 5 * 1e3
            if first is None:
                first = U
            R = U.double() @ H64 @ U.double().t()
            res = float((R - torch.eye(K, device=dev, dtype=torch.float64)).norm() / K ** 0.5)
            err = f" rel err vs fp64 trio {float((U.double() - ref).norm() / ref.norm()):.2e}" if ref is not None else ""
            print(f"K={K} {form:7s}: {ms:8.2f} ms  |U H U^T - I|_F/sqrt(K)={res:.2e}{err}  rel diff to the C-ABI form: {float((U - first).norm() / first.norm()):.2e}", flush=True)


if __name__ == "__main__":
    main()
