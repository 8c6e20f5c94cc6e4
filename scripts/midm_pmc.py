"""Mid-M fused GEMM and one-launch decode groups, a few launches each, for rocprofv3 --pmc passes (scripts/gpu_pmc_midm.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_compressor_amd.torch.algorithms.weight_only.modules import MI355XWeightOnlyLinear, woq_linear_group  # noqa: E402
from neural_compressor_amd.torch.algorithms.weight_only.utility import quant_tensor  # noqa: E402

dev = torch.device("cuda:0")


def packed(N, K):
    w = torch.randn(N, K, device=dev) * 0.02
    iw, sc, _ = quant_tensor(w, bits=4, group_size=128, scheme="sym", return_int=True)
    m = MI355XWeightOnlyLinear(K, N, bits=4, group_size=128, device=dev)
    m.pack(iw, sc, None, None)
    m.bias = None
    return m


m = packed(4096, 4096)
for M in (128, 256, 512, 1024):
    x = torch.randn(M, 4096, device=dev, dtype=torch.bfloat16)
    for _ in range(6):
        m(x)
qkv = [packed(4096, 4096) for _ in range(3)]
x1 = torch.randn(1, 4096, device=dev, dtype=torch.bfloat16)
for _ in range(6):
    woq_linear_group(x1, qkv)
torch.cuda.synchronize()
