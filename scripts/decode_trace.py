"""Decode launches under rocprofv3 --kernel-trace: per-kernel durations of the cold rings bench.py times (single modules and module groups),
next to the per-launch time of the graph replay (which also contains the gap between dependent kernels).
usage: rocprofv3 --kernel-trace --stats ... -- python scripts/decode_trace.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
rows = bench.bench_gemv_groups(dev)
for r in rows:
    print({k: r[k] for k in ("group", "cold_graph_ms", "cold_graph_frac", "single_calls_cold_graph_ms")})
single = bench.bench_dequant_gemm(dev, [(1, 4096, 4096), (1, 11008, 4096), (1, 4096, 11008)])
for r in single:
    print({k: r.get(k) for k in ("M", "N", "K", "graph_ms", "cold_graph_ms", "cold_graph_frac")})
