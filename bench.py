#!/usr/bin/env python
"""bench.py -- BASELINE.json metric: "GPTQ quantize wall-clock (s) + INT4->bf16 dequant-GEMM TFLOPS, Llama-2-7B g128".

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

A STEP = the full GPTQ treatment of ONE Llama-2-7B transformer block (hidden 4096, ffn 11008, 32 heads, bf16,
random init) with the BASELINE calibration set (128 samples x 2048 tokens, synthetic ids):
    block forward over all samples with the Hessian hooks (inc_gptq_hessian_accum, one H per distinct input)
    -> Cholesky-inverse + blocked column loop for the 7 Linears (inc_gptq_quant_block / inc_gptq_lazy_update)
    -> second forward with the quantised weights (feeds the next block) -> on-device packing (inc_woq_pack),
driven through the public API objects (prepare -> run_fn -> RAWGPTQuantizer.quantize_block).
`value` = wall-clock seconds to GPTQ-quantise Llama-2-7B (32 such blocks) = 32 * T / K, T = time of the K timed steps.
With N > 1 the N ranks quantise ONE model together (neural_compressor_amd/distributed.py); when the script is started WITHOUT
a torchrun environment it launches its own N ranks (torch.distributed.run, one per GPU, backend "nccl" = RCCL) and fails if the
process group it ends up in does not have N ranks.  Two modes (--mgpu-mode):
  layer (default for N > 1; BASELINE north_star, SURVEY 8(e) mode B): one transformer block per GPU.  Samples are sharded; every
      rank forwards its samples through the float blocks of a round (N consecutive blocks), the block inputs travel to the block's
      owner over RCCL / xGMI (point-to-point, INC_MI355X_GPTQ_ACT_EXCHANGE=broadcast for the broadcast form), each rank quantises
      ITS block on the full calibration set.  Every block is calibrated on the FLOAT model's activations -- a documented deviation
      from the reference's sequential scheme (gptq.py:749-762); per block the result is bit-identical to a single process run of
      the same mode.  A STEP is then one ROUND (N blocks, one per rank): value = ceil(32 / N) * T / K.
  exact (mode "sample+rows", exact reference semantics): calibration samples are sharded (block forwards + Hessian accumulation
      run on 128/N samples per rank, no activation crosses GPUs), the i-th distinct Hessian of the block is reduced to rank
      i % N, factorised there and its factor broadcast, every column loop runs row-sharded and the codes / scales are
      all-gathered; every rank ends with the same packed block.
Total work is fixed -> "strong" scaling; nothing is divided by N.
`e2e` times the thing the north_star names once more, without extrapolation: a full 32-block model through
prepare -> run_fn (calibration capture) -> convert, wall-clock.  The second half of the metric, the fused INT4->bf16
dequant-GEMM, is timed on the BASELINE shapes and reported in `dequant_gemm`; `roofline` describes the kernel that
dominates the step (chosen from the measured breakdown); `awq_block` / `smoothquant_block` are BASELINE configs #3 / #4
at full layer size (one block each).
"""

import argparse
import copy
import json
import os
import sys
import time

import torch

T_START = time.perf_counter()
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BF16_MFMA_PEAK_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md: dense bf16 MFMA ~2.5 PFLOP/s
HBM_PEAK_GBS = 8000.0           # HBM3E 8 TB/s spec
F32_MFMA_PEAK_TFLOPS = 157.0    # exact-fp32 matrix instruction (SURVEY 8(d) peak denominators); the bare loop measures 154.6 (tools/f32mfma_lab, profiles/r6)


class KernelClock:
    """HIP-event timing of selected C-ABI calls on the stream they are launched on (torch's current stream)."""

    def __init__(self):
        self.pairs = {}
        self.work = {}

    def wrap(self, module, name, key_fn, work_fn):
        orig = getattr(module, name)
        clock = self

        def timed(*a, **k):
            key = key_fn(*a, **k)
            if key is None or not clock.enabled:
                return orig(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig(*a, **k)
            e1.record()
            clock.pairs.setdefault(key, []).append((e0, e1))
            clock.work[key] = clock.work.get(key, 0.0) + work_fn(*a, **k)
            return out

        setattr(module, name, timed)

    enabled = False

    def reset(self):
        self.pairs, self.work = {}, {}

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for key, pairs in self.pairs.items():
            ms = sum(a.elapsed_time(b) for a, b in pairs)
            out[key] = dict(launches=len(pairs), total_ms=ms, avg_ms=ms / len(pairs), work=self.work[key])
        return out


def _pmc_traffic(key):
    """HBM bytes per launch of the roofline kernel.  PMC counters cannot be read from inside the process being timed:
    they come from separate `rocprofv3 --pmc` passes over THIS command (scripts/gpu_pmc_bench.sh: FETCH_SIZE doubled as
    the MI355X guide prescribes for wide coalesced reads on gfx950, plus WRITE_SIZE), whose per-launch means are
    committed under profiles/ (newest round first); null when no PMC pass has been recorded for this kernel."""
    for rnd in ("r6_pmc", "r5_pmc", "r4_pmc", "r3_pmc", "r2_pmc", "r1_pmc"):
        path = os.path.join(ROOT, "profiles", rnd, "bench_traffic.json")
        try:
            with open(path) as f:
                v = json.load(f).get(key, {}).get("traffic_bytes_per_launch")
            if v is not None:
                return v, f"profiles/{rnd}/bench_traffic.json"
        except Exception:
            pass
    return None, None


WORKLOADS = {
    # name: (hidden, ffn, heads, kv heads, blocks of the whole model, BASELINE.json config it stands for)
    "llama2-7b": (4096, 11008, 32, 32, 32, "Llama-2-7B GPTQ INT4 group_size=128 sym, 128 calib samples, 1 MI355X (configs[1], the headline)"),
    "llama2-70b": (8192, 28672, 64, 8, 80, "Llama-2-70B GPTQ INT4 g128, layers sharded across 8xMI355X (configs[4]); one block per step here"),
}
WORKLOAD = "llama2-7b"


def build_model(n_layers, device):
    from transformers import LlamaConfig, LlamaForCausalLM

    hidden, ffn, heads, kv, _, _ = WORKLOADS[WORKLOAD]
    cfg = LlamaConfig(
        hidden_size=hidden, intermediate_size=ffn, num_hidden_layers=n_layers, num_attention_heads=heads,
        num_key_value_heads=kv, vocab_size=32000, max_position_embeddings=4096, rms_norm_eps=1e-5,
        tie_word_embeddings=False,
    )
    torch.manual_seed(0)
    with torch.device(device):
        model = LlamaForCausalLM(cfg)
    model = model.to(torch.bfloat16)
    model.eval()
    model.config.use_cache = False
    return model


def bench_dequant_gemm(device, shapes, iters=20):
    from neural_compressor_amd.torch.algorithms.weight_only.modules import MI355XWeightOnlyLinear
    from neural_compressor_amd.torch.algorithms.weight_only.utility import quant_tensor

    res = []
    for (M, N, K) in shapes:
        torch.manual_seed(0)
        w = torch.randn(N, K, device=device) * 0.02
        iw, sc, _ = quant_tensor(w, bits=4, group_size=128, scheme="sym", return_int=True)
        m = MI355XWeightOnlyLinear(K, N, bits=4, group_size=128, device=device)
        m.pack(iw, sc, None, None)
        m.bias = None
        del w, iw
        x = torch.randn(M, K, device=device, dtype=torch.bfloat16)
        # steady state: the chip needs some tens of milliseconds of matrix work to settle its clocks after the (VALU-only) packing
        # above -- the first large shape of the list read 10 % low with 20 warm-up calls (2 ms); median of three timed batches
        for _ in range(150 if M >= 1024 else 3):
            m(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        batches = []
        for _ in range(3):
            e0.record()
            for _ in range(iters):
                m(x)
            e1.record()
            torch.cuda.synchronize()
            batches.append(e0.elapsed_time(e1) / iters)
        ms = sorted(batches)[1]
        dense_ms, unfused_ms = None, None
        if M >= 128:
            # the library's DENSE bf16 GEMM (hipBLASLt through torch) on the same shape, in the same process and thermal state, timed in
            # alternation with the fused kernel (a reference point, not a product path).  M >= 1024 also times the UN-FUSED route the
            # fusion competes with: inc_woq_dequant (recover()) into a dense bf16 weight, then the library GEMM on it, per call.
            from neural_compressor_amd import ops as _ops

            wd = torch.randn(N, K, device=device, dtype=torch.bfloat16) * 0.02
            for _ in range(20):
                torch.nn.functional.linear(x, wd)

            def unfused():
                w_ = _ops.woq_dequant(m.qweight, m.scales, m.qzeros, None, N, K, 128, 4, torch.bfloat16)
                return torch.nn.functional.linear(x, w_)

            legs = [("fused", lambda: m(x)), ("dense", lambda: torch.nn.functional.linear(x, wd))] + ([("unfused", unfused)] if M >= 1024 else [])
            got = {k: [] for k, _ in legs}
            for _ in range(3):
                for k, fn in legs:
                    e0.record()
                    for _ in range(iters):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    got[k].append(e0.elapsed_time(e1) / iters)
            ms = min(ms, sorted(got["fused"])[1])
            dense_ms = sorted(got["dense"])[1]
            if "unfused" in got:
                unfused_ms = sorted(got["unfused"])[1]
            del wd
        graph_ms = None
        if M <= 512:
            # launch-bound regime: the same `iters` calls captured once in a hipGraph and replayed -- what a decode loop does
            # (no Python / ctypes time between the kernels); `ms` above is the eager per-call time including the host side
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(3):
                        m(x)
                torch.cuda.current_stream().wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    for _ in range(iters):
                        m(x)
                graph.replay()
                torch.cuda.synchronize()
                reps = 5
                e0.record()
                for _ in range(reps):
                    graph.replay()
                e1.record()
                torch.cuda.synchronize()
                graph_ms = e0.elapsed_time(e1) / (reps * iters)
                del graph
            except Exception as e:  # pragma: no cover - report, never fake
                graph_ms = None
                print(f"[bench] hipGraph timing of M={M} failed: {type(e).__name__}: {e}", file=sys.stderr)
        cold_ms, ring_n, ring_mib = None, 0, 0.0
        if M <= 64:
            # HBM-bound rows, measured honestly: the replay above re-reads ONE 8-22 MiB weight tensor, which lives in the L2 /
            # Infinity Cache (256 MiB) after the first call.  Here the graph walks a RING of modules with distinct packed weights
            # (>= 512 MiB together, about one model's worth of layers), so every call's weight bytes come from HBM -- what a decode
            # step over a whole model sees.  `graph_ms` stays as the cache-resident figure.
            try:
                per = m.qweight.numel() * 4 + m.scales.numel() * 2 + m.qzeros.numel() * 4
                ring_n = int(min(128, max(2, -(-(512 << 20) // per))))
                ring = [m] + [copy.deepcopy(m) for _ in range(ring_n - 1)]
                ring_mib = ring_n * per / 2**20
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(2):
                        for mod in ring:
                            mod(x)
                torch.cuda.current_stream().wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    for mod in ring:
                        mod(x)
                graph.replay()
                torch.cuda.synchronize()
                reps = 5
                e0.record()
                for _ in range(reps):
                    graph.replay()
                e1.record()
                torch.cuda.synchronize()
                cold_ms = e0.elapsed_time(e1) / (reps * ring_n)
                del graph, ring
            except Exception as e:  # pragma: no cover - report, never fake
                cold_ms = None
                print(f"[bench] cold-weight ring timing of M={M} failed: {type(e).__name__}: {e}", file=sys.stderr)
        flops = 2.0 * M * N * K
        G = K // 128
        bytes_ = N * K / 2 + G * N * 2 + G * (N // 8) * 4 + M * K * 2 + M * N * 2
        tflops = flops / ms / 1e9
        gbs = bytes_ / ms / 1e6
        bound = "mfma" if M >= 128 else "hbm"
        row = dict(M=M, N=N, K=K, ms=round(ms, 4), tflops=round(tflops, 2), gbs=round(gbs, 1), bound=bound,
                   frac=round(tflops / BF16_MFMA_PEAK_TFLOPS if bound == "mfma" else gbs / HBM_PEAK_GBS, 4))
        if dense_ms is not None:
            row.update(hipblaslt_dense_bf16_ms=round(dense_ms, 4), hipblaslt_dense_bf16_tflops=round(flops / dense_ms / 1e9, 2),
                       vs_hipblaslt_dense=round(dense_ms / ms, 4))
        if unfused_ms is not None:
            # recover() + library GEMM per call against the fused kernel (> 1: the fusion wins time, not only memory)
            row.update(unfused_dequant_plus_hipblaslt_ms=round(unfused_ms, 4), fused_vs_unfused=round(unfused_ms / ms, 4))
        if graph_ms is not None:
            row.update(graph_ms=round(graph_ms, 4), graph_tflops=round(flops / graph_ms / 1e9, 2), graph_gbs=round(bytes_ / graph_ms / 1e6, 1),
                       graph_frac=round((flops / graph_ms / 1e9) / BF16_MFMA_PEAK_TFLOPS if bound == "mfma" else (bytes_ / graph_ms / 1e6) / HBM_PEAK_GBS, 4))
        if cold_ms is not None:
            # the HBM fraction to quote for an HBM-bound row is the COLD one (`cold_graph_frac`: distinct weights per call); `frac` keeps
            # its meaning of rounds 1-4 (eager single call, host side included) and `graph_frac` is the cache-resident replay
            row.update(cold_graph_ms=round(cold_ms, 4), cold_graph_gbs=round(bytes_ / cold_ms / 1e6, 1),
                       cold_graph_frac=round((bytes_ / cold_ms / 1e6) / HBM_PEAK_GBS, 4), cold_ring_modules=ring_n, cold_ring_mib=round(ring_mib, 1))
        res.append(row)
    return res


def bench_int8_rows(device):
    """Weight-only INT8 (BASELINE config #1's packed format, bits = 8, per-channel scales) forward at the BASELINE layer sizes: decode (M = 1,
    cold ring of distinct modules >= 512 MiB, bytes = the packed bytes) and prefill (M = 4096, TFLOP/s)."""
    from neural_compressor_amd.torch.algorithms.weight_only.modules import MI355XWeightOnlyLinear
    from neural_compressor_amd.torch.algorithms.weight_only.utility import quant_tensor

    res = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for N, K in ((4096, 4096), (11008, 4096)):
        torch.manual_seed(0)
        w = torch.randn(N, K, device=device) * 0.02
        iw, sc, _ = quant_tensor(w, bits=8, group_size=-1, scheme="sym", return_int=True)
        m = MI355XWeightOnlyLinear(K, N, bits=8, group_size=-1, device=device)
        m.pack(iw, sc, None, None)
        m.bias = None
        del w, iw
        per = m.qweight.numel() * 4 + m.scales.numel() * 2 + m.qzeros.numel() * 4
        row = dict(N=N, K=K, bits=8, group_size=-1, packed_bytes=int(per))
        try:
            ring_n = int(max(2, -(-(512 << 20) // per)))
            ring = [m] + [copy.deepcopy(m) for _ in range(ring_n - 1)]
            x1 = torch.randn(1, K, device=device, dtype=torch.bfloat16)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    for mod in ring:
                        mod(x1)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for mod in ring:
                    mod(x1)
            graph.replay()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                graph.replay()
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / (5 * ring_n)
            row.update(decode_cold_ms=round(t, 5), decode_cold_gbs=round(per / t / 1e6, 1), decode_cold_hbm_frac=round(per / t / 1e6 / HBM_PEAK_GBS, 4))
            del graph, ring
        except Exception as e:  # pragma: no cover - report, never fake
            print(f"[bench] INT8 decode timing failed: {type(e).__name__}: {e}", file=sys.stderr)
        x = torch.randn(4096, K, device=device, dtype=torch.bfloat16)
        for _ in range(30):
            m(x)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            m(x)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20
        row.update(prefill_M=4096, prefill_ms=round(t, 4), prefill_tflops=round(2.0 * 4096 * N * K / t / 1e9, 1),
                   prefill_frac=round(2.0 * 4096 * N * K / t / 1e9 / BF16_MFMA_PEAK_TFLOPS, 4))
        res.append(row)
        del m, x
        torch.cuda.empty_cache()
    return res


def bench_gemv_groups(device):
    """Decode (M = 1) of the module groups that share x -- q / k / v and gate / up of a Llama-2-7B block -- as ONE launch per group
    (inc_woq_gemm_multi via woq_linear_group), measured COLD: a hipGraph over a ring of groups with distinct packed weights (>= 512 MiB),
    next to the same ring issued as single calls.  Bytes = the packed bytes of all modules of the group (SURVEY 8(d) per module)."""
    from neural_compressor_amd.torch.algorithms.weight_only.modules import MI355XWeightOnlyLinear, woq_linear_group
    from neural_compressor_amd.torch.algorithms.weight_only.utility import quant_tensor

    res = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for tag, n, N, K in (("qkv_3x4096x4096", 3, 4096, 4096), ("gateup_2x11008x4096", 2, 11008, 4096)):
        torch.manual_seed(0)
        w = torch.randn(N, K, device=device) * 0.02
        iw, sc, _ = quant_tensor(w, bits=4, group_size=128, scheme="sym", return_int=True)
        m = MI355XWeightOnlyLinear(K, N, bits=4, group_size=128, device=device)
        m.pack(iw, sc, None, None)
        m.bias = None
        del w, iw
        G = K // 128
        per = N * K / 2 + G * N * 2 + G * (N // 8) * 4
        bytes_ = n * per + K * 2 + n * N * 2
        ring_n = int(max(2, -(-(512 << 20) // int(n * per))))
        ring = [[copy.deepcopy(m) for _ in range(n)] for _ in range(ring_n)]
        x = torch.randn(1, K, device=device, dtype=torch.bfloat16)
        row = dict(group=tag, modules=n, M=1, N=N, K=K, bytes=int(bytes_), cold_ring_groups=ring_n, cold_ring_mib=round(ring_n * n * per / 2**20, 1))
        for key, fn in (("cold_graph_ms", lambda grp: woq_linear_group(x, grp)), ("single_calls_cold_graph_ms", lambda grp: [mm(x) for mm in grp])):
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(2):
                        for grp in ring:
                            fn(grp)
                torch.cuda.current_stream().wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                keep = []
                with torch.cuda.graph(graph):
                    for grp in ring:
                        keep.append(fn(grp))
                graph.replay()
                torch.cuda.synchronize()
                reps = 5
                e0.record()
                for _ in range(reps):
                    graph.replay()
                e1.record()
                torch.cuda.synchronize()
                row[key] = round(e0.elapsed_time(e1) / (reps * ring_n), 5)
                del graph, keep
            except Exception as e:  # pragma: no cover - report, never fake
                row[key] = None
                print(f"[bench] group decode timing of {tag} failed: {type(e).__name__}: {e}", file=sys.stderr)
        if row.get("cold_graph_ms"):
            row["cold_graph_gbs"] = round(bytes_ / row["cold_graph_ms"] / 1e6, 1)
            row["cold_graph_frac"] = round(bytes_ / row["cold_graph_ms"] / 1e6 / HBM_PEAK_GBS, 4)
        if row.get("single_calls_cold_graph_ms"):
            row["single_calls_cold_graph_frac"] = round(bytes_ / row["single_calls_cold_graph_ms"] / 1e6 / HBM_PEAK_GBS, 4)
        res.append(row)
        del ring, m
        torch.cuda.empty_cache()
    return res


def bench_w8a8_gemm(device, shapes, iters=20):
    """BASELINE config #4's kernel (SmoothQuant W8A8, Llama-2-13B shapes): the whole W8A8Linear forward = activation
    quantisation (HBM-bound) + INT8 MFMA GEMM; `gemm_ms` is the GEMM alone."""
    from neural_compressor_amd import ops
    from neural_compressor_amd.torch.algorithms.smooth_quant import W8A8Linear

    res = []
    for (M, N, K) in shapes:
        torch.manual_seed(0)
        lin = torch.nn.Linear(K, N, bias=False, device=device, dtype=torch.bfloat16)
        lin.weight.data.normal_(0, 0.02)
        x = torch.randn(M, K, device=device, dtype=torch.bfloat16)
        m = W8A8Linear.from_float(lin, x.float().min(dim=0)[0], x.float().max(dim=0)[0], device=device)
        del lin
        xq = ops.sq_quant_act(x, None, m.act_scale.item(), m.act_zp.item(), m.kp)
        out = {}
        for tag, fn in (("fwd", lambda: m(x)), ("gemm", lambda: ops.w8a8_gemm(xq, m.qweight, m.alpha, m.corr, None, torch.bfloat16))):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            out[tag] = e0.elapsed_time(e1) / iters
        ops_ = 2.0 * M * N * K
        res.append(dict(M=M, N=N, K=K, fwd_ms=round(out["fwd"], 4), gemm_ms=round(out["gemm"], 4),
                        gemm_tops=round(ops_ / out["gemm"] / 1e9, 1), fwd_tops=round(ops_ / out["fwd"] / 1e9, 1)))
    return res


def _cpu_baseline_worker(threads):
    """Runs in a child process: times the oracle (CPU restatement of the reference) on a bounded sample of the workload:
    every distinct kernel shape of one transformer block ONCE (nothing is scaled between shapes)."""
    from oracle import woq_oracle as O

    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(0)
    out = {}
    hess = {}
    for K in (4096, 11008):
        x = torch.randn(1, 2048, K, generator=g)
        H, n = torch.zeros(K, K), 0
        H, n = O.gptq_add_batch(H, n, x)  # untimed warm-up (thread pool, page faults)
        reps = 3 if K == 4096 else 2
        t0 = time.time()
        for _ in range(reps):
            H, n = O.gptq_add_batch(H, n, x)
        out[f"t_add_{K}"] = (time.time() - t0) / reps
        hess[K] = H
    for N, K in ((4096, 4096), (11008, 4096), (4096, 11008)):
        W = torch.randn(N, K, generator=g) * 0.02
        t0 = time.time()
        O.gptq_fasterquant(W, hess[K], bits=4, sym=True, blocksize=128, percdamp=0.01, groupsize=128)
        out[f"t_fq_{N}x{K}"] = time.time() - t0
    # one fp32 block forward of one 2048-token sample (the reference runs the float block twice per sample, gptq.py:660-760)
    from transformers import LlamaConfig
    from transformers.models.llama.modeling_llama import LlamaDecoderLayer, LlamaRotaryEmbedding

    cfg = LlamaConfig(hidden_size=4096, intermediate_size=11008, num_hidden_layers=1, num_attention_heads=32, num_key_value_heads=32,
                      vocab_size=32000, max_position_embeddings=4096)
    layer = LlamaDecoderLayer(cfg, 0).float().eval()
    rot = LlamaRotaryEmbedding(config=cfg)
    h = torch.randn(1, 2048, 4096, generator=g) * 0.1
    pos = torch.arange(2048).unsqueeze(0)
    with torch.no_grad():
        pe = rot(h, pos)
        layer(h, position_embeddings=pe, position_ids=pos)  # warm-up
        t0 = time.time()
        layer(h, position_embeddings=pe, position_ids=pos)
        out["t_fwd"] = time.time() - t0
    # pack / unpack / recover / quant_tensor of one 4096 x 4096 g128 layer (per-layer CPU figures)
    import numpy as np

    w = torch.randn(4096, 4096, generator=g) * 0.02
    t0 = time.time()
    iw, sc, _ = O.quant_tensor(w.clone(), bits=4, group_size=128, scheme="sym", return_int=True)
    out["t_quant_tensor_4096"] = time.time() - t0
    iwn, scn = iw.numpy().astype(np.int32), sc.numpy()
    t0 = time.time()
    qw, qz, s16 = O.woq_pack_optimum(iwn, scn, None, 4)
    out["t_pack_4096"] = time.time() - t0
    t0 = time.time()
    O.woq_unpack_optimum(qw, qz, 4096, 4096, 32, 4)
    out["t_unpack_4096"] = time.time() - t0
    t0 = time.time()
    rec = O.woq_recover(qw, s16, qz, 4096, 4096, 4, 128)
    out["t_recover_4096"] = time.time() - t0
    # the reference's forward on this host (modules.py:594-610): recover() once, then F.linear on the CACHED fp32 weight -- the CPU figure
    # SURVEY 8(d) asks for beside the fused dequant-GEMM.  One 2048-token calibration sample per shape (bounded: < 1 s each).
    wrec = torch.from_numpy(np.asarray(rec)).float()
    # M in {1, 16, 512, 4096} at 4096 x 4096, and the prefill row (M = 4096) of the two 11008 shapes -- each beside the GPU row
    for (Nl, Kl, wl, Ms) in ((4096, 4096, wrec, (1, 16, 512, 4096)), (11008, 4096, torch.randn(11008, 4096, generator=g) * 0.02, (4096,)),
                             (4096, 11008, torch.randn(4096, 11008, generator=g) * 0.02, (4096,))):
        for Ml in Ms:
            xl = torch.randn(Ml, Kl, generator=g)
            torch.nn.functional.linear(xl, wl)  # warm-up
            reps = 3 if Ml >= 512 else 20
            t0 = time.time()
            for _ in range(reps):
                torch.nn.functional.linear(xl, wl)
            out[f"t_linear_{Ml}x{Nl}x{Kl}"] = (time.time() - t0) / reps
    # BASELINE config #1 on these cores: RTN INT8 per-channel over every Linear of an OPT-125M-shaped stack (72 modules; rtn.py:68 ->
    # quant_tensor + pack), the oracle's restatement of the reference's CPU adaptor
    from tests.model_zoo import opt125m_like

    model = opt125m_like()
    t0 = time.time()
    nmod = 0
    for name, mod in model.named_modules():
        if isinstance(mod, torch.nn.Linear) and name != "lm_head":
            iw_, sc_, _ = O.quant_tensor(mod.weight.data.clone(), bits=8, group_size=-1, scheme="sym", return_int=True)
            O.woq_pack_optimum(iw_.numpy().astype(np.int32), sc_.numpy(), None, 8)
            nmod += 1
    out["t_rtn_config1"] = time.time() - t0
    out["rtn_config1_modules"] = nmod
    del model
    # pack / unpack / recover at 11008 x 4096 (the per_layer rows that had no CPU figure)
    w2 = torch.randn(11008, 4096, generator=g) * 0.02
    iw2, sc2, _ = O.quant_tensor(w2, bits=4, group_size=128, scheme="sym", return_int=True)
    iw2n, sc2n = iw2.numpy().astype(np.int32), sc2.numpy()
    t0 = time.time()
    qw2, qz2, s2 = O.woq_pack_optimum(iw2n, sc2n, None, 4)
    out["t_pack_11008"] = time.time() - t0
    t0 = time.time()
    O.woq_unpack_optimum(qw2, qz2, 11008, 4096, 32, 4)
    out["t_unpack_11008"] = time.time() - t0
    t0 = time.time()
    O.woq_recover(qw2, s2, qz2, 11008, 4096, 4, 128)
    out["t_recover_11008"] = time.time() - t0
    print(json.dumps(out), flush=True)


def bench_ceilings(device):
    """Measured ceilings of THIS chip in THIS run: a stream triad (a <- b + s c, fp32, 3 x 1 GiB) and a bare bf16 MFMA
    loop on random operands, next to the datasheet figures the `frac` fields are priced against."""
    from neural_compressor_amd import ops

    n = 256 * 1024 * 1024
    a = torch.empty(n, dtype=torch.float32, device=device)
    b = torch.randn(n, dtype=torch.float32, device=device)
    c = torch.randn(n, dtype=torch.float32, device=device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        ops.probe_hbm_triad(a, b, c, 2.0)
    e0.record()
    for _ in range(5):
        ops.probe_hbm_triad(a, b, c, 2.0)
    e1.record()
    torch.cuda.synchronize()
    triad = 5 * 12.0 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9
    # copy ceiling: 1 GiB read + 1 GiB written, every variant of the probe (loads in flight per lane x non-temporal x grid shape), best kept
    copies = {}
    for variant in range(16):
        ops.probe_hbm_copy(a, b, variant)
        e0.record()
        for _ in range(3):
            ops.probe_hbm_copy(a, b, variant)
        e1.record()
        torch.cuda.synchronize()
        copies[variant] = round(3 * 8.0 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
    best = max(copies, key=copies.get)
    del a, b, c
    src = torch.randn(64 * 1024, dtype=torch.bfloat16, device=device)
    sink = torch.zeros(4096 * 256, dtype=torch.float32, device=device)
    flops = ops.probe_mfma_bf16(src, sink, 4096, 2000)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(3):
        flops = ops.probe_mfma_bf16(src, sink, 4096, 2000)
    e1.record()
    torch.cuda.synchronize()
    mfma = 3 * flops / (e0.elapsed_time(e1) * 1e-3) / 1e12
    torch.cuda.empty_cache()
    return dict(hbm_stream_triad_gbs=round(triad, 1), hbm_copy_gbs=copies[best], hbm_copy_variant=best, hbm_copy_all_variants=copies,
                hbm_spec_gbs=HBM_PEAK_GBS, bf16_mfma_loop_tflops=round(mfma, 1),
                bf16_mfma_spec_tflops=BF16_MFMA_PEAK_TFLOPS,
                note="measured in this run: triad = inc_probe_hbm_triad over 3 x 1 GiB fp32 (12 n bytes per call); copy = inc_probe_hbm_copy, 1 GiB "
                     "read + 1 GiB written, best of its 16 variants (variant = 4 log2(loads in flight per lane) + 2 non-temporal + 1 flat grid); MFMA loop = "
                     "inc_probe_mfma_bf16, 16 waves per CU of back-to-back v_mfma_f32_32x32x16_bf16 on random operands (the chip "
                     "clocks to its power budget: zero operands would read higher); every `frac` in this line is against the spec figures")


def bench_per_layer(device, cpu):
    """SURVEY 8(d) "report per-layer": the calls of one Llama-2-7B Linear on the GPU, each alone on the chip, next to the same
    calls of the CPU baseline (`cpu` = the cpu_baseline object of this run, or None)."""
    from neural_compressor_amd import ops
    from neural_compressor_amd.torch.algorithms.weight_only import gptq as G
    from neural_compressor_amd.torch.algorithms.weight_only.modules import MI355XWeightOnlyLinear
    from neural_compressor_amd.torch.algorithms.weight_only.utility import quant_tensor

    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(fn, reps=3, warm=1):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps

    def c(key):
        return None if not cpu or cpu.get(key) is None else cpu[key]

    out = {}
    torch.manual_seed(0)
    Hs = {}
    for K in (4096, 11008):
        TOK = G.HessianAccumulator.STAGE_TOKENS  # one launch of the driver: a stage of tokens (32 samples of 2048)
        x = torch.randn(TOK, K, device=device, dtype=torch.bfloat16)
        H = torch.zeros(K, K, device=device)
        t = timed(lambda: ops.gptq_hessian_accum(H, x, 0.5, 0.5), reps=5, warm=2)
        out[f"hessian_K{K}"] = dict(gpu_s_per_sample=round(t / (TOK / 2048), 6), cpu_s_per_sample=c(f"t_add_{K}"), tokens_per_launch=TOK,
                                    tflops=round(2.0 * TOK * K * K / t / 1e12, 1), frac=round(2.0 * TOK * K * K / t / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4))
        acc = G.HessianAccumulator(K, device)
        acc.add_batch(x.view(TOK // 2048, 2048, K))
        Hs[K] = acc
        del x, H
    # the three solves of a block, set up side by side and timed INTERLEAVED and repeated (profiles/NOTES.md: a single cold-started
    # timing of a latency chain reads the chip's clock state, not the kernel): min and median of REPS rounds over the shapes
    REPS = 5
    setups = []
    for N, K in ((4096, 4096), (11008, 4096), (4096, 11008)):
        layer = torch.nn.Linear(K, N, bias=False, device=device, dtype=torch.bfloat16)
        layer.weight.data.normal_(0, 0.02)
        W0 = layer.weight.data.clone()
        Hs[K].flush()

        # (a) the whole solve: damp + inverse-Cholesky factor + column loop (GPTQ.fasterquant), a fresh accumulator copy each time
        def solve(layer=layer, W0=W0, K=K):
            a = G.HessianAccumulator(K, device)
            a.H, a._n = Hs[K].H.clone(), Hs[K]._n
            gq = G.GPTQ(layer, device=device, accumulator=a)
            gq.configure(dict(bits=4, sym=True, dtype="int"))
            gq.fasterquant(W0, blocksize=128, percdamp=0.01, groupsize=128)

        # (b) the column loop alone, with the factor in hand (inc_gptq_quantize_layer)
        a = G.HessianAccumulator(K, device)
        a.H, a._n = Hs[K].H.clone(), Hs[K]._n
        Hinv, dead, _ = a.inverse_factor(0.01, False)
        scale = torch.empty(N, K // 128, device=device)
        zero = torch.empty_like(scale)
        codes = torch.empty(N, K, dtype=torch.uint8, device=device)
        Q = torch.empty(N, K, dtype=torch.bfloat16, device=device)
        err = torch.empty(2, N, 128, device=device)
        side = G._lookahead_stream(device)
        w32s = [ops.gptq_prepare_weight(W0, dead) for _ in range(REPS + 1)]  # the loop consumes its working copy: one per timed call

        def loop(i, Hinv=Hinv, scale=scale, zero=zero, codes=codes, Q=Q, err=err, side=side, w32s=w32s):
            ops.gptq_quantize_layer(w32s[i], Hinv, scale, zero, None, None, codes, Q, err, 128, 128, 128, 4, True, ops.GPTQ_DYNAMIC_GROUPS, aux_stream=side)

        setups.append((N, K, solve, loop))
    t_solve = {i: [] for i in range(len(setups))}
    t_loop = {i: [] for i in range(len(setups))}
    for rep in range(REPS + 1):  # round 0 is the warm-up
        for i, (N, K, solve, loop) in enumerate(setups):
            torch.cuda.synchronize()
            e0.record()
            loop(rep)
            e1.record()
            torch.cuda.synchronize()
            if rep:
                t_loop[i].append(e0.elapsed_time(e1) * 1e-3)
            if rep < 3:
                e0.record()
                solve()
                e1.record()
                torch.cuda.synchronize()
                if rep:
                    t_solve[i].append(e0.elapsed_time(e1) * 1e-3)
    for i, (N, K, _, _) in enumerate(setups):
        lo, med = min(t_loop[i]), sorted(t_loop[i])[len(t_loop[i]) // 2]
        out[f"fasterquant_{N}x{K}"] = dict(gpu_s=round(min(t_solve[i]), 5), cpu_s=c(f"t_fq_{N}x{K}"), column_loop_s_min=round(lo, 5), column_loop_s_median=round(med, 5),
                                           column_loop_us_per_column_min=round(lo * 1e6 / K, 3), column_loop_us_per_column_median=round(med * 1e6 / K, 3),
                                           column_loop_hbm_frac=round(2.0 * N * K * 4 / med / 1e9 / HBM_PEAK_GBS, 4),
                                           # the loop's flops are its trailing update (gptq.py:1304), N * 128 * (K - i2) * 2 per block on the exact-fp32 MFMA
                                           trailing_update_tflops=round(sum(2.0 * N * 128 * (K - i2) for i2 in range(128, K, 128)) / med / 1e12, 1),
                                           trailing_update_f32_mfma_frac=round(sum(2.0 * N * 128 * (K - i2) for i2 in range(128, K, 128)) / med / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                                           reps=REPS)
    del setups
    torch.cuda.empty_cache()
    # pack / unpack / recover / quant_tensor of one 4096 x 4096 g128 layer (bytes: SURVEY 8(d))
    N = K = 4096
    w = torch.randn(N, K, device=device) * 0.02
    t_q = timed(lambda: quant_tensor(w.clone(), bits=4, group_size=128, scheme="sym", return_int=True))
    t_q -= timed(lambda: w.clone())
    iw, sc, _ = quant_tensor(w.clone(), bits=4, group_size=128, scheme="sym", return_int=True)
    m = MI355XWeightOnlyLinear(K, N, bits=4, group_size=128, device=device)

    def timed_gpu(fn, calls=10, reps=5):
        """GPU time per call: `calls` calls captured in one hipGraph and replayed (the module methods spend 30-100 us of Python per
        call, more than their kernels: eager timing measured the host)."""
        try:
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
            return e0.elapsed_time(e1) * 1e-3 / (reps * calls)
        except Exception as e:  # pragma: no cover - report, never fake
            print(f"[bench] hipGraph timing failed ({type(e).__name__}: {e}); eager timing instead", file=sys.stderr)
            return timed(fn, reps=10, warm=2)

    def timed_gpu_ring(make, n, reps=5):
        """GPU time per call with the bytes coming from HBM: `n` calls on `n` DISTINCT argument sets (make(i) -> callable), their
        outputs kept alive so that every call also writes fresh memory, captured in one hipGraph and replayed.  (timed_gpu replays ONE
        <= 180 MB working set, which stays in the 256 MiB Infinity Cache: its GB/s can exceed what HBM delivers.)"""
        try:
            fns = [make(i) for i in range(n)]
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for f in fns:
                    f()
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            keep = []
            with torch.cuda.graph(g):
                for f in fns:
                    keep.append(f())
            g.replay()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) * 1e-3 / (reps * n)
            del g, keep, fns
            torch.cuda.empty_cache()
            return t
        except Exception as e:  # pragma: no cover - report, never fake
            print(f"[bench] cold ring timing failed ({type(e).__name__}: {e})", file=sys.stderr)
            torch.cuda.empty_cache()
            return None

    def ring_of(mod, n):
        return [mod] + [copy.deepcopy(mod) for _ in range(n - 1)]

    t_p = timed_gpu(lambda: m.pack(iw, sc, None, None))
    t_u = timed_gpu(lambda: m.unpack())
    t_r = timed_gpu(lambda: m.recover())
    # cold (HBM) figures: 8 x (64 MiB of int32 codes + 8.4 MiB packed) for pack, 16 x (8.4 MiB packed + 32 MiB out) for unpack / recover
    iws = [iw] + [iw.clone() for _ in range(7)]
    mods8 = ring_of(m, 8)
    cold = {}
    cold["pack_4096x4096"] = timed_gpu_ring(lambda i: (lambda: mods8[i].pack(iws[i], sc, None, None)), 8)
    del iws, mods8
    mods16 = ring_of(m, 16)
    cold["unpack_4096x4096"] = timed_gpu_ring(lambda i: (lambda: mods16[i].unpack()), 16)
    cold["recover_4096x4096"] = timed_gpu_ring(lambda i: (lambda: mods16[i].recover()), 16)
    del mods16
    ws = [w] + [w.clone() for _ in range(7)]
    cold["quant_tensor_4096x4096"] = timed_gpu_ring(lambda i: (lambda: quant_tensor(ws[i], bits=4, group_size=128, scheme="sym", return_int=True)), 8)
    del ws
    # the same three at 11008 x 4096 (2.7 x the bytes: the fixed cost of a launch weighs less)
    N2 = 11008
    w2 = torch.randn(N2, K, device=device) * 0.02
    iw2, sc2, _ = quant_tensor(w2, bits=4, group_size=128, scheme="sym", return_int=True)
    m2 = MI355XWeightOnlyLinear(K, N2, bits=4, group_size=128, device=device)
    for key, fn, byts in (("pack_11008x4096", lambda: m2.pack(iw2, sc2, None, None), 4.0 * N2 * K + N2 * K / 2),
                          ("unpack_11008x4096", lambda: m2.unpack(), N2 * K / 2 + 2.0 * N2 * K),
                          ("recover_11008x4096", lambda: m2.recover(), N2 * K / 2 + 2.0 * N2 * K)):
        t = timed_gpu(fn)
        out[key] = dict(gpu_s=round(t, 6), cpu_s=c("t_" + key.split("_")[0] + "_11008"), gbs=round(byts / t / 1e9, 1), hbm_frac=round(byts / t / 1e9 / HBM_PEAK_GBS, 4))
    iw2s = [iw2] + [iw2.clone() for _ in range(3)]
    m2s4 = ring_of(m2, 4)
    cold["pack_11008x4096"] = timed_gpu_ring(lambda i: (lambda: m2s4[i].pack(iw2s[i], sc2, None, None)), 4)
    del iw2s, m2s4
    m2s6 = ring_of(m2, 6)
    cold["unpack_11008x4096"] = timed_gpu_ring(lambda i: (lambda: m2s6[i].unpack()), 6)
    cold["recover_11008x4096"] = timed_gpu_ring(lambda i: (lambda: m2s6[i].recover()), 6)
    del m2s6
    del w2, iw2, sc2, m2
    for key, t, byts, ck in (("quant_tensor_4096x4096", t_q, 2.0 * N * K * 4, "t_quant_tensor_4096"), ("pack_4096x4096", t_p, 4.0 * N * K + N * K / 2, "t_pack_4096"),
                             ("unpack_4096x4096", t_u, N * K / 2 + 2.0 * N * K, "t_unpack_4096"), ("recover_4096x4096", t_r, N * K / 2 + 2.0 * N * K, "t_recover_4096")):
        out[key] = dict(gpu_s=round(t, 6), cpu_s=c(ck), gbs=round(byts / t / 1e9, 1), hbm_frac=round(byts / t / 1e9 / HBM_PEAK_GBS, 4))
    # `gpu_s` / `gbs` / `hbm_frac` above replay ONE working set (cache-resident: `cache_resident_*`); the HBM figures are the cold ones
    byts_of = {"quant_tensor_4096x4096": 2.0 * N * K * 4, "pack_4096x4096": 4.0 * N * K + N * K / 2, "unpack_4096x4096": N * K / 2 + 2.0 * N * K,
               "recover_4096x4096": N * K / 2 + 2.0 * N * K, "pack_11008x4096": 4.0 * N2 * K + N2 * K / 2, "unpack_11008x4096": N2 * K / 2 + 2.0 * N2 * K,
               "recover_11008x4096": N2 * K / 2 + 2.0 * N2 * K}
    for key, t in cold.items():
        row = out[key]
        row["cache_resident_gpu_s"], row["cache_resident_gbs"], row["cache_resident_hbm_frac"] = row["gpu_s"], row["gbs"], row["hbm_frac"]
        if t is not None:
            row.update(gpu_s=round(t, 6), gbs=round(byts_of[key] / t / 1e9, 1), hbm_frac=round(byts_of[key] / t / 1e9 / HBM_PEAK_GBS, 4),
                       basis="hipGraph replay over a ring of distinct tensors (>= 512 MiB): bytes come from HBM")
        else:
            row["basis"] = "cache-resident replay only (the cold ring failed)"
    torch.cuda.empty_cache()
    return out


def cpu_baseline(timeout_s=420):
    """The oracle (a restatement of the reference's CPU arithmetic) timed on this host's cores on a bounded sample of the same
    workload -- one call of every distinct shape of a Llama-2-7B block: add_batch [1,2048,4096] and [1,2048,11008], fasterquant
    4096x4096, 11008x4096 and 4096x11008, one fp32 block forward of one sample -- multiplied out to the 32-block job (the full
    CPU run is ~2 h).  Runs in a child process under a hard timeout so that a host with an unusual CPU
    quota can never stall the bench line."""
    import subprocess

    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, 32))  # torch's intra-op pool stops scaling on these shapes well before 32 threads
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="")
    code = f"import sys; sys.path.insert(0, {ROOT!r}); import bench; bench._cpu_baseline_worker({threads})"
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout_s, env=env, cwd=ROOT)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        m = json.loads(line)
    except Exception as e:  # timeout / crash: report it, never fake a number
        return dict(value=None, unit="s", cores=threads, kind="port", sample=f"CPU baseline failed or timed out after {timeout_s}s: {type(e).__name__}")
    # per block as the reference does it (7 separate Hessians, gptq.py:670-688: 6 inputs of K=4096, 1 of K=11008; 7 solves;
    # the block forward twice per sample): counts only, every term measured
    hess = 128 * (6 * m["t_add_4096"] + m["t_add_11008"])
    solve = 4 * m["t_fq_4096x4096"] + 2 * m["t_fq_11008x4096"] + m["t_fq_4096x11008"]
    fwd = 2 * 128 * m["t_fwd"]
    per_block = hess + solve + fwd
    return dict(
        value=round(32 * per_block, 1), unit="s", cores=threads, kind="port",
        sample=(f"oracle (CPU restatement of the reference) on this host, {threads} threads, one call per distinct shape: GPTQ.add_batch "
                f"[1,2048,4096] = {m['t_add_4096']:.3f} s, [1,2048,11008] = {m['t_add_11008']:.3f} s; GPTQ.fasterquant g128 4096x4096 = "
                f"{m['t_fq_4096x4096']:.2f} s, 11008x4096 = {m['t_fq_11008x4096']:.2f} s, 4096x11008 = {m['t_fq_4096x11008']:.2f} s; fp32 block "
                f"forward of one 2048-token sample = {m['t_fwd']:.2f} s; x (128 samples x 7 Hessians + 7 solves + 2 x 128 forwards) x 32 blocks"),
        per_block_s=dict(hessians=round(hess, 2), solves=round(solve, 2), forwards=round(fwd, 2)),
        # the reference's packed-module forward on these cores: fp32 F.linear on the cached recovered weight (modules.py:594-610)
        f_linear=[dict(M=m_, N=n_, K=k_, ms=round(m[f"t_linear_{m_}x{n_}x{k_}"] * 1e3, 3), tflops=round(2.0 * m_ * n_ * k_ / m[f"t_linear_{m_}x{n_}x{k_}"] / 1e12, 4))
                  for (m_, n_, k_) in ((1, 4096, 4096), (16, 4096, 4096), (512, 4096, 4096), (4096, 4096, 4096), (4096, 11008, 4096), (4096, 4096, 11008))
                  if f"t_linear_{m_}x{n_}x{k_}" in m],
        **{k: round(v, 4) for k, v in m.items()},
    )


def calib_ids(samples, seq):
    g = torch.Generator().manual_seed(1)
    return [torch.randint(0, 32000, (1, seq), generator=g) for _ in range(samples)]


def bench_e2e(device, args, rank, world, note):
    """The north_star's own quantity, not extrapolated: Llama-2-7B-shaped model (32 blocks), GPTQ INT4 g128 sym,
    `samples` x `seq` calibration tokens, wall-clock of prepare -> run_fn (embedding + capture of block-0 inputs) ->
    convert (32 blocks, packing included).  Model construction (random init) is outside, like loading a checkpoint."""
    from neural_compressor_amd import distributed as D
    from neural_compressor_amd.torch.algorithms.weight_only.modules import MI355XWeightOnlyLinear
    from neural_compressor_amd.torch.quantization import GPTQConfig, convert, prepare

    model = build_model(args.e2e_blocks, device)
    ids = calib_ids(args.samples, args.seq)
    mine = D.shard_samples(len(ids), rank, world) if world > 1 else range(len(ids))
    cfg = GPTQConfig(bits=4, group_size=128, use_sym=True, block_size=128, percdamp=0.01, act_order=False)
    torch.cuda.synchronize()
    if D.live():
        torch.distributed.barrier()
    t0 = time.perf_counter()
    with torch.no_grad():
        model = prepare(model, cfg)
        for j in mine:
            model(ids[j].to(device))
        t1 = time.perf_counter()
        model = convert(model)
    torch.cuda.synchronize()
    if D.live():
        torch.distributed.barrier()
    wall = D.barrier_max_time(time.perf_counter() - t0, device=device)
    packed = sum(isinstance(m, MI355XWeightOnlyLinear) for m in model.modules())
    note(f"e2e: {args.e2e_blocks} blocks in {wall:.2f}s ({packed} packed modules)")
    out = dict(wall_s=round(wall, 3), blocks=args.e2e_blocks, packed_modules=packed, prepare_and_capture_s=round(t1 - t0, 3),
               samples=args.samples, seq_len=args.seq, n_gpus=world,
               what="prepare -> run_fn -> convert of a Llama-2-7B-shaped model, wall-clock incl. capture and packing")
    del model
    torch.cuda.empty_cache()
    return out


def bench_awq_sq_e2e(device, note):
    """BASELINE configs #3 and #4 un-extrapolated (--e2e-configs; ~4 minutes, not part of the default line): a Llama-2-7B-shaped model
    (32 blocks) through prepare -> 128 x 2048 calibration tokens -> convert with AWQConfig(INT4 g128, auto-scale + auto-clip), and a
    Llama-2-13B-shaped model (40 blocks, hidden 5120, ffn 13824) through SmoothQuantConfig(alpha 0.5) with 32 x 2048 tokens; wall-clock."""
    from transformers import LlamaConfig, LlamaForCausalLM

    from neural_compressor_amd.torch.quantization import AWQConfig, SmoothQuantConfig, convert, prepare

    out = {}
    g = torch.Generator().manual_seed(1)
    for tag, (hidden, inter, heads, layers), n, seq, cfg in (
        ("awq_e2e", (4096, 11008, 32, 32), 128, 2048, AWQConfig(bits=4, group_size=128, use_sym=False, use_auto_scale=True, use_auto_clip=True)),
        ("smoothquant_e2e", (5120, 13824, 40, 40), 32, 2048, SmoothQuantConfig(alpha=0.5, folding=False, scale_sharing=True)),
    ):
        mc = LlamaConfig(hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers, num_attention_heads=heads,
                         num_key_value_heads=heads, vocab_size=32000, max_position_embeddings=4096, tie_word_embeddings=False)
        torch.manual_seed(0)
        with torch.device(device):
            model = LlamaForCausalLM(mc)
        model = model.to(torch.bfloat16).eval()
        model.config.use_cache = False
        if tag == "smoothquant_e2e":
            cfg.set_local("lm_head", SmoothQuantConfig(w_dtype="fp32"))
        ids = [torch.randint(0, 32000, (1, seq), generator=g) for _ in range(n)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            model = prepare(model, cfg, example_inputs=ids[0].to(device))
            for x in ids:
                model(x.to(device))
            model = convert(model)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[tag] = dict(wall_s=round(dt, 2), blocks=layers, samples=n, seq_len=seq, hidden=hidden, ffn=inter,
                        what="prepare -> calibration forwards -> convert of the whole model, wall-clock")
        note(f"{tag}: {dt:.1f}s for {layers} blocks")
        del model, ids
        torch.cuda.empty_cache()
    return out


def bench_awq_sq_blocks(device, note):
    """BASELINE configs #3 (AWQ INT4 g128, auto-scale + auto-clip; one Llama-2-7B-shaped block, 128 x 2048 tokens: the headline's set) and #4
    (SmoothQuant W8A8 calibrate + convert; one Llama-2-13B-shaped block, 32 x 2048 tokens): wall-clock per block."""
    from transformers import LlamaConfig, LlamaForCausalLM

    from neural_compressor_amd.torch.quantization import AWQConfig, SmoothQuantConfig, convert, prepare

    def llama(hidden, inter, heads):
        cfg = LlamaConfig(hidden_size=hidden, intermediate_size=inter, num_hidden_layers=1, num_attention_heads=heads,
                          num_key_value_heads=heads, vocab_size=32000, max_position_embeddings=4096, tie_word_embeddings=False)
        torch.manual_seed(0)
        with torch.device(device):
            m = LlamaForCausalLM(cfg)
        m = m.to(torch.bfloat16).eval()
        m.config.use_cache = False
        return m

    out = {}
    g = torch.Generator().manual_seed(1)
    for tag, dims, n, seq, cfg in (
        ("awq_block", (4096, 11008, 32), 128, 2048, AWQConfig(bits=4, group_size=128, use_sym=False, use_auto_scale=True, use_auto_clip=True)),
        ("smoothquant_block", (5120, 13824, 40), 32, 2048, SmoothQuantConfig(alpha=0.5, folding=False, scale_sharing=True)),
    ):
        ids = [torch.randint(0, 32000, (1, seq), generator=g) for _ in range(n)]
        if tag == "smoothquant_block":
            cfg.set_local("lm_head", SmoothQuantConfig(w_dtype="fp32"))
        times = []
        for _ in range(2):  # the first pass pays the libraries' first-call costs for these shapes (a 32-block model pays them once)
            model = llama(*dims)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.no_grad():
                model = prepare(model, copy.deepcopy(cfg), example_inputs=ids[0].to(device))
                for x in ids:
                    model(x.to(device))
                model = convert(model)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
            del model
            torch.cuda.empty_cache()
        dt = times[-1]
        out[tag] = dict(seconds_per_block=round(dt, 3), first_pass_s=round(times[0], 3), samples=n, seq_len=seq, hidden=dims[0],
                        ffn=dims[1], model_estimate_s=round(dt * (32 if tag == "awq_block" else 40), 1))
        if tag == "awq_block":
            out[tag]["floor"] = "GEMM flops the reference's grid searches prescribe (awq.py:341,454); DESIGN.md section 6"
        note(f"{tag}: {dt:.2f}s (first pass {times[0]:.2f}s)")
    return out


def bench_rtn_config1(device, note):
    """BASELINE config #1 (OPT-125M RTN INT8 weight-only, rtn.py:68) on the GPU: quantize(model, RTNConfig(bits=8, group_size=-1)) over
    the OPT-125M-shaped stack the parity test uses (72 Linears, weights resident in HBM), wall-clock incl. packing; the CPU port of the
    same job is timed in cpu_baseline (t_rtn_config1)."""
    from neural_compressor_amd.torch.quantization import RTNConfig, quantize
    from tests.model_zoo import opt125m_like

    times = []
    for _ in range(3):
        model = opt125m_like().to(device)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        q = quantize(model, RTNConfig(bits=8, group_size=-1, use_layer_wise=False))
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        del model, q
    note(f"config #1 (RTN INT8, OPT-125M-shaped): {min(times) * 1e3:.1f} ms")
    return dict(seconds=round(min(times), 4), first_pass_s=round(times[0], 4), modules=72, params_m=85.0,
                what="quantize(model, RTNConfig(bits=8, group_size=-1)) on an OPT-125M-shaped stack, weights in HBM, packing included")


def _run_child(mode, timeout_s=600):
    """Some measurements want a FRESH process (no streams created yet, chip not just out of a full-power phase: profiles/NOTES.md):
    re-run this script with --child MODE and read the JSON line it prints."""
    import subprocess

    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", mode], capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        return json.loads(line)
    except Exception as e:  # report, never fake
        print(f"[bench] child '{mode}' failed: {type(e).__name__}: {e}", file=sys.stderr)
        return None


def compact_line(full):
    """The ONE stdout line: the contract's keys, `roofline` and `cpu_baseline` as flat scalars (the driver's record keeps the scalars of
    these two objects and drops nested lists), everything else in short form.  The full record goes to --detail-file."""
    out = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "rccl_ranks", "dist_backend", "steps", "warmup", "ms_per_step", "higher_is_better",
                                "scaling", "vs_baseline", "dtype", "data", "config", "value_is", "rounds_only_s", "layer_round_ms", "sanity") if k in full}
    roof = dict(full.get("roofline") or {})
    cpu = full.get("cpu_baseline") or {}
    rows = full.get("dequant_gemm") or []
    loop = (full.get("ceilings") or {}).get("bf16_mfma_loop_tflops")
    fl = {(f["M"], f["N"], f["K"]): f["tflops"] for f in (cpu.get("f_linear") or [])}
    gem = []
    for r in rows:
        tag = f"{r['M']}x{r['N']}x{r['K']}"
        if r["bound"] == "mfma":
            # second half of BASELINE's metric: fused INT4->bf16 dequant-GEMM per shape -- TFLOP/s, of the spec peak, against hipBLASLt's
            # dense bf16 GEMM timed in alternation, and (M >= 1024) against the un-fused route recover() + hipBLASLt
            roof[f"gemm_{tag}_tflops"] = r["tflops"]
            roof[f"gemm_{tag}_frac"] = r["frac"]
            if r.get("vs_hipblaslt_dense") is not None:
                roof[f"gemm_{tag}_vs_hipblaslt"] = r["vs_hipblaslt_dense"]
            if r.get("fused_vs_unfused") is not None:
                roof[f"gemm_{tag}_vs_unfused"] = r["fused_vs_unfused"]
            if loop and r["M"] >= 1024:
                roof[f"gemm_{tag}_of_mfma_loop"] = round(r["tflops"] / loop, 4)
        elif r.get("cold_graph_frac") is not None:
            # HBM-bound rows: the COLD figure (ring of distinct modules), the only one that is HBM bandwidth
            roof[f"gemv_{tag}_hbm_frac_cold"] = r["cold_graph_frac"]
            roof[f"gemv_{tag}_us_cold"] = round(r["cold_graph_ms"] * 1e3, 2)
        if (r["M"], r["N"], r["K"]) in fl:
            roof[f"cpu_f_linear_{tag}_tflops"] = fl[(r["M"], r["N"], r["K"])]
        gem.append([r["M"], r["N"], r["K"], r["tflops"], r["frac"]])
    big = [r["frac"] for r in rows if r["M"] >= 4096]
    if big:
        roof["gemm_frac_min_M_ge_4096"], roof["gemm_frac_max_M_ge_4096"] = min(big), max(big)
    for r in full.get("dequant_gemv_groups") or []:
        if r.get("cold_graph_frac") is not None:  # ONE launch for the modules that share x (inc_woq_gemm_multi), cold
            roof[f"gemv_group_{r['group']}_hbm_frac_cold"] = r["cold_graph_frac"]
            roof[f"gemv_group_{r['group']}_us_cold"] = round(r["cold_graph_ms"] * 1e3, 2)
            roof[f"gemv_group_{r['group']}_single_calls_hbm_frac_cold"] = r.get("single_calls_cold_graph_frac")
    for r in full.get("int8_weight_only") or []:  # weight-only INT8 (config #1's format): cold decode and prefill
        tag = f"{r['N']}x{r['K']}"
        if r.get("decode_cold_hbm_frac") is not None:
            roof[f"int8_gemv_1x{tag}_hbm_frac_cold"] = r["decode_cold_hbm_frac"]
            roof[f"int8_gemv_1x{tag}_us_cold"] = round(r["decode_cold_ms"] * 1e3, 2)
        roof[f"int8_gemm_4096x{tag}_frac"] = r.get("prefill_frac")
    per = full.get("per_layer") or {}
    for k, v in per.items():
        if k.startswith("fasterquant_"):
            roof[f"column_loop_{k[12:]}_us_per_column_median"] = v.get("column_loop_us_per_column_median")
            roof[f"column_loop_{k[12:]}_us_per_column_min"] = v.get("column_loop_us_per_column_min")
            roof[f"column_loop_{k[12:]}_f32_mfma_frac"] = v.get("trailing_update_f32_mfma_frac")
        elif k.startswith(("unpack", "recover", "pack", "quant_tensor")):
            roof[f"{k}_hbm_frac_cold"] = v.get("hbm_frac")
    ceil_ = full.get("ceilings") or {}
    for k in ("hbm_copy_gbs", "hbm_stream_triad_gbs", "bf16_mfma_loop_tflops"):
        if ceil_.get(k) is not None:
            roof[f"measured_{k}"] = ceil_[k]
    out["roofline"] = roof
    if cpu:
        out["cpu_baseline"] = {k: v for k, v in cpu.items() if not isinstance(v, (list, dict))}
        if len(str(out["cpu_baseline"].get("sample", ""))) > 118:
            out["cpu_baseline"]["sample"] = "oracle on this host: one call per distinct shape of a block, x counts x 32 blocks (detail file has the terms)"
    if full.get("e2e"):
        out["e2e"] = {k: full["e2e"][k] for k in ("wall_s", "blocks", "packed_modules", "prepare_and_capture_s") if k in full["e2e"]}
    out["kernel_breakdown"] = {k: v["avg_ms"] for k, v in (full.get("kernel_breakdown") or {}).items() if v.get("launches")}
    if gem:
        out["dequant_gemm"] = dict(columns=["M", "N", "K", "tflops", "frac_eager"], rows=gem)
    if full.get("w8a8_gemm"):
        out["w8a8_gemm"] = dict(columns=["M", "N", "K", "gemm_tops", "fwd_tops"], rows=[[r["M"], r["N"], r["K"], r["gemm_tops"], r["fwd_tops"]] for r in full["w8a8_gemm"]])
    for k in ("awq_block", "smoothquant_block"):
        if full.get(k):
            out[k] = {kk: full[k][kk] for kk in ("seconds_per_block", "samples", "seq_len", "model_estimate_s") if kk in full[k]}
    for k in ("awq_e2e", "smoothquant_e2e", "rtn_config1"):
        if full.get(k):
            out[k] = {kk: vv for kk, vv in full[k].items() if not isinstance(vv, str)}
    if per:
        out["per_layer"] = {k: ([v.get("gpu_s_per_sample"), v.get("frac")] if k.startswith("hessian") else
                                [v.get("gpu_s"), v.get("hbm_frac") if "hbm_frac" in v else v.get("column_loop_s_median")]) for k, v in per.items()}
    if full.get("dequant_gemm_replicas"):
        out["dequant_gemm_replicas"] = full["dequant_gemm_replicas"]
    out["allocator"] = {k: v for k, v in (full.get("allocator") or {}).items() if k in ("hipMalloc_calls", "hipFree_calls", "reserved_GiB")}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--seq", type=int, default=2048)
    ap.add_argument("--e2e-blocks", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gemm", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true")
    ap.add_argument("--no-per-layer", action="store_true")
    ap.add_argument("--e2e-configs", action="store_true", help="also run BASELINE configs #3 (AWQ, 32 blocks) and #4 (SmoothQuant, 40 blocks) end to end (~4 min)")
    ap.add_argument("--mgpu-mode", choices=("layer", "exact"), default="layer", help="N > 1: one block per GPU on float activations (north_star) | exact reference semantics")
    ap.add_argument("--layer-on-one-gpu", action="store_true", help="N = 1: time the layer-per-GPU mode's round (float forward + quantise, no exchange)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="llama2-7b", help="llama2-70b: BASELINE config #5's block shape (hidden 8192, ffn 28672, GQA 64:8)")
    ap.add_argument("--detail-file", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"),
                    help="the full per-row record (cold / graph / cache-resident variants of every row); the stdout line is its compact form")
    ap.add_argument("--full-line", action="store_true", help="print the full record on stdout instead of the compact line")
    ap.add_argument("--child", choices=("per_layer",), default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.child == "per_layer":
        # a fresh process on an otherwise idle chip (the parent waits): the measured ceilings and the per-layer calls
        device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(device)
        print(json.dumps(dict(ceilings=bench_ceilings(device), per_layer=bench_per_layer(device, None))), flush=True)
        return
    global WORKLOAD
    WORKLOAD = args.workload
    if args.workload != "llama2-7b":
        args.no_extra_configs = True  # (the AWQ / SmoothQuant block configs and the e2e model are 7B / 13B-shaped)
        args.no_e2e = True
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as a plain script: launch the N ranks ourselves (one process per GPU, RCCL) and hand their output through
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print(f"[bench] --gpus {args.gpus} without a torchrun environment: launching {' '.join(cmd)}", file=sys.stderr, flush=True)
        sys.exit(subprocess.call(cmd))

    def note(msg):  # progress on stderr: the single JSON line on stdout stays clean
        print(f"[bench +{time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)

    from neural_compressor_amd import distributed as D
    from neural_compressor_amd import ops
    from neural_compressor_amd.torch.quantization import GPTQConfig, prepare

    backend = os.environ.get("INC_MI355X_DIST_BACKEND") or "nccl"
    if args.gpus > 1 and backend == "nccl" and torch.cuda.device_count() < args.gpus:
        raise SystemExit(f"--gpus {args.gpus}: this node shows {torch.cuda.device_count()} GPU(s); RCCL needs one device per rank "
                         "(INC_MI355X_DIST_BACKEND=gloo lets test ranks share a device)")
    rank, world, local_rank = D.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the process group has {world} rank(s): refusing to report a number for another job size")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # --layer-on-one-gpu: the layer-per-GPU mode with a "world" of one rank (no exchange) -- the per-round terms of the N-GPU projection
    # `live`: the multi-GPU paths run -- N > 1 ranks, or ONE rank with INC_MI355X_DIST_SINGLE_RANK=1 (RCCL bring-up on a 1-GPU box: the
    # process group, the collective wrappers and the drivers' exchange / broadcast code execute over a world of one rank)
    live = D.live()
    layer_mode = args.mgpu_mode == "layer" and (live or args.layer_on_one_gpu)
    if live:
        # ONE model, N ranks.  layer: one block per rank on the float model's activations; exact: samples sharded, Hessians reduced to
        # their owner rank, factors broadcast, row-sharded solves
        os.environ["INC_MI355X_GPTQ_MULTI_GPU"] = "layer" if layer_mode else "sample+rows"
    dist_backend = torch.distributed.get_backend() if live else None
    # ranks that really are one process per GPU over RCCL ("nccl"); null when the ranks talk gloo (test runs sharing one device)
    rccl_ranks = (torch.distributed.get_world_size() if dist_backend == "nccl" else None) if live else 1

    # (layer mode on N > 1 ranks: one more round of blocks than is timed -- every round also runs the NEXT round's float forwards and
    # posts its exchange before it quantises, so the last timed round needs a successor to do the same work as the others)
    n_blocks = (args.warmup + args.steps + (1 if (layer_mode and live) else 0)) * (world if layer_mode else 1)
    note(f"building {n_blocks}-block Llama-2-7B-shaped model on {device}")
    model = build_model(n_blocks, device)  # same seed on every rank: the ranks hold replicas of the one model
    ids = calib_ids(args.samples, args.seq)
    mine = D.shard_samples(len(ids), rank, world) if world > 1 else range(len(ids))

    cfg = GPTQConfig(bits=4, group_size=128, use_sym=True, block_size=128, percdamp=0.01, act_order=False)
    model = prepare(model, cfg)
    with torch.no_grad():
        for j in mine:  # run_fn: embeddings only, block-0 inputs captured in HBM (gptq.py:413-433 semantics)
            model(ids[j].to(device))
    note(f"calibration inputs captured ({len(mine)} of {len(ids)} samples on this rank)")
    rq = model.quantizer.gptq_quantizer
    assert ((rq.layer_ctx if layer_mode else rq.dist_ctx) is not None) == live
    rq.remove_prepare_for_calibration()
    if layer_mode:
        rq.independent_setup()
    blocks = rq.gptq_related_blocks["transformers"]
    round_timing = {}

    clock = KernelClock()
    clock.wrap(ops, "gptq_hessian_accum", lambda H, x, b, a: f"hessian_K{x.shape[1]}", lambda H, x, b, a: 2.0 * x.shape[0] * x.shape[1] ** 2)
    # the driver's normal path: ALL Hessians of one stacked forward in one launch (inc_gptq_hessian_accum_multi)
    clock.wrap(ops, "gptq_hessian_accum_multi", lambda items: "hessian_multi_K" + "+".join(str(x.shape[1]) for _, x, _, _ in items),
               lambda items: sum(2.0 * x.shape[0] * x.shape[1] ** 2 for _, x, _, _ in items))
    clock.wrap(ops, "gptq_quant_block", lambda w, *a: "quant_block", lambda w, *a: 2.0 * w.shape[0] * w.shape[1] * 4)
    clock.wrap(ops, "gptq_quant_block_params", lambda w, *a: "quant_block", lambda w, *a: 2.0 * w.shape[0] * w.shape[1] * 4)  # find_params fused in
    clock.wrap(ops, "gptq_lazy_update", lambda w, h, e, i1, c: "lazy_update", lambda w, h, e, i1, c: 2.0 * w.shape[0] * c * max(w.shape[1] - i1 - c, 0))
    # look-ahead column loop: the next block's 128 columns on the main stream, the rest of the trailing matrix on the second stream
    clock.wrap(ops, "gptq_lazy_update_cols", lambda w, h, e, i1, c, c0, c1: "lazy_update_next" if c0 == i1 + c else "lazy_update_rest",
               lambda w, h, e, i1, c, c0, c1: 2.0 * w.shape[0] * c * max(c1 - c0, 0))

    # the column loop as ONE C-ABI call per solve (inc_gptq_quantize_layer): timed on the main stream, the launches above stay empty
    clock.wrap(ops, "gptq_quantize_layer", lambda w, *a, **k: f"quantize_layer_{w.shape[0]}x{w.shape[1]}",
               lambda w, *a, **k: float(w.shape[0]) * w.shape[1] * (128 + w.shape[1]))

    # float weights of the last block the timed region quantises (the sanity check after the timed region compares against them)
    float_w = {} if layer_mode else {n: m.weight.data.clone() for n, m in blocks[args.warmup + args.steps - 1].named_modules() if isinstance(m, torch.nn.Linear)}
    with torch.no_grad():
        def step(i):  # exact / single GPU: one block; layer: one round = `world` blocks, one per rank
            if layer_mode:
                rq.independent_round(blocks, i * world)
            else:
                rq.quantize_block(blocks[i], i)

        for i in range(args.warmup):
            step(i)
            torch.cuda.synchronize()
            note(f"warmup step {i} done")
        torch.cuda.synchronize()
        if live:
            torch.distributed.barrier()
        clock.enabled = True
        if layer_mode and os.environ.get("INC_MI355X_BENCH_ROUND_TIMING", "1" if world == 1 else "0") == "1":
            rq._layer_state["timing"] = round_timing  # device-synchronised phase times of every round (three extra syncs per round)
        trace = os.environ.get("INC_MI355X_TRACE_RANGES", "0") == "1"  # phase markers for scripts/step_timeline.py
        mem0 = torch.cuda.memory_stats(device)
        t0 = time.perf_counter()
        for i in range(args.warmup, args.warmup + args.steps):
            if trace:
                ops.trace_marker(1)
            step(i)
        if trace:
            ops.trace_marker(1)
        torch.cuda.synchronize()
        if live:
            torch.distributed.barrier()
        elapsed = time.perf_counter() - t0
        clock.enabled = False
    note(f"timed region done: {elapsed:.2f}s for {args.steps} steps")
    # sanity of what the timed steps produced (several streams share the work: a lifetime bug would show up as garbage, not as an error):
    if layer_mode:
        rq._drain_prefetched()  # the look-ahead exchange of the round after the last timed one: wait for it, drop its buffers
    # every packed module of the last quantised block dequantises to finite values near its float weight
    sanity = None
    if not layer_mode:
        from neural_compressor_amd.torch.algorithms.weight_only.modules import MI355XWeightOnlyLinear

        ref = float_w
        worst = 0.0
        for name, m in blocks[args.warmup + args.steps - 1].named_modules():
            if isinstance(m, MI355XWeightOnlyLinear):
                w = m.recover().float()
                w0 = ref[name].float()
                rel = float((w - w0).norm() / w0.norm())
                worst = max(worst, rel if bool(torch.isfinite(w).all()) else float("inf"))
        sanity = dict(last_block_max_rel_weight_error=round(worst, 4), ok=bool(worst < 0.25))
        if not sanity["ok"]:
            raise SystemExit(f"bench: the last quantised block does not dequantise near its float weights ({sanity}): refusing to report a time")
        del ref
    mem1 = torch.cuda.memory_stats(device)
    # caching-allocator activity inside the timed region: hipMalloc / hipFree calls reach the driver only through "segment" events
    allocator = {k: int(mem1.get(v, 0) - mem0.get(v, 0)) for k, v in (("hipMalloc_calls", "num_device_alloc"), ("hipFree_calls", "num_device_free"),
                                                                   ("alloc_retries", "num_alloc_retries"), ("block_allocations", "allocation.all.allocated"))}
    allocator["reserved_GiB"] = round(mem1.get("reserved_bytes.all.current", 0) / 2**30, 2)
    allocator["per"] = f"{args.steps} timed steps"
    elapsed = D.barrier_max_time(elapsed, device=device)
    ms_per_step = elapsed * 1e3 / args.steps
    # whole job = a 32-block model.  exact / single GPU: 32 steps (N ranks work on the SAME block); layer: ceil(32 / N) rounds
    hidden, ffn, heads, kv_heads, model_blocks, baseline_cfg = WORKLOADS[WORKLOAD]
    steps_per_model = -(-model_blocks // world) if layer_mode else model_blocks
    value = steps_per_model * elapsed / args.steps
    value_is = f"{steps_per_model} x the timed step"

    kern = clock.summary()
    breakdown = {k: dict(launches=v["launches"], total_ms=round(v["total_ms"], 3), avg_ms=round(v["avg_ms"], 4)) for k, v in kern.items()}
    # dominant own kernel of the step -> roofline (MFMA-bound Hessian syrk; algorithmic flops = 2*T*K^2 per launch)
    hess = {k: v for k, v in kern.items() if k.startswith("hessian")}
    roofline = None
    if hess:
        dom = max(hess, key=lambda k: hess[k]["total_ms"])
        v = hess[dom]
        achieved = v["work"] / (v["total_ms"] * 1e-3) / 1e12
        Ks = [int(t) for t in dom.split("K")[-1].split("+")]
        tokens = int(round(v["work"] / v["launches"] / sum(2.0 * k * k for k in Ks)))
        # the kernel multiplies the upper-triangular 256x256 tiles only: executed / algorithmic flops of the launch
        tiles = [(-(-k // 256)) for k in Ks]
        executed = sum(t * (t + 1) / 2 * 256 * 256 for t in tiles) / sum(float(k) * k for k in Ks)
        all_work = sum(x["work"] for x in hess.values())
        all_ms = sum(x["total_ms"] for x in hess.values())
        traffic, traffic_src = _pmc_traffic(dom)
        multi = dom.startswith("hessian_multi")
        roofline = dict(kernel=("hessian_syrk_tr_256_multi_kernel<bf16>" if multi else "hessian_syrk_tr_256_kernel<bf16>") +
                               f" K={dom.split('K')[-1]}; flops 2*T*K^2 per Hessian", bound="mfma",
                        achieved=round(achieved, 2), peak=BF16_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                        frac=round(achieved / BF16_MFMA_PEAK_TFLOPS, 4), traffic=traffic, traffic_source=traffic_src,
                        avg_launch_ms=round(v["avg_ms"], 4), launches=v["launches"], tokens_per_launch=tokens,
                        executed_frac=round(achieved * executed / BF16_MFMA_PEAK_TFLOPS, 4),
                        all_launches_frac=round(all_work / (all_ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4),
                        note=f"algorithmic = full X^T X (SURVEY 8d); the syrk executes the upper tiles only ({executed:.3f}): executed_frac")

    result = dict(
        metric=f"{WORKLOAD.replace('-', '_')}_gptq_int4_g128_quantize_wall_clock", value=round(value, 3), unit="s", n_gpus=world, rccl_ranks=rccl_ranks,
        dist_backend=dist_backend,
        steps=args.steps, warmup=args.warmup, ms_per_step=round(ms_per_step, 2), higher_is_better=False,
        scaling="strong", vs_baseline=None, dtype="bf16", data="synthetic",
        config=dict(workload=f"{WORKLOAD} GPTQ INT4 g128 sym, {args.samples}x{args.seq} calib tokens; step = one block (7 Linears), value = {model_blocks} blocks",
                    baseline_config=baseline_cfg, samples=args.samples, seq_len=args.seq, block_size=128, percdamp=0.01,
                    linears=f"q/o [{hidden},{hidden}], k/v [{hidden * kv_heads // heads},{hidden}], gate/up [{ffn},{hidden}], down [{hidden},{ffn}]",
                    capture_pass="first forward ends at the last hooked Linear (gptq.CAPTURE_EARLY_STOP; bit-identical model; DESIGN.md 6)",
                    arithmetic="bf16 activations/weights (MFMA, fp32 accumulate), fp32 Hessian + Cholesky + column loop, int4 codes",
                    parallelism=(("single GPU" if not layer_mode else "single GPU running the layer-per-GPU mode's round (no exchange)") if world == 1 else
                                 (f"ONE model on {world} ranks, one block per rank per round, inputs sent to the owner over RCCL (mode layer; DESIGN.md 7)" if layer_mode else
                                  f"ONE model on {world} ranks: samples sharded, Hessians reduced to owners, row-sharded column loop (mode exact)")),
                    steps_per_model=steps_per_model),
        roofline=roofline, kernel_breakdown=breakdown, allocator=allocator, sanity=sanity,
    )
    if round_timing:
        result["layer_round_ms"] = {k.replace("_s", ""): round(v * 1e3 / args.steps, 2) for k, v in round_timing.items() if k != "_"}
    del model, rq, blocks
    torch.cuda.empty_cache()
    if not args.no_e2e:
        result["e2e"] = bench_e2e(device, args, rank, world, note)
        if live and result["e2e"]["blocks"] == model_blocks:
            # N > 1: the whole job is what `e2e` ran -- capture on every rank, ceil(32 / N) rounds (or 32 sharded blocks), AND the
            # broadcasts of the packed blocks at the end, max over ranks between two barriers.  The round time stays in ms_per_step.
            result["value"] = result["e2e"]["wall_s"]
            result["value_is"] = "e2e.wall_s: prepare -> capture -> all rounds -> packed-block broadcasts of the whole model (max over ranks)"
            result["rounds_only_s"] = round(value, 3)
    result.setdefault("value_is", value_is)
    big = [(4096, 4096, 4096), (4096, 11008, 4096), (4096, 4096, 11008), (8192, 4096, 4096)]
    if not args.no_gemm and live:
        # north_star: the 4096x4096 / 11008x4096 linears "at 1, 2, 4 and 8 GPUs": the forward does not shard, so every rank runs the
        # four BASELINE shapes as a replica at the same time (replicas, reported separately); per-GPU and aggregate
        torch.distributed.barrier()
        mine_rows = bench_dequant_gemm(device, big)
        t = torch.tensor([r["tflops"] for r in mine_rows], dtype=torch.float64, device="cpu" if dist_backend == "gloo" else device)
        gathered = [torch.zeros_like(t) for _ in range(world)]
        torch.distributed.all_gather(gathered, t)
        if rank == 0:
            per = torch.stack(gathered).cpu()
            result["dequant_gemm_replicas"] = [dict(M=m_, N=n_, K=k_, per_gpu_tflops=[round(float(v), 1) for v in per[:, i]],
                                                    aggregate_tflops=round(float(per[:, i].sum()), 1),
                                                    min_frac_of_bf16_peak=round(float(per[:, i].min()) / BF16_MFMA_PEAK_TFLOPS, 4))
                                               for i, (m_, n_, k_) in enumerate(big)]
    if rank == 0 and not args.no_gemm:
        shapes = list(big)
        shapes += [(m, 4096, 4096) for m in (1, 16, 32, 64, 128, 256, 512)] + [(1, 11008, 4096), (1, 4096, 11008)]
        if WORKLOAD == "llama2-70b":
            shapes = [(4096, 8192, 8192), (4096, 28672, 8192), (4096, 8192, 28672), (1, 8192, 8192), (1, 28672, 8192), (1, 8192, 28672)]
        result["dequant_gemm"] = bench_dequant_gemm(device, shapes)
        if WORKLOAD == "llama2-7b":
            result["dequant_gemv_groups"] = bench_gemv_groups(device)
            result["int8_weight_only"] = bench_int8_rows(device)
        note("dequant-GEMM shapes timed")
        result["w8a8_gemm"] = bench_w8a8_gemm(device, [(4096, 5120, 5120), (4096, 13824, 5120), (4096, 5120, 13824)])
        note("W8A8 shapes timed")
    if rank == 0 and not args.no_per_layer:
        # BEFORE the CPU baseline (which leaves the chip idle and down-clocked for half a minute) and in a fresh process: ceilings + per-layer
        torch.cuda.empty_cache()
        child = _run_child("per_layer") if WORKLOAD == "llama2-7b" else dict(ceilings=bench_ceilings(device), per_layer=None)
        if child:
            result["ceilings"] = child["ceilings"]
            if child.get("per_layer"):
                result["per_layer"] = child["per_layer"]
        note("ceilings + per-layer figures done (fresh process)")
    if rank == 0 and world == 1 and not args.no_extra_configs:
        result.update(bench_awq_sq_blocks(device, note))
        result["rtn_config1"] = bench_rtn_config1(device, note)
        if args.e2e_configs:
            result.update(bench_awq_sq_e2e(device, note))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline()
        note("cpu baseline done")
        cpu = result["cpu_baseline"]
        cpu_key = {"hessian_K4096": "t_add_4096", "hessian_K11008": "t_add_11008", "quant_tensor_4096x4096": "t_quant_tensor_4096",
                   "pack_4096x4096": "t_pack_4096", "unpack_4096x4096": "t_unpack_4096", "recover_4096x4096": "t_recover_4096",
                   "pack_11008x4096": "t_pack_11008", "unpack_11008x4096": "t_unpack_11008", "recover_11008x4096": "t_recover_11008"}
        for k, row in (result.get("per_layer") or {}).items():
            ck = cpu_key.get(k) or (k.replace("fasterquant_", "t_fq_") if k.startswith("fasterquant_") else None)
            if ck and cpu.get(ck) is not None:
                row["cpu_s_per_sample" if k.startswith("hessian") else "cpu_s"] = cpu[ck]
        if result.get("rtn_config1") and cpu.get("t_rtn_config1") is not None:
            result["rtn_config1"]["cpu_port_s"] = cpu["t_rtn_config1"]
    if rank == 0:
        line = result if args.full_line else compact_line(result)
        try:
            os.makedirs(os.path.dirname(args.detail_file), exist_ok=True)
            with open(args.detail_file, "w") as f:
                json.dump(result, f, indent=1)
            if not args.full_line:
                line["detail_file"] = os.path.relpath(args.detail_file, ROOT)
        except OSError as e:  # a read-only checkout: the compact line is still complete on its own
            print(f"[bench] could not write {args.detail_file}: {e}", file=sys.stderr)
        print(json.dumps(line))
    if live:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
