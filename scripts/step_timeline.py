#!/usr/bin/env python3
"""One-step timeline of bench.py's GPTQ step from a `rocprofv3 --kernel-trace [--memory-copy-trace]` CSV.

usage: step_timeline.py <dir with *_kernel_trace.csv> [out.md]

bench.py (INC_MI355X_TRACE_RANGES=1) launches inc_trace_marker_kernel with Grid_Size_X = 64 * id on the main stream:
  id 1 = a step begins (one more after the last step); 2p / 2p+1 = phase p of quantize_block begins / ends
  (p: 2 capture forward + Hessians, 3 solves issued, 4 second forward, 5 wait for the solves, 6 packing).
For every complete step: span, union of busy intervals (= span - GPU idle), per-stream busy time, busy time per kernel class,
the same per phase, and how long the column-loop kernels ran beside factorisation / forward kernels.
"""
import collections
import csv
import glob
import json
import os
import sys

PHASES = {2: "capture forward + Hessians", 3: "solves issued (factor + column loops)", 4: "second forward", 5: "wait for solves / checks", 6: "packing"}


def classify(name):
    n = name
    if "inc_trace_marker" in n:
        return "marker"
    if "hessian_syrk" in n or "hessian_tail" in n or "hessian_final" in n:
        return "hessian (own)"
    if "chol_" in n or "ifac_" in n or "f32gemm" in n or "bf16x3_gemm" in n or "hessian_mirror" in n or "hessian_diag" in n:
        return "factorisation (own)"
    if "gptq_quant_block" in n or "gptq_lazy_update" in n or "gptq_find_params" in n or "gptq_prepare" in n or "gptq_hessian_finalize" in n:
        return "column loop (own)"
    if "woq_pack" in n or "pack_" in n or "woq_" in n:
        return "pack / woq (own)"
    if "Cijk_" in n:
        # Tensile names: Cijk_<A layout>_<B layout>_<types>_...; fp32 GEMMs carry the type string "S_B" / "SB", bf16 ones "BBS_BH" / "B_B"
        head = n.split("Cijk_")[1].split("_MT")[0]
        return "library GEMM fp32 (factorisation)" if ("_S_B_" in "_" + head + "_" or head.endswith("_SB") or "_SB_" in head) else "library GEMM bf16 (model forward)"
    if "attention" in n.lower() or "fmha" in n.lower() or "flash" in n.lower() or "sdpa" in n.lower() or "attn_fwd" in n:
        return "attention (model forward)"
    if "elementwise" in n or "reduce_kernel" in n or "rms" in n.lower() or "norm" in n.lower() or "CatArray" in n or "index" in n or "copy" in n.lower() or "softmax" in n.lower():
        return "elementwise / norm / copies (torch)"
    return "other"


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s > ce:
            tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    if cs is not None:
        tot += ce - cs
    return tot


def overlap(a, b):
    """total time intervals of `a` spend inside the union of `b`"""
    b = sorted(b)
    merged = []
    for s, e in b:
        if merged and s <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], e)
        else:
            merged.append([s, e])
    tot = 0
    for s, e in a:
        for ms, me in merged:
            if me <= s:
                continue
            if ms >= e:
                break
            tot += min(e, me) - max(s, ms)
    return tot


def main():
    d = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else None
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit("no kernel trace under " + d)
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    ks = []
    for r in rows:
        ks.append(dict(name=r["Kernel_Name"], s=int(r["Start_Timestamp"]), e=int(r["End_Timestamp"]), stream=r.get("Stream_Id", "?"),
                       queue=r.get("Queue_Id", "?"), grid=int(r.get("Grid_Size_X", 0) or 0)))
    ks.sort(key=lambda k: k["s"])
    copies = []
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            copies.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    marks = [(k["s"], k["grid"] // 64) for k in ks if "inc_trace_marker" in k["name"]]
    starts = [t for t, i in marks if i == 1]
    if len(starts) < 2:
        sys.exit(f"need >= 2 step markers, found {len(starts)}")
    lines = []
    summary = []
    for si in range(len(starts) - 1):
        t0, t1 = starts[si], starts[si + 1]
        step = [k for k in ks if k["s"] >= t0 and k["s"] < t1 and "inc_trace_marker" not in k["name"]]
        span = t1 - t0
        busy = union([(k["s"], min(k["e"], t1)) for k in step])
        cls = collections.defaultdict(list)
        for k in step:
            cls[classify(k["name"])].append((k["s"], k["e"]))
        streams = collections.defaultdict(list)
        for k in step:
            streams[(k["queue"], k["stream"])].append((k["s"], k["e"]))
        cp = [(s, e) for s, e in copies if s >= t0 and s < t1]
        rec = dict(step=si, span_ms=span / 1e6, busy_ms=busy / 1e6, idle_ms=(span - busy) / 1e6, kernels=len(step),
                   sum_of_kernel_ms=sum(k["e"] - k["s"] for k in step) / 1e6, copies_ms=union(cp) / 1e6, n_copies=len(cp),
                   classes={c: dict(busy_ms=union(v) / 1e6, sum_ms=sum(e - s for s, e in v) / 1e6, launches=len(v)) for c, v in cls.items()},
                   streams={f"queue {q} / stream {s}": dict(busy_ms=union(v) / 1e6, launches=len(v)) for (q, s), v in streams.items()})
        # phases
        pm = [(t, i) for t, i in marks if t0 <= t < t1 and i >= 4]
        phases = []
        open_ = {}
        for t, i in pm:
            p = i // 2
            if i % 2 == 0:
                open_[p] = t
            elif p in open_:
                a, b = open_.pop(p), t
                inside = [k for k in step if k["s"] >= a and k["s"] < b]
                ub = union([(max(k["s"], a), min(k["e"], b)) for k in step if k["e"] > a and k["s"] < b])
                top = collections.Counter()
                for k in inside:
                    top[classify(k["name"])] += k["e"] - k["s"]
                phases.append(dict(phase=PHASES.get(p, str(p)), span_ms=(b - a) / 1e6, busy_ms=ub / 1e6, idle_ms=(b - a - ub) / 1e6, launched=len(inside),
                                   sum_ms_by_class={c: v / 1e6 for c, v in top.most_common(6)}))
        rec["phases"] = phases
        # who runs beside the column loop
        col = cls.get("column loop (own)", [])
        rec["column_loop_beside_ms"] = {c: overlap(col, v) / 1e6 for c, v in cls.items() if c not in ("column loop (own)", "marker")}
        rec["column_loop_sum_ms"] = sum(e - s for s, e in col) / 1e6
        names = collections.Counter()
        cnt = collections.Counter()
        for k in step:
            short = k["name"].split("(")[0][-70:]
            names[short] += k["e"] - k["s"]
            cnt[short] += 1
        rec["top_kernels"] = [dict(name=n, sum_ms=v / 1e6, launches=cnt[n]) for n, v in names.most_common(18)]
        # the model's own elementwise / copy kernels by what they are (their names differ only inside the template arguments)
        tn, tc = collections.Counter(), collections.Counter()
        for k in step:
            if classify(k["name"]).startswith("elementwise"):
                full = k["name"].replace("at::native::", "").replace("(anonymous namespace)::", "").replace("void ", "")
                key = full[:150]
                tn[key] += k["e"] - k["s"]
                tc[key] += 1
        rec["torch_kernels"] = [dict(name=n, sum_ms=v / 1e6, launches=tc[n]) for n, v in tn.most_common(16)]
        summary.append(rec)
    js = json.dumps(summary, indent=1)
    for rec in summary:
        lines.append(f"## step {rec['step']}: span {rec['span_ms']:.1f} ms, GPU busy (union) {rec['busy_ms']:.1f} ms, idle {rec['idle_ms']:.1f} ms, "
                     f"{rec['kernels']} kernels, sum of kernel time {rec['sum_of_kernel_ms']:.1f} ms, {rec['n_copies']} copies {rec['copies_ms']:.2f} ms")
        lines.append("")
        lines.append("| class | busy (union) ms | sum ms | launches |")
        lines.append("|---|---|---|---|")
        for c, v in sorted(rec["classes"].items(), key=lambda kv: -kv[1]["sum_ms"]):
            lines.append(f"| {c} | {v['busy_ms']:.2f} | {v['sum_ms']:.2f} | {v['launches']} |")
        lines.append("")
        lines.append("| stream | busy ms | launches |")
        lines.append("|---|---|---|")
        for c, v in sorted(rec["streams"].items(), key=lambda kv: -kv[1]["busy_ms"]):
            lines.append(f"| {c} | {v['busy_ms']:.2f} | {v['launches']} |")
        lines.append("")
        lines.append("| phase (main-stream markers) | span ms | busy ms | idle ms | launches | kernel time by class (ms) |")
        lines.append("|---|---|---|---|---|---|")
        for p in rec["phases"]:
            lines.append(f"| {p['phase']} | {p['span_ms']:.2f} | {p['busy_ms']:.2f} | {p['idle_ms']:.2f} | {p['launched']} | "
                         + ", ".join(f"{c} {v:.1f}" for c, v in p["sum_ms_by_class"].items()) + " |")
        lines.append("")
        lines.append(f"column-loop kernels: {rec['column_loop_sum_ms']:.2f} ms of kernel time, of which beside: "
                     + ", ".join(f"{c} {v:.2f}" for c, v in sorted(rec["column_loop_beside_ms"].items(), key=lambda kv: -kv[1]) if v > 0.01))
        lines.append("")
        lines.append("top kernels: " + "; ".join(f"{t['name']} {t['sum_ms']:.1f} ms x{t['launches']}" for t in rec["top_kernels"][:12]))
        lines.append("")
        lines.append("| torch elementwise / norm / copy kernel | ms | launches |")
        lines.append("|---|---|---|")
        for t in rec.get("torch_kernels", []):
            lines.append(f"| `{t['name']}` | {t['sum_ms']:.2f} | {t['launches']} |")
        lines.append("")
    text = "\n".join(lines)
    print(text)
    if out:
        with open(out, "w") as f:
            f.write("# GPTQ step timeline (scripts/step_timeline.py over rocprofv3 --kernel-trace; profiled runs are slower than un-profiled ones)\n\n" + text)
        with open(os.path.splitext(out)[0] + ".json", "w") as f:
            f.write(js)


if __name__ == "__main__":
    main()
