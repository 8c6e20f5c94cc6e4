#!/bin/bash
# One gpurun call of round 5.  usage: scripts/gpu_r5.sh <action> [<action> ...]   (run in order; `kbench`, `pytest` and `benchab` take one argument)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD
BENCH_MIN="--no-cpu-baseline --no-gemm --no-extra-configs --no-e2e --no-per-layer"
while [[ $# -gt 0 ]]; do
  case $1 in
    tests)
      timeout 1700 python -m pytest tests -m gpu -q -s --timeout=900 -p no:cacheprovider --durations=30 > gpurun_out/pytest_gpu.log 2>&1
      echo "pytest exit $?" | tee -a gpurun_out/pytest_gpu.log
      grep -E "passed|failed|Error|error" gpurun_out/pytest_gpu.log | tail -30
      timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" | tee -a gpurun_out/smoke.log
      tail -2 gpurun_out/smoke.log ;;
    pytest)
      shift
      timeout 1500 python -m pytest tests -m gpu -q -s --timeout=900 -p no:cacheprovider -k "$1" > gpurun_out/pytest_sel.log 2>&1
      echo "pytest -k '$1' exit $?" | tee -a gpurun_out/pytest_sel.log
      grep -E "passed|failed|Error|error|assert" gpurun_out/pytest_sel.log | tail -30 ;;
    bench)
      timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2> gpurun_out/bench.err
      echo "bench exit $?"; tail -c 7000 gpurun_out/bench.log; tail -25 gpurun_out/bench.err ;;
    kbench)
      shift
      timeout 600 tools/kbench $1 > gpurun_out/kbench_$1.log 2>&1; echo "kbench $1 exit $?"; tail -70 gpurun_out/kbench_$1.log ;;
    harness)
      # the variants prepared at the end of round 4 (profiles/NOTES.md "Prepared for round 5"): one log per kbench mode
      for m in hessian qlayer strip decode; do
        timeout 300 tools/kbench $m > gpurun_out/kbench_r5_$m.log 2>&1; echo "kbench $m exit $?"
      done
      grep -E "spread|rolling|prio|transpose-read 2x64|FAIL" gpurun_out/kbench_r5_hessian.log | tail -40
      grep -E "QLAYER" gpurun_out/kbench_r5_qlayer.log | tail -12
      grep -E "median|FAIL" gpurun_out/kbench_r5_strip.log | tail -60
      grep -E "median|FAIL" gpurun_out/kbench_r5_decode.log | tail -40 ;;
    timeline)
      # one-step timelines (kernel trace + copies) of the bench step, late solve off / on
      for late in 1; do
        rm -rf "$R/gpurun_out/tl_late$late"
        ( cd /tmp && INC_MI355X_TRACE_RANGES=1 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv \
            -d "$R/gpurun_out/tl_late$late" -o tl -- python "$R/bench.py" --steps 2 --warmup 2 $BENCH_MIN > "$R/gpurun_out/tl_late$late.log" 2> "$R/gpurun_out/tl_late$late.err" )
        echo "timeline late=$late exit $?"; grep -o '"ms_per_step": [0-9.]*' gpurun_out/tl_late$late.log
        python3 scripts/step_timeline.py gpurun_out/tl_late$late gpurun_out/step_timeline_late$late.md > /dev/null 2> gpurun_out/step_timeline_late$late.err || tail -3 gpurun_out/step_timeline_late$late.err
        head -12 gpurun_out/step_timeline_late$late.md | cut -c1-300
        find gpurun_out/tl_late$late -name "*.csv" -size +20M -delete   # (keep the merge-back small)
      done ;;
    benchab)
      shift
      for v in 1 0 1 0; do
        env "$1=$v" timeout 300 python bench.py --steps 4 --warmup 1 $BENCH_MIN > gpurun_out/bench_ab.log 2> gpurun_out/bench_ab.err
        python3 - $1 $v <<'PY' | tee -a gpurun_out/bench_ab_summary.txt
import json, sys
for line in open("gpurun_out/bench_ab.log"):
    if line.startswith("{"):
        d = json.loads(line)
        print(sys.argv[1], "=", sys.argv[2], "ms/step", d["ms_per_step"], {k.replace("quantize_layer_", "ql_").replace("hessian_multi_", "h_"): v["avg_ms"] for k, v in d["kernel_breakdown"].items()}, d.get("allocator"))
PY
      done ;;
    final)
      # the round's evidence set: the bench line the driver will see, rocprofv3 kernel stats of the reduced bench, the PMC traffic pass,
      # the tile A/B of the harness and the lab candidates
      timeout 900 python bench.py > gpurun_out/c_bench.log 2> gpurun_out/c_bench.err; echo "bench exit $?"
      rm -rf "$R/gpurun_out/prof"
      ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -o r5 -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-extra-configs --no-per-layer > "$R/gpurun_out/prof_bench.log" 2> "$R/gpurun_out/prof_bench.err" )
      echo "prof exit $?"; f=$(find gpurun_out/prof -name "*kernel_stats*" | head -1); cp "$f" gpurun_out/d_bench_kernel_stats.csv; head -5 gpurun_out/d_bench_kernel_stats.csv | cut -c1-200
      find gpurun_out/prof -name "*.csv" -size +20M -delete
      bash scripts/gpu_pmc_bench.sh > gpurun_out/pmc_bench.log 2>&1; tail -12 gpurun_out/pmc_bench.log
      find gpurun_out/pmc_bench -name "*.csv" -size +20M -delete
      timeout 300 tools/kbench hpf > gpurun_out/kbench_hpf_final.log 2>&1; tail -14 gpurun_out/kbench_hpf_final.log
      ( tools/hess_lab 16384 11008 0; tools/hess_lab 16384 4096 0 ) > gpurun_out/hess_lab_32x32.log 2>&1; cat gpurun_out/hess_lab_32x32.log ;;
    chol)
      timeout 300 python scripts/chol_time.py > gpurun_out/chol_time.log 2>&1; tail -12 gpurun_out/chol_time.log ;;
    choltrace)
      rm -rf "$R/gpurun_out/prof_chol"
      ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/prof_chol" -o c -- python "$R/scripts/chol_time.py" --form ${CHOL_FORM:-cabi1} ${CHOL_K:-11008} > "$R/gpurun_out/prof_chol.log" 2>&1 )
      echo "choltrace exit $?"; tail -2 gpurun_out/prof_chol.log
      python3 - gpurun_out/prof_chol/c_kernel_trace.csv <<'PY' | tee gpurun_out/chol_trace_summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last factorisation = from the last ifac_flip_in to the last ifac_flip_out
ins = [i for i, r in enumerate(rows) if "ifac_flip_in" in r["Kernel_Name"]]
outs = [i for i, r in enumerate(rows) if "ifac_flip_out" in r["Kernel_Name"]]
run = rows[ins[-1]: outs[-1] + 1]
t0, t1 = int(run[0]["Start_Timestamp"]), int(run[-1]["End_Timestamp"])
busy = collections.Counter(); cnt = collections.Counter()
for r in run:
    n = r["Kernel_Name"]
    if "f32gemm" in n:
        g = int(r["Grid_Size_X"]) // 256
        tile = "128" if "Li128ELi128" in n or "<128, 128" in n else "64"
        k = f"f32gemm<{tile}> {'NT' if ('Lb1' in n or 'true' in n) else 'NN'} " + ("big(>=256 wg)" if g >= 256 else "small(<256 wg)")
    else:
        k = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50]
    busy[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[k] += 1
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in run)
tot = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: tot += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
tot += ce - cs
print(f"span {(t1 - t0) / 1e6:.2f} ms, device busy {tot / 1e6:.2f} ms, idle {(t1 - t0 - tot) / 1e6:.2f} ms, kernels {len(run)}")
for k, v in busy.most_common(14):
    print(f"  {v / 1e6:8.2f} ms {cnt[k]:5d} x {k}   ({v / cnt[k] / 1e3:.1f} us each)")
PY
      find gpurun_out/prof_chol -name "*.csv" -size +20M -delete ;;
    prof)
      rm -rf "$R/gpurun_out/prof"
      ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -o r5 -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-extra-configs --no-per-layer > "$R/gpurun_out/prof_bench.log" 2> "$R/gpurun_out/prof_bench.err" )
      echo "prof exit $?"; find gpurun_out/prof -name "*kernel_stats*" | head -3
      find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete ;;
    bench2)
      for mode in layer exact; do
        INC_MI355X_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --mgpu-mode $mode --steps 2 --warmup 1 --samples 32 --seq 1024 --no-cpu-baseline --no-extra-configs --e2e-blocks 4 > gpurun_out/bench_n2_$mode.log 2> gpurun_out/bench_n2_$mode.err
        echo "bench --gpus 2 ($mode) exit $?"; tail -c 2500 gpurun_out/bench_n2_$mode.log; tail -8 gpurun_out/bench_n2_$mode.err
      done ;;
    rccl1)
      # RCCL bring-up on this 1-GPU box: a one-rank nccl group through the package's wrappers and both multi-GPU drivers, then
      # bench.py's own multi-GPU paths (both modes) over that group
      INC_MI355X_DIST_SINGLE_RANK=1 timeout 600 python scripts/rccl_single_rank.py > gpurun_out/rccl_single_rank.log 2>&1; echo "rccl_single_rank exit $?"; tail -8 gpurun_out/rccl_single_rank.log
      for mode in layer exact; do
        INC_MI355X_DIST_SINGLE_RANK=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 timeout 900 python bench.py --gpus 1 --mgpu-mode $mode --steps 2 --warmup 1 \
          --no-cpu-baseline --no-extra-configs --no-per-layer --e2e-blocks 4 > gpurun_out/bench_rccl1_$mode.log 2> gpurun_out/bench_rccl1_$mode.err
        echo "bench --gpus 1 over a one-rank RCCL group ($mode) exit $?"; tail -c 1500 gpurun_out/bench_rccl1_$mode.log; tail -5 gpurun_out/bench_rccl1_$mode.err
      done ;;
    *) echo "unknown action $1" ;;
  esac
  shift
done
