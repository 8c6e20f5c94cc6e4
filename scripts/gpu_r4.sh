#!/bin/bash
# One gpurun call of round 4.  usage: scripts/gpu_r4.sh <action> [<action> ...]   (run in order; `kbench`, `pytest` and `benchab` take one argument)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD
BENCH_MIN="--no-cpu-baseline --no-gemm --no-extra-configs --no-e2e --no-per-layer"
while [[ $# -gt 0 ]]; do
  case $1 in
    tests)
      timeout 1700 python -m pytest tests -m gpu -q -s --timeout=900 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
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
    timeline)
      # one-step timelines (kernel trace + copies) of the bench step, late solve off / on
      for late in 0 1; do
        rm -rf "$R/gpurun_out/tl_late$late"
        ( cd /tmp && INC_MI355X_TRACE_RANGES=1 INC_MI355X_GPTQ_LATE_SOLVE=$late timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv \
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
    chol)
      timeout 300 python scripts/chol_time.py > gpurun_out/chol_time.log 2>&1; tail -12 gpurun_out/chol_time.log ;;
    prof)
      rm -rf "$R/gpurun_out/prof"
      ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -o r4 -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-extra-configs --no-per-layer > "$R/gpurun_out/prof_bench.log" 2> "$R/gpurun_out/prof_bench.err" )
      echo "prof exit $?"; find gpurun_out/prof -name "*kernel_stats*" | head -3
      find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete ;;
    bench2)
      for mode in layer exact; do
        INC_MI355X_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --mgpu-mode $mode --steps 2 --warmup 1 --samples 32 --seq 1024 --no-cpu-baseline --no-extra-configs --e2e-blocks 4 > gpurun_out/bench_n2_$mode.log 2> gpurun_out/bench_n2_$mode.err
        echo "bench --gpus 2 ($mode) exit $?"; tail -c 2500 gpurun_out/bench_n2_$mode.log; tail -8 gpurun_out/bench_n2_$mode.err
      done ;;
    *) echo "unknown action $1" ;;
  esac
  shift
done
