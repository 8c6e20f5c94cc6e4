#!/bin/bash
# One gpurun call of round 6.  usage: scripts/gpu_r6.sh <action> [<action> ...]   (run in order; `kbench`, `pytest` and `py` take one argument)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD
BENCH_MIN="--no-cpu-baseline --no-gemm --no-extra-configs --no-e2e --no-per-layer"
while [[ $# -gt 0 ]]; do
  case $1 in
    tests)
      timeout 2400 python -m pytest tests -m gpu -q -s --timeout=1200 -p no:cacheprovider --durations=25 > gpurun_out/pytest_gpu.log 2>&1
      echo "pytest exit $?" | tee -a gpurun_out/pytest_gpu.log
      grep -E "passed|failed|Error|error" gpurun_out/pytest_gpu.log | tail -30
      timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" | tee -a gpurun_out/smoke.log
      tail -2 gpurun_out/smoke.log ;;
    pytest)
      shift
      timeout 1800 python -m pytest tests -m gpu -q -s --timeout=1200 -p no:cacheprovider -k "$1" > gpurun_out/pytest_sel.log 2>&1
      echo "pytest -k '$1' exit $?" | tee -a gpurun_out/pytest_sel.log
      grep -E "^\[|passed|failed|Error|error|assert" gpurun_out/pytest_sel.log | cut -c1-600 | tail -60 ;;
    bench)
      timeout 1200 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2> gpurun_out/bench.err
      echo "bench exit $?"; tail -c 9000 gpurun_out/bench.log; tail -30 gpurun_out/bench.err ;;
    benchmin)
      timeout 600 python bench.py --steps 3 --warmup 1 $BENCH_MIN > gpurun_out/bench_min.log 2> gpurun_out/bench_min.err
      echo "bench exit $?"; tail -c 4000 gpurun_out/bench_min.log; tail -10 gpurun_out/bench_min.err ;;
    kbench)
      shift
      timeout 600 tools/kbench $1 > gpurun_out/kbench_$1.log 2>&1; echo "kbench $1 exit $?"; tail -70 gpurun_out/kbench_$1.log ;;
    py)
      shift
      n=$(basename "$1" .py)
      timeout 900 python $1 > gpurun_out/$n.log 2>&1; echo "$1 exit $?"; tail -60 gpurun_out/$n.log ;;
    prof)
      rm -rf "$R/gpurun_out/prof"
      ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -o r6 -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-extra-configs --no-per-layer > "$R/gpurun_out/prof_bench.log" 2> "$R/gpurun_out/prof_bench.err" )
      echo "prof exit $?"; f=$(find gpurun_out/prof -name "*kernel_stats*" | head -1); cp "$f" gpurun_out/bench_kernel_stats.csv; head -8 gpurun_out/bench_kernel_stats.csv | cut -c1-220
      find gpurun_out/prof -name "*.csv" -size +20M -delete ;;
    pmc)
      bash scripts/gpu_pmc_bench.sh > gpurun_out/pmc_bench.log 2>&1; tail -12 gpurun_out/pmc_bench.log
      find gpurun_out/pmc_bench -name "*.csv" -size +20M -delete ;;
    bench2)
      for mode in layer exact; do
        INC_MI355X_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --mgpu-mode $mode --steps 2 --warmup 1 --samples 32 --seq 1024 --no-cpu-baseline --no-extra-configs --no-per-layer --e2e-blocks 4 \
          --detail-file gpurun_out/bench_n2_${mode}_detail.json > gpurun_out/bench_n2_$mode.log 2> gpurun_out/bench_n2_$mode.err
        echo "bench --gpus 2 ($mode) exit $?"; tail -c 2500 gpurun_out/bench_n2_$mode.log; tail -8 gpurun_out/bench_n2_$mode.err
      done ;;
    *) echo "unknown action $1" ;;
  esac
  shift
done
