#!/bin/bash
# One gpurun call: parity tests + smoke + short bench (+ optional rocprof).  Everything is logged under gpurun_out/.
# usage: scripts/gpu_check.sh [tests|bench|prof|all]
set -u
what=${1:-all}
mkdir -p gpurun_out
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx|Compute Unit" | head -6 > gpurun_out/device.txt
if [[ $what == tests || $what == all ]]; then
  timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
  tail -40 gpurun_out/pytest_gpu.log
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
  tail -5 gpurun_out/smoke.log
fi
if [[ $what == bench || $what == all ]]; then
  timeout 480 python bench.py --steps 2 --warmup 1 > gpurun_out/bench.log 2> gpurun_out/bench.err
  echo "bench exit $?"; tail -3 gpurun_out/bench.log; tail -5 gpurun_out/bench.err
fi
if [[ $what == prof || $what == all ]]; then
  ( cd /tmp && timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o r1 -- python "$OLDPWD/bench.py" --steps 1 --warmup 1 --no-cpu-baseline > "$OLDPWD/gpurun_out/prof_bench.log" 2> "$OLDPWD/gpurun_out/prof_bench.err" )
  echo "prof exit $?"
  find gpurun_out/prof -name "*stats*" | head
fi
