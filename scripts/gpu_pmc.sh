#!/bin/bash
# rocprofv3 PMC passes over the kernel-only workload (tools/kbench prof): HBM bytes and MFMA busy per dispatch.
# Separate passes (TCC slot limits; gpurun refuses --pmc together with the trace domains other than kernel-trace).
set -u
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY"; do
  tag=$(echo "$pass" | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$ROOT/gpurun_out/pmc/$tag" -o pmc -- "$ROOT/tools/kbench" prof > "$ROOT/gpurun_out/pmc/$tag.log" 2>&1
  echo "pass [$pass] exit $?"
done
find "$ROOT/gpurun_out/pmc" -name "*.csv" | head -20
