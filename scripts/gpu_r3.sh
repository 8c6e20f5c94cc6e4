#!/bin/bash
# One gpurun call of round 3.  usage: scripts/gpu_r3.sh [tests] [bench] [kbench <args>] [prof]   (any subset, in order)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD
while [[ $# -gt 0 ]]; do
  case $1 in
    tests)
      timeout 1700 python -m pytest tests -m gpu -q -s --timeout=900 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
      echo "pytest exit $?" | tee -a gpurun_out/pytest_gpu.log
      grep -E "^\[|passed|failed|Error|error" gpurun_out/pytest_gpu.log | tail -60
      timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" | tee -a gpurun_out/smoke.log
      tail -2 gpurun_out/smoke.log ;;
    newtests)
      timeout 1700 python -m pytest tests/test_gpu_baseline_parity.py tests/test_gpu_models.py -m gpu -q -s --timeout=900 -p no:cacheprovider -k "baseline or awq or row_sharded or sample_and_row or vs_oracle or staged or running_mean" > gpurun_out/pytest_new.log 2>&1
      echo "pytest(new) exit $?" | tee -a gpurun_out/pytest_new.log
      grep -E "^\[|passed|failed|Error|error|assert" gpurun_out/pytest_new.log | tail -80 ;;
    bench)
      timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2> gpurun_out/bench.err
      echo "bench exit $?"; tail -c 6000 gpurun_out/bench.log; tail -25 gpurun_out/bench.err ;;
    kbench)
      shift
      timeout 600 tools/kbench $1 > gpurun_out/kbench_$1.log 2>&1; echo "kbench $1 exit $?"; tail -60 gpurun_out/kbench_$1.log ;;
    gptqtests)
      timeout 1500 python -m pytest tests -m gpu -q -s --timeout=900 -p no:cacheprovider -k "gptq or fasterquant or column_loop or baseline" > gpurun_out/pytest_gptq.log 2>&1
      echo "pytest(gptq) exit $?" | tee -a gpurun_out/pytest_gptq.log
      grep -E "passed|failed|Error|error|assert" gpurun_out/pytest_gptq.log | tail -20 ;;
    sq)
      timeout 900 python -m pytest tests/test_gpu_sq.py tests/test_gpu_nf4.py -m gpu -q -s --timeout=900 -p no:cacheprovider > gpurun_out/pytest_sq.log 2>&1
      echo "pytest(sq) exit $?" | tee -a gpurun_out/pytest_sq.log
      grep -E "passed|failed|Error|error|assert|smoothquant" gpurun_out/pytest_sq.log | tail -30 ;;
    layer)
      timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_parity.py -m gpu -q -s --timeout=900 -p no:cacheprovider -k "layer_per_gpu or non_positive_definite" > gpurun_out/pytest_layer.log 2>&1
      echo "pytest(layer) exit $?" | tee -a gpurun_out/pytest_layer.log
      grep -E "passed|failed|Error|error|assert" gpurun_out/pytest_layer.log | tail -20 ;;
    bench2)
      # the N-rank driver on ONE GPU: two ranks share the device, gloo between them (RCCL refuses two ranks per device)
      for mode in layer exact; do
        INC_MI355X_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --mgpu-mode $mode --steps 2 --warmup 1 --samples 32 --seq 1024 --no-cpu-baseline --no-gemm --no-extra-configs --e2e-blocks 4 > gpurun_out/bench_n2_$mode.log 2> gpurun_out/bench_n2_$mode.err
        echo "bench --gpus 2 ($mode) exit $?"; tail -c 1500 gpurun_out/bench_n2_$mode.log; tail -8 gpurun_out/bench_n2_$mode.err
      done
      timeout 300 python bench.py --gpus 2 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_n2_nccl.log 2>&1; echo "bench --gpus 2 (nccl on one GPU: must refuse) exit $?"; tail -3 gpurun_out/bench_n2_nccl.log ;;
    gidx)
      timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -s --timeout=600 -p no:cacheprovider -k "g_idx or fused_gemm" > gpurun_out/pytest_gidx.log 2>&1
      echo "pytest(gidx) exit $?" | tee -a gpurun_out/pytest_gidx.log
      grep -E "passed|failed|Error|error|assert" gpurun_out/pytest_gidx.log | tail -12 ;;
    pmc)
      bash scripts/gpu_pmc.sh prof > gpurun_out/pmc_kbench.log 2>&1; echo "pmc kbench exit $?"; tail -12 gpurun_out/pmc_kbench.log ;;
    pmcbench)
      bash scripts/gpu_pmc_bench.sh > gpurun_out/pmc_bench.log 2>&1; echo "pmc bench exit $?"; tail -12 gpurun_out/pmc_bench.log ;;
    benchlayer1)
      timeout 600 python bench.py --steps 3 --warmup 1 --layer-on-one-gpu --no-cpu-baseline --no-gemm --no-extra-configs --no-e2e --no-per-layer > gpurun_out/bench_layer1.log 2> gpurun_out/bench_layer1.err
      echo "bench layer-on-one-gpu exit $?"; tail -c 2500 gpurun_out/bench_layer1.log; tail -5 gpurun_out/bench_layer1.err ;;
    qtrace)
      ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/prof_qlayer" -o q -- "$R/tools/kbench" qlayer > "$R/gpurun_out/prof_qlayer.log" 2>&1 )
      echo "qtrace exit $?"; ls gpurun_out/prof_qlayer | head ;;
    chol)
      timeout 300 python scripts/chol_trace.py 11008 > gpurun_out/chol_plain.log 2>&1; tail -7 gpurun_out/chol_plain.log
      timeout 300 python scripts/chol_trace.py 4096 > gpurun_out/chol_plain4096.log 2>&1; tail -4 gpurun_out/chol_plain4096.log
      timeout 900 python -m pytest tests -m gpu -q -s --timeout=900 -p no:cacheprovider -k "chol" > gpurun_out/pytest_chol.log 2>&1
      echo "pytest(chol) exit $?"; grep -E "passed|failed|Error|error|assert" gpurun_out/pytest_chol.log | tail -8 ;;
    choltrace)
      timeout 300 python scripts/chol_trace.py 11008 > gpurun_out/chol_plain.log 2>&1; tail -6 gpurun_out/chol_plain.log
      ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/prof_chol" -o c -- python "$R/scripts/chol_trace.py" 11008 > "$R/gpurun_out/prof_chol.log" 2>&1 )
      echo "choltrace exit $?"; tail -6 gpurun_out/prof_chol.log
      python3 - gpurun_out/prof_chol/c_kernel_trace.csv <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last run = after the last big gap: split on chol_diag sequences of 86
diag = [i for i, r in enumerate(rows) if "chol_diag" in r["Kernel_Name"]]
last = diag[-86]
# extend to the end of that run: up to the last kernel before the residual GEMMs (take until 40 kernels after the last diag)
run = rows[last - 3 : diag[-1] + 120]
t0, t1 = int(run[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in run)
busy = collections.Counter(); cnt = collections.Counter()
for r in run:
    n = r["Kernel_Name"]
    k = "chol_diag" if "chol_diag" in n else ("gemm" if ("Cijk" in n or "gemm" in n.lower()) else n[:40])
    busy[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[k] += 1
# union of busy intervals
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in run)
tot = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: tot += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
tot += ce - cs
print(f"span {(t1 - t0) / 1e6:.2f} ms, device busy {tot / 1e6:.2f} ms, idle {(t1 - t0 - tot) / 1e6:.2f} ms, kernels {len(run)}")
for k, v in busy.most_common(12):
    print(f"  {v / 1e6:8.2f} ms {cnt[k]:5d} x {k}")
PY
      ;;
    benchab)
      for v in 1 0 1 0; do
        env "${ABVAR:-INC_MI355X_CHOL_LOOKAHEAD}=$v" timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-gemm --no-extra-configs --no-e2e --no-per-layer > gpurun_out/bench_ab.log 2> gpurun_out/bench_ab.err
        python3 - $v <<'PY'
import json, sys
for line in open("gpurun_out/bench_ab.log"):
    if line.startswith("{"):
        d = json.loads(line)
        print("variant", sys.argv[1], "ms/step", d["ms_per_step"], {k.replace("quantize_layer_", "ql_"): v["avg_ms"] for k, v in d["kernel_breakdown"].items()})
PY
      done ;;
    awqtests)
      timeout 1200 python -m pytest tests -m gpu -q -s --timeout=900 -p no:cacheprovider -k "awq" > gpurun_out/pytest_awq.log 2>&1
      echo "pytest(awq) exit $?" | tee -a gpurun_out/pytest_awq.log
      grep -E "passed|failed|Error|error|assert" gpurun_out/pytest_awq.log | tail -12 ;;
    awq)
      timeout 600 python tools/awq_block_prof.py 2 > gpurun_out/awq_phases.log 2> gpurun_out/awq_phases.err; echo "awq phases exit $?"; cat gpurun_out/awq_phases.log; tail -3 gpurun_out/awq_phases.err
      ( cd /tmp && INC_MI355X_AWQ_TIMING=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_awq" -o awq -- python "$R/tools/awq_block_prof.py" 1 > "$R/gpurun_out/prof_awq.log" 2>&1 )
      echo "awq prof exit $?"; f=$(find gpurun_out/prof_awq -name "*kernel_stats*" | head -1); python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", round(tot / 1e6, 1))
for r in rows[:25]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.1f} ms {int(r["Calls"]):6d} calls  {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:110]}')
PY
      ;;
    prof)
      rm -rf "$R/gpurun_out/prof"
      ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -o r3 -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-extra-configs --no-per-layer > "$R/gpurun_out/prof_bench.log" 2> "$R/gpurun_out/prof_bench.err" )
      echo "prof exit $?"; find gpurun_out/prof -name "*kernel_stats*" | head -3 ;;
  esac
  shift
done
