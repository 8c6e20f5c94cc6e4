#!/bin/bash
# A/B of environment settings inside ONE gpurun call (same box, interleaved): ROUNDS=2 scripts/ab_env.sh "VAR=a" "VAR=b" ["-" = nothing set]
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
R=${ROUNDS:-2}
MIN="--no-cpu-baseline --no-gemm --no-extra-configs --no-e2e --no-per-layer"
for r in $(seq 1 $R); do
  i=0
  for kv in "$@"; do
    i=$((i + 1))
    if [[ "$kv" == "-" ]]; then
      timeout 600 python bench.py --steps 4 --warmup 1 $MIN --detail-file gpurun_out/abe_${i}_$r.json > gpurun_out/abe_${i}_$r.log 2> gpurun_out/abe_${i}_$r.err
    else
      env $kv timeout 600 python bench.py --steps 4 --warmup 1 $MIN --detail-file gpurun_out/abe_${i}_$r.json > gpurun_out/abe_${i}_$r.log 2> gpurun_out/abe_${i}_$r.err
    fi
    python - "$i" "$r" "$kv" <<'PY'
import json, sys
leg, r, kv = sys.argv[1:4]
try:
    d = json.load(open(f"gpurun_out/abe_{leg}_{r}.json"))
    kb = d.get("kernel_breakdown", {})
    print(f"round {r} [{kv}]: ms_per_step {d['ms_per_step']:.2f}  " + "  ".join(f"{k.replace('quantize_layer_', 'ql_').replace('hessian_multi_K4096+4096+4096+11008', 'hess')} {v['avg_ms']:.2f}" for k, v in kb.items()))
except Exception as e:
    print(leg, r, "failed", e)
PY
  done
done
