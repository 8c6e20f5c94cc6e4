#!/bin/bash
# A/B of several builds of the library inside ONE gpurun call (same box, interleaved): ROUNDS=2 scripts/ab_lib.sh <A.so> <B.so> [<C.so> ...]
# Each leg copies its .so over neural_compressor_amd/libinc_mi355x.so and runs the minimal bench (4 steps); the product .so is restored at the end.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
R=${ROUNDS:-2}
cp neural_compressor_amd/libinc_mi355x.so /tmp/product.so
MIN="--no-cpu-baseline --no-gemm --no-extra-configs --no-e2e --no-per-layer"
for r in $(seq 1 $R); do
  i=0
  for so in "$@"; do
    i=$((i + 1))
    cp "$so" neural_compressor_amd/libinc_mi355x.so
    timeout 600 python bench.py --steps 4 --warmup 1 $MIN --detail-file gpurun_out/ab_${i}_$r.json > gpurun_out/ab_${i}_$r.log 2> gpurun_out/ab_${i}_$r.err
    python - "$i" "$r" "$so" <<'PY'
import json, sys
leg, r, so = sys.argv[1:4]
try:
    d = json.load(open(f"gpurun_out/ab_{leg}_{r}.json"))
    kb = d.get("kernel_breakdown", {})
    print(f"round {r} {so}: ms_per_step {d['ms_per_step']:.2f}  " + "  ".join(f"{k.replace('quantize_layer_', 'ql_').replace('hessian_multi_K4096+4096+4096+11008', 'hess')} {v['avg_ms']:.2f}" for k, v in kb.items()))
except Exception as e:
    print(leg, r, "failed", e)
PY
  done
done
cp /tmp/product.so neural_compressor_amd/libinc_mi355x.so
