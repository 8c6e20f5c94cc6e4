#!/bin/bash
# PMC passes over the bench workload itself (1 step): HBM-side bytes per launch of the Hessian syrk kernel.
# FETCH_SIZE and WRITE_SIZE cannot share a pass (TCC slots); only --kernel-trace goes with --pmc.
set -u
mkdir -p gpurun_out/pmc_bench
export TMPDIR=/tmp
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp
for pass in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$ROOT/gpurun_out/pmc_bench/$pass" -o pmc -- python "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-gemm --no-e2e --no-extra-configs --no-per-layer > "$ROOT/gpurun_out/pmc_bench/$pass.log" 2>&1
  echo "pass [$pass] exit $?"
done
python - <<'PY'
import csv, collections, glob, json, os
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
agg = collections.defaultdict(lambda: collections.defaultdict(list))
extra = collections.defaultdict(list)
for d in glob.glob(os.path.join(root, "gpurun_out/pmc_bench/*/pmc_counter_collection.csv")):
    for r in csv.DictReader(open(d)):
        name = r["Kernel_Name"]
        if "hessian_syrk_tr_256_multi" in name or "hessian_syrk_16bit_256_multi" in name:
            # the bench's Llama block launch: 3 x 136 + 946 = 1354 tiles of 512 threads (1280 + 74 x 3 = 1502 workgroups with the tail split)
            key = "hessian_multi_K4096+4096+4096+11008" if int(r["Grid_Size"]) in (512 * 1354, 512 * 1502) else "hessian_multi_other"
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        elif "hessian_tail_finalize" in name:  # second kernel of the same C-ABI call: its bytes belong to the launch
            extra[r["Counter_Name"]].append(float(r["Counter_Value"]))
        elif "hessian_syrk_tr_256" in name or "hessian_syrk_16bit_256" in name:
            key = "hessian_K11008" if int(r["Grid_Size"]) > 512 * 200 else "hessian_K4096"
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in agg.items():
    add = k == "hessian_multi_K4096+4096+4096+11008"
    f = (sum(c["FETCH_SIZE"]) + (sum(extra["FETCH_SIZE"]) if add else 0.0)) / max(len(c["FETCH_SIZE"]), 1) * 1024.0 if c.get("FETCH_SIZE") else None
    w = (sum(c["WRITE_SIZE"]) + (sum(extra["WRITE_SIZE"]) if add else 0.0)) / max(len(c["WRITE_SIZE"]), 1) * 1024.0 if c.get("WRITE_SIZE") else None
    out[k] = dict(fetch_size_bytes=f, write_size_bytes=w, launches=len(c.get("FETCH_SIZE", [])),
                  traffic_bytes_per_launch=(2.0 * f + w) if f is not None and w is not None else None,
                  note="traffic = 2*FETCH_SIZE + WRITE_SIZE (KB -> bytes): gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X guide)")
json.dump(out, open(os.path.join(root, "gpurun_out/pmc_bench/bench_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
