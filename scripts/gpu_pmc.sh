#!/bin/bash
# rocprofv3 PMC passes over the kernel-only workload (tools/kbench prof): HBM bytes and MFMA busy per dispatch.
# Separate passes (TCC slot limits; gpurun refuses --pmc together with the trace domains other than kernel-trace).
set -u
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
MODE="${1:-prof}"   # tools/kbench mode to profile: prof (default) | profstrip
rm -rf "$ROOT/gpurun_out/pmc"; mkdir -p "$ROOT/gpurun_out/pmc"
cd /tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS"; do
  tag=$(echo "$pass" | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$ROOT/gpurun_out/pmc/$tag" -o pmc -- "$ROOT/tools/kbench" $MODE > "$ROOT/gpurun_out/pmc/$tag.log" 2>&1
  echo "pass [$pass] exit $?"
done
find "$ROOT/gpurun_out/pmc" -name "*.csv" | head -20

python - <<'PY'
import collections, csv, glob, json, os, re
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in glob.glob(os.path.join(root, "gpurun_out/pmc/*/pmc_counter_collection.csv")):
    for r in csv.DictReader(open(d)):
        name = re.sub(r"^void |\(anonymous namespace\)::|\(.*$", "", r["Kernel_Name"])
        agg[f"{name} grid={r['Grid_Size']}"][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"dispatches": max(len(v) for v in cs.values())} for k, cs in agg.items()}
json.dump(out, open(os.path.join(root, "gpurun_out/pmc/pmc_per_dispatch_means.json"), "w"), indent=1, sort_keys=True)
for k, v in sorted(out.items()):
    print(k, {c: round(x, 1) for c, x in v.items()})
PY
