#!/bin/bash
# FETCH_SIZE of the batched Hessian launch per tile form (tools/kbench hpf runs the product tile and its A/B partners back to back):
# one rocprofv3 --pmc pass (only --kernel-trace beside it), per-kernel means by template arguments and grid size.
set -u
mkdir -p gpurun_out/pmc_hpf
export TMPDIR=/tmp
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp
timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$ROOT/gpurun_out/pmc_hpf/FETCH_SIZE" -o pmc -- "$ROOT/tools/kbench" hpf > "$ROOT/gpurun_out/pmc_hpf/run.log" 2>&1
echo "pass exit $?"
python3 - <<'PY'
import csv, collections, glob, os, re
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
agg = collections.defaultdict(list)
for d in glob.glob(os.path.join(root, "gpurun_out/pmc_hpf/*/pmc_counter_collection.csv")):
    for r in csv.DictReader(open(d)):
        n = r["Kernel_Name"]
        if "hessian_syrk_tr_256_multi" in n and r["Counter_Name"] == "FETCH_SIZE":
            m = re.search(r"multi_kernel<([^>]*)>", n)
            agg[(m.group(1) if m else n[:60], int(r["Grid_Size"]) // 512)].append(float(r["Counter_Value"]) * 1024.0)
with open(os.path.join(root, "gpurun_out/pmc_hpf/summary.txt"), "w") as f:
    for k, v in sorted(agg.items()):
        line = f"tile <{k[0]}> grid {k[1]} workgroups: {len(v)} launches, 2 x FETCH_SIZE = {2 * sum(v) / len(v) / 1e9:.2f} GB per launch (min {2 * min(v) / 1e9:.2f}, max {2 * max(v) / 1e9:.2f})"
        print(line); f.write(line + "\n")
PY
find "$ROOT/gpurun_out/pmc_hpf" -name "*.csv" -size +20M -delete
