#!/bin/bash
# rocprofv3 PMC passes (separate passes; only --kernel-trace beside --pmc) over `tools/kbench colloop`: the trailing update of the GPTQ
# column loop -- strip form (gptq_lazy_update_v4_kernel) against the tile form (gptq_lazy_update_v3_kernel<128>), whole-range launches
# at 4096^2 / 11008 x 4096 / 4096 x 11008.
set -u
export TMPDIR=/tmp
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
rm -rf "$ROOT/gpurun_out/pmc_lazy"; mkdir -p "$ROOT/gpurun_out/pmc_lazy"
cd /tmp
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo "$pass" | tr ' ' '_' | cut -c1-40)
  timeout 400 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$ROOT/gpurun_out/pmc_lazy/$tag" -o pmc -- "$ROOT/tools/kbench" colloop > "$ROOT/gpurun_out/pmc_lazy/$tag.log" 2>&1
  echo "pass [$pass] exit $?"
done
python - <<'PY'
import collections, csv, glob, json, os, re
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in glob.glob(os.path.join(root, "gpurun_out/pmc_lazy/*/pmc_counter_collection.csv")):
    for r in csv.DictReader(open(d)):
        if "gptq_lazy_update_v" not in r["Kernel_Name"]:
            continue
        name = re.sub(r"^void |\(anonymous namespace\)::|\(.*$", "", r["Kernel_Name"])
        if int(r["Grid_Size"]) < 256 * 256:  # whole-range launches of the big shapes only (>= 256 workgroups)
            continue
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"dispatches": max(len(v) for v in cs.values())} for k, cs in agg.items()}
for k, v in out.items():
    if v.get("SQ_WAVE_CYCLES"):
        v["share_issuing"] = round(v.get("SQ_ACTIVE_INST_ANY", 0) / v["SQ_WAVE_CYCLES"], 3)
        v["share_waiting"] = round(v.get("SQ_WAIT_ANY", 0) / v["SQ_WAVE_CYCLES"], 3)
        v["share_issue_stalled"] = round(v.get("SQ_WAIT_INST_ANY", 0) / v["SQ_WAVE_CYCLES"], 3)
    if v.get("GRBM_GUI_ACTIVE") and v.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        v["mfma_busy_per_simd_over_kernel_cycles"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (v["GRBM_GUI_ACTIVE"] / 8.0), 3)
    if v.get("FETCH_SIZE") is not None and v.get("WRITE_SIZE") is not None:
        v["traffic_mb_2fetch_plus_write"] = round((2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0 / 1e6, 1)
json.dump(out, open(os.path.join(root, "gpurun_out/pmc_lazy/pmc_per_dispatch_means.json"), "w"), indent=1, sort_keys=True)
for k, v in sorted(out.items()):
    print(k, {c: (round(x, 3) if isinstance(x, float) else x) for c, x in v.items()})
PY
find "$ROOT/gpurun_out/pmc_lazy" -name "*.csv" -size +5M -delete
