#!/bin/bash
# rocprofv3 kernel stats of the inverse-Cholesky factorisation alone (scripts/chol_time.py K)
set -u
R="${GRAFT_REPO_ROOT:-$PWD}"
export TMPDIR=/tmp
rm -rf "$R/gpurun_out/prof_chol"; mkdir -p "$R/gpurun_out/prof_chol"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_chol" -o c -- python "$R/scripts/chol_time.py" ${1:-11008} > "$R/gpurun_out/prof_chol/run.log" 2>&1 )
echo "exit $?"; tail -3 "$R/gpurun_out/prof_chol/run.log"
python - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/prof_chol/**/*kernel_stats.csv",recursive=True)
rows=list(csv.DictReader(open(f[0])))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
print("total ms",tot/1e6, "kernels", sum(int(r["Calls"]) for r in rows))
for r in rows[:18]: print(round(int(r["TotalDurationNs"])/1e6,2),"ms",r["Calls"],round(float(r["AverageNs"])/1e3,1),"us",r["Name"][:100])
PY
