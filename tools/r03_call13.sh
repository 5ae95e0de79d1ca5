#!/bin/bash
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp && rm -rf profp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profp -- python $GRAFT_REPO_ROOT/tools/bench_prefill.py > $GRAFT_REPO_ROOT/gpurun_out/r03_prof_prefill.log 2>&1
find /tmp/profp -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r03_prefill_kernel_stats.csv \;
python - <<'PY'
import csv, os
rows = list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r03_prefill_kernel_stats.csv")))
for r in rows[:12]:
    print(r["Name"][:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
tail -3 $GRAFT_REPO_ROOT/gpurun_out/r03_prof_prefill.log
