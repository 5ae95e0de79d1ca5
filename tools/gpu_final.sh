#!/bin/bash
# Round 2, final GPU call: full suite, roofline-only kernel stats, PMC passes for the dominant kernel, the full bench line.
export PYTHONPATH=$PWD
R=$GRAFT_REPO_ROOT
echo "== gpu suite =="; timeout 1000 python -m pytest tests -m gpu -q --timeout 400 -p no:cacheprovider 2>&1 | tail -6 | tee gpurun_out/r02_gpu_tests.log
echo "== smoke =="; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee -a gpurun_out/r02_gpu_tests.log
cd /tmp && export TMPDIR=/tmp
echo "== rocprofv3 --kernel-trace --stats: bench.py --roofline-only =="
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02_roofline -- python $R/bench.py --roofline-only > $R/gpurun_out/r02_roofline_only_stdout.log 2>&1
tail -1 $R/gpurun_out/r02_roofline_only_stdout.log | cut -c1-600
echo "== PMC passes (one counter per run) =="
for cnt in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $cnt --output-format csv -d $R/gpurun_out/pmc_r02_$cnt -- python $R/tools/pmc_target.py > $R/gpurun_out/pmc_r02_$cnt.log 2>&1
  tail -1 $R/gpurun_out/pmc_r02_$cnt.log
done
cd $R
F=$(find gpurun_out/pmc_r02_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find gpurun_out/pmc_r02_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python tools/pmc_summary.py "$F" "$W" gemv_lnf_kernel gpurun_out/r02_pmc_dominant_kernel.json | tail -12
S=$(find gpurun_out/prof_r02_roofline -name "*kernel_stats.csv" | head -1); cp "$S" gpurun_out/r02_roofline_only_kernel_stats.csv; head -8 gpurun_out/r02_roofline_only_kernel_stats.csv | cut -c1-200
echo "== full bench (driver command, wall budget 800 s) =="
JB_BENCH_BUDGET_S=800 JB_BENCH_TIMELINE=1 timeout 1100 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r02_bench_full_stderr.log | tail -1 > gpurun_out/r02_bench_full_1gpu.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02_bench_full_1gpu.json"))
b = d["breakdown"]
print({k: d[k] for k in ("value", "steps", "warmup", "ms_per_step", "steps_requested")})
print("roofline", d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["roofline"]["traffic"])
print({k: v for k, v in b.items() if k != "timeline"})
print("cpu", d["cpu_baseline"]["kind"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
print("level-0 windows:", [(x[1], x[2], x[3]) for x in b.get("timeline", []) if x[0] == 0][:4], "...")
PY
