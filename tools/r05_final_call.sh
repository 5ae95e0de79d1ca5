#!/bin/bash
# Final call of round 5 on the final tree: suite -> smoke -> bench.py as the driver runs it (a warm-up job, then a timed 20-second
# job in the same process) -> rocprofv3 of --roofline-only -> the window's glue (conditioner / prefill in situ).
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 700 python -u -m pytest tests -q -m gpu -p no:cacheprovider --durations=6 > gpurun_out/r05f_gpu_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/r05f_gpu_tests.log | tail -3
timeout 300 python -u __graft_entry__.py smoke > gpurun_out/r05f_smoke.log 2>&1; tail -2 gpurun_out/r05f_smoke.log
JB_BENCH_TIMELINE=1 timeout 700 python -u bench.py --gpus 1 --seconds 20 --steps 1 --warmup 1 > gpurun_out/r05f_bench_20s_1gpu.json 2> gpurun_out/r05f_bench_20s_1gpu.err
cut -c1-600 gpurun_out/r05f_bench_20s_1gpu.json
python - <<PY
import json
d = json.load(open("gpurun_out/r05f_bench_20s_1gpu.json"))
b = d["breakdown"]
print("value", d["value"], "steps", b["step_seconds"], "L2/L1/L0 done at", b.get("level2_codes_done_at_s"), b.get("level1_codes_done_at_s"), b.get("level0_codes_done_at_s"),
      "form", b.get("level0_launch_form"), b.get("level0_in_situ_comparison_ms_per_step"))
print("roofline", d.get("roofline")); print("cpu", d.get("cpu_baseline"))
PY
cd /tmp && rm -rf /tmp/prof_roof && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_roof -- \
    python -u $GRAFT_REPO_ROOT/bench.py --roofline-only > $GRAFT_REPO_ROOT/gpurun_out/r05f_roofline_only_stdout.json 2> /tmp/prof_roof.err
cd $GRAFT_REPO_ROOT
cp $(find /tmp/prof_roof -name "*kernel_stats.csv" | head -1) gpurun_out/r05f_roofline_only_kernel_stats.csv 2>/dev/null
head -4 gpurun_out/r05f_roofline_only_kernel_stats.csv | cut -c1-160
timeout 200 python -u tools/window_glue.py --steps 512 --windows 2 > gpurun_out/r05f_window_glue.log 2>&1; tail -16 gpurun_out/r05f_window_glue.log
