#!/bin/bash
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 120 tools/persist_probe 30 2>&1 | grep -E "^chain" > gpurun_out/r03_persist_probe_v4.log; cat gpurun_out/r03_persist_probe_v4.log
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_baseline_configs.py -x -q -s -m gpu \
  -k "5b_head_size or gemv_ln_folded or config5" > gpurun_out/r03_5b_tests.log 2>&1
echo "rc=$?"; tail -8 gpurun_out/r03_5b_tests.log
timeout 300 python tools/bench_engine.py 5b --batch 3 --steps 32 > gpurun_out/r03_bench_engine_5b.log 2>&1; cat gpurun_out/r03_bench_engine_5b.log
cd /tmp && rm -rf prof5b && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof5b -- python $GRAFT_REPO_ROOT/tools/bench_engine.py 5b --batch 3 --steps 8 --eager > $GRAFT_REPO_ROOT/gpurun_out/r03_prof5b.log 2>&1
find /tmp/prof5b -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r03_5b_kernel_stats.csv \;
head -12 $GRAFT_REPO_ROOT/gpurun_out/r03_5b_kernel_stats.csv | cut -c1-200
