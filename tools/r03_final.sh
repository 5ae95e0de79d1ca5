#!/bin/bash
# Final GPU call of round 3: the GPU suite as two workers on the one GPU, smoke(), rocprofv3 summary of bench.py --roofline-only.
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
JB_TEST_GPU_MEM_FRACTION=0.45 timeout 430 python -m pytest tests -m gpu -q -n 2 -p no:cacheprovider > gpurun_out/r03_gpu_tests_final.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/r03_gpu_tests_final.log | cut -c1-300
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a gpurun_out/r03_gpu_tests_final.log
R=$PWD
cd /tmp && rm -rf profr && timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profr -- python $R/bench.py --roofline-only > $R/gpurun_out/r03_roofline_only_stdout.json 2> $R/gpurun_out/r03_roofline_only.err
find /tmp/profr -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r03_roofline_only_kernel_stats.csv \;
head -6 $R/gpurun_out/r03_roofline_only_kernel_stats.csv | cut -c1-160; cut -c1-600 $R/gpurun_out/r03_roofline_only_stdout.json
