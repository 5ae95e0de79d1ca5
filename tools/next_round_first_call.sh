#!/bin/bash
# First GPU call of the next round (run through gpurun from the repo root; ~9 minutes of box time):
#   1. the whole GPU suite on the tree the round starts from (the wide-value tree was validated file by file, never in
#      one run: profiles/README.md, r02_wide_* rows);
#   2. the PMC passes on the dominant kernel in its wide-value shapes (roofline.traffic is null until this exists);
#   3. the driver's bench command against a short wall budget (first full-length measurement of the wide-value tree).
# Then (separate, ~1 minute): `git checkout wip/pipelined-launch -- tools/pipelined_launch_probe.hip`, build it with hipcc and run
# it -- DESIGN.md section 8 item 1 says what its numbers decide.
# Outputs land in gpurun_out/; copy what is to be judged into profiles/ (r03_*).
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 420 python -m pytest tests -q -x -m gpu > gpurun_out/r03_gpu_tests.log 2>&1; tail -3 gpurun_out/r03_gpu_tests.log
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_$c
    timeout 120 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_$c -- python tools/pmc_target.py --wide > gpurun_out/pmc_$c.log 2>&1
    find gpurun_out/pmc_$c -name "*counter_collection.csv" -exec cp {} gpurun_out/r03_pmc_${c}_counter_collection_wide.csv \;
done
python tools/pmc_summary.py gpurun_out/r03_pmc_FETCH_SIZE_counter_collection_wide.csv gpurun_out/r03_pmc_WRITE_SIZE_counter_collection_wide.csv \
    gemv_lnf gpurun_out/r03_pmc_dominant_kernel_wide.json --wide | tail -8
JB_BENCH_BUDGET_S=420 JB_BENCH_TIMELINE=1 timeout 560 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench_full_1gpu.json 2> gpurun_out/r03_bench_full_1gpu.err
cut -c1-600 gpurun_out/r03_bench_full_1gpu.json
