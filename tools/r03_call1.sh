#!/bin/bash
# Round 3, first GPU call (~8 minutes of box time): decide the decode-step architecture by measurement.
#   1. tools/persist_probe            persistent step (flag rows + sc1 hand-offs + run-ahead weights) vs the launch chain
#   2. tools/pipelined_launch_probe   software-pipelined launches on 1 / 2 / 3 streams of one graph (parked in round 2)
#   3. PMC passes (FETCH_SIZE, WRITE_SIZE) on the dominant kernel in its wide-value shapes -> roofline.traffic
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
rocm-smi --showclocks 2>/dev/null | head -20 > gpurun_out/r03_clocks.txt
timeout 180 tools/persist_probe 30 > gpurun_out/r03_persist_probe.log 2>&1; echo "persist_probe rc=$?"; cat gpurun_out/r03_persist_probe.log
timeout 180 tools/pipelined_launch_probe 30 > gpurun_out/r03_pipelined_launch_probe.log 2>&1; echo "pipelined rc=$?"; cat gpurun_out/r03_pipelined_launch_probe.log
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_$c
    timeout 240 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_$c -- python tools/pmc_target.py --wide > gpurun_out/pmc_$c.log 2>&1
    find gpurun_out/pmc_$c -name "*counter_collection.csv" -exec cp {} gpurun_out/r03_pmc_${c}_counter_collection_wide.csv \;
done
python tools/pmc_summary.py gpurun_out/r03_pmc_FETCH_SIZE_counter_collection_wide.csv gpurun_out/r03_pmc_WRITE_SIZE_counter_collection_wide.csv \
    gemv_lnf gpurun_out/r03_pmc_dominant_kernel_wide.json --wide | tail -12
