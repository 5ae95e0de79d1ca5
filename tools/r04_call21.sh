#!/bin/bash
# Round 4, GPU call 21: the PMC passes (HBM bytes per launch of the dominant kernel) repeated on this round's library -- one counter
# per run, no trace domains, as MI355X_MICROARCH.md prescribes.
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_$c
    timeout 100 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_$c -- python tools/pmc_target.py --wide > gpurun_out/pmc_$c.log 2>&1
    find gpurun_out/pmc_$c -name "*counter_collection.csv" -exec cp {} gpurun_out/r04_pmc_${c}_counter_collection_wide.csv \;
    rm -rf gpurun_out/pmc_$c
done
python tools/pmc_summary.py gpurun_out/r04_pmc_FETCH_SIZE_counter_collection_wide.csv gpurun_out/r04_pmc_WRITE_SIZE_counter_collection_wide.csv \
    gemv_lnf gpurun_out/r04_pmc_dominant_kernel_wide.json --wide | tail -12
echo done
