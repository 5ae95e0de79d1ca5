#!/bin/bash
# Round 4, GPU call 9: kernel statistics of the prefill / conditioner micro-benchmark, then the 20-second job on the final tree.
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
O=$PWD/gpurun_out
cd /tmp && rm -rf /tmp/prof_pf && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pf -- python -u $GRAFT_REPO_ROOT/tools/bench_prefill.py > $O/r04_prof_prefill.log 2>&1
f=$(find /tmp/prof_pf -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r04_prefill_kernel_stats.csv && head -9 $O/r04_prefill_kernel_stats.csv | cut -c1-150
cd $GRAFT_REPO_ROOT
JB_BENCH_BUDGET_S=480 JB_BENCH_TIMELINE=1 timeout 700 python -u bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_bench_full_1gpu_final2.json 2> $O/r04_bench_full_1gpu_final2.err; cut -c1-330 $O/r04_bench_full_1gpu_final2.json; grep -i "timed out\|fell back\|Traceback" -A3 $O/r04_bench_full_1gpu_final2.err | head
echo done
