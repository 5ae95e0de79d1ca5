#!/bin/bash
# Round 4, GPU call 23 (what is left of the budget): kernel statistics of the 5b_lyrics top prior's decode step on this round's library.
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
O=$PWD/gpurun_out
cd /tmp && rm -rf /tmp/prof5b && timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof5b -- python -u $GRAFT_REPO_ROOT/tools/bench_engine.py 5b --batch 3 --steps 16 > $O/r04_bench_engine_5b.log 2>&1
f=$(find /tmp/prof5b -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r04_5b_kernel_stats.csv
grep -v amdgpu.ids $O/r04_bench_engine_5b.log | tail -4; grep "gemv\|attn" $O/r04_5b_kernel_stats.csv | cut -c1-130 | head -8
echo done
