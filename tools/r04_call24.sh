#!/bin/bash
# Round 4, GPU call 24 (branch wip/pipe-5b, the last of the budget): pipelined launches for multi-head engines -- bit-identity at the
# 5b_lyrics geometry, then the 5b top prior's decode step plain and pipelined.
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
O=$PWD/gpurun_out
timeout 55 python -u -m pytest tests/test_hip_engine.py -q -p no:cacheprovider -x -k "pipelined_launches_equal" > $O/r04_pipe5b_tests.log 2>&1; tail -15 $O/r04_pipe5b_tests.log | cut -c1-200
timeout 50 python -u tools/bench_engine.py 5b --batch 3 --steps 48 --pipelined 1 > $O/r04_pipe5b_bench_engine.log 2>&1; grep -v amdgpu.ids $O/r04_pipe5b_bench_engine.log | tail -6 | cut -c1-200
echo done
