#!/bin/bash
# Round 4, GPU call 24 (branch wip/pipe-5b, the last of the budget): pipelined launches for multi-head engines -- bit-identity at the
# 5b_lyrics geometry, then the 5b top prior's decode step plain and pipelined.
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
O=$PWD/gpurun_out
timeout 55 python -u -m pytest tests/test_hip_engine.py -q -p no:cacheprovider -x -k "pipelined_launches_equal" > $O/r04_pipe5b_tests.log 2>&1; tail -15 $O/r04_pipe5b_tests.log | cut -c1-200
timeout 50 python -u tools/bench_engine.py 5b --batch 3 --steps 48 --pipelined 1 > $O/r04_pipe5b_bench_engine.log 2>&1; grep -v amdgpu.ids $O/r04_pipe5b_bench_engine.log | tail -6 | cut -c1-200
# (round 5, first call on this branch) the 8-wave long-row kernels: tests, then the 5b step with them, plain and pipelined
# timeout 100 python -u -m pytest tests/test_hip_kernels.py tests/test_hip_engine.py -q -p no:cacheprovider -k 'gemv_long or pipelined_launches_equal'
# for p in 0 1; do python -u -c "import sys; sys.argv=['x','5b','--batch','3','--steps','48','--pipelined',str($p)]; from jukebox_amd import _lib as L; L.lib().jb_tune_gemv_long(1); from tools import bench_engine; bench_engine.main()"; done
echo done
