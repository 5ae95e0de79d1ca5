#!/bin/bash
# Round 4, GPU call 12: a level-0 window stage by stage with the f16-split conv stacks in place (tools/window_glue.py), the
# conditioner timed in both forms on the final kernel, and the split-kernel tests on the final kernel.
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
O=$PWD/gpurun_out
timeout 100 python -u tools/bench_conditioner.py > $O/r04_bench_conditioner.log 2>&1; cat $O/r04_bench_conditioner.log
timeout 200 python -u tools/window_glue.py --steps 512 --windows 3 > $O/r04_window_glue_split.log 2>&1; tail -16 $O/r04_window_glue_split.log
timeout 150 python -u -m pytest tests/test_hip_kernels.py -q -p no:cacheprovider -s -k "conv_stack or gemm_split" > $O/r04_split_kernel_tests.log 2>&1; tail -8 $O/r04_split_kernel_tests.log
echo done
