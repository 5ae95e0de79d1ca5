#!/bin/bash
# Round 4, GPU call 22: the overflow flag of the f16-split conv kernel (kernel tests, the model-level cases that run through the
# sampler's end-of-job check, the conditioner's timing with the check in the kernel).
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
O=$PWD/gpurun_out
timeout 120 python -u -m pytest tests/test_hip_kernels.py -q -p no:cacheprovider -k "conv_stack or gemm_split" > $O/r04_split_overflow_tests.log 2>&1; tail -4 $O/r04_split_overflow_tests.log
timeout 200 python -u -m pytest tests/test_hip_models.py tests/test_hip_baseline_configs.py tests/test_sample_windows.py -q -p no:cacheprovider -k "vqvae or end_to_end or primed_mode or pipelined_levels or conditioner_full_size or config1 or upsamplers" >> $O/r04_split_overflow_tests.log 2>&1; tail -4 $O/r04_split_overflow_tests.log
timeout 100 python -u tools/bench_conditioner.py 2>&1 | grep conditioner
echo done
