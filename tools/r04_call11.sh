#!/bin/bash
# Round 4, GPU call 11: the f16-split conv stacks (jb_gemm_args.w_split) -- kernel tests, the full-size conditioner and the
# VQ-VAE cases that now run on it, the conditioner timed in the three forms, and the tests the knob removal touched.
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
O=$PWD/gpurun_out
timeout 150 python -u -m pytest tests/test_hip_kernels.py -q -p no:cacheprovider -s -k "conv_stack or gemm_split or gemm_plain or gemm_epilogues" > $O/r04_split_kernel_tests.log 2>&1; tail -12 $O/r04_split_kernel_tests.log
timeout 100 python -u tools/bench_conditioner.py > $O/r04_bench_conditioner.log 2>&1; cat $O/r04_bench_conditioner.log
timeout 300 python -u -m pytest tests/test_hip_baseline_configs.py tests/test_hip_models.py tests/test_hip_engine.py -q -p no:cacheprovider -s --durations=8 -k "conditioner_full_size or config1 or vqvae or end_to_end or teacher_forced_agreement or pipelined_launches_equal" > $O/r04_split_model_tests.log 2>&1; tail -25 $O/r04_split_model_tests.log
echo done
