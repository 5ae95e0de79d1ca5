#!/bin/bash
# Round 4, GPU call 2: the new kernels' tests (LDS-DMA GEMM, tiled transpose prefill attention, completion protocol V4 +
# host-synchronous pipelined decode), the full-size cases against their digests, micro-benchmarks, then the benchmark job.
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
echo "== kernel + engine tests"
timeout 500 python -u -m pytest tests/test_hip_kernels.py tests/test_hip_engine.py -q -m gpu -p no:cacheprovider --durations=8 > $O/r04_kernel_engine_tests.log 2>&1; tail -14 $O/r04_kernel_engine_tests.log
echo "== full-size cases against the digests"
timeout 600 python -u -m pytest tests/test_hip_baseline_configs.py -q -m gpu -p no:cacheprovider --durations=12 -s > $O/r04_full_size_tests.log 2>&1; grep -E "prefill of sample 0|passed|failed|Error|assert|^[0-9.]+s " $O/r04_full_size_tests.log | tail -30
echo "== bench_engine (plain / pipelined V4)"
JB_PIPE_DEBUG=1 timeout 200 python -u tools/bench_engine.py up --pipelined 1 --steps 512 > $O/r04_bench_engine_up_v4.log 2>&1; tail -9 $O/r04_bench_engine_up_v4.log
echo "== bench_prefill"
timeout 200 python -u tools/bench_prefill.py > $O/r04_bench_prefill.log 2>&1; tail -6 $O/r04_bench_prefill.log
echo "== 6-second job"
JB_BENCH_TIMELINE=1 timeout 400 python -u bench.py --seconds 6 --steps 1 --warmup 0 --no-cpu-baseline > $O/r04_bench_6s_1gpu.json 2> $O/r04_bench_6s_1gpu.err; cut -c1-1500 $O/r04_bench_6s_1gpu.json; tail -3 $O/r04_bench_6s_1gpu.err
ms=$(python -c "import json;print(json.load(open('$O/r04_bench_6s_1gpu.json'))['ms_per_step'])" 2>/dev/null || echo 999999)
echo "6-second job: $ms ms"
if python -c "import sys; sys.exit(0 if float('$ms') < 79000 else 1)"; then
  echo "== 20-second job (one step)"
  JB_BENCH_BUDGET_S=420 JB_BENCH_TIMELINE=1 timeout 600 python -u bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_bench_full_1gpu.json 2> $O/r04_bench_full_1gpu.err; cut -c1-1200 $O/r04_bench_full_1gpu.json; tail -3 $O/r04_bench_full_1gpu.err
fi
echo done
