#!/bin/bash
# Round 4, GPU call 8: the software-pipelined tap GEMM (conv stacks): tests, conditioner / prefill timings, kernel statistics.
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
O=$PWD/gpurun_out
timeout 300 python -u -m pytest tests/test_hip_kernels.py tests/test_hip_models.py tests/test_hip_baseline_configs.py -q -m gpu -p no:cacheprovider -k "gemm or conv or vqvae or conditioner or upsampler or end_to_end or config1" 2>&1 | tail -4
timeout 200 python -u tools/bench_prefill.py > $O/r04_bench_prefill_pipelined_taps.log 2>&1; tail -3 $O/r04_bench_prefill_pipelined_taps.log
JB_BENCH_TIMELINE=1 timeout 330 python -u bench.py --seconds 6 --steps 1 --warmup 0 --no-cpu-baseline > $O/r04_bench_6s_1gpu_final.json 2> $O/r04_bench_6s_1gpu_final.err; cut -c1-260 $O/r04_bench_6s_1gpu_final.json
python - <<PY
import json
d = json.load(open("$O/r04_bench_6s_1gpu_final.json"))
for x in d["breakdown"].get("timeline", []):
    print("   ", x, round(x[3] - x[2], 2))
PY
echo done
