#!/bin/bash
# First GPU call of the next round: validate and measure everything that was written after round 1's GPU budget ran out.
#   gpurun --timeout 900 -- 'bash tools/next_round_first_call.sh > gpurun_out/first_call.log 2>&1; tail -40 gpurun_out/first_call.log'
export PYTHONPATH=$PWD
echo "== default suite =="; timeout 300 python -m pytest tests -m gpu -q --timeout 200 -p no:cacheprovider 2>&1 | tail -5
echo "== experimental / gated tests =="; JB_EXPERIMENTAL=1 timeout 300 python -m pytest tests/test_hip_experimental.py tests/test_hip_models.py -q --timeout 200 -p no:cacheprovider -k "experimental or gemv_pair or fused_pairs or prefill_v2 or teacher_forced" 2>&1 | tail -15
echo "== decode step: default vs 3 launches per layer =="
for f in 0 1; do echo JB_FUSED_PAIRS=$f; JB_FUSED_PAIRS=$f timeout 90 python tools/bench_engine.py up --steps 128 2>&1 | tail -2; done
JB_FUSED_PAIRS=1 timeout 90 python tools/bench_engine.py 1b --steps 128 2>&1 | tail -1
echo "== prefill: attention v1 vs v2 =="
for v in 0 1; do echo JB_PREFILL_V2=$v; JB_PREFILL_V2=$v timeout 90 python tools/bench_prefill.py 2>&1 | tail -3; done
