#!/bin/bash
# Round 2, GPU call 8: full GPU suite on the cleaned-up tree + transposed cache mirror on / off
export PYTHONPATH=$PWD
echo "== gpu suite =="; timeout 1000 python -m pytest tests -m gpu -q --timeout 400 -p no:cacheprovider 2>&1 | tail -8 | tee gpurun_out/r02_gpu_tests_tail.log
echo "== decode step: transposed mirror on / off =="
for m in 1 0; do echo "JB_TRANSPOSE_MIRROR=$m"; JB_TRANSPOSE_MIRROR=$m timeout 100 python tools/bench_engine.py up --steps 256 2>&1 | tail -1; done
timeout 100 python tools/bench_engine.py 1b --steps 128 2>&1 | tail -1
echo "== prefill =="; timeout 100 python tools/bench_prefill.py 2>&1 | tail -3
