#!/bin/bash
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tools/bench_engine.py 5b --batch 3 --steps 32 > gpurun_out/r03_bench_engine_5b.log 2>&1; cat gpurun_out/r03_bench_engine_5b.log
timeout 600 python -m pytest tests/test_hip_baseline_configs.py tests/test_hip_models.py -x -q -m gpu -k "config5 or separated_encoder" 2>&1 | tail -3
timeout 200 python tools/bench_engine.py up --steps 64 2>&1 | tail -3
