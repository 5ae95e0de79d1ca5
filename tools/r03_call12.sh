#!/bin/bash
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_hip_baseline_configs.py -x -q -s -m gpu -k "whole_window" > gpurun_out/r03_golden_window.log 2>&1; tail -4 gpurun_out/r03_golden_window.log | cut -c1-600
timeout 200 python tools/bench_prefill.py 2>&1 | tail -5
