#!/bin/bash
export PYTHONPATH=$PWD TMPDIR=/tmp JB_PIPE_TIMEOUT_MS=100
timeout 120 python -m pytest tests/test_hip_engine.py -x -q -m gpu -k "pipelined_launches" 2>&1 | tail -2
timeout 100 python tools/bench_engine.py up --steps 128 --pipelined 1 2>&1 | tail -1
timeout 100 python tools/bench_engine.py up --steps 128 --pipelined 1 --prio -1 2>&1 | tail -1
