#!/bin/bash
export PYTHONPATH=$PWD TMPDIR=/tmp
timeout 300 python -m pytest tests/test_hip_engine.py -x -q -m gpu -k "pipelined_launches or seeded_model" 2>&1 | tail -3
JB_PIPE_DEBUG=1 timeout 200 python tools/bench_engine.py up --steps 64 --pipelined 1 2>&1 | tail -5 | cut -c1-330
timeout 200 python tools/bench_engine.py up --steps 128 --pipelined 1 2>&1 | tail -1
