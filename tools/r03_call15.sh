#!/bin/bash
export PYTHONPATH=$PWD TMPDIR=/tmp JB_PIPE_TIMEOUT_MS=100
echo "pipelined from the start: $(timeout 100 python tools/bench_engine.py up --steps 96 --pipelined 1 2>&1 | tail -1)"
echo "plain graph first, then pipelined: $(timeout 100 python tools/bench_engine.py up --steps 96 --pipelined 1 --plain-first 32 2>&1 | tail -1)"
