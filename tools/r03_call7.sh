#!/bin/bash
export PYTHONPATH=$PWD TMPDIR=/tmp
JB_PIPE_DEBUG=1 timeout 200 python tools/bench_engine.py up --steps 64 --pipelined 1 2>&1 | tail -5
