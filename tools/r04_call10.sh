#!/bin/bash
# Round 4, GPU call 10: where a level-0 window's non-decode time goes (tools/window_glue.py), pipelined and plain.
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 python -u tools/window_glue.py --steps 512 --windows 3 > gpurun_out/r04_window_glue.log 2>&1; tail -25 gpurun_out/r04_window_glue.log
