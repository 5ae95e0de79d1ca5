#!/bin/bash
# Round 4, GPU call 17: the level pipeline with the lowest level's window decoded in plain chunks while upper levels run.
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 python -u -m pytest tests/test_hip_models.py -q -p no:cacheprovider --durations=5 -k "pipelined_levels or end_to_end or primed_mode" > gpurun_out/r04_recheck_tests.log 2>&1; tail -12 gpurun_out/r04_recheck_tests.log
echo done
