#!/bin/bash
export PYTHONPATH=$PWD TMPDIR=/tmp JB_PIPE_TIMEOUT_MS=50
mkdir -p gpurun_out
timeout 170 python -u tools/pipe_in_job.py --seconds 2.2 > gpurun_out/r03_pipe_in_job.log 2>&1
echo "rc=$?"
grep -v "^Sampling\|^Ancestral\|^Primed\|Loading\|amdgpu.ids" gpurun_out/r03_pipe_in_job.log | cut -c1-400 | tail -70
