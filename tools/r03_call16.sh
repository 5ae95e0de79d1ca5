#!/bin/bash
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp && rm -rf profj && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profj -- python $GRAFT_REPO_ROOT/tools/profile_job.py > $GRAFT_REPO_ROOT/gpurun_out/r03_profile_job.log 2>&1
find /tmp/profj -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r03_full_job_kernel_stats.csv \;
tail -2 $GRAFT_REPO_ROOT/gpurun_out/r03_profile_job.log | cut -c1-300
head -25 $GRAFT_REPO_ROOT/gpurun_out/r03_full_job_kernel_stats.csv | cut -c1-160
