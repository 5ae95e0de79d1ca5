#!/bin/bash
# Round 4, GPU call 16: V12 of the probe (contiguous shards, every consumer wave waits for its own eighth of the block) beside V5.
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 60 tools/pipelined_launch_probe 60 5 5; timeout 60 tools/pipelined_launch_probe 60 12 12; timeout 60 tools/pipelined_launch_probe 60 5 5 ) 2>&1 | grep -v "3 graphs\|aborted: a poll" > gpurun_out/r04_pipelined_launch_probe_v12.log
cat gpurun_out/r04_pipelined_launch_probe_v12.log
echo done
