#!/bin/bash
# Round 4, GPU call 14: the phase of the pipelined step on 60 / 120 / 240 workgroups (protocol 1 = V5 of the probe).
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 120 tools/pipelined_launch_probe 40 5 5 0 1 > gpurun_out/r04_pipelined_launch_probe_wg_sweep.log 2>&1
cat gpurun_out/r04_pipelined_launch_probe_wg_sweep.log
echo done
