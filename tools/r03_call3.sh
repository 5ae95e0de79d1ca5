#!/bin/bash
# Round 3, call 3: pipelined launches -- flag-row protocol, eager vs one-graph-per-stream replay, 2 and 3 streams.
mkdir -p gpurun_out
timeout 240 tools/pipelined_launch_probe 30 > gpurun_out/r03_pipelined_launch_probe_v3.log 2>&1; echo "rc=$?"; cat gpurun_out/r03_pipelined_launch_probe_v3.log
