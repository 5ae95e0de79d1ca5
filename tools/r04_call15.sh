#!/bin/bash
# Round 4, GPU call 15: V11 of the probe -- protocol 1's flags without the producer's drain, tagged 16-byte words catching the stores
# still in flight -- beside V5 (protocol 1) and V10 (tags only).
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 60 tools/pipelined_launch_probe 40 5 5; timeout 60 tools/pipelined_launch_probe 40 10 11 ) 2>&1 | grep -v "3 graphs" > gpurun_out/r04_pipelined_launch_probe_v11.log
cat gpurun_out/r04_pipelined_launch_probe_v11.log
echo done
