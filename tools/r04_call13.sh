#!/bin/bash
# Round 4, GPU call 13: completion protocols without a completion word -- the activations carry their own validity (V8 / V9 / V10 of
# tools/pipelined_launch_probe.hip) -- against protocol 1 of the engine (V5 in the probe).
export TMPDIR=/tmp
mkdir -p gpurun_out
O=$PWD/gpurun_out
( timeout 60 tools/pipelined_launch_probe 40 5 5; timeout 120 tools/pipelined_launch_probe 40 8 10 ) > $O/r04_pipelined_launch_probe_tagged.log 2>&1
cat $O/r04_pipelined_launch_probe_tagged.log
echo done
