#!/bin/bash
# Round 3, call 2: does a side-stream L2 prefetcher shorten the phases of the launch chain?
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 240 tools/persist_probe 30 > gpurun_out/r03_persist_probe_v2.log 2>&1; echo "persist_probe rc=$?"; cat gpurun_out/r03_persist_probe_v2.log
