#!/bin/bash
# Round 4, GPU calls 18 / 19 (branch wip/pipe-v12): per-wave dependencies in the engine (18: contiguous shards; 19: permuted tiles under mod-8 shards) -- the pipelined tests, then
# the upsampler step on the main build and on this build of the library, alternating.
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
O=$PWD/gpurun_out
timeout 150 python -u -m pytest tests/test_hip_engine.py -q -p no:cacheprovider -k "pipelined" > $O/r04_v12_engine_tests.log 2>&1; tail -6 $O/r04_v12_engine_tests.log
for rep in 1 2; do
  for lib in libjukebox_hip_main.so libjukebox_hip.so; do
    echo "== $lib" >> $O/r04_v12_bench_engine.log
    JB_PIPE_DEBUG=1 timeout 60 python -u tools/ab_bench_engine.py $PWD/jukebox_amd/csrc/$lib up --pipelined 1 --steps 512 2>&1 | grep -v amdgpu.ids >> $O/r04_v12_bench_engine.log
  done
done
cat $O/r04_v12_bench_engine.log
echo done
