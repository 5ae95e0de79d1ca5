#!/bin/bash
# Round 2, GPU call 3: where do the 5.8 us per launch go?  kernarg placement, clocks, per-slot trace, CU masks, host enqueue cost.
export PYTHONPATH=$PWD
echo "== box =="; ls oracle/ | head; ls oracle/_ref 2>&1 | head -3; nproc; rocm-smi --showclocks 2>&1 | grep -iE "sclk|mclk|fclk" | head -6
echo "== fixed tests =="; timeout 300 python -m pytest tests/test_hip_models.py -q -x --timeout 250 -p no:cacheprovider -k "pipelined or end_to_end" 2>&1 | tail -4
echo "== decode step baseline (split off by default) =="; timeout 100 python tools/bench_engine.py up --steps 256 2>&1 | tail -2
echo "== HIP_FORCE_DEV_KERNARG =="
for v in 0 1; do echo "HIP_FORCE_DEV_KERNARG=$v"; HIP_FORCE_DEV_KERNARG=$v timeout 100 python tools/bench_engine.py up --steps 256 2>&1 | tail -2; done
echo "== clocks =="; timeout 100 python tools/clock_probe.py 2>&1 | tail -6
echo "== perf level high =="; rocm-smi --setperflevel high 2>&1 | tail -2; rocm-smi --showclocks 2>&1 | grep -iE "sclk" | head -2
timeout 100 python tools/bench_engine.py up --steps 256 2>&1 | tail -1; timeout 60 python tools/clock_probe.py 2>&1 | tail -3
rocm-smi --setperflevel auto 2>&1 | tail -1
echo "== per-slot trace =="
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r02_up -- python $GRAFT_REPO_ROOT/tools/bench_engine.py up --steps 48 > $GRAFT_REPO_ROOT/gpurun_out/prof_r02_up.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/slot_stats.py gpurun_out/prof_r02_up 72 2>&1 | tail -22
echo "== cu masks / concurrency / host enqueue =="; timeout 400 python tools/cu_mask_probe.py 2>&1 | tail -30
