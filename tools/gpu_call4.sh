#!/bin/bash
# Round 2, GPU call 4: decode kernels with every request of a phase issued up front (bias / residual / c1 / activations / value rows)
export PYTHONPATH=$PWD
echo "== kernel + engine tests =="; timeout 600 python -m pytest tests/test_hip_kernels.py tests/test_hip_engine.py tests/test_hip_models.py -m gpu -q -x --timeout 300 -p no:cacheprovider 2>&1 | tail -6
echo "== decode step =="; timeout 100 python tools/bench_engine.py up --steps 256 2>&1 | tail -2
timeout 100 python tools/bench_engine.py 1b --steps 128 2>&1 | tail -1
JB_ATTN_SPLIT_MIN_KEYS=1 timeout 100 python tools/bench_engine.py up --steps 256 2>&1 | tail -1
echo "== reference cpu leg =="; timeout 200 python oracle/time_reference.py --budget-s 45 2>&1 | tail -4 | cut -c1-600
echo "== per-slot trace =="
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r02_up2 -- python $GRAFT_REPO_ROOT/tools/bench_engine.py up --steps 48 > $GRAFT_REPO_ROOT/gpurun_out/prof_r02_up2.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/slot_stats.py gpurun_out/prof_r02_up2 72 2>&1 | tail -16
