#!/bin/bash
# Round 4, GPU call 20: main's library after the branch experiment (smoke + the pipelined tests), and RCCL next to pipelined launches
# in one process (tools/rccl_coexist.py).
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
O=$PWD/gpurun_out
timeout 100 python -u -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 100 python -u -m pytest tests/test_hip_engine.py -q -p no:cacheprovider -k "pipelined" 2>&1 | tail -2
timeout 150 python -u tools/rccl_coexist.py > $O/r04_rccl_coexist.log 2>&1; grep -v amdgpu.ids $O/r04_rccl_coexist.log | tail -12
echo done
