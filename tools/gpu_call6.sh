#!/bin/bash
# Round 2, GPU call 6: three-launch decode layer; level-pipeline variants with a per-window timeline
export PYTHONPATH=$PWD
echo "== fused-layer tests =="; timeout 600 python -m pytest tests/test_hip_kernels.py tests/test_hip_engine.py -m gpu -q -x --timeout 300 -p no:cacheprovider -k "fused or three_launch" 2>&1 | tail -15
echo "== decode step: 3 launches vs 5 =="
for f in 1 0; do echo "JB_FUSED3=$f"; JB_FUSED3=$f timeout 100 python tools/bench_engine.py up --steps 256 2>&1 | tail -2; done
JB_FUSED3=1 timeout 100 python tools/bench_engine.py 1b --steps 128 2>&1 | tail -1
echo "== pipeline variants (6 s job, 1 step, 5-launch layer for comparability) =="
show() { python -c "
import json,sys
d=json.load(sys.stdin); b=d['breakdown']
print(sys.argv[1], d['value'], d['ms_per_step'], {k:b[k] for k in b if k.endswith('_done_at_s')})
tl=[x for x in b.get('timeline',[]) if x[0]==0]
print('   level-0 windows (start, begin_s, end_s, lookahead):', [(x[1], x[2], x[3], x[4]) for x in tl])
tl1=[x for x in b.get('timeline',[]) if x[0]==1]
print('   level-1 windows:', [(x[1], x[2], x[3]) for x in tl1])
" "$1"; }
export JB_BENCH_TIMELINE=1 JB_FUSED3=0
JB_CU_PARTITION=0 timeout 300 python bench.py --seconds 6 --steps 1 --no-cpu-baseline 2>/dev/null | tail -1 | show "no partition:"
JB_LOOKAHEAD=0 timeout 300 python bench.py --seconds 6 --steps 1 --no-cpu-baseline 2>/dev/null | tail -1 | show "partition, no look-ahead:"
timeout 300 python bench.py --seconds 6 --steps 1 --no-cpu-baseline 2>/dev/null | tail -1 | show "partition + look-ahead:"
