#!/bin/bash
# Round 2, GPU call 7: why is the three-launch layer not faster? (per-kernel stats); partition / look-ahead variants
export PYTHONPATH=$PWD
echo "== fused-layer tests =="; timeout 600 python -m pytest tests/test_hip_kernels.py tests/test_hip_engine.py -m gpu -q --timeout 300 -p no:cacheprovider -k "fused or three_launch" 2>&1 | tail -8
echo "== kernel stats, three-launch step =="
cd /tmp && export TMPDIR=/tmp
JB_FUSED3=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r02_f3 -- python $GRAFT_REPO_ROOT/tools/bench_engine.py up --steps 48 > $GRAFT_REPO_ROOT/gpurun_out/prof_r02_f3.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_r02_f3/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:12]:
    print(r["Name"][:70], "calls", r["Calls"], "avg_ns", r["AverageNs"], "min", r["MinNs"], "pct", r["Percentage"])
PY
echo "== pipeline variants (6 s job, 1 step, 5-launch layer) =="
show() { python -c "
import json,sys
d=json.load(sys.stdin); b=d['breakdown']
print(sys.argv[1], d['value'], d['ms_per_step'], {k:b[k] for k in b if k.endswith('_done_at_s')})
tl=[x for x in b.get('timeline',[]) if x[0]==0]
print('   level-0 windows (start, begin_s, end_s, lookahead):', [(x[1], x[2], x[3], x[4]) for x in tl])
" "$1"; }
export JB_BENCH_TIMELINE=1 JB_FUSED3=0
JB_LOOKAHEAD=0 timeout 300 python bench.py --seconds 6 --steps 1 --no-cpu-baseline 2>/dev/null | tail -1 | show "L0 0-127 | L1 128-255 | L2 unmasked, no look-ahead:"
JB_LOOKAHEAD_CUS=224-255 timeout 300 python bench.py --seconds 6 --steps 1 --no-cpu-baseline 2>/dev/null | tail -1 | show "same + look-ahead on 32 CUs:"
echo "== prefill chunk 2048 =="; timeout 100 python tools/bench_prefill.py 2048 2>&1 | tail -3
