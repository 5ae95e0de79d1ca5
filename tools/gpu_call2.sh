#!/bin/bash
# Round 2, GPU call 2: the rewritten decode step (key-split attention + merging c_proj, fused tail), sampler, prefill v2 default.
export PYTHONPATH=$PWD
echo "== gpu suite =="; timeout 900 python -m pytest tests -m gpu -q -x --timeout 400 -p no:cacheprovider 2>&1 | tail -25
echo "== decode step: key split variants (upsampler, N=16) =="
for v in "off" "2,2" "4,2" "4,1" "3,2"; do
  if [ "$v" = "off" ]; then export JB_ATTN_SPLIT_OFF=1; unset JB_ATTN_SPLIT; else export JB_ATTN_SPLIT_OFF=0; export JB_ATTN_SPLIT=$v; fi
  echo "split=$v"; timeout 120 python tools/bench_engine.py up --steps 256 2>&1 | tail -2
done
unset JB_ATTN_SPLIT; export JB_ATTN_SPLIT_OFF=0
echo "== 1b top =="; timeout 120 python tools/bench_engine.py 1b --steps 128 2>&1 | tail -2
echo "== prefill =="; timeout 120 python tools/bench_prefill.py 2>&1 | tail -3
echo "== bench 6 s =="; JB_BENCH_BUDGET_S=600 timeout 700 python bench.py --seconds 6 --steps 2 --warmup 0 2>&1 | tail -3
