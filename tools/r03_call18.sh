#!/bin/bash
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
JB_BENCH_TIMELINE=1 timeout 230 python -u bench.py --seconds 6 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r03_bench_6s_plain_untapped.json 2> gpurun_out/r03_bench_6s_plain_untapped.err; echo "rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_bench_6s_plain_untapped.json'))
print(d['value'], d['ms_per_step'], {k:v for k,v in d['breakdown'].items() if k!='timeline'})
for x in d['breakdown'].get('timeline',[]): print(x)
print(d['roofline'])
PY
tail -3 gpurun_out/r03_bench_6s_plain_untapped.err | cut -c1-300
