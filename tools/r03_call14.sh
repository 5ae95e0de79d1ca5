#!/bin/bash
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 python tools/bench_prefill.py 2>&1 | tail -3
timeout 120 python -m pytest tests/test_hip_engine.py -x -q -m gpu -k "pipelined_launches" 2>&1 | tail -1
JB_PIPELINE_LAUNCHES=1 JB_PIPE_TIMEOUT_MS=300 JB_BENCH_TIMELINE=1 timeout 400 python bench.py --seconds 6 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r03_bench_6s_pipelined.json 2> gpurun_out/r03_bench_6s_pipelined.err; echo "rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03_bench_6s_pipelined.json"))
b=d["breakdown"]; print(d["value"], d["ms_per_step"], {k:v for k,v in b.items() if k!="timeline"})
PY
