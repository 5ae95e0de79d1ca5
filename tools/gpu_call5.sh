#!/bin/bash
# Round 2, GPU call 5: CU-partitioned level pipeline + look-ahead prefill + early audio decode; reference CPU leg
export PYTHONPATH=$PWD
echo "== model tests =="; timeout 600 python -m pytest tests/test_hip_models.py -m gpu -q -x --timeout 300 -p no:cacheprovider 2>&1 | tail -8
echo "== bench 6 s (2 steps) =="; JB_BENCH_BUDGET_S=900 timeout 1000 python bench.py --seconds 6 --steps 2 --warmup 1 2>gpurun_out/bench6_stderr.log | tail -1 > gpurun_out/bench6.json; grep -v "Sampling\|sampling" gpurun_out/bench6_stderr.log | tail -15; python - <<'PY'
import json
d = json.load(open("gpurun_out/bench6.json"))
print({k: d[k] for k in ("value", "steps", "ms_per_step")}, d["breakdown"], d["cpu_baseline"]["kind"], d["cpu_baseline"]["value"], d["cpu_baseline"].get("cores"))
PY
echo "== bench 6 s without look-ahead / without CU partition =="
JB_CU_PARTITION=0 JB_BENCH_BUDGET_S=400 timeout 500 python bench.py --seconds 6 --steps 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('no partition:', d['value'], d['ms_per_step'], d['breakdown'])"
