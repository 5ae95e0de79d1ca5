#!/bin/bash
# Round 4, GPU call 7: the 20-second job with the upper levels confined to 64 compute units while they run (level 0 pipelined on
# the other 192 from its first step, on all 256 once it is alone) against 214.95 s for the default path.
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
JB_CONFINE_UPPER_CUS=64 JB_BENCH_BUDGET_S=330 JB_BENCH_TIMELINE=1 timeout 600 python -u bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/r04_bench_full_1gpu_confine64.json 2> $O/r04_bench_full_1gpu_confine64.err
python - <<PY
import json
try:
    d = json.load(open("$O/r04_bench_full_1gpu_confine64.json"))
    b = d["breakdown"]
    print("value", d["value"], "ms", d["ms_per_step"], {k: v for k, v in b.items() if k != "timeline"})
    for x in b.get("timeline", []):
        print("   ", x, round(x[3] - x[2], 2))
except Exception as e:
    print("no result:", e)
PY
grep -i "fell back\|timed out\|error\|Traceback" -A3 $O/r04_bench_full_1gpu_confine64.err | head -12
echo done
