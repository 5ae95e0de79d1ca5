#!/bin/bash
# Round 4, GPU call 6: completion protocol 1 with the own-count race fixed + recovery from a pipelined timeout (tests), the
# confinement experiment once more, then the 20-second job on the default path.
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 200 python -u -m pytest tests/test_hip_engine.py -q -m gpu -p no:cacheprovider -k "pipelined" 2>&1 | tail -4
echo "== engine: protocol 1 (fixed)"
JB_PIPE_DEBUG=1 timeout 200 python -u tools/bench_engine.py up --pipelined 1 --steps 1024 > $O/r04_bench_engine_up_proto1_fixed.log 2>&1; grep -E "graph=True|whole step|c_attn|c_fc |attention|c_proj" $O/r04_bench_engine_up_proto1_fixed.log
echo "== 6-second job, JB_CONFINE_UPPER_CUS=64"
JB_CONFINE_UPPER_CUS=64 JB_BENCH_TIMELINE=1 timeout 330 python -u bench.py --seconds 6 --steps 1 --warmup 0 --no-cpu-baseline > $O/r04_bench_6s_confine64_fixed.json 2> $O/r04_bench_6s_confine64_fixed.err
python - <<PY
import json
try:
    d = json.load(open("$O/r04_bench_6s_confine64_fixed.json"))
    b = d["breakdown"]
    print("value", d["value"], "ms", d["ms_per_step"], {k: v for k, v in b.items() if k != "timeline"})
    for x in b.get("timeline", []):
        print("   ", x, round(x[3] - x[2], 2))
except Exception as e:
    print("no result:", e)
PY
grep -i "fell back\|timed out\|error\|Traceback" -A3 $O/r04_bench_6s_confine64_fixed.err | head -12
echo "== 20-second job (default path)"
JB_BENCH_BUDGET_S=420 JB_BENCH_TIMELINE=1 timeout 600 python -u bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_bench_full_1gpu_c.json 2> $O/r04_bench_full_1gpu_c.err; cut -c1-400 $O/r04_bench_full_1gpu_c.json; grep -i "timed out\|fell back\|Traceback" -A3 $O/r04_bench_full_1gpu_c.err | head; tail -1 $O/r04_bench_full_1gpu_c.err
echo done
