#!/bin/bash
# Round 4, GPU call 4: the 6-second job -- baseline (completion protocol 1) against the experiment that confines the upper
# levels to n compute units once level 0 starts and runs level 0 pipelined from its first step.
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
for n in 0 64 96; do
  echo "== 6-second job, JB_CONFINE_UPPER_CUS=$n"
  JB_CONFINE_UPPER_CUS=$n JB_PIPE_TIMEOUT_MS=500 JB_BENCH_TIMELINE=1 timeout 330 python -u bench.py --seconds 6 --steps 1 --warmup 0 --no-cpu-baseline > $O/r04_bench_6s_confine$n.json 2> $O/r04_bench_6s_confine$n.err
  python - <<PY
import json
try:
    d = json.load(open("$O/r04_bench_6s_confine$n.json"))
    b = d["breakdown"]
    print("value", d["value"], "ms", d["ms_per_step"], {k: v for k, v in b.items() if k != "timeline"})
    for x in b.get("timeline", []):
        print("   ", x, round(x[3] - x[2], 2))
except Exception as e:
    print("no result:", e)
PY
  tail -2 $O/r04_bench_6s_confine$n.err
done
echo done
