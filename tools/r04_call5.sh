#!/bin/bash
# Round 4, GPU call 5: the confinement experiment once more with the fallback reason printed (and under GPU_MAX_HW_QUEUES=8),
# conv-stack / prefill micro-benchmark after the out-of-range-tap skip, then the 20-second job on the default path.
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
for q in "" 8; do
  echo "== 6-second job, JB_CONFINE_UPPER_CUS=64 GPU_MAX_HW_QUEUES='$q'"
  if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; else unset GPU_MAX_HW_QUEUES; fi
  JB_CONFINE_UPPER_CUS=64 JB_PIPE_TIMEOUT_MS=500 JB_BENCH_TIMELINE=1 timeout 330 python -u bench.py --seconds 6 --steps 1 --warmup 0 --no-cpu-baseline > $O/r04_bench_6s_confine64_q$q.json 2> $O/r04_bench_6s_confine64_q$q.err
  python - <<PY
import json
try:
    d = json.load(open("$O/r04_bench_6s_confine64_q$q.json"))
    b = d["breakdown"]
    print("value", d["value"], "ms", d["ms_per_step"], {k: v for k, v in b.items() if k != "timeline"})
    for x in b.get("timeline", []):
        print("   ", x, round(x[3] - x[2], 2))
except Exception as e:
    print("no result:", e)
PY
  grep -i "fell back\|error\|Traceback" -A3 $O/r04_bench_6s_confine64_q$q.err | head -12
done
unset GPU_MAX_HW_QUEUES
echo "== bench_prefill"
timeout 200 python -u tools/bench_prefill.py > $O/r04_bench_prefill_tapskip.log 2>&1; tail -4 $O/r04_bench_prefill_tapskip.log
timeout 120 python -u -m pytest tests/test_hip_kernels.py tests/test_hip_models.py -q -m gpu -p no:cacheprovider -k "conv or vqvae or conditioner or upsampler" 2>&1 | tail -3
echo "== 20-second job (default path)"
JB_BENCH_BUDGET_S=420 JB_BENCH_TIMELINE=1 timeout 600 python -u bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_bench_full_1gpu_b.json 2> $O/r04_bench_full_1gpu_b.err; cut -c1-400 $O/r04_bench_full_1gpu_b.json; tail -2 $O/r04_bench_full_1gpu_b.err
echo done
