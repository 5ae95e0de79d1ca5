#!/bin/bash
# The GPU call that protects the round's measurements (through gpurun from the repo root):
#   tools/round_call.sh [tag] [bench seconds] [bench steps]
#     1. the whole GPU suite                 -> gpurun_out/<tag>_gpu_tests.log
#     2. smoke()                             -> gpurun_out/<tag>_smoke.log
#     3. bench.py AS THE DRIVER RUNS IT: one warm-up job, then several timed jobs IN ONE PROCESS (round 4's lines were all first
#        jobs of a fresh process and hid a 13 % steady-state loss).  Default here: 6 s of audio, 2 timed steps (~4 GPU-minutes);
#        `tools/round_call.sh r05 20 2` is the full-length form (~12 minutes), bare `python bench.py --steps 20 --warmup 5`
#        the driver's own (~17 minutes)      -> gpurun_out/<tag>_bench_<seconds>s_1gpu.json
#     4. rocprofv3 kernel statistics of `bench.py --roofline-only` (the dominant kernel's average duration must agree with the
#        line's roofline.avg_launch_us)       -> gpurun_out/<tag>_roofline_only_kernel_stats.csv
#     5. a level-0 window stage by stage through the sampler's own entry point (conditioner, prefill, decode, everything else)
#                                             -> gpurun_out/<tag>_window_glue.log
# (Round 5's last full call was this with `r05f 20 1`: profiles/r05f_*.)
# Copy what is to be judged into profiles/.  Always `python -u` and a `timeout` of your own: a call that runs into gpurun's
# limit is lost.
TAG=${1:-r06}; SECS=${2:-6}; STEPS=${3:-2}
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -u -m pytest tests -q -m gpu -p no:cacheprovider --durations=8 -s > gpurun_out/${TAG}_gpu_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/${TAG}_gpu_tests.log | tail -3
timeout 300 python -u __graft_entry__.py smoke > gpurun_out/${TAG}_smoke.log 2>&1; tail -2 gpurun_out/${TAG}_smoke.log
JB_BENCH_TIMELINE=1 timeout 1500 python -u bench.py --gpus 1 --seconds $SECS --steps $STEPS --warmup 1 \
    > gpurun_out/${TAG}_bench_${SECS}s_1gpu.json 2> gpurun_out/${TAG}_bench_${SECS}s_1gpu.err
cut -c1-900 gpurun_out/${TAG}_bench_${SECS}s_1gpu.json
python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_bench_${SECS}s_1gpu.json"))
b = d["breakdown"]
print("steps", b["step_seconds"], "L2/L1/L0 done at", b.get("level2_codes_done_at_s"), b.get("level1_codes_done_at_s"), b.get("level0_codes_done_at_s"),
      "form", b.get("level0_launch_form"), b.get("level0_in_situ_comparison_ms_per_step"))
PY
cd /tmp && rm -rf /tmp/prof_roof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_roof -- \
    python -u $GRAFT_REPO_ROOT/bench.py --roofline-only > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_roofline_only_stdout.json 2> /tmp/prof_roof.err
cd $GRAFT_REPO_ROOT
cp $(find /tmp/prof_roof -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_roofline_only_kernel_stats.csv 2>/dev/null
head -4 gpurun_out/${TAG}_roofline_only_kernel_stats.csv | cut -c1-200; cat gpurun_out/${TAG}_roofline_only_stdout.json | cut -c1-500
timeout 200 python -u tools/window_glue.py --steps 512 --windows 2 > gpurun_out/${TAG}_window_glue.log 2>&1; tail -16 gpurun_out/${TAG}_window_glue.log
