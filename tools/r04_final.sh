#!/bin/bash
# Round 4, final GPU call: the whole GPU suite in one process, smoke(), a rocprofv3 summary of the miniature job, prefill with
# one 4096-position chunk per window, then the driver's bench command against a wall budget that fits one full step.
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
O=$PWD/gpurun_out
timeout 900 python -u -m pytest tests -q -m gpu -p no:cacheprovider --durations=15 > $O/r04_gpu_tests_final.log 2>&1; tail -22 $O/r04_gpu_tests_final.log
timeout 120 python -u -c "import __graft_entry__ as g; g.smoke()" > $O/r04_smoke.log 2>&1; tail -3 $O/r04_smoke.log
echo "== prefill chunk 2048 / 4096"
timeout 150 python -u tools/bench_prefill.py 2048 2>&1 | grep prefill; timeout 150 python -u tools/bench_prefill.py 4096 2>&1 | grep prefill
echo "== rocprofv3: miniature of the whole job"
cd /tmp && rm -rf /tmp/prof_job && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_job -- python -u $GRAFT_REPO_ROOT/tools/profile_job.py > $O/r04_profile_job.log 2>&1
f=$(find /tmp/prof_job -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r04_full_job_kernel_stats.csv && head -12 $O/r04_full_job_kernel_stats.csv | cut -c1-160
cd $GRAFT_REPO_ROOT
echo "== bench.py --gpus 1 --steps 20 --warmup 5 (wall budget 480 s)"
JB_BENCH_BUDGET_S=480 JB_BENCH_TIMELINE=1 timeout 700 python -u bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_bench_full_1gpu_final.json 2> $O/r04_bench_full_1gpu_final.err; cut -c1-330 $O/r04_bench_full_1gpu_final.json; grep -i "timed out\|fell back\|Traceback" -A3 $O/r04_bench_full_1gpu_final.err | head
echo done
