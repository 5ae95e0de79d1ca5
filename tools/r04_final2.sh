#!/bin/bash
# Round 4, last GPU call on the final tree (f16-split conv stacks, mid-window switch to pipelined launches): the whole GPU suite in
# one process, smoke(), the driver's bench command against a wall budget that fits one full step, then -- if the lease still has
# room -- a rocprofv3 summary of the miniature job (every kernel of the path, gemm_split_kernel among them).
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
O=$PWD/gpurun_out
timeout 420 python -u -m pytest tests -q -m gpu -p no:cacheprovider --durations=12 > $O/r04_gpu_tests_final2.log 2>&1; tail -18 $O/r04_gpu_tests_final2.log
timeout 100 python -u -c "import __graft_entry__ as g; g.smoke()" > $O/r04_smoke2.log 2>&1; tail -3 $O/r04_smoke2.log
echo "== bench.py --gpus 1 --steps 20 --warmup 5 (wall budget 450 s)"
JB_BENCH_BUDGET_S=450 JB_BENCH_TIMELINE=1 timeout 520 python -u bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_bench_full_1gpu_final2.json 2> $O/r04_bench_full_1gpu_final2.err; cut -c1-400 $O/r04_bench_full_1gpu_final2.json; grep -i "timed out\|fell back\|Traceback" -A3 $O/r04_bench_full_1gpu_final2.err | head
echo "== rocprofv3: miniature of the whole job"
cd /tmp && rm -rf /tmp/prof_job && timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_job -- python -u $GRAFT_REPO_ROOT/tools/profile_job.py > $O/r04_profile_job2.log 2>&1
f=$(find /tmp/prof_job -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r04_full_job_kernel_stats2.csv && head -8 $O/r04_full_job_kernel_stats2.csv | cut -c1-150
echo done
