#!/bin/bash
# Round 4, GPU call 1: probes (single-queue AQL pipelining, prefill-GEMM candidates), the in-job diagnosis of the pipelined
# launches, and the new parity cases.  Every step under its own timeout; logs in gpurun_out/r04_*.
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
rocm-smi --showproductname 2>/dev/null | head -8 > $O/r04_box.txt; nproc >> $O/r04_box.txt
echo "== pipelined_launch_probe V1 (+ AQL modes)"; timeout 120 tools/pipelined_launch_probe 40 1 1 > $O/r04_pipelined_launch_probe_v1_aql.log 2>&1; tail -40 $O/r04_pipelined_launch_probe_v1_aql.log
echo "== pipelined_launch_probe V4"; timeout 120 tools/pipelined_launch_probe 40 4 4 > $O/r04_pipelined_launch_probe_v4_aql.log 2>&1; grep -E "2 graphs|1 stream" $O/r04_pipelined_launch_probe_v4_aql.log
echo "== gemm_glds_probe"; timeout 200 tools/gemm_glds_probe > $O/r04_gemm_glds_probe.log 2>&1; cat $O/r04_gemm_glds_probe.log
echo "== bench_engine long pipelined call"; JB_PIPE_TIMEOUT_MS=200 timeout 300 python -u tools/bench_engine.py up --pipelined 1 --steps 2048 > $O/r04_bench_engine_up_pipelined_2048.log 2>&1; tail -5 $O/r04_bench_engine_up_pipelined_2048.log
echo "== new parity cases"
timeout 600 python -u -m pytest tests/test_hip_kernels.py tests/test_hip_baseline_configs.py tests/test_hip_engine.py -q -m gpu -p no:cacheprovider --durations=15 \
  -k "cross_attention or lyric_encoder or order10 or second_window or pipelined_launches" > $O/r04_new_parity_tests.log 2>&1; tail -30 $O/r04_new_parity_tests.log
echo "== pipe_in_job (early creation of the pair; B = late creation)"
JB_PIPE_TIMEOUT_MS=50 timeout 400 python -u tools/pipe_in_job.py --seconds 6 > $O/r04_pipe_in_job.log 2>&1; grep -v "^Sampling\|^Ancestral\|^Primed\|Loading" $O/r04_pipe_in_job.log | tail -60
slow=$(grep "^A  job" $O/r04_pipe_in_job.log | awk '{for(i=1;i<=NF;i++) if($i=="->") print ($(i+1) > 2.0) ? 1 : 0}')
if [ "$slow" != "0" ]; then
  echo "== pipe_in_job slow or failed: once more with the runtime's warnings and the default queue limit"
  GPU_MAX_HW_QUEUES=4 AMD_LOG_LEVEL=2 JB_PIPE_TIMEOUT_MS=50 timeout 400 python -u tools/pipe_in_job.py --seconds 6 > $O/r04_pipe_in_job_q4.log 2> $O/r04_pipe_in_job_q4.err
  grep -v "^Sampling\|^Ancestral\|^Primed\|Loading" $O/r04_pipe_in_job_q4.log | tail -40
  grep -c . $O/r04_pipe_in_job_q4.err; grep -i "packet capture\|argument pool" $O/r04_pipe_in_job_q4.err | sort | uniq -c | head; head -c 200000 $O/r04_pipe_in_job_q4.err > $O/r04_pipe_in_job_q4.err.head; rm -f $O/r04_pipe_in_job_q4.err
fi
echo done
