#!/bin/bash
# Round 4, GPU call 3: completion-protocol variants (probe + engine), full-size cases with the decode check on torch's GPU
# kernels, rocprofv3 summaries (roofline-only run; pipelined engine), BASELINE config 5 end to end (6 s of audio).
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
O=$PWD/gpurun_out
echo "== completion protocols V1 V5 V6 V7 (probe)"
for v in 1 6 5 7; do timeout 60 tools/pipelined_launch_probe 40 $v $v | grep -E "2 graphs|no wait" ; done > $O/r04_pipelined_launch_probe_protocols.log 2>&1; cat $O/r04_pipelined_launch_probe_protocols.log
echo "== engine: protocol 0 / 1"
for p in 0 1; do JB_PIPE_PROTO=$p JB_PIPE_DEBUG=1 timeout 200 python -u tools/bench_engine.py up --pipelined 1 --steps 512 > $O/r04_bench_engine_up_proto$p.log 2>&1; grep -E "graph=True|whole step|c_attn|c_fc " $O/r04_bench_engine_up_proto$p.log; done
timeout 120 python -u -m pytest tests/test_hip_engine.py -q -m gpu -p no:cacheprovider -k pipelined_launches 2>&1 | tail -3
echo "== full-size cases (decode check on torch's GPU kernels)"
timeout 500 python -u -m pytest tests/test_hip_baseline_configs.py -q -m gpu -p no:cacheprovider --durations=12 -k "full_size or full_depth or second_window or order10 or geometry_fast" > $O/r04_full_size_tests_gpu_port.log 2>&1; tail -16 $O/r04_full_size_tests_gpu_port.log
echo "== rocprofv3: roofline-only run"
cd /tmp && rm -rf /tmp/prof_rl && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rl -- python -u $GRAFT_REPO_ROOT/bench.py --roofline-only > $O/r04_roofline_only_stdout.json 2> $O/r04_roofline_only.err
f=$(find /tmp/prof_rl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r04_roofline_only_kernel_stats.csv && head -6 $O/r04_roofline_only_kernel_stats.csv | cut -c1-200
echo "== rocprofv3: pipelined engine"
rm -rf /tmp/prof_pipe && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pipe -- python -u $GRAFT_REPO_ROOT/tools/bench_engine.py up --pipelined 1 --steps 24 > $O/r04_prof_pipelined_engine.log 2>&1
f=$(find /tmp/prof_pipe -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r04_pipelined_engine_kernel_stats.csv && head -8 $O/r04_pipelined_engine_kernel_stats.csv | cut -c1-200
cd $GRAFT_REPO_ROOT
echo "== BASELINE config 5: 5b_lyrics, 3 samples, 6 s of audio, end to end"
JB_BENCH_TIMELINE=1 timeout 600 python -u bench.py --model 5b_lyrics --samples-per-gpu 3 --seconds 6 --steps 1 --warmup 0 --no-cpu-baseline > $O/r04_bench_5b_lyrics_6s_1gpu.json 2> $O/r04_bench_5b_lyrics_6s_1gpu.err; cut -c1-900 $O/r04_bench_5b_lyrics_6s_1gpu.json; tail -4 $O/r04_bench_5b_lyrics_6s_1gpu.err
echo done
