#!/bin/bash
# First GPU call of the next round (through gpurun from the repo root): the whole GPU suite and the 20-second job -- the state
# every later change is compared with.  Outputs land in gpurun_out/; copy what is to be judged into profiles/ (r05_*).
# Always `python -u` and a `timeout` of your own: a call that runs into gpurun's limit is lost.
#
# Where round 4 left the job (profiles/r04_bench_full_1gpu*.json): 215 s = 9 s until level 0 starts + 68 s
# with level 1 beside it (plain chains, 2.28 ms per step each) + 20 windows of 6.9 s alone (4096 x 1.56 ms pipelined + 0.5 s).
# Candidates, by what they could take off the 215 s, with what is already known:
#   1. the pipelined phase (5.2-5.6 us: 2.9 inputs seen -> stores issued, 1.4 of it the sc1 fetch of the 61-KB activation
#      block; ~2.3 until the consumer sees the flags): every idea inside this decomposition has now been measured --
#      data-carrying flags, "stores issued" flags, 60 / 240 workgroups, per-wave dependencies (built into the engine on branch
#      wip/pipe-v12: no gain).  What is left is structural: pipelined forms of the multi-head / 16-wave kernels (top priors, 5b),
#      or a fused pipelined pair of the two upsampler levels (~6 s of the job for a lock-step scheduler).
#   2. prefill GEMM 670-745 TFLOP/s -> the guide's 8-phase 256x256 structure (counted vmcnt, raw barriers, 128 KB of LDS):
#      tools/gemm_glds_probe.hip takes a new tile variant and checks it bit for bit; worth ~1.5 s of the job.
#   3. conv stacks: DONE in round 4 (gemm_split_kernel, conditioner 133 -> 77 ms); staging both operands through LDS (128 x 128
#      tile, split once per tile) would reach ~45 ms: ~0.8 s of the job.
#   4. 5b_lyrics decode (4.06-4.16 ms, 33 % of HBM; profiles/r04_5b_kernel_stats.csv: 12.0 / 14.6 / 6.6 / 6.5 us per launch).
#      Pipelined launches for multi-head engines exist on branch wip/pipe-5b (PIPE form of the MFMA decode attention, 16- / 4-wave
#      PIPE projections, eligibility rule, pad launch for an odd launch count): bit-identical in 5 cases, and NO faster at 5b
#      (4.15 ms): a 16-wave workgroup fills a compute unit, so the next launch is not resident while this one streams.  The same
#      branch carries the kernel that should change that, written after the budget was spent and NEVER RUN: gemv_long_kernel (8
#      waves walking their k-tiles through two register stages of 5 fragments, >= 10 requests in flight per wave by the ISA, 2
#      workgroups per compute unit, all 300 column tiles resident; jb_tune_gemv_long(1)) with its tests
#      (test_gemv_long_rows_on_8_waves, the [..-1] cases of test_pipelined_launches_equal_the_plain_chain); the commented lines at
#      the end of tools/r04_call24.sh on the branch are its first call.  Merge only what the measurement keeps.
#   5. one prefill chunk per window (measured: no faster, 244 vs 240 ms) would let all wide-value layers share ONE S-wide V buffer:
#      -9 GB per upsampler engine; memory only.
# Dead ends measured in round 4 (do not repeat): flag bytes / flag words in one line; one barrier-less AQL queue of our own;
# the pair of streams made early; an event join in the caller's queue; confining the upper levels to 64 compute units; both
# upsampler levels pipelined on disjoint halves of the chip (an attention workgroup needs an EMPTY compute unit: with fewer
# than ~184 compute units the waiting projection's 120 workgroups leave too few).
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -u -m pytest tests -q -m gpu -p no:cacheprovider --durations=10 > gpurun_out/r05_gpu_tests.log 2>&1; tail -14 gpurun_out/r05_gpu_tests.log
JB_BENCH_BUDGET_S=480 JB_BENCH_TIMELINE=1 timeout 700 python -u bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_full_1gpu.json 2> gpurun_out/r05_bench_full_1gpu.err
cut -c1-600 gpurun_out/r05_bench_full_1gpu.json
