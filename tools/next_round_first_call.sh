#!/bin/bash
# First GPU calls of the next round (through gpurun from the repo root).  Outputs land in gpurun_out/; copy what is to be
# judged into profiles/ (r04_*).  Always `python -u` and a `timeout` of your own: a call that runs into gpurun's limit is lost.
#
#   call A (~4 min): why are software-pipelined launches slow inside the three-level job?  (finding 3)
#       AMD_LOG_LEVEL=2 JB_PIPE_TIMEOUT_MS=50 timeout 200 python -u tools/pipe_in_job.py --seconds 6 > gpurun_out/r04_pipe_in_job.log 2>&1
#     (AMD_LOG_LEVEL=2: the runtime's warnings -- grep the log for "Packet capture failed" / "Failed to allocate kernel argument
#     pool": hipGraphLaunch has a slow per-node path when an executable graph could not get its pre-built packets, and the
#     in-job numbers -- 2.96 ms per step = 290 x 10 us, whatever the GPU does -- look like a host-bound enqueue.  The tool prints the
#     host's enqueue time next to the total: equal -> the graph launches are the bottleneck, not the GPU.)
#     and once more with GPU_MAX_HW_QUEUES=4 in front (the package raises the runtime's default of 4 to 8 -- jukebox_amd/__init__.py --
#     which was only ever tested in tools/bench_engine.py, never inside the job: with two priority classes in use that is up to
#     16 pooled hardware queues + the two CU-mask queues, close to what the hardware scheduler maps at once).
#     and with --late (streams and graphs made when the launches are switched on, as in this round's slow runs; without it
#     they are made at the engine's first decode, as in the one fast run).
#     reads: A (job's engine as left) vs A' (plain) vs B (fresh streams) vs D (new engine) vs E (worker thread), each with the
#     per-call / per-step split and the per-slot stamps.  B fast -> create the pair when the level becomes the only one running;
#     D slow too -> process state (count HSA queues: rocprofv3 --hsa-trace of a 64-step call); only A slow -> engine state.
#   call A' (~2.5 min each, only once call A says the launches are fast inside the job): the 6-second bench with
#       JB_PIPELINE_LAUNCHES=1                                              (level 0 pipelined while it runs alone)
#       JB_PIPELINE_LAUNCHES=1 JB_PIPELINE_WHEN=always JB_PIPE_RESERVE_CUS=64  (from its first step, 8 CUs per XCD left to levels 2 / 1)
#     against 79.8 s plain; level 0 alone is 163 of the 242 s of the 20-second job, level 0 altogether 206 s.
#   call A'' (~1 min): completion protocol V4 of tools/pipelined_launch_probe.hip (8 shard tickets, the shard's last arriver stores one
#     BYTE of the slot's 8-byte flag word, the consumer polls that word) against V1 (one ticket, one flag):
#       tools/pipelined_launch_probe 40 1 1 | grep "2 graphs"; tools/pipelined_launch_probe 40 4 4 | grep "2 graphs"
#     V4 below V1's 4.63-4.74 us per phase -> replace the engine's two-level ticket (jb_pipe_publish: 1.9 us of a launch's 4.8):
#     the engine form is on the local branch wip/pipe-v4 (common.h + engine.hip; engines of >= 8 samples) --
#     `tools/bench_engine.py up --pipelined 1` against 1.585 ms, then test_pipelined_launches_equal_the_plain_chain.
#   call B (~1.5 min): the prefill-GEMM candidate against the library kernel, bit-for-bit and timed
#       hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I tools tools/gemm_glds_probe.hip -L jukebox_amd/csrc -ljukebox_hip \
#             -Wl,-rpath,$PWD/jukebox_amd/csrc -o tools/gemm_glds_probe     (build HERE, the binary travels)
#       timeout 120 tools/gemm_glds_probe > gpurun_out/r04_gemm_glds_probe.log 2>&1
#     EQUAL + >= 800 TFLOP/s on the 32768-row shapes -> the kernel inside jb_gemm is ready on the local branch wip/gemm-glds
#     (git checkout wip/gemm-glds -- jukebox_amd/csrc/gemm.hip; JB_GEMM_GLDS=1 selects it): prefill tests + tools/bench_prefill.py
#     with and without the variable, then make it the default for flat fp16 problems.
#   call B' (~2 min): the transpose pattern's prefill attention with 16-byte tile copies -- local branch wip/prefill-attn-vec (one file):
#       git checkout wip/prefill-attn-vec -- jukebox_amd/csrc/attention.hip && python -m jukebox_amd.csrc.build
#       timeout 200 python -u -m pytest tests/test_hip_kernels.py tests/test_hip_engine.py -q -m gpu -k "prefill or engine" ; python -u tools/bench_prefill.py
#     keep it if the tests pass and the window's prefill gets shorter (attn_prefill_kernel<f16,30>: 96 calls x 1.18 ms in
#     profiles/r03_full_job_kernel_stats.csv), `git checkout main -- jukebox_amd/csrc/attention.hip` otherwise.
#   call C (this script, ~11 min): the whole GPU suite in ONE process (two xdist workers were no faster: the long cases wait for
#     the CPU oracle) and the driver's bench command against a short wall budget.
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
timeout 700 python -u -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r04_gpu_tests.log 2>&1; tail -3 gpurun_out/r04_gpu_tests.log
JB_BENCH_BUDGET_S=420 JB_BENCH_TIMELINE=1 timeout 560 python -u bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_full_1gpu.json 2> gpurun_out/r04_bench_full_1gpu.err
cut -c1-600 gpurun_out/r04_bench_full_1gpu.json
