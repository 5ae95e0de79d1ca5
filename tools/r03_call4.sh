#!/bin/bash
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
nproc > gpurun_out/r03_box.txt; free -g | head -2 >> gpurun_out/r03_box.txt
timeout 1500 python -m pytest tests/test_hip_engine.py::test_seeded_model_vs_oracle tests/test_hip_baseline_configs.py -x -q -s -m gpu \
  -k "config4 or config2_small_prior_full_size or config3_1b_lyrics_top_prior_full" > gpurun_out/r03_parity_tests.log 2>&1
echo "rc=$?"; tail -30 gpurun_out/r03_parity_tests.log; cat gpurun_out/r03_box.txt
