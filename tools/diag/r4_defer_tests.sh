#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r4_defer_tests.log; : > $L
timeout -k 10 900 python -m pytest tests/test_towers_gpu.py tests/test_configs_gpu.py tests/test_fullsize_gpu.py tests/test_trajectory_gpu.py tests/test_amp_gpu.py tests/test_checkpoint_gpu.py tests/test_edge_batches_gpu.py -m gpu -q >> $L 2>&1
grep -E "^FAILED|^ERROR|passed|failed|matched-oracle parity|worst" $L | cut -c1-900
