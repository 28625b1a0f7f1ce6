#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_leaves.py tests/test_hip_configs.py -m gpu -x -q -k "chirp" > gpurun_out/r05_chirp_tests.log 2>&1 || { tail -30 gpurun_out/r05_chirp_tests.log; exit 1; }
tail -2 gpurun_out/r05_chirp_tests.log
echo "== default grid"; bash tools/lab/ab_libs.sh --image 480,208,308 --osf 640/480 --steps 10 --no-extras --no-cpu-baseline || exit 1
