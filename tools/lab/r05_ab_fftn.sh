#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_leaves.py -m gpu -x -q -k "fft" > gpurun_out/r05_fftn_tests.log 2>&1 || { tail -30 gpurun_out/r05_fftn_tests.log; exit 1; }
tail -2 gpurun_out/r05_fftn_tests.log
for cfg in "320 8" "480 2" "640 1" "432 2"; do set -- $cfg; echo "== fftn $1^3 x $2"; bash tools/lab/ab_libs.sh --config 2 --image $1 --batch $2 --steps 10 --no-cpu-baseline || exit 1; done
