#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_leaves.py tests/test_hip_configs.py tests/test_hip_operators.py tests/test_hip_pics.py tests/test_hip_dist.py -m gpu -x -q > gpurun_out/r05_head_tests.log 2>&1 || { tail -30 gpurun_out/r05_head_tests.log; exit 1; }
tail -2 gpurun_out/r05_head_tests.log
echo "== headline"; bash tools/lab/ab_libs.sh || exit 1
echo "== config 2"; bash tools/lab/ab_libs.sh --config 2 --steps 20 --no-cpu-baseline || exit 1
echo "== osf 1.25"; bash tools/lab/ab_libs.sh --osf 1.25 --steps 20 --no-extras --no-cpu-baseline || exit 1
echo "== config 5"; bash tools/lab/ab_libs.sh --config 5 --shard 0/1 --steps 5 --no-cpu-baseline || exit 1
echo "== default grid"; bash tools/lab/ab_libs.sh --image 480,208,308 --osf 640/480 --steps 10 --no-extras --no-cpu-baseline || exit 1
