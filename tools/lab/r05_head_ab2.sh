#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_configs.py -m gpu -x -q -k "sense or coil or chirp or config2 or config4" > gpurun_out/r05_head_tests2.log 2>&1 || { tail -30 gpurun_out/r05_head_tests2.log; exit 1; }
tail -2 gpurun_out/r05_head_tests2.log
echo "== headline"; bash tools/lab/ab_libs.sh || exit 1
echo "== config 2"; bash tools/lab/ab_libs.sh --config 2 --steps 20 --no-cpu-baseline || exit 1
echo "== osf 1.25"; bash tools/lab/ab_libs.sh --osf 1.25 --steps 20 --no-extras --no-cpu-baseline || exit 1
echo "== config 5"; bash tools/lab/ab_libs.sh --config 5 --shard 0/1 --steps 5 --no-cpu-baseline || exit 1
