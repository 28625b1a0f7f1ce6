#!/bin/bash
set -x
mkdir -p gpurun_out/r06c
timeout -k 10 600 python -m pytest tests/test_hip_gridsep.py -q > gpurun_out/r06c/test_gridsep.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r06c/test_gridsep.log
python bench.py --no-extras --no-cpu-baseline --parity --steps 10 > gpurun_out/r06c/headline.json 2> gpurun_out/r06c/headline.log
grep -E "grid_|csrmm|pack|ms/step|parity|setup" gpurun_out/r06c/headline.log
python bench.py --spokes-scale 8 --no-extras --no-cpu-baseline --parity --steps 5 > gpurun_out/r06c/dense.json 2> gpurun_out/r06c/dense.log
grep -E "grid_|csrmm|pack|ms/step|parity|setup" gpurun_out/r06c/dense.log
python bench.py --width 3 --no-extras --no-cpu-baseline --parity --steps 10 > gpurun_out/r06c/w3.json 2> gpurun_out/r06c/w3.log
grep -E "grid_|csrmm|pack|ms/step|parity|setup" gpurun_out/r06c/w3.log
