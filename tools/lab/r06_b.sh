#!/bin/bash
set -x
mkdir -p gpurun_out/r06b
python -m pytest tests/test_hip_gridsep.py -x -q > gpurun_out/r06b/test_gridsep.log 2>&1; echo "pytest rc=$?" 
tail -5 gpurun_out/r06b/test_gridsep.log
python bench.py --no-extras --no-cpu-baseline --parity --steps 10 > gpurun_out/r06b/headline.json 2> gpurun_out/r06b/headline.log
grep -E "grid_gather|csrmm|ms/step|parity" gpurun_out/r06b/headline.log
python bench.py --spokes-scale 8 --no-extras --no-cpu-baseline --steps 5 > gpurun_out/r06b/dense.json 2> gpurun_out/r06b/dense.log
grep -E "grid_gather|csrmm|ms/step|parity" gpurun_out/r06b/dense.log
python bench.py --width 3 --no-extras --no-cpu-baseline --parity --steps 10 > gpurun_out/r06b/w3.json 2> gpurun_out/r06b/w3.log
grep -E "grid_gather|csrmm|ms/step|parity" gpurun_out/r06b/w3.log
