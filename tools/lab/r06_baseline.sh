#!/bin/bash
# round 6, first GPU call: where the round starts -- headline, dense trajectory, and the reference's default kernel width 3
set -x
mkdir -p gpurun_out/r06a
python bench.py --no-extras --no-cpu-baseline --parity --steps 10 > gpurun_out/r06a/base_headline.json 2> gpurun_out/r06a/base_headline.log &&
python bench.py --spokes-scale 8 --no-extras --no-cpu-baseline --steps 5 > gpurun_out/r06a/base_dense.json 2> gpurun_out/r06a/base_dense.log &&
python bench.py --width 3 --no-extras --no-cpu-baseline --parity --steps 10 > gpurun_out/r06a/w3_headline.json 2> gpurun_out/r06a/w3_headline.log &&
python bench.py --width 3 --image 480,208,308 --osf 640/480 --no-extras --no-cpu-baseline --parity --steps 5 > gpurun_out/r06a/w3_default_grid.json 2> gpurun_out/r06a/w3_default_grid.log
echo "rc=$?"
tail -3 gpurun_out/r06a/*.log
