#!/bin/bash
mkdir -p gpurun_out/r06c2
timeout -k 10 500 python -m pytest tests/test_hip_leaves.py tests/test_hip_configs.py tests/test_hip_operators.py -q -k "chirp or default_grid" > gpurun_out/r06c2/t.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r06c2/t.log
python bench.py --image 480,208,308 --osf 640/480 --no-extras --no-cpu-baseline --parity --steps 5 > gpurun_out/r06c2/dg.json 2> gpurun_out/r06c2/s.log
grep -E "fft_|ms/step|parity" gpurun_out/r06c2/s.log | awk '{print $2, $6, $NF}' | tr '\n' ';'; echo
timeout -k 10 200 python tools/lab/chirp_fft.py 2>&1 | tail -4
