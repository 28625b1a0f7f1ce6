#!/bin/bash
mkdir -p gpurun_out/r06f
timeout -k 10 600 python -m pytest tests/test_hip_gridsep.py tests/test_hip_stress.py -q -x > gpurun_out/r06f/t.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r06f/t.log
for a in "--image 480,208,308 --osf 640/480" "--image 480,208,308 --osf 640/480 --width 3" "--width 3" "--osf 1.25 --width 3"; do
python bench.py $a --no-extras --no-cpu-baseline --parity --steps 5 > gpurun_out/r06f/b.json 2> gpurun_out/r06f/b.log || { tail -5 gpurun_out/r06f/b.log; continue; }
python -c "
import json; d=json.load(open('gpurun_out/r06f/b.json')); print('$a', round(d['ms_per_step'],3), d['parity_rel_err']['vs_float64_evaluation'], {k:round(v['avg_ms'],2) for k,v in d['kernels'].items()})"
done
