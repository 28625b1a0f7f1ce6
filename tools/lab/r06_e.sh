#!/bin/bash
mkdir -p gpurun_out/r06e
timeout -k 10 300 python -m pytest tests/test_hip_gridsep.py -q -x > gpurun_out/r06e/test_gridsep.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r06e/test_gridsep.log
for sc in 1 8; do
  python bench.py --no-extras --no-cpu-baseline --parity --steps 5 --spokes-scale $sc > gpurun_out/r06e/mfma_$sc.json 2> gpurun_out/r06e/s.log
  echo "spokes x$sc: $(grep -E 'grid_scatter_sep' gpurun_out/r06e/s.log | awk '{print $6}') ms scatter; $(grep -E 'ms/step' gpurun_out/r06e/s.log | awk '{print $4}') ms/step $(grep parity gpurun_out/r06e/s.log)"
done
python bench.py --no-extras --no-cpu-baseline --parity --steps 5 --width 3 > gpurun_out/r06e/mfma_w3.json 2> gpurun_out/r06e/s.log
echo "w3: $(grep -E 'grid_scatter_sep' gpurun_out/r06e/s.log | awk '{print $6}') ms scatter; $(grep -E 'ms/step' gpurun_out/r06e/s.log | awk '{print $4}') ms/step $(grep parity gpurun_out/r06e/s.log)"
