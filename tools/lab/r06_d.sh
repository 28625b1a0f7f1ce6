#!/bin/bash
# sweep of the share scatter's brick shape / task size on the headline and the dense trajectory
mkdir -p gpurun_out/r06d
timeout -k 10 300 python -m pytest tests/test_hip_gridsep.py -q -x > gpurun_out/r06d/test_gridsep.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r06d/test_gridsep.log
for shape in "(8,2,1024,1024)" "(4,4,1024,1024)" "(4,2,1024,1024)" "(2,2,1024,1024)" "(8,2,256,256)" "(8,2,4096,4096)" "(4,2,256,512)"; do
  for sc in 1 8; do
    python tools/run_with_tuning.py "share_shape={8:$shape}" -- --no-extras --no-cpu-baseline --steps 5 --spokes-scale $sc > gpurun_out/r06d/s_${shape//[(),]/_}_$sc.json 2> gpurun_out/r06d/s.log
    echo "shape $shape spokes x$sc: $(grep -E 'grid_scatter_sep' gpurun_out/r06d/s.log | awk '{print $6}') ms scatter; $(grep -E 'ms/step' gpurun_out/r06d/s.log | awk '{print $4}') ms/step"
  done
done
