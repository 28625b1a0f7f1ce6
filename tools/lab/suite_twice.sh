#!/bin/bash
# the GPU suite twice in one call (the one intermittent failure of round 3 happened INSIDE a suite run): summaries appended to a log
for i in 1 2; do
  python -m pytest tests -m gpu -q -x > gpurun_out/suite_run_$i.log 2>&1
  echo "suite run $i on $(hostname) at $(date -u +%H:%M:%S): $(tail -1 gpurun_out/suite_run_$i.log)" | tee -a gpurun_out/r04_suite_repeats.log
done
