#!/bin/bash
mkdir -p gpurun_out/r06f
for dbg in 0 1 2 4 3 7; do
  python tools/run_with_tuning.py sep_dbg=$dbg -- --no-extras --no-cpu-baseline --steps 5 --spokes-scale 8 > gpurun_out/r06f/d$dbg.json 2> gpurun_out/r06f/s.log
  echo "dbg $dbg (1 no mfma, 2 no gather, 4 no flush): $(grep -E 'grid_scatter_sep' gpurun_out/r06f/s.log | awk '{print $6}') ms scatter"
done
