#!/bin/bash
# A/B on one box: large arrays as the best-placed 1-GB-step window of one allocation (placement_window_gb=8) against the best of three
# candidate allocations (round 5)
mkdir -p gpurun_out/r06h
for rep in 1 2; do
  for w in 8 0; do
    python tools/run_with_tuning.py placement_window_gb=$w -- --no-extras --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r06h/w${w}_$rep.json 2> gpurun_out/r06h/s.log
    echo "window_gb=$w rep $rep: $(grep -E 'ms/step' gpurun_out/r06h/s.log | awk '{print $4}') ms/step; y passes $(grep -E 'fft_pad_y|fft_crop_y' gpurun_out/r06h/s.log | awk '{print $6}' | tr '\n' ' ') z $(grep -E 'fft_pad_z|fft_crop_z' gpurun_out/r06h/s.log | awk '{print $6}' | tr '\n' ' ') x $(grep -E 'fft_pad_x|fft_crop_x' gpurun_out/r06h/s.log | awk '{print $6}' | tr '\n' ' ')"
    python -c "import json;d=json.load(open('gpurun_out/r06h/w${w}_$rep.json'));print('   placement', d['config']['placement'])"
  done
done
