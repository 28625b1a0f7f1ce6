#!/bin/bash
# (the option fft.touch_table existed for that one run only -- profiles/r06_touch_table_ab.txt keeps the result; this script is its record)
# round 6, VERDICT r5 item 6: the cropped z pass waits for its support records (an HBM trip) before it can issue its loads.  A/B on one
# box: the support table read once by a 5-microsecond kernel just before the pass (memory-side cache warm) against the plain pass.
mkdir -p gpurun_out
for i in 1 2; do
for v in 0 1; do
timeout -k 10 300 python tools/run_with_tuning.py opt:fft.touch_table=$v -- --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r06_touch_${v}_$i.json 2> gpurun_out/r06_touch_${v}_$i.log || exit 1
python - <<PY
import json
d=json.load(open('gpurun_out/r06_touch_${v}_$i.json'))
print('touch_table=$v run $i: eval', round(d['ms_per_step'],3), {n:x['avg_ms'] for n,x in d['kernels'].items() if n in ('fft_crop_z','fft_pad_z','fft_crop_y')})
PY
done; done
