#!/bin/bash
# the headline in separate processes, back to back on one box: evaluation ms, dominant-kernel fraction, per-pass averages, placement
set -o pipefail
mkdir -p gpurun_out
for i in 1 2 3 4 5 6; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/rep_$i.json 2> gpurun_out/rep_$i.log || { tail -20 gpurun_out/rep_$i.log; exit 1; }
  python - <<PY
import json
d = json.load(open('gpurun_out/rep_$i.json'))
print('run $i', round(d['ms_per_step'], 4), 'evals/s', round(d['value'], 1), 'dominant', d['roofline']['kernel'][-9:], round(d['roofline']['frac'], 4), ' '.join('%s %.3f' % (k[4:], v['avg_ms']) for k, v in d['kernels'].items() if k.startswith('fft_')), d['config']['placement'][0][1], flush=True)
PY
done
