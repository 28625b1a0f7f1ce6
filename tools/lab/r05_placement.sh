#!/bin/bash
# round 5: large arrays placed by probing: the best against the WORST of six candidates (lab switch), separate processes, alternating
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2 3; do
for pick in best worst; do
  timeout -k 10 300 python tools/run_with_tuning.py placement_candidates=6 placement_pick="'$pick'" -- --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/pl_$pick.json 2> gpurun_out/pl_$pick.log || { tail -20 gpurun_out/pl_$pick.log; exit 1; }
  python - <<PY
import json
d = json.load(open('gpurun_out/pl_$pick.json'))
print('%-5s' % '$pick', round(d['ms_per_step'], 4), ' '.join('%s %.4f' % (k[4:], v['avg_ms']) for k, v in d['kernels'].items() if k.startswith('fft_')), d['config'].get('placement'), flush=True)
PY
done
done
