#!/bin/bash
# round 5: where the 0.63 ms of k_grid_bricks<8, 8, PAIR, REALW> go -- ablated instantiations (results are WRONG by design; timing only)
#   1 no LDS read-add-write   2 no bpermutes, no accumulation   3 no stores in the flush   4 no panel-row loads   5 no flush at all
set -o pipefail
mkdir -p gpurun_out
for abl in 0 1 2 3 4 5 0; do
  IG_LAB_ABL=$abl timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r05_abl$abl.json 2> gpurun_out/r05_abl$abl.log || { tail -20 gpurun_out/r05_abl$abl.log; exit 1; }
  python - <<PY
import json
d = json.load(open('gpurun_out/r05_abl$abl.json'))
print('abl $abl', round(d['ms_per_step'], 4), ' '.join('%s %.3f' % (k, v['avg_ms']) for k, v in d['kernels'].items() if 'csrmm' in k or 'pack' in k))
PY
done
