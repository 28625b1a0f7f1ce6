#!/bin/bash
# round 5: full support tables (bitmaps) on chirp-z z axes -- parity, then the default grid with bitmaps / hulls only
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_leaves.py tests/test_hip_configs.py tests/test_hip_pics.py -m gpu -x -q -k "chirp or default_oversampling or padded or reference_drivers or support" > gpurun_out/r05g_tests.log 2>&1 || { tail -40 gpurun_out/r05g_tests.log; exit 1; }
tail -2 gpurun_out/r05g_tests.log
for bm in True False; do   # False: the grid without its table, as in round 4
  timeout -k 10 400 python tools/run_with_tuning.py support_chirp=$bm -- --image 480,208,308 --osf 640/480 --steps 10 --no-extras > gpurun_out/r05g_default_grid_bitmaps_$bm.json 2> gpurun_out/r05g_default_grid_bitmaps_$bm.log || { tail -20 gpurun_out/r05g_default_grid_bitmaps_$bm.log; exit 1; }
  python - <<PY
import json
d = json.load(open('gpurun_out/r05g_default_grid_bitmaps_$bm.json'))
print('640x277x410 bitmaps=$bm', round(d['ms_per_step'], 3), d['config'].get('support_flagged_frac'), d['parity_rel_err']['vs_float64_evaluation'], ' '.join('%s %.3f' % (k, v['avg_ms']) for k, v in d['kernels'].items()))
PY
done
(time timeout -k 10 300 python tools/lab/pics_default_grid.py) > gpurun_out/r05g_pics_default_grid.log 2>&1 || { tail -20 gpurun_out/r05g_pics_default_grid.log; exit 1; }
tail -6 gpurun_out/r05g_pics_default_grid.log
