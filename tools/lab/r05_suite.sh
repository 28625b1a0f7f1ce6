#!/bin/bash
# round 5: full GPU suite, then the reference driver's default grid (640 x 277 x 410) with the support hulls, torchrun rehearsal
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r05d_gputest.log 2>&1 || { tail -40 gpurun_out/r05d_gputest.log; exit 1; }
tail -2 gpurun_out/r05d_gputest.log
timeout -k 10 400 python bench.py --image 480,208,308 --osf 640/480 --steps 10 --no-extras > gpurun_out/r05d_bench_default_grid.json 2> gpurun_out/r05d_bench_default_grid.log || { tail -30 gpurun_out/r05d_bench_default_grid.log; exit 1; }
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05d_bench_default_grid.json'))
print('640x277x410', d['ms_per_step'], d['value'], 'setup', d['setup_s'], d['config'].get('support_table'), d['config'].get('support_flagged_frac'), d.get('parity_rel_err'))
for k, v in d['kernels'].items():
    print("   %-22s %.4f ms  frac %s" % (k, v["avg_ms"], v.get("frac_of_peak")))
PY
