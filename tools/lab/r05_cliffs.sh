#!/bin/bash
# round 5, GPU call 2: the full GPU suite with the coil-count tests, then evaluations at coil counts that are no power of two,
# and the default line with its dense-trajectory extra
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest ${R05_TESTS:-tests} -m gpu -x -q > gpurun_out/r05b_gputest.log 2>&1 || { tail -40 gpurun_out/r05b_gputest.log; exit 1; }
tail -2 gpurun_out/r05b_gputest.log
timeout -k 10 300 python bench.py --tree recipe --steps 5 --no-extras --no-cpu-baseline > gpurun_out/r05b_bench_tree_recipe.json 2> gpurun_out/r05b_bench_tree_recipe.log || { tail -20 gpurun_out/r05b_bench_tree_recipe.log; exit 1; }
python -c "import json;d=json.load(open('gpurun_out/r05b_bench_tree_recipe.json'));print('tree recipe', round(d['ms_per_step'],3), 'setup_s', d['setup_s'], d['roofline']['frac'])"
(time timeout -k 10 300 python tools/lab/pics_default_grid.py) > gpurun_out/r05b_pics_default_grid.log 2>&1 || { tail -20 gpurun_out/r05b_pics_default_grid.log; exit 1; }
tail -6 gpurun_out/r05b_pics_default_grid.log
for c in 12 6 3 5 7 9; do
  timeout -k 10 300 python bench.py --coils $c --steps 10 --no-extras --no-cpu-baseline > gpurun_out/r05b_bench_coils$c.json 2> gpurun_out/r05b_bench_coils$c.log || { tail -20 gpurun_out/r05b_bench_coils$c.log; exit 1; }
  python -c "import json;d=json.load(open('gpurun_out/r05b_bench_coils$c.json'));print('coils $c', round(d['ms_per_step'],3), d['config']['coil_chunk_widths'], 'setup', d['setup_s'])"
done
timeout -k 10 300 python bench.py --osf 1.25 --coils 12 --steps 10 --no-extras --no-cpu-baseline > gpurun_out/r05b_bench_osf125_coils12.json 2> gpurun_out/r05b_bench_osf125_coils12.log || { tail -20 gpurun_out/r05b_bench_osf125_coils12.log; exit 1; }
python -c "import json;d=json.load(open('gpurun_out/r05b_bench_osf125_coils12.json'));print('osf1.25 coils 12', round(d['ms_per_step'],3), d['config']['coil_chunk_widths'])"
timeout -k 10 300 python bench.py --osf 1.25 --coils 9 --steps 10 --no-extras --no-cpu-baseline > gpurun_out/r05b_bench_osf125_coils9.json 2> gpurun_out/r05b_bench_osf125_coils9.log || { tail -20 gpurun_out/r05b_bench_osf125_coils9.log; exit 1; }
python -c "import json;d=json.load(open('gpurun_out/r05b_bench_osf125_coils9.json'));print('osf1.25 coils 9', round(d['ms_per_step'],3), d['config']['coil_chunk_widths'])"
timeout -k 10 600 python bench.py --steps 10 --warmup 3 --no-config5 --no-leaf-configs > gpurun_out/r05b_bench_default_dense.json 2> gpurun_out/r05b_bench_default_dense.log || { tail -30 gpurun_out/r05b_bench_default_dense.log; exit 1; }
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05b_bench_default_dense.json'))
print('default', d['ms_per_step'], d['roofline']['frac'], 'flagged', d['config'].get('support_flagged_frac'))
x = d.get('dense_trajectory', {})
print('dense', x.get('ms_per_step'), x.get('setup_s'), x.get('parity_rel_err'), x.get('error'), (x.get('config') or {}).get('support_flagged_frac'))
for k, v in (x.get('kernels') or {}).items():
    print("   %-22s %.4f ms  frac %s" % (k, v["avg_ms"], v.get("frac_of_peak")))
PY
