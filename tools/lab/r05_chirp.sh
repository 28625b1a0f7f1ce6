#!/bin/bash
# round 5: the chirp-z kernel in packed arithmetic -- parity, then the reference driver's default grid again
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_leaves.py tests/test_hip_configs.py tests/test_hip_pics.py -m gpu -x -q -k "chirp or default_oversampling or generated or a_x_b or AxB or smooth or padded or reference_drivers" > gpurun_out/r05f_tests.log 2>&1 || { tail -40 gpurun_out/r05f_tests.log; exit 1; }
tail -2 gpurun_out/r05f_tests.log
timeout -k 10 300 python tools/lab/chirp_fft.py > gpurun_out/r05f_chirp_fft.log 2>&1 || { tail -20 gpurun_out/r05f_chirp_fft.log; exit 1; }
tail -6 gpurun_out/r05f_chirp_fft.log
timeout -k 10 400 python bench.py --image 480,208,308 --osf 640/480 --steps 10 --no-extras > gpurun_out/r05f_bench_default_grid.json 2> gpurun_out/r05f_bench_default_grid.log || { tail -30 gpurun_out/r05f_bench_default_grid.log; exit 1; }
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05f_bench_default_grid.json'))
print('640x277x410', d['ms_per_step'], d['value'], 'setup', d['setup_s'], d.get('parity_rel_err'))
for k, v in d['kernels'].items():
    print("   %-22s %.4f ms  frac %s" % (k, v["avg_ms"], v.get("frac_of_peak")))
PY
timeout -k 10 300 python bench.py --osf 1.25 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r05f_osf125.json 2> gpurun_out/r05f_osf125.log || exit 1
python -c "import json;d=json.load(open('gpurun_out/r05f_osf125.json'));print('osf125', d['ms_per_step'])"
(time timeout -k 10 300 python tools/lab/pics_default_grid.py) > gpurun_out/r05f_pics_default_grid.log 2>&1 || { tail -20 gpurun_out/r05f_pics_default_grid.log; exit 1; }
tail -6 gpurun_out/r05f_pics_default_grid.log
