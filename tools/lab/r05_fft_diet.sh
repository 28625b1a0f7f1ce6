#!/bin/bash
# round 5, first GPU call: the instruction diet of k_fft_2stage -- parity of every transform test, then the headline / osf 1.25 /
# config 2 / config 5 timings with the per-kernel table
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_leaves.py tests/test_hip_configs.py tests/test_hip_operators.py -m gpu -x -q -k "fft or FFT or sense or padded or cropped or config or chirp or golden" > gpurun_out/r05a_tests.log 2>&1 || { tail -30 gpurun_out/r05a_tests.log; exit 1; }
tail -3 gpurun_out/r05a_tests.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r05a_bench_cfg4.json 2> gpurun_out/r05a_bench_cfg4.log || { tail -20 gpurun_out/r05a_bench_cfg4.log; exit 1; }
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05a_bench_cfg4.json'))
print('cfg4', d['ms_per_step'], d['roofline']['frac'])
for k, v in d.get("kernels", {}).items():
    print("   %-22s %.4f ms  frac %.3f" % (k, v["avg_ms"], v.get("frac_of_peak", 0)))
PY
timeout -k 10 200 python bench.py --config 2 --steps 20 > gpurun_out/r05a_bench_cfg2.json 2> gpurun_out/r05a_bench_cfg2.log || exit 1
python -c "import json;d=json.load(open('gpurun_out/r05a_bench_cfg2.json'));print('cfg2', d['ms_per_step'], d['roofline']['frac'])"
timeout -k 10 300 python bench.py --osf 1.25 --steps 10 --no-extras --no-cpu-baseline > gpurun_out/r05a_bench_osf125.json 2> gpurun_out/r05a_bench_osf125.log || exit 1
python -c "import json;d=json.load(open('gpurun_out/r05a_bench_osf125.json'));print('osf125', d['ms_per_step'])"
timeout -k 10 300 python bench.py --config 5 --shard 0/1 --steps 5 --no-cpu-baseline > gpurun_out/r05a_bench_cfg5.json 2> gpurun_out/r05a_bench_cfg5.log || exit 1
python -c "import json;d=json.load(open('gpurun_out/r05a_bench_cfg5.json'));print('cfg5', d['ms_per_step'])"
