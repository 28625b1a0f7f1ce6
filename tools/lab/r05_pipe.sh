#!/bin/bash
# round 5: the two-tile pipelined passes (fft.pipe) -- parity, A/B timing on one box -- and the CG graph
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_operators.py tests/test_hip_leaves.py tests/test_hip_configs.py -m gpu -x -q -k "pipelined or padded or cropped or sense or zpadfft or config4 or support" > gpurun_out/r05c_tests.log 2>&1 || { tail -40 gpurun_out/r05c_tests.log; exit 1; }
tail -2 gpurun_out/r05c_tests.log
for rep in 1 2; do
for pipe in 1 0; do
  timeout -k 10 300 python tools/run_with_tuning.py opt:fft.pipe=$pipe -- --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r05c_bench_pipe$pipe.json 2> gpurun_out/r05c_bench_pipe$pipe.log || { tail -20 gpurun_out/r05c_bench_pipe$pipe.log; exit 1; }
  python - <<PY
import json
d = json.load(open('gpurun_out/r05c_bench_pipe$pipe.json'))
print('pipe $pipe', round(d['ms_per_step'], 4), ' '.join('%s %.3f' % (k[4:], v['avg_ms']) for k, v in d['kernels'].items() if k.startswith('fft_')))
PY
done
done
timeout -k 10 300 python tools/run_with_tuning.py opt:fft.pipe=1 -- --config 5 --shard 0/1 --steps 5 --no-cpu-baseline > gpurun_out/r05c_cfg5.json 2> gpurun_out/r05c_cfg5.log || exit 1
python -c "import json;d=json.load(open('gpurun_out/r05c_cfg5.json'));print('cfg5', d['ms_per_step'])"
timeout -k 10 400 python -X faulthandler tools/cg_bench.py 40 > gpurun_out/r05c_cg.log 2>&1
tail -8 gpurun_out/r05c_cg.log
CG_PROFILE=1 timeout -k 10 300 python -X faulthandler tools/cg_bench.py 40 > gpurun_out/r05c_cg_prof.log 2>&1
tail -25 gpurun_out/r05c_cg_prof.log
