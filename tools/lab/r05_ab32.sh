#!/bin/bash
# round 5: 32-column tiles on the A x B passes (osf 1.25) A/B, support hulls on the chirp-z grid A/B, torchrun rehearsal over gloo
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_operators.py tests/test_hip_leaves.py tests/test_hip_configs.py -m gpu -x -q -k "a_x_b or padded or cropped or reference_drivers or every_coil or chirp or zpadfft or support" > gpurun_out/r05e_tests.log 2>&1 || { tail -40 gpurun_out/r05e_tests.log; exit 1; }
tail -2 gpurun_out/r05e_tests.log
for rep in 1 2; do
for w in 1 0; do
  timeout -k 10 300 python tools/run_with_tuning.py opt:fft.ab_w32=$w -- --osf 1.25 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r05e_osf125_w32_$w.json 2> gpurun_out/r05e_osf125_w32_$w.log || { tail -20 gpurun_out/r05e_osf125_w32_$w.log; exit 1; }
  python - <<PY
import json
d = json.load(open('gpurun_out/r05e_osf125_w32_$w.json'))
print('osf1.25 ab_w32=$w', round(d['ms_per_step'], 4), ' '.join('%s %.3f' % (k[4:], v['avg_ms']) for k, v in d['kernels'].items() if k.startswith('fft_')))
PY
done
done
for h in True False; do
  timeout -k 10 400 python tools/run_with_tuning.py support_hulls=$h -- --image 480,208,308 --osf 640/480 --steps 10 --no-extras --no-cpu-baseline > gpurun_out/r05e_default_grid_hulls_$h.json 2> gpurun_out/r05e_default_grid_hulls_$h.log || { tail -20 gpurun_out/r05e_default_grid_hulls_$h.log; exit 1; }
  python - <<PY
import json
d = json.load(open('gpurun_out/r05e_default_grid_hulls_$h.json'))
print('640x277x410 hulls=$h', round(d['ms_per_step'], 3), ' '.join('%s %.3f' % (k[4:], v['avg_ms']) for k, v in d['kernels'].items() if k.startswith('fft_')))
PY
done
for img in "640,480,480" "480,480,480"; do
  timeout -k 10 400 python bench.py --image 384,384,384 --osf 1.25 --steps 10 --no-extras --no-cpu-baseline > gpurun_out/r05e_osf125_384.json 2> gpurun_out/r05e_osf125_384.log || { tail -5 gpurun_out/r05e_osf125_384.log; exit 1; }
done
python -c "import json;d=json.load(open('gpurun_out/r05e_osf125_384.json'));print('384^3 on 480^3', d['ms_per_step'])"
INDIGO_BENCH_DIST_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --no-config5 > gpurun_out/r05e_torchrun_2rank_gloo.json 2> gpurun_out/r05e_torchrun_2rank_gloo.log || { tail -20 gpurun_out/r05e_torchrun_2rank_gloo.log; exit 1; }
python -c "import json;d=json.load(open('gpurun_out/r05e_torchrun_2rank_gloo.json'));print('torchrun gloo 2 ranks: valid JSON,', d['n_gpus'], d['ms_per_step'])"
