#!/bin/bash
# A/B of two builds of the library on ONE box, alternating: tools/lab/libindigo_hip_base.so (the previous build) against indigo_amd/lib/libindigo_hip.so.
# usage: bash tools/lab/ab_libs.sh [bench args ...]      (default: the headline, 20 steps)
set -o pipefail
mkdir -p gpurun_out
ARGS=${@:-"--steps 20 --warmup 5 --no-extras --no-cpu-baseline"}
for rep in 1 2; do
for which in base new; do
  if [ $which = base ]; then export INDIGO_HIP_LIB=$PWD/tools/lab/libindigo_hip_base.so; else unset INDIGO_HIP_LIB; fi
  timeout -k 10 300 python bench.py $ARGS > gpurun_out/ab_$which.json 2> gpurun_out/ab_$which.log || { tail -20 gpurun_out/ab_$which.log; exit 1; }
  python - <<PY
import json
d = json.load(open('gpurun_out/ab_$which.json'))
print('%-5s' % '$which', round(d['ms_per_step'], 4), ' '.join('%s %.4f' % (k[4:] if k.startswith('fft_') else k[6:], v['avg_ms']) for k, v in d['kernels'].items() if k.startswith('fft_') or k.startswith('csrmm_g') or k.startswith('csrmm_b')), flush=True)
PY
done
done
