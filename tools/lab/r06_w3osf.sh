#!/bin/bash
mkdir -p gpurun_out/r06w
run() { # name, tuning..., -- bench args
  name=$1; shift
  timeout -k 10 300 python tools/run_with_tuning.py "$@" > gpurun_out/r06w/$name.json 2> gpurun_out/r06w/$name.log || { echo "$name FAILED"; tail -5 gpurun_out/r06w/$name.log; return; }
  python -c "
import json; d=json.load(open('gpurun_out/r06w/$name.json')); print('$name', round(d['ms_per_step'],3), '%.2e' % d['parity_rel_err']['vs_float64_evaluation'], {k:round(v['avg_ms'],3) for k,v in d['kernels'].items() if 'scatter' in k or 'bricks_conj' in k})"
}
B="--osf 1.25 --width 3 --steps 3 --warmup 1 --no-extras --no-cpu-baseline --parity"
for rep in 1 2 3 4; do
run stored_$rep separable=False -- $B
run c256_$rep "share_shape={8:(4,4,256,1024),4:(4,4,256,1024)}" -- $B
run c128_$rep "share_shape={8:(4,4,128,1024),4:(4,4,128,1024)}" -- $B
run c64_$rep "share_shape={8:(4,4,64,1024),4:(4,4,64,1024)}" -- $B
done
