#!/bin/bash
# config 3 adjoint (register-image scatter): brick shapes / task shapes; prints the scatter kernel's time per variant
run() {
  timeout -k 10 300 python tools/run_with_tuning.py "$@" -- --config 3 --steps 5 > gpurun_out/ws.json 2> gpurun_out/ws.log || { echo "FAILED $*"; tail -3 gpurun_out/ws.log; return; }
  python -c "
import json,sys
d=json.loads(open('gpurun_out/ws.json').read().strip().splitlines()[-1])
a=d['kernels']['adjoint']
print('%-70s adjoint %.3f ms  scatter %.3f  zero %.3f  pack %.3f  err %.2e' % (' '.join(sys.argv[1:]), d['config']['adjoint_ms'], a['csrmm_bricks_wide_conj']['avg_ms'], a['bricks_wide_zero']['avg_ms'], a['pack_panel']['avg_ms'], d['config']['adjoint_parity_rel_err_vs_float64']))" "$@"
}
for v in "$@"; do run $v; done
