#!/bin/bash
# 8-coil adjoint gridding with 8-byte entries: brick shapes x support-table granularity (the unrolled flushes exist for 16 x 2 x 2 only)
for tile in 4 8; do
for shape in "(2,2,4096,4096)" "(2,4,4096,4096)" "(4,2,4096,4096)" "(1,2,4096,4096)" "(2,1,4096,4096)"; do
  timeout -k 10 300 python tools/run_with_tuning.py "brick_shape={4: (2,4,4096,4096), 8: $shape}" "support_tile={8: $tile, 4: 8}" -- --steps 10 --no-extras --no-cpu-baseline > gpurun_out/bs.json 2> gpurun_out/bs.log || { echo "FAILED $shape tile $tile"; tail -3 gpurun_out/bs.log; continue; }
  python -c "
import json
d=json.loads(open('gpurun_out/bs.json').read().strip().splitlines()[-1])
print('8 coils, table $tile, brick $shape: %.3f ms' % d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items() if 'brick' in k or 'gather' in k or '_z' in k})"
done; done
