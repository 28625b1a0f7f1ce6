#!/bin/bash
# adjoint gridding (brick-binned scatter) for several brick shapes, 4-coil rank and 8-coil tree
for shape in "(2,2,4096,4096)" "(4,2,4096,4096)" "(4,4,4096,4096)" "(2,4,4096,4096)"; do
  timeout -k 10 300 python tools/run_with_tuning.py "brick_shape={4: $shape, 8: $shape}" -- --shard 0/2 --steps 10 --no-extras --no-cpu-baseline > gpurun_out/bs.json 2> gpurun_out/bs.log || { echo "FAILED $shape"; tail -3 gpurun_out/bs.log; continue; }
  python -c "
import json
d=json.loads(open('gpurun_out/bs.json').read().strip().splitlines()[-1])
print('4 coils, brick $shape: %.3f ms' % d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items() if 'brick' in k or 'gather' in k})"
done
for shape in "(4,2,4096,4096)" "(4,4,4096,4096)"; do
  timeout -k 10 300 python tools/run_with_tuning.py "brick_shape={4: $shape, 8: $shape}" -- --steps 10 --no-extras --no-cpu-baseline > gpurun_out/bs.json 2> gpurun_out/bs.log || { echo "FAILED $shape"; tail -3 gpurun_out/bs.log; continue; }
  python -c "
import json
d=json.loads(open('gpurun_out/bs.json').read().strip().splitlines()[-1])
print('8 coils, brick $shape: %.3f ms' % d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items() if 'brick' in k or 'gather' in k})"
done
