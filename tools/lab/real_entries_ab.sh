#!/bin/bash
# real-weight formats (8-byte brick entries, 12-byte slot entries, 4-byte gather values): tests, then the headline and the 1- / 2- / 4-coil
# shards with and without them (same box)
python -m pytest tests/test_hip_leaves.py tests/test_hip_operators.py tests/test_hip_configs.py tests/test_hip_stress.py -q -x -k "brick or slot or sense or config4 or config5 or stress or gridding" > gpurun_out/t_real.log 2>&1; tail -3 gpurun_out/t_real.log
show() { python -c "
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], round(d['ms_per_step'],4), round(d['roofline']['frac'],4), (d.get('parity_rel_err') or {}).get('vs_float64_evaluation'))
print('   ', ', '.join('%s %.4f' % (k, v['avg_ms']) for k,v in d['kernels'].items() if 'csrmm' in k))
" $1; }
for sh in none 0/2 0/4 0/8; do
  for v in a b; do
    extra=""; [ $sh != none ] && extra="--shard $sh"
    f=gpurun_out/real_${v}_$(echo $sh | tr / of).json
    if [ $v = a ]; then python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline $extra > $f 2> ${f%.json}.log
    else python tools/run_with_tuning.py real_entries=False -- --steps 20 --warmup 5 --no-extras --no-cpu-baseline $extra > $f 2> ${f%.json}.log; fi
    show $f
  done
done
