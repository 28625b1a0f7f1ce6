#!/bin/bash
python -m pytest tests/test_hip_leaves.py tests/test_hip_configs.py -q -x -k "real_weight or wide_panel_adjoint or config3" > gpurun_out/t_real3.log 2>&1; tail -3 gpurun_out/t_real3.log
for v in a b; do
  f=gpurun_out/real_wide_$v.json
  if [ $v = a ]; then python bench.py --config 3 --steps 8 --no-cpu-baseline > $f 2> ${f%.json}.log
  else python tools/run_with_tuning.py real_entries=False -- --config 3 --steps 8 --no-cpu-baseline > $f 2> ${f%.json}.log; fi
  python -c "
import json,sys
d=json.load(open('$f'))
print('$v fwd', round(d['ms_per_step'],4), 'adj', round(d['config']['adjoint_ms'],4), d['config']['adjoint_frac_of_peak_reference_model'], d['config']['adjoint_parity_rel_err_vs_float64'])
print('   ', {k: v['avg_ms'] for k,v in d['kernels']['adjoint'].items()})
"
done
