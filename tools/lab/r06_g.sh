#!/bin/bash
# rehearsal of the direct all-reduce route with 2 and 4 ranks on the ONE GPU of the box (INDIGO_BENCH_DIST_BACKEND=gloo only means
# "all ranks share GPU 0" here: the collective is the library's own IPC route, no gloo, no RCCL)
mkdir -p gpurun_out/r06g
for n in 2 4; do
  INDIGO_BENCH_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus $n --comm direct --no-config5 --no-cpu-baseline --steps 5 > gpurun_out/r06g/direct_$n.json 2> gpurun_out/r06g/direct_$n.log
  echo "n=$n rc=$? $(grep -E 'communicator|ms/step' gpurun_out/r06g/direct_$n.log | tr '\n' ' ')"
  python -c "import json;d=json.load(open('gpurun_out/r06g/direct_$n.json'));print(d['n_gpus'], d['ms_per_step'], d.get('comm'))"
done
