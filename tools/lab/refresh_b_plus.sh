#!/bin/bash
# part B of the refresh + a second default line on the same box + the multi-rank extra (with and without its time limit), gloo rehearsal
bash tools/refresh_profiles.sh r04 B > gpurun_out/r04_refresh_B.log 2>&1; tail -32 gpurun_out/r04_refresh_B.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_default_b.json 2> gpurun_out/r04_bench_default_b.log; tail -1 gpurun_out/r04_bench_default_b.log
INDIGO_BENCH_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r04_selflaunch_2rank_with_cfg5.json 2> gpurun_out/r04_selflaunch_2rank_with_cfg5.log; echo rc=$?
python -c "import json;d=json.load(open('gpurun_out/r04_selflaunch_2rank_with_cfg5.json'));print(d['n_gpus'], d['ms_per_step'], d['config5'].get('ms_per_step'), d['config5'].get('error'))"
INDIGO_BENCH_EXTRA_LIMIT=2 INDIGO_BENCH_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r04_selflaunch_2rank_giveup.json 2> gpurun_out/r04_selflaunch_2rank_giveup.log; echo rc=$?
python -c "import json;d=json.load(open('gpurun_out/r04_selflaunch_2rank_giveup.json'));print(d['n_gpus'], d['ms_per_step'], d['config5'])"
