#!/bin/bash
# everything under profiles/ that round 2 cites, re-measured in one GPU call (copy gpurun_out/r02_* into profiles/ afterwards)
mkdir -p gpurun_out
timeout -k 10 250 tools/profile_config.sh r02_cfg4 || exit 1
timeout -k 10 300 tools/profile_config.sh r02_cfg5 --config 5 --shard 0/4 --steps 5 || exit 1
cp gpurun_out/r02_cfg4_pmc_traffic.json gpurun_out/r02_cfg5_pmc_traffic.json profiles/      # bench.py reads the PMC traffic from profiles/
timeout -k 10 400 python bench.py --steps 20 > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.log || exit 1
tail -3 gpurun_out/r02_bench_default.log
for s in 0/1 0/2 0/4 0/8; do t=$(echo $s | sed "s|/|of|"); timeout -k 10 300 python bench.py --config 5 --shard $s --steps 5 --no-cpu-baseline > gpurun_out/r02_bench_cfg5_shard_$t.json 2> gpurun_out/r02_bench_cfg5_shard_$t.log || exit 1; python -c "import json;print('cfg5 $t', json.load(open('gpurun_out/r02_bench_cfg5_shard_$t.json'))['ms_per_step'])"; done
for s in 0/2 0/4 0/8; do t=$(echo $s | sed "s|/|of|"); timeout -k 10 300 python bench.py --config 4 --shard $s --steps 10 --no-cpu-baseline --no-extras > gpurun_out/r02_bench_cfg4_shard_$t.json 2> gpurun_out/r02_bench_cfg4_shard_$t.log || exit 1; python -c "import json;print('cfg4 $t', json.load(open('gpurun_out/r02_bench_cfg4_shard_$t.json'))['ms_per_step'])"; done
timeout -k 10 300 python tools/cg_bench.py 30 > gpurun_out/r02_cg.log 2>&1 || exit 1
tail -2 gpurun_out/r02_cg.log
for cfg in "320 8" "480 2" "640 1" "432 2"; do set -- $cfg; timeout -k 10 200 python bench.py --config 2 --image $1 --batch $2 --steps 10 > gpurun_out/r02_bench_fft$1.json 2>gpurun_out/r02_bench_fft$1.log || exit 1; done
timeout -k 10 400 tools/profile_config.sh r02_cfg3 --config 3 --steps 5 || exit 1
timeout -k 10 300 tools/profile_config.sh r02_cfg2 --config 2 || exit 1
timeout -k 10 400 python bench.py --config 3 --steps 5 > gpurun_out/r02_bench_cfg3.json 2> gpurun_out/r02_bench_cfg3.log || exit 1
timeout -k 10 200 python bench.py --config 1 > gpurun_out/r02_bench_cfg1.json 2> gpurun_out/r02_bench_cfg1.log || exit 1
timeout -k 10 200 python bench.py --config 2 --steps 20 > gpurun_out/r02_bench_cfg2.json 2> gpurun_out/r02_bench_cfg2.log || exit 1
python -c "import json;d=json.load(open('gpurun_out/r02_bench_cfg2.json'));print('cfg2', d['ms_per_step'], d['roofline']['frac'])"
