#!/bin/bash
# everything under profiles/ that the current round cites, re-measured in two GPU calls: tools/refresh_profiles.sh r05 A|B
# (copy gpurun_out/${R}_* into profiles/ afterwards; the PMC summaries record the hash of the kernel sources they belong to)
# Three parts (a gpurun call is limited to 20 minutes): A = rocprofv3 stats + PMC of configs 4, 5, 3, 2; D = the default line (after A's
# summaries are in profiles/); B = per-rank shares, CG, the recipe's trees, osf 1.25; C = coil counts, the reference driver's default grid,
# plain transforms, configs 1-3 alone, the self-launch rehearsals.
R=${1:-r06}
PART=${2:-AB}          # A, D, B, C as below; W = the width-3 subset of A and C
mkdir -p gpurun_out
if [[ $PART == *A* ]]; then
timeout -k 10 300 tools/profile_config.sh ${R}_cfg4 || exit 1
timeout -k 10 300 tools/profile_config.sh ${R}_cfg5 --config 5 --shard 0/4 --steps 5 || exit 1
timeout -k 10 400 tools/profile_config.sh ${R}_cfg3 --config 3 --steps 5 || exit 1
timeout -k 10 300 tools/profile_config.sh ${R}_cfg2 --config 2 || exit 1
# (round 6) the reference's default kernel half-width 3 and the densely sampled trajectory: kernel stats + PMC traffic of the gridding kernels there
timeout -k 10 300 tools/profile_config.sh ${R}_cfg4_w3 --width 3 --steps 5 || exit 1
timeout -k 10 400 tools/profile_config.sh ${R}_cfg4_dense --spokes-scale 8 --steps 5 || exit 1
cp gpurun_out/${R}_cfg4_pmc_traffic.json gpurun_out/${R}_cfg5_pmc_traffic.json gpurun_out/${R}_cfg3_pmc_traffic.json gpurun_out/${R}_cfg2_pmc_traffic.json profiles/      # bench.py reads the PMC traffic from profiles/
fi
if [[ $PART == *W* ]]; then
# (round 6) only what depends on the share format: the width-3 profile and bench lines
timeout -k 10 300 tools/profile_config.sh ${R}_cfg4_w3 --width 3 --steps 5 || exit 1
timeout -k 10 400 python bench.py --image 480,208,308 --osf 640/480 --width 3 --steps 10 --no-extras --no-cpu-baseline --parity > gpurun_out/${R}_bench_width3_default_grid.json 2> gpurun_out/${R}_bench_width3_default_grid.log || exit 1
python -c "import json;d=json.load(open('gpurun_out/${R}_bench_width3_default_grid.json'));print('640x277x410 width 3', d['ms_per_step'], d['parity_rel_err'])"
timeout -k 10 300 python bench.py --width 3 --steps 10 --no-extras --no-cpu-baseline --parity > gpurun_out/${R}_bench_width3.json 2> gpurun_out/${R}_bench_width3.log || exit 1
python -c "import json;d=json.load(open('gpurun_out/${R}_bench_width3.json'));print('headline width 3', d['ms_per_step'], d['parity_rel_err'])"
timeout -k 10 300 python bench.py --width 3 --osf 1.25 --steps 10 --no-extras --no-cpu-baseline --parity > gpurun_out/${R}_bench_width3_osf125.json 2> gpurun_out/${R}_bench_width3_osf125.log || exit 1
python -c "import json;d=json.load(open('gpurun_out/${R}_bench_width3_osf125.json'));print('osf 1.25 width 3', d['ms_per_step'], d['parity_rel_err'])"
fi
if [[ $PART == *D* ]]; then
# the default line as the driver runs it (headline + config 5 + configs 2 and 3 + the dense-trajectory extra), after part A's PMC
# summaries have been copied into profiles/
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${R}_bench_default.json 2> gpurun_out/${R}_bench_default.log || exit 1
tail -3 gpurun_out/${R}_bench_default.log
python -c "import json;d=json.load(open('gpurun_out/${R}_bench_default.json'));print('default', d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic_stale'], 'dense', d['dense_trajectory'].get('ms_per_step'), 'width3', d['width3'].get('ms_per_step'), 'cfg5', d['config5'].get('ms_per_step'), 'cfg3', d['config3'].get('ms_per_step'), 'cfg2', d['config2'].get('ms_per_step'))"
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-dense > gpurun_out/${R}_bench_default_run2.json 2> gpurun_out/${R}_bench_default_run2.log || exit 1
python -c "import json;d=json.load(open('gpurun_out/${R}_bench_default_run2.json'));print('default run 2', d['ms_per_step'], d['roofline']['frac'])"
fi
if [[ $PART == *B* ]]; then
for s in 0/1 0/2 0/4 0/8; do t=$(echo $s | sed "s|/|of|"); timeout -k 10 300 python bench.py --config 5 --shard $s --steps 5 --no-cpu-baseline > gpurun_out/${R}_bench_cfg5_shard_$t.json 2> gpurun_out/${R}_bench_cfg5_shard_$t.log || exit 1; python -c "import json;print('cfg5 $t', json.load(open('gpurun_out/${R}_bench_cfg5_shard_$t.json'))['ms_per_step'])"; done
for s in 0/2 0/4 0/8; do t=$(echo $s | sed "s|/|of|"); timeout -k 10 300 python bench.py --config 4 --shard $s --steps 10 --no-cpu-baseline --no-extras > gpurun_out/${R}_bench_cfg4_shard_$t.json 2> gpurun_out/${R}_bench_cfg4_shard_$t.log || exit 1; python -c "import json;print('cfg4 $t', json.load(open('gpurun_out/${R}_bench_cfg4_shard_$t.json'))['ms_per_step'])"; done
CG_PROFILE=1 timeout -k 10 300 python tools/cg_bench.py 30 > gpurun_out/${R}_cg.log 2>&1 || exit 1
tail -2 gpurun_out/${R}_cg.log
for t in o3 recipe; do timeout -k 10 300 python bench.py --tree $t --steps 5 --no-extras --no-cpu-baseline > gpurun_out/${R}_bench_tree_$t.json 2> gpurun_out/${R}_bench_tree_$t.log || exit 1; python -c "import json;print('tree $t', json.load(open('gpurun_out/${R}_bench_tree_$t.json'))['ms_per_step'])"; done
for t in zpadfft o3 recipe; do timeout -k 10 300 python bench.py --osf 1.25 --tree $t --steps 10 --no-extras --no-cpu-baseline > gpurun_out/${R}_bench_osf125_$t.json 2> gpurun_out/${R}_bench_osf125_$t.log || exit 1; python -c "import json;print('osf 1.25 tree $t', json.load(open('gpurun_out/${R}_bench_osf125_$t.json'))['ms_per_step'])"; done
fi
if [[ $PART == *C* ]]; then
for c in 12 6 3 7; do timeout -k 10 300 python bench.py --coils $c --steps 10 --no-extras --no-cpu-baseline > gpurun_out/${R}_bench_coils$c.json 2> gpurun_out/${R}_bench_coils$c.log || exit 1; python -c "import json;d=json.load(open('gpurun_out/${R}_bench_coils$c.json'));print('coils $c', d['ms_per_step'], d['config']['coil_chunk_widths'])"; done
timeout -k 10 400 python bench.py --image 480,208,308 --osf 640/480 --steps 10 --no-extras > gpurun_out/${R}_bench_default_grid_640x277x410.json 2> gpurun_out/${R}_bench_default_grid_640x277x410.log || exit 1
python -c "import json;d=json.load(open('gpurun_out/${R}_bench_default_grid_640x277x410.json'));print('640x277x410', d['ms_per_step'], d['parity_rel_err'])"
# (round 6) the reference driver's OWN defaults: its scan size, its oversampling AND its kernel half-width 3 (Backend.NUFFT, backend.py:403)
timeout -k 10 400 python bench.py --image 480,208,308 --osf 640/480 --width 3 --steps 10 --no-extras --no-cpu-baseline --parity > gpurun_out/${R}_bench_width3_default_grid.json 2> gpurun_out/${R}_bench_width3_default_grid.log || exit 1
python -c "import json;d=json.load(open('gpurun_out/${R}_bench_width3_default_grid.json'));print('640x277x410 width 3', d['ms_per_step'], d['parity_rel_err'])"
timeout -k 10 300 python bench.py --width 3 --steps 10 --no-extras --no-cpu-baseline --parity > gpurun_out/${R}_bench_width3.json 2> gpurun_out/${R}_bench_width3.log || exit 1
python -c "import json;d=json.load(open('gpurun_out/${R}_bench_width3.json'));print('headline width 3', d['ms_per_step'], d['parity_rel_err'])"
timeout -k 10 300 python bench.py --width 3 --osf 1.25 --steps 10 --no-extras --no-cpu-baseline --parity > gpurun_out/${R}_bench_width3_osf125.json 2> gpurun_out/${R}_bench_width3_osf125.log || exit 1
python -c "import json;d=json.load(open('gpurun_out/${R}_bench_width3_osf125.json'));print('osf 1.25 width 3', d['ms_per_step'], d['parity_rel_err'])"
for cfg in "320 8" "480 2" "640 1" "432 2" "512 8"; do set -- $cfg; timeout -k 10 200 python bench.py --config 2 --image $1 --batch $2 --steps 10 > gpurun_out/${R}_bench_fft$1.json 2>gpurun_out/${R}_bench_fft$1.log || exit 1; python -c "import json;d=json.load(open('gpurun_out/${R}_bench_fft$1.json'));print('fft $1', d['ms_per_step'], d['roofline']['frac'])"; done
timeout -k 10 400 python bench.py --config 3 --steps 5 > gpurun_out/${R}_bench_cfg3.json 2> gpurun_out/${R}_bench_cfg3.log || exit 1
timeout -k 10 200 python bench.py --config 1 > gpurun_out/${R}_bench_cfg1.json 2> gpurun_out/${R}_bench_cfg1.log || exit 1
timeout -k 10 200 python bench.py --config 2 --steps 20 > gpurun_out/${R}_bench_cfg2.json 2> gpurun_out/${R}_bench_cfg2.log || exit 1
python -c "import json;d=json.load(open('gpurun_out/${R}_bench_cfg2.json'));print('cfg2', d['ms_per_step'], d['roofline']['frac'])"
python -c "import json;d=json.load(open('gpurun_out/${R}_bench_cfg3.json'));print('cfg3 fwd', d['ms_per_step'], d['roofline']['frac'], 'adj', d['config']['adjoint_ms'], d['config']['adjoint_frac_of_peak_reference_model'])"
# the self-launching multi-rank path, rehearsed over gloo on this one GPU (all ranks on GPU 0; numbers are NOT scaling results)
for n in 2 4; do INDIGO_BENCH_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus $n --steps 3 --warmup 1 --no-config5 > gpurun_out/${R}_bench_selflaunch_${n}rank_gloo.json 2> gpurun_out/${R}_bench_selflaunch_${n}rank_gloo.log || exit 1; python -c "import json;d=json.load(open('gpurun_out/${R}_bench_selflaunch_${n}rank_gloo.json'));print('self-launch gloo', d['n_gpus'], d['ms_per_step'])"; done
# ... and of the library's own all-reduce route (no RCCL, no gloo: HIP IPC windows), all ranks on GPU 0
for n in 2 4; do INDIGO_BENCH_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus $n --comm direct --steps 3 --warmup 1 --no-config5 --no-cpu-baseline > gpurun_out/${R}_bench_selflaunch_${n}rank_direct_one_gpu.json 2> gpurun_out/${R}_bench_selflaunch_${n}rank_direct_one_gpu.log || exit 1; python -c "import json;d=json.load(open('gpurun_out/${R}_bench_selflaunch_${n}rank_direct_one_gpu.json'));print('self-launch direct', d['n_gpus'], d['ms_per_step'], d['comm'])"; done
timeout -k 10 200 python tools/lab/chirp_fft.py > gpurun_out/${R}_chirp_fft.log 2>&1 || exit 1
tail -1 gpurun_out/${R}_chirp_fft.log
fi
