#!/bin/bash
# rocprofv3 evidence for one bench configuration, written under gpurun_out/ (copy what is to be judged into profiles/):
#   tools/profile_config.sh TAG [bench.py arguments...]        e.g.  tools/profile_config.sh r02_cfg2 --config 2
# Three separate runs of the same command (the guide's HBM recipe: kernel trace + stats alone, then one --pmc pass per
# counter; FETCH_SIZE and WRITE_SIZE do not fit one pass), then tools/pmc_summary.py applies the gfx950 corrections.
# The program itself follows `--` (no env / shell wrappers: the profiler's library has already initialised the GPU).
set -e
TAG=$1; shift
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras "$@" > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o f --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 "$@" > /dev/null 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o w --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 "$@" > /dev/null 2> $OUT/write.log
S=$(find $OUT/stats -name 's_kernel_stats.csv' | head -1)
F=$(find $OUT/fetch -name 'f_counter_collection.csv' | head -1)
W=$(find $OUT/write -name 'w_counter_collection.csv' | head -1)
cp "$S" gpurun_out/${TAG}_kernel_stats.csv
python3 tools/pmc_summary.py "$F" "$W" gpurun_out/${TAG}_pmc_traffic.json > gpurun_out/${TAG}_pmc_traffic.txt
cp $OUT/bench_under_rocprof.json gpurun_out/${TAG}_bench_under_rocprof.json
head -12 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-160
