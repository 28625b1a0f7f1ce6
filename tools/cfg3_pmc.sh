#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the config-3 kernels for one setting of INDIGO_HIP_WIDE_TASKS (first argument) [+ other env as VAR=VALUE arguments]
export TMPDIR=/tmp
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o f --output-format csv -- python3 bench.py --config 3 --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2> $OUT/fetch.log || { tail -5 $OUT/fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o w --output-format csv -- python3 bench.py --config 3 --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2> $OUT/write.log || { tail -5 $OUT/write.log; exit 1; }
F=$(find $OUT/fetch -name 'f_counter_collection.csv' | head -1)
W=$(find $OUT/write -name 'w_counter_collection.csv' | head -1)
echo "== $TAG $@"
python3 tools/pmc_summary.py "$F" "$W" gpurun_out/pmc_${TAG}.json
