#!/bin/bash
# per-kernel SQ counters: tools/pmc_kernel.sh TAG REGEX "COUNTERS..." -- bench args     (one rocprofv3 --pmc pass)
TAG=$1; RE=$2; CNT=$3; shift 3; shift
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/pmck_$TAG
mkdir -p $OUT
rocprofv3 --pmc $CNT --kernel-include-regex "$RE" -d $OUT -o p --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras "$@" > /dev/null 2> $OUT/log.txt
F=$(find $OUT -name 'p_counter_collection.csv' | head -1)
python3 - "$F" <<'P'
import csv, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r'k_\w+(<[^>]*>)?', r['Kernel_Name'])
    acc[m.group(0) if m else r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c, v in d.items():
        print("   %-28s %16.0f  (avg over %d launches)" % (c, sum(v) / len(v), len(v)))
P
