#!/bin/bash
# SQ counters of the gridding kernels (one --pmc pass per group), averaged per dispatch.  usage: r06_sq.sh <tag> <bench args...>
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/sq_$TAG
mkdir -p $OUT
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT" "SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $OUT/g$i -o c --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 "$@" > /dev/null 2> $OUT/g$i.log || { tail -3 $OUT/g$i.log; continue; }
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/g$i/**/c_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    for tag in ("k_grid_bricks<", "k_grid_scatter_mfma<", "k_grid_scatter_sep<", "k_grid_gather_sep<", "k_grid_gather_mfma<", "k_csrmm_gather_v<"):
        if tag in k:
            acc[tag][r["Counter_Name"]] += float(r["Counter_Value"]); n[(tag, r["Counter_Name"])] += 1
for k in acc:
    print(k, {c: round(v / n[(k, c)]) for c, v in acc[k].items()})
PY
done
