#!/bin/bash
# SQ counters (one --pmc pass per group of four) of the brick gridding kernel and of the padded y/z FFT pass, averaged per dispatch;
# the sums run over 32 shader engines (SQ_BUSY_CYCLES / 32 = the kernel's duration in clocks)
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_brick
mkdir -p $OUT
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT" "SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $OUT/g$i -o c --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > /dev/null 2> $OUT/g$i.log || { tail -5 $OUT/g$i.log; continue; }
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/g$i/**/c_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "k_grid_bricks<" in k or "k_fft_2stage<32, 16, 16, 32, false, 0, true, 1" in k or "k_fft_2stage<32, 16, 16, 16, false, 0, true, 2" in k:
        k = "bricks" if "k_grid_bricks<" in k else "fft_pad_yz(W32,half-in)" if "true, 1" in k else "fft_crop_z(W16,half-out)"
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in acc:
    print(k, {c: round(v / n[(k, c)]) for c, v in acc[k].items()})
PY
done
