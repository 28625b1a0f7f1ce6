mkdir -p gpurun_out
for dbg in 3 4 8 16 27; do
  INDIGO_HIP_BRICK_DEBUG=$dbg timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline --no-config5 > gpurun_out/r02i_bench_dbg$dbg.json 2> gpurun_out/r02i_bench.log || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/r02i_bench_dbg$dbg.json"))
print("dbg$dbg", d["ms_per_step"], d["kernels"]["csrmm_bricks_conj"]["avg_ms"])
PY
done
