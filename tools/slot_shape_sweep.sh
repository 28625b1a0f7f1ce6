#!/bin/bash
# adjoint gridding of 1-, 2- and 4-coil ranks: slot-format scatter for several brick shapes / task sizes against the routes it replaces
for sh in "0/8" "0/4" "0/2"; do
  for tun in "slots=()" "slots=(1,2,4) bricks=(8,) slot_shape=(2,2,256,128)" "slots=(1,2,4) bricks=(8,) slot_shape=(4,4,256,64)" "slots=(1,2,4) bricks=(8,) slot_shape=(2,4,256,96)" "slots=(1,2,4) bricks=(8,) slot_shape=(4,4,512,128)"; do
    timeout -k 10 300 python tools/run_with_tuning.py $tun -- --shard $sh --steps 10 --no-extras --no-cpu-baseline > gpurun_out/ss.json 2> gpurun_out/ss.log || { echo "FAILED $sh $tun"; tail -4 gpurun_out/ss.log; continue; }
    python - "$sh" "$tun" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/ss.json').read().strip().splitlines()[-1])
print("shard %s %-60s %.3f ms" % (sys.argv[1], sys.argv[2], d['ms_per_step']), {k: round(v['avg_ms'], 3) for k, v in d['kernels'].items() if 'conj' in k or 'rows_' in k or 'pack' in k}, flush=True)
PY
  done
done
