# Exchange rounds of the A x B kernels: lib80.so / lib160.so = the library built after `IG_AB_LDS_BUDGET=81920|163840 python tools/gen_ab_list.py`
# (copied to indigo_amd/lib/lab/ for the run; output: profiles/r04_ab_rounds_sweep.txt).  The shipped list is the 160 KB one.
for v in lib80 lib160; do
  echo "== $v"
  export INDIGO_HIP_LIB=$PWD/indigo_amd/lib/lab/$v.so
  for cfg in "320 8" "384 4" "432 2" "480 2" "512 2" "576 1" "600 1" "640 1"; do set -- $cfg; python bench.py --config 2 --image $1 --batch $2 --steps 10 --no-cpu-baseline > gpurun_out/fft_$v$1.json 2>gpurun_out/fft_$v$1.log; python -c "import json;d=json.load(open('gpurun_out/fft_$v$1.json'));print('fft $1', round(d['ms_per_step'],3), round(d['roofline']['frac'],3), d['parity_rel_err'], d['config']['plan'][:60])"; done
  python tools/lab/chirp_fft.py 2>&1 | tail -2
  python - <<'PY'
import numpy as np, time
from indigo_amd.backends import get_backend
from indigo_amd.util import rand64c
B = get_backend("hip")
for n in (1000, 1024, 960, 768):
    x = rand64c(n, 64, 64, seed=1)      # axis 0 contiguous; and strided axes
    for shape in ((n, 64, 64), (64, n, 64), (64, 64, n)):
        x = rand64c(*shape, seed=2)
        xd = B.copy_array(x.reshape(-1, 1)).reshape(shape + (1,)) if False else None
    # strided + contiguous in one 3-D transform of (n, 32, n)? too big; use (n, n, 8)
    shape = (n, n, 8)
    x = rand64c(*shape, seed=3)
    x_d = B.copy_array(np.asfortranarray(x.reshape(shape + (1,), order='F')))
    y_d = B.zero_array(x_d.shape, x_d.dtype)
    B.fftn(y_d, x_d); B.barrier()
    t0 = time.perf_counter()
    for _ in range(5): B.fftn(y_d, x_d)
    B.barrier()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    ref = np.fft.fftn(x.astype(np.complex128), axes=(0, 1, 2))
    err = np.linalg.norm(y_d.to_host().reshape(shape, order='F') - ref) / np.linalg.norm(ref)
    print("fftn %s: %.3f ms, err %.2e" % (shape, ms, err))
PY
done
