#!/bin/bash
# round 5: the adjoint stress harness on the final kernels -- the round-3 tree (7 coils as 2 + 2 + 2 + 1), mixed widths (6 as 4 + 2),
# a padded chunk (7 as 4 + 4), the 8-coil round format; fresh processes, idle pauses, rebuilds
set -o pipefail
mkdir -p gpurun_out
{
timeout -k 10 400 python tests/stress_adjoint.py --reps 120 --procs 3 --idle 3 --rebuild 40 --coils 7 --chunk 2 || exit 1
timeout -k 10 300 python tests/stress_adjoint.py --reps 100 --procs 2 --idle 2 --rebuild 50 --coils 6 --chunk 4 || exit 1
timeout -k 10 300 python tests/stress_adjoint.py --reps 100 --procs 2 --idle 2 --rebuild 50 --coils 7 --chunk 4 || exit 1
timeout -k 10 300 python tests/stress_adjoint.py --reps 100 --procs 2 --rebuild 50 --coils 8 --chunk 8 || exit 1
} > gpurun_out/r05_stress_adjoint.log 2>&1
rc=$?
grep -c -i "deviation" gpurun_out/r05_stress_adjoint.log
tail -12 gpurun_out/r05_stress_adjoint.log
exit $rc
