#!/bin/bash
# Per-kernel register / spill / occupancy table for one HIP source: tools/kernel_regs.sh indigo_amd/csrc/ig_fft.hip [filter]
R=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-slp-vectorize $EXTRA -I$R/include -I$R/indigo_amd/csrc \
  -c "$1" -o /tmp/_regs.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys, re, subprocess
rows = []; cur = None
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur); continue
    for key in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]", "VGPR Spill"):
        m = re.search(re.escape(key) + r": (\d+)", line)
        if m and cur is not None: cur[key.split()[0] + ("Spill" if "Spill" in key and key.startswith("VGPR ") else "")] = m.group(1)
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for r in rows:
    if flt in r["name"]:
        n = re.sub(r"\(anonymous namespace\)::", "", r["name"]).split("(")[0]
        print("%-62s vgpr %4s agpr %3s scratch %4s occ %s" % (n[:62], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize"), r.get("Occupancy")))
' "$2"
