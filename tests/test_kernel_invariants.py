"""Build-time invariants of hand-scheduled kernels, checked on the compiler's assembly (no GPU needed).

k_bricks_wide64r (indigo_amd/csrc/ig_spmm.hip) keeps its brick image, its panel rows and its entries in v72..v255 behind the
compiler's back (amdgpu_num_vgpr(72), assembly blocks with literal register numbers) and addresses the image through the VGPR
index mode with M0 written inside those blocks.  That is only sound while the compiler's own instructions stay below v72 and
never touch M0, and while the kernel is given all 256 registers and no scratch."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_register_image_kernel_owns_its_registers(tmp_path):
    out = tmp_path / "spmm.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "indigo_amd", "csrc"), "--cuda-device-only", "-S", os.path.join(ROOT, "indigo_amd", "csrc", "ig_spmm.hip"),
           "-o", str(out)]
    subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    text = out.read_text().splitlines()
    kernels = 0
    for nt in (2, 4):
        start = next(i for i, line in enumerate(text) if re.match(r"^_ZN\S*k_bricks_wide64rILi%dE\S*:" % nt, line))
        end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
        in_asm, compiler_lines = False, 0
        for line in text[start:end]:
            if "#ASMSTART" in line:
                in_asm = True
                continue
            if "#ASMEND" in line:
                in_asm = False
                continue
            if in_asm or line.strip().startswith(";"):
                continue
            compiler_lines += 1
            for m in re.finditer(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]", line):
                hi = int(m.group(1)) if m.group(1) else int(m.group(3))
                assert hi < 72, "compiler code touches v%d in k_bricks_wide64r<%d>: %s" % (hi, nt, line.strip())
            assert not re.search(r"\bm0\b", line), "compiler code touches M0 in k_bricks_wide64r<%d>: %s" % (nt, line.strip())
        assert compiler_lines > 500
        kernels += 1
    assert kernels == 2
    # the kernel descriptors: 256 VGPRs (the image is real), no scratch
    meta = "\n".join(text)
    blocks = re.findall(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", meta, flags=re.S)
    seen = 0
    for blk in blocks:
        if "k_bricks_wide64rILi" not in blk:
            continue
        seen += 1
        assert re.search(r"\.vgpr_count:\s+256\b", blk) and re.search(r"\.agpr_count:\s+0\b", blk)
        assert re.search(r"\.private_segment_fixed_size:\s+0\b", blk) and re.search(r"\.vgpr_spill_count:\s+0\b", blk)
    assert seen == 2
