"""Build-time invariants of hand-scheduled kernels, checked on the compiler's assembly (no GPU needed).

k_bricks_wide64r (indigo_amd/csrc/ig_spmm.hip) keeps its brick image, its panel rows and its entries in v72..v255 behind the
compiler's back (amdgpu_num_vgpr(72), assembly blocks with literal register numbers) and addresses the image through the VGPR
index mode with M0 written inside those blocks.  That is only sound while the compiler's own instructions stay below v72 and
never touch M0, and while the kernel is given all 256 registers and no scratch.  k_csrmm_runs64r does the same with v56..v127."""
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
    def check(pattern, cap, name, min_lines):
        """the compiler's own instructions of the kernel whose mangled name matches `pattern` stay below v<cap> and never name M0"""
        start = next(i for i, line in enumerate(text) if re.match(pattern, line))
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
                assert hi < cap, "compiler code touches v%d in %s: %s" % (hi, name, line.strip())
            assert not re.search(r"\bm0\b", line), "compiler code touches M0 in %s: %s" % (name, line.strip())
        assert compiler_lines > min_lines, (name, compiler_lines)

    # the adjoint's register image: both brick shapes, complex and real-weight entries
    for nt in (2, 4):
        for rw in (0, 1):
            check(r"^_ZN\S*k_bricks_wide64rILi%dELb%dE\S*:" % (nt, rw), 72, "k_bricks_wide64r<%d, %d>" % (nt, rw), 500)
    # the forward's run kernel (results and the ring of panel rows in v56..v127): beta == 0 / != 0, real / complex weights
    for bm in (0, 1):
        for rw in (0, 1):
            check(r"^_ZN\S*k_csrmm_runs64rILi%dELb%dE\S*:" % (bm, rw), 56, "k_csrmm_runs64r<%d, %d>" % (bm, rw), 200)
    # the kernel descriptors: all the registers the kernels claim (256 / 128: the images are real), no scratch
    meta = "\n".join(text)
    blocks = re.findall(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", meta, flags=re.S)
    seen = {"k_bricks_wide64rILi": 0, "k_csrmm_runs64rILi": 0}
    for blk in blocks:
        for key, regs in (("k_bricks_wide64rILi", 256), ("k_csrmm_runs64rILi", 128)):
            if key not in blk:
                continue
            seen[key] += 1
            assert re.search(r"\.vgpr_count:\s+%d\b" % regs, blk) and re.search(r"\.agpr_count:\s+0\b", blk), blk[:200]
            assert re.search(r"\.private_segment_fixed_size:\s+0\b", blk) and re.search(r"\.vgpr_spill_count:\s+0\b", blk), blk[:200]
    assert seen == {"k_bricks_wide64rILi": 4, "k_csrmm_runs64rILi": 4}, seen
