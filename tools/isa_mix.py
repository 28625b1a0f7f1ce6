#!/usr/bin/env python3
"""Instruction mix of device kernels in a hipcc --save-temps assembly file: tools/isa_mix.py file.s [substring of the demangled name ...]"""
import collections
import re
import subprocess
import sys


def main():
    lines = open(sys.argv[1]).read().splitlines()
    want = sys.argv[2:]
    i = 0
    while i < len(lines):
        m = re.match(r'^(_Z\w+):\s*; @', lines[i])
        if not m:
            i += 1
            continue
        name = m.group(1)
        j = i + 1
        while j < len(lines) and not lines[j].startswith('.Lfunc_end'):
            j += 1
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r'\(anonymous namespace\)::', '', dem).split('(')[0]
        if not want or any(w in dem for w in want):
            c = collections.Counter()
            for line in lines[i + 1:j]:
                mm = re.match(r'\s+([a-z_0-9]+)(\s|$)', line)
                if not mm:
                    continue
                op = mm.group(1)
                cls = ('vmem_ld' if op.startswith(('buffer_load', 'global_load')) else 'vmem_st' if op.startswith(('buffer_store', 'global_store'))
                       else 'vmem_atomic' if 'atomic' in op else 'lds' if op.startswith('ds_') else 'smem' if op.startswith('s_load') or op.startswith('s_buffer_load')
                       else 'salu' if op.startswith('s_') else 'valu_pk' if op.startswith('v_pk') else 'valu')
                c[cls] += 1
                c['total'] += 1
                if op in ('s_waitcnt', 's_barrier', 's_nop', 'v_mov_b32', 'v_bfe_i32', 'v_or_b32', 'v_readlane_b32'):
                    c[op] += 1
            print(dem[:90], dict(sorted(c.items())))
        i = j


if __name__ == "__main__":
    main()
