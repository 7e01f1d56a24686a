#!/usr/bin/env python3
"""development aid: the sequence of global loads, vmcnt waits, barriers and branch labels of one kernel of an assembly listing
(hipcc -S --cuda-device-only), compressed — shows at a glance whether a loop waits for ALL outstanding loads before a barrier.
usage: isa_waits.py file.s <mangled-name-substring> [...]"""
import re
import sys

s = open(sys.argv[1]).read()
for tag in sys.argv[2:]:
    for m in re.finditer(r'^(_Z\S*' + re.escape(tag) + r'\S*):', s, re.M):
        i, j = m.start(), s.find('s_endpgm', m.start())
        lines = [l.strip() for l in s[i:j].splitlines()]
        out = []
        for l in lines:
            k = None
            if l.startswith('s_waitcnt') and 'vmcnt' in l:
                k = 'W' + re.search(r'vmcnt\((\d+)\)', l).group(1)
            elif l.startswith('s_barrier'):
                k = 'BAR'
            elif l.startswith('buffer_load') or l.startswith('global_load'):
                k = 'L' + l.split()[0].split('_')[-1][-2:]
            elif l.startswith('.LBB'):
                k = '[' + l.rstrip(':').split('_')[-1] + ']'
            elif l.startswith('s_cbranch') or l.startswith('s_branch'):
                k = 'br'
            elif l.startswith('v_mfma'):
                k = 'M'
            if k is None:
                continue
            if out and out[-1][0] == k:
                out[-1][1] += 1
            else:
                out.append([k, 1])
        print(m.group(1)[:110], len(lines))
        print(' '.join(f"{k}x{n}" if n > 1 else k for k, n in out))
        print()
