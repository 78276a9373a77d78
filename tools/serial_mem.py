#!/usr/bin/env python3
"""Finds kernels whose code alternates global load -> s_waitcnt vmcnt(0) -> global store several times in a row: loads the
compiler had to keep behind the previous store (possible aliasing) are issued one memory round trip at a time.
   python tools/serial_mem.py file.s [min_alternations=3]"""
import re, subprocess, sys
lines = open(sys.argv[1]).read().split('\n')
need = int(sys.argv[2]) if len(sys.argv) > 2 else 3
i = 0
while i < len(lines):
    m = re.match(r'^(_Z\w+):\s', lines[i])
    if not m:
        i += 1
        continue
    j = i + 1
    seq = []
    while j < len(lines) and not lines[j].strip().startswith('.amdhsa_kernel') and not re.match(r'^_Z\w+:\s', lines[j]):
        s = lines[j].strip()
        if s.startswith(('global_load', 'buffer_load')): seq.append('L')
        elif s.startswith(('global_store', 'buffer_store')): seq.append('S')
        elif s.startswith('s_waitcnt') and 'vmcnt(0)' in s: seq.append('W')
        j += 1
    txt = re.sub(r'(.)\1+', r'\1', ''.join(seq))          # runs collapsed
    best = max((len(x.group(0)) // 3 for x in re.finditer(r'(?:LWS)+', txt)), default=0)
    if best >= need:
        name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
        print(f"{best:3d} x (load, wait, store)  {name[:140]}")
    i = j
