#!/usr/bin/env python3
"""Register / LDS / scratch budget of every kernel in a `hipcc --save-temps` assembly file.
   python tools/kinfo.py file.s [substring ...]"""
import re, subprocess, sys
s = open(sys.argv[1]).read()
pats = sys.argv[2:]
names, bodies = [], []
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    names.append(m.group(1)); bodies.append(m.group(2))
dn = subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.strip().split('\n')
for n, b in zip(dn, bodies):
    if pats and not any(p in n for p in pats):
        continue
    g = lambda k: (re.search(r'\.amdhsa_' + k + r'\s+(\S+)', b) or [None, None])[1]
    print(f"{n[:90]:90s} vgpr {g('next_free_vgpr'):>4s} agpr_off {g('accum_offset'):>4s} sgpr {g('next_free_sgpr'):>4s} "
          f"lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size')}")
