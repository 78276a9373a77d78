#!/usr/bin/env python3
"""Instruction mix of kernels in a `hipcc --save-temps` assembly file (static counts, whole kernel body).
   python tools/icount.py file.s substring [substring ...]"""
import collections, re, subprocess, sys
lines = open(sys.argv[1]).read().split('\n')
pats = sys.argv[2:]
i = 0
while i < len(lines):
    m = re.match(r'^(_Z\w+):\s', lines[i])
    if not m:
        i += 1
        continue
    name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
    j = i + 1
    while j < len(lines) and not lines[j].strip().startswith('.amdhsa_kernel') and not re.match(r'^_Z\w+:\s', lines[j]):
        j += 1
    if any(p in name for p in pats):
        c = collections.Counter()
        other = collections.Counter()
        for ln in lines[i + 1:j]:
            mm = re.match(r'\s+([a-z][a-z_0-9]+)\s', ln)
            if not mm:
                continue
            op = mm.group(1)
            if op.startswith('v_'):
                c['VALU'] += 1
                if re.match(r'v_(pk_)?(add|sub|mul|fma|fmac|mad)_f(32|64)', op): c['valu_fp'] += 1
                elif 'mov' in op or 'cndmask' in op or 'accvgpr' in op: c['valu_mov'] += 1
                else: other[op] += 1
            elif op.startswith('ds_'): c['DS'] += 1; other[op] += 1
            elif op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')): c['VMEM'] += 1
            elif op.startswith('s_waitcnt'): c['waitcnt'] += 1
            elif op.startswith('s_barrier'): c['barrier'] += 1
            elif op.startswith('s_'): c['SALU'] += 1
        print(name[:120])
        print('   ', dict(c))
        print('   ', other.most_common(12))
    i = j
