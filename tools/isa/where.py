#!/usr/bin/env python3
"""Developer tool: where do scratch (spill) instructions sit in a kernel's ISA relative to its MFMA block and its loops?
usage: where.py file.s kernel_name_substring"""
import sys, re
from collections import Counter
f, name = sys.argv[1], sys.argv[2]
lines = open(f).read().split('\n')
start = [i for i, l in enumerate(lines) if name in l and l.split(':')[0].startswith('_Z') and ':' in l and not l.startswith('\t')][0]
end = [i for i, l in enumerate(lines) if i > start and 's_endpgm' in l]
# the kernel ends at the .Lfunc_end label
fe = [i for i, l in enumerate(lines) if i > start and l.startswith('.Lfunc_end')][0]
body = lines[start:fe]
ins = [l for l in body if l.startswith('\t') and not l.strip().startswith('.') and not l.strip().startswith(';')]
print('instructions', len(ins))
sc = [i for i, l in enumerate(body) if 'scratch_' in l]
mf = [i for i, l in enumerate(body) if 'v_mfma' in l]
print('scratch ops', len(sc), 'mfma', len(mf), 'mfma line range', mf[0] if mf else None, mf[-1] if mf else None, 'body lines', len(body))
c = Counter(i // 250 for i in sc)
for k in sorted(c):
    lo = k * 250
    tag = 'MFMA region' if mf and mf[0] <= lo + 125 <= mf[-1] else ''
    print(f'{lo:6d} {c[k]:3d} {tag}')
for key in ('global_load', 'global_store', 'flat_load', 'flat_store', 'ds_read', 'ds_write', 's_waitcnt vmcnt', 'v_mfma', 's_swappc', 'v_readlane', 'v_writelane'):
    print(key, sum(1 for l in ins if key in l))
