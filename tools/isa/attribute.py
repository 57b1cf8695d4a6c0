#!/usr/bin/env python3
"""Developer tool: static instruction mix of one kernel, attributed to source lines.
Build the ISA with line tables (-gline-tables-only -S), then
    attribute.py file.s kernel_name_substring [--blocks] [--by-line FILE]
prints per source file / coarse line range the number of VALU / SALU / VMEM / LDS / MFMA / scratch instructions, and with --blocks
the same per basic block (label) in program order, so that the hot loops can be read off."""
import re
import sys
from collections import Counter, defaultdict


def klass(op):
    if op.startswith('v_mfma'):
        return 'mfma'
    if op.startswith('scratch_'):
        return 'scratch'
    if op.startswith(('global_', 'flat_', 'buffer_')):
        return 'vmem'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('s_waitcnt'):
        return 'wait'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


def main():
    f, name = sys.argv[1], sys.argv[2]
    want_blocks = '--blocks' in sys.argv
    by_line = sys.argv[sys.argv.index('--by-line') + 1] if '--by-line' in sys.argv else None
    lines = open(f).read().split('\n')
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
        if m:
            files[int(m.group(1))] = m.group(2)
    start = [i for i, l in enumerate(lines) if name in l and l.startswith('_Z') and ':' in l][0]
    fe = [i for i, l in enumerate(lines) if i > start and l.startswith('.Lfunc_end')][0]
    cur = (0, 0)
    label = 'entry'
    per_src = defaultdict(Counter)
    per_blk = []
    blk = None
    for l in lines[start:fe]:
        s = l.strip()
        m = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
        if m:
            cur = (int(m.group(1)), int(m.group(2)))
            continue
        if re.match(r'^\.LBB[0-9_]+:', l):
            label = l.split(':')[0]
            blk = None
            continue
        if l and not l[0].isspace() and ':' in l:
            continue
        if not l.startswith('\t') or s.startswith(('.', ';')) or not s:
            continue
        op = s.split()[0]
        k = klass(op)
        fn = files.get(cur[0], '?')
        per_src[(fn, cur[1])][k] += 1
        if blk is None:
            blk = [label, Counter(), Counter()]
            per_blk.append(blk)
        blk[1][k] += 1
        blk[2][(fn, cur[1])] += 1
    keys = ['valu', 'salu', 'vmem', 'lds', 'mfma', 'scratch', 'wait']
    # per file, coarse 25-line buckets
    agg = defaultdict(Counter)
    for (fn, ln), c in per_src.items():
        agg[(fn, ln // 25 * 25)].update(c)
    print('%-24s %6s ' % ('file', 'line') + ' '.join('%7s' % k for k in keys))
    for (fn, ln) in sorted(agg):
        c = agg[(fn, ln)]
        print('%-24s %6d ' % (fn, ln) + ' '.join('%7d' % c[k] for k in keys))
    tot = Counter()
    for c in per_src.values():
        tot.update(c)
    print('%-24s %6s ' % ('TOTAL', '') + ' '.join('%7d' % tot[k] for k in keys))
    if by_line:
        print()
        for (fn, ln) in sorted(per_src):
            if fn == by_line:
                c = per_src[(fn, ln)]
                print('%-24s %6d ' % (fn, ln) + ' '.join('%7d' % c[k] for k in keys))
    if want_blocks:
        print()
        for lab, c, src in per_blk:
            top = ', '.join('%s:%d(%d)' % (a[0].replace('.cuh', ''), a[1], n) for a, n in src.most_common(3))
            print('%-14s ' % lab + ' '.join('%5d' % c[k] for k in keys) + '  ' + top)


if __name__ == '__main__':
    main()
