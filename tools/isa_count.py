#!/usr/bin/env python3
"""Static instruction histogram of selected tile kernels in a hipcc -S dump."""
import re, sys, collections
s = open(sys.argv[1]).read()
pats = sys.argv[2:] or ['Li14ELi3ELi11ELi4ELi0ELb1', 'Li13ELi0ELi13ELi4ELi0ELb0']
for f in re.split(r'\n(?=_ZN10sventt_hip11tile_kernel[^\n]*:\s)', s):
    name = f.split(':')[0]
    if not f.startswith('_ZN') or not any(p in name for p in pats):
        continue
    body = f.split('s_endpgm')[0]
    ins = [l.split()[0] for l in body.split('\n') if re.match(r'\s+(v_|s_|ds_|global_|buffer_)', l)]
    c = collections.Counter(ins)
    tot = lambda p: sum(n for i, n in c.items() if i.startswith(p))
    m = re.search(r'TileNTTILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)', name)
    vg = re.search(r'\.vgpr_count:\s+(\d+)', f)
    print('== TileNTT<%s>' % ','.join(m.groups()), 'VALU', tot('v_'), 'SALU', tot('s_'), 's_nop', c['s_nop'],
          'LDS', tot('ds_'), 'VMEM', tot('global_') + tot('buffer_'))
    print('  ', ' '.join('%s:%d' % (i.replace('_e32', '').replace('_e64', ''), n) for i, n in sorted(c.items(), key=lambda x: -x[1])[:24]))
