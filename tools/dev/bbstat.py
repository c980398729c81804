"""Dev: per-basic-block instruction mix of one kernel in a gfx950 .s file (blocks that contain MFMAs).
usage: python tools/dev/bbstat.py file.s kernel-name-substring [dump-label]"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
start = [i for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l) and pat in l][0]
end = [i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm')][0]
blocks, cur, name = [], [], 'entry'
for l in lines[start:end]:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        blocks.append((name, cur)); cur = []; name = m.group(1)
    else:
        t = l.split(';')[0].strip()
        if t and not t.startswith('.'):
            cur.append(t)
blocks.append((name, cur))
def kind(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('v_accvgpr'): return 'acc'
    if op.startswith('v_exp'): return 'exp'
    if op.startswith('v_cvt_pk'): return 'cvtpk'
    if op.startswith('v_'): return 'valu'
    if op.startswith('ds_'): return 'ds'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_nop'): return 'nop'
    if op.startswith('s_barrier'): return 'bar'
    if op.startswith('s_'): return 'salu'
    if op.startswith('global_') or op.startswith('buffer_'): return 'vmem'
    if op.startswith('scratch_'): return 'scratch'
    return 'other'
for n, b in blocks:
    c = Counter(kind(t.split()[0]) for t in b)
    if c['mfma'] > 0 or c['scratch'] > 0:
        print("%-10s %4d  %s" % (n, len(b), ' '.join('%s=%d' % kv for kv in sorted(c.items()))))
if len(sys.argv) > 3:
    for n, b in blocks:
        if n == sys.argv[3]:
            print('\n'.join(b))
