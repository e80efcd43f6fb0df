"""Instruction mix of one kernel per barrier-delimited segment of a hipcc -S dump:
python tools/isa_phases.py file.s <kernel-name-substring>"""
import collections
import re
import sys

s = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
start = next(i for i, l in enumerate(s) if re.match(r'^_Z\S+:', l) and pat in l)
end = next(i for i in range(start, len(s)) if s[i].startswith('.Lfunc_end'))
seg, segs = [], []
for l in s[start + 1:end]:
    t = l.strip()
    if not t or t.startswith(('.', ';')) or t.endswith(':'):
        continue
    op = t.split()[0]
    seg.append(op)
    if op == 's_barrier':
        segs.append(seg)
        seg = []
segs.append(seg)


def cls(op):
    if op.startswith('v_') and 'f64' in op: return 'f64'
    if op in ('v_readlane_b32', 'v_writelane_b32'): return 'lane(spill)'
    if op.startswith('v_'): return 'valu_other'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_')): return 'vmem'
    if op == 's_waitcnt': return 'waitcnt'
    if op == 's_nop': return 'nop'
    if op.startswith('s_'): return 'salu'
    return 'other'


for i, g in enumerate(segs):
    c = collections.Counter(cls(o) for o in g)
    print(f"segment {i}: {len(g):5d} instr  " + "  ".join(f"{k}={v}" for k, v in sorted(c.items())))
    if len(sys.argv) > 3:
        print("    " + ", ".join(f"{k}:{v}" for k, v in collections.Counter(o for o in g if cls(o) == 'valu_other').most_common(12)))
