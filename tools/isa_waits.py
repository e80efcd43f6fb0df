"""Show where a kernel waits on vector memory: python tools/isa_waits.py file.s <kernel-name-substring>"""
import re, sys
s = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
start = next(i for i, l in enumerate(s) if re.match(r'^_Z\S+:', l) and pat in l)
end = next(i for i in range(start, len(s)) if s[i].startswith('.Lfunc_end'))
body = s[start:end]
run = None
for i, l in enumerate(body):
    t = l.strip()
    if t.startswith(('global_load', 'global_store', 'buffer_')):
        k = 'LD' if 'load' in t else 'ST'
        if run and run[0] == k:
            run[2] += 1
        else:
            if run: print(f"{run[1]:5d}     {run[0]} x{run[2]}")
            run = [k, i, 1]
        continue
    if t.startswith('s_waitcnt') and 'vmcnt' in t or 's_barrier' in t:
        if run: print(f"{run[1]:5d}     {run[0]} x{run[2]}"); run = None
        print(f"{i:5d} {'=== barrier' if 's_barrier' in t else t}")
if run: print(f"{run[1]:5d}     {run[0]} x{run[2]}")
