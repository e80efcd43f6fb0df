"""Count instructions of one kernel in a hipcc -S dump: python tools/isa_count.py file.s substring"""
import sys, re, collections
s = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
start = next(i for i,l in enumerate(s) if l.startswith('_Z') and pat in l and l.rstrip().endswith(('d',':')) or (pat in l and re.match(r'^_Z\S+:', l)))
end = next(i for i in range(start, len(s)) if '.amdhsa_kernel' in s[i] or s[i].startswith('.Lfunc_end'))
ins = []
for l in s[start+1:end]:
    t = l.strip()
    if not t or t.startswith(('.', ';')) or t.endswith(':'): continue
    ins.append(t.split()[0])
c = collections.Counter(ins)
print("total", len(ins), " f64:", sum(v for k,v in c.items() if 'f64' in k), " valu:", sum(v for k,v in c.items() if k.startswith('v_')),
      " vmem:", sum(v for k,v in c.items() if k.startswith(('global_','buffer_','flat_'))), " salu:", sum(v for k,v in c.items() if k.startswith('s_')))
for k,v in c.most_common(int(sys.argv[3]) if len(sys.argv)>3 else 40): print(f"  {k:30s}{v}")
