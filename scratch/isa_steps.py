# per-barrier-interval instruction mix of a kernel in a hipcc -S dump: python isa_steps.py file.s symbol-substring
import re, sys
from collections import Counter
s = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
start = next(i for i, l in enumerate(s) if l.startswith('_Z') and pat in l and l.rstrip().split(':')[0].endswith('E') and ':' in l)
end = next(i for i in range(start, len(s)) if s[i].startswith('.Lfunc_end'))
lines = [l.strip() for l in s[start + 1:end] if l.strip() and not l.strip().startswith((';', '.'))]
segs, cur = [], []
for l in lines:
    cur.append(l)
    if l.startswith('s_barrier'): segs.append(cur); cur = []
segs.append(cur)
print('intervals', len(segs), 'instructions', len(lines))
for i, sg in enumerate(segs):
    c = Counter(l.split()[0] for l in sg)
    g = lambda p: sum(v for k, v in c.items() if k.startswith(p))
    print(i, 'n', len(sg), 'pk_fma', c['v_pk_fma_f32'], 'mov', g('v_mov') + g('v_pk_mov') + g('v_accvgpr'), 'ds_r', g('ds_read'), 'ds_w', g('ds_write'), 'gl', g('global_load'), 'gs', g('global_store'),
          'salu', g('s_'), 'vcmp', g('v_cmp'), 'cnd', g('v_cndmask'), 'valu_other', sum(v for k, v in c.items() if k.startswith('v_') and not k.startswith(('v_pk_fma', 'v_mov', 'v_pk_mov', 'v_accvgpr', 'v_cmp', 'v_cndmask'))))
