import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
for m in re.finditer(r'^(_ZN\S*' + pat + r'\S*):[^\n]*\n(.*?)\.Lfunc_end\d+:', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    def g(k):
        r = re.search(re.escape(name) + r'\.' + k + r', (\d+)', s)
        return r.group(1) if r else '?'
    code = re.search(r'; codeLenInByte = (\d+)', s[m.end():m.end() + 4000])
    print(name[20:60], 'vgpr', g('num_vgpr'), 'sgpr', g('numbered_sgpr'), 'scratch', g('private_seg_size'),
          '| pk_fma', body.count('v_pk_fma_f32'), 'fma', body.count('v_fma_f32'), 'fmac', body.count('v_fmac_f32'), 'pk_mul', body.count('v_pk_mul_f32'),
          'mov', len(re.findall(r'v_mov_b32|v_pk_mov', body)), 'vmcnt0', len(re.findall(r'vmcnt\(0\)', body)), 'waitcnt', body.count('s_waitcnt'),
          'ds_read', len(re.findall(r'ds_read', body)), 'ds_write', len(re.findall(r'ds_write', body)), 'gload', len(re.findall(r'global_load', body)), 'code', code and code.group(1))
