#!/usr/bin/env python3
"""Static instruction histogram per kernel from a hipcc -save-temps gfx950 .s file."""
import collections, re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ''
funcs = re.split(r'\n\s*\.globl\s+', s)
for f in funcs[1:]:
    name = f.split('\n', 1)[0].strip()
    if pat and not re.search(pat, name): continue
    body = f[:f.find('.end_amdhsa_kernel')] if '.end_amdhsa_kernel' in f else f
    ins = []
    for l in body.split('\n'):
        t = l.strip()
        if not l.startswith('\t') or not t or t.startswith('.') or t.startswith(';'): continue
        ins.append(t.split()[0])
    c = collections.Counter(ins); tot = sum(c.values()); grp = collections.Counter()
    for k, v in c.items():
        if re.match(r'v_(div|rcp|sqrt|rsq)', k): g = 'div/rcp/sqrt'
        elif k.startswith('ds_'): g = 'lds'
        elif re.match(r'(global|buffer|flat|scratch)', k): g = 'vmem'
        elif k == 's_waitcnt': g = 'waitcnt'
        elif k.startswith('s_'): g = 'salu'
        elif re.match(r'v_(cndmask|cmp)', k): g = 'cmp/sel'
        elif re.match(r'v_(mul_lo|mul_hi|mad_u|mad_i|add_u|sub_u|add_co|lshl|lshr|ashr|and_|or_|xor|bfe|subrev_u|addc|subb|add3|lshl_add|mad_u64|add_lshl|and_or)', k): g = 'int'
        else: g = 'fp/other valu'
        grp[g] += v
    print(name[:70], 'static', tot, dict(grp))
    print('   top:', c.most_common(16))
