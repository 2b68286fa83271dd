#!/usr/bin/env python3
"""Loops of one kernel in a hipcc -S gfx950 .s file: for every backward branch the instruction mix of the range it spans (vector ALU,
SGPR-spill traffic = v_readlane / v_writelane, v_mov, LDS, VMEM, waits). Usage: asm_loops.py file.s <kernel name regex> [min instructions]"""
import collections, re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
minins = int(sys.argv[3]) if len(sys.argv) > 3 else 200
for f in re.split(r'\n\s*\.globl\s+', s)[1:]:
    name = f.split('\n', 1)[0].strip()
    if not re.search(pat, name):
        continue
    body = f[:f.find('.end_amdhsa_kernel')] if '.end_amdhsa_kernel' in f else f
    lines = body.split('\n')
    labels, ins = {}, []
    for l in lines:
        t = l.strip()
        m = re.match(r'^(\.LBB\d+_\d+):', t)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if not l.startswith('\t') or not t or t.startswith('.') or t.startswith(';'):
            continue
        ins.append(t)
    print(name[:90], 'instructions', len(ins))
    loops = []
    for i, t in enumerate(ins):
        m = re.match(r's_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', t)
        if m:
            tgt = labels.get(m.group(1) or m.group(2))
            if tgt is not None and tgt <= i and i - tgt >= minins:
                loops.append((tgt, i))
    for a, b in loops:
        c = collections.Counter(x.split()[0] for x in ins[a:b + 1])
        valu = sum(v for k, v in c.items() if k.startswith('v_'))
        print(f'  loop [{a}, {b}] {b - a + 1} instructions: VALU {valu}, v_readlane {c["v_readlane_b32"]}, v_writelane {c["v_writelane_b32"]}, v_mov {c["v_mov_b32_e32"] + c["v_mov_b32_dpp"]}, '
              f'ds {sum(v for k, v in c.items() if k.startswith("ds_"))}, vmem {sum(v for k, v in c.items() if re.match("(global|buffer|scratch|flat)_", k))}, '
              f's_waitcnt {c["s_waitcnt"]}, salu {sum(v for k, v in c.items() if k.startswith("s_"))}')
