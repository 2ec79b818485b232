"""Developer aid: list the loops of a gfx950 .s file with their instruction counts, DPP/transcendental/scratch counts."""
import re, sys, collections
lines = open(sys.argv[1]).read().split('\n')
labels = {}
for i, l in enumerate(lines):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
loops = []
for i, l in enumerate(lines):
    m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
def stats(a, b):
    ops = collections.Counter()
    for l in lines[a:b]:
        l = l.strip()
        if not l or l[0] in ';.': continue
        ops[l.split()[0]] += 1
    tot = sum(ops.values())
    dpp = sum(v for k, v in ops.items() if 'dpp' in k)
    trans = sum(v for k, v in ops.items() if re.match(r'v_(rcp|rsq|sqrt|log|exp)_', k))
    scr = sum(v for k, v in ops.items() if 'scratch' in k)
    ds = sum(v for k, v in ops.items() if k.startswith('ds_'))
    nop = ops.get('s_nop', 0)
    mov = ops.get('v_mov_b32_e32', 0)
    return tot, dpp, trans, scr, ds, nop, mov, ops
for a, b in sorted(loops, key=lambda t: t[0] - t[1])[:6]:
    tot, dpp, trans, scr, ds, nop, mov, ops = stats(a, b)
    print(f"loop lines {a}-{b}: {tot} instrs, dpp {dpp}, trans {trans}, scratch {scr}, ds {ds}, s_nop {nop}, v_mov {mov}; slots~{tot + dpp + 3*trans}")
    if len(sys.argv) > 2:
        for k, v in ops.most_common(18): print(f"      {v:5d} {k}")
