"""Developer aid: instruction mix of the big loops (sweep loop, class loop) of a pair-kernel .s file (-save-temps=obj),
plus the register / spill figures of every kernel in it:  python tools/asm_sweep.py file.s"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
labels = {}
for i, l in enumerate(lines):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
for i, l in enumerate(lines):
    m = re.search(r's_(c?branch\w*)\s+(\.LBB\d+_\d+)', l)
    if m and m.group(2) in labels and labels[m.group(2)] < i and i - labels[m.group(2)] > 200:
        a = labels[m.group(2)]
        seg = [x for x in lines[a:i] if x.strip() and x.strip()[0] not in ';.']
        cnt = lambda pat: sum(bool(re.search(pat, x)) for x in seg)
        print(f"loop {a}-{i}: {len(seg)} instrs | fma {cnt(r'v_fma|v_fmac')} mul {cnt(r'v_mul_f')} dpp {cnt('dpp')} swizzle {cnt('ds_swizzle')} "
              f"ds_other {cnt(r'ds_(?!swizzle)')} permlane {cnt('permlane')} cndmask {cnt('v_cndmask')} trans {cnt(r'v_(rcp|rsq|sqrt|log|exp)_')} "
              f"mov {cnt(r'v_mov_b32_e32')} waitcnt {cnt('s_waitcnt')} nop {cnt('s_nop')} scratch {cnt('scratch')}")
for l in lines:
    if re.search(r'vgpr_spill_count|\.vgpr_count|\.name:|group_segment_fixed_size', l): print(l.strip())
