#!/usr/bin/env python3
"""Instruction census of the strip kernel's block-loop bodies from hipcc -S output.
usage: isa_body.py file.s mangled-name-substring"""
import re, sys, collections
src = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
# function text
start = next(i for i, l in enumerate(src) if l.startswith('_ZN') and key in l and re.match(r'^_ZN\S*:', l))
end = next(i for i in range(start, len(src)) if src[i].strip().startswith('.amdhsa_kernel') or src[i].strip() == 's_endpgm')
fn = src[start:end]
# basic blocks
blocks, cur = [], []
for l in fn:
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'):
        if t.startswith('.LBB'):
            blocks.append(cur); cur = []
        continue
    if re.match(r'^\.?LBB\S*:', t):
        blocks.append(cur); cur = []
        continue
    cur.append(t.split(';')[0].strip())
blocks.append(cur)
for bi, b in enumerate(blocks):
    nm = sum(1 for i in b if i.startswith('v_mfma'))
    if nm < 32:
        continue
    c = collections.Counter()
    for i in b:
        op = i.split()[0]
        if op.startswith('v_mfma'): k = 'mfma'
        elif op == 's_nop': k = 's_nop'
        elif op == 's_waitcnt': k = 's_waitcnt'
        elif op.startswith('v_mov') or op.startswith('v_accvgpr'): k = 'v_mov/acc'
        elif op.startswith('ds_'): k = 'ds'
        elif op.startswith('buffer_') or op.startswith('global_'): k = 'vmem'
        elif op.startswith('v_readlane') or op.startswith('v_writelane'): k = 'lane'
        elif op.startswith('v_'): k = 'valu'
        elif op.startswith('s_'): k = 'salu'
        else: k = op
        c[k] += 1
    # per-slot sizes
    sizes, n = [], 0
    for i in b:
        if i.startswith('v_mfma'):
            sizes.append(n); n = 0
        else:
            n += 1
    sizes.append(n)
    print(f'block {bi}: {len(b)} instr, {nm} mfma :', dict(c))
    print('   slot sizes:', sizes)
print('blocks with mfma:', [(bi, len(b), sum(1 for i in b if i.startswith('v_mfma'))) for bi, b in enumerate(blocks) if any(i.startswith('v_mfma') for i in b)])
tot = collections.Counter()
for b in blocks:
    if not any(i.startswith('v_mfma') for i in b): continue
    for i in b:
        op = i.split()[0]
        if op.startswith('v_mfma'): k = 'mfma'
        elif op in ('s_nop', 's_waitcnt', 's_barrier'): k = op
        elif op.startswith('v_mov') or op.startswith('v_accvgpr'): k = 'v_mov/acc'
        elif op.startswith('ds_'): k = 'ds'
        elif op.startswith('buffer_') or op.startswith('global_'): k = 'vmem'
        elif op.startswith('v_readlane') or op.startswith('v_writelane'): k = 'lane'
        elif op.startswith('v_'): k = 'valu'
        elif op.startswith('s_'): k = 'salu'
        else: k = op
        tot[k] += 1
print('all mfma blocks:', sum(tot.values()), dict(tot))
