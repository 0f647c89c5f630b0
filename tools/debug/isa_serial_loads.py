#!/usr/bin/env python3
"""Finds compiler-serialised loads in `hipcc -S` output: a vector-memory load whose data is waited for (`s_waitcnt vmcnt(0)`) within a
few instructions, with no other load in between -- one dependent round trip per load.  usage: isa_serial_loads.py file.s [window]"""
import re, sys
src = open(sys.argv[1]).read().split('\n')
win = int(sys.argv[2]) if len(sys.argv) > 2 else 8
name, res = None, {}
ins = []
def flush():
    if name is None: return
    loads = [i for i, t in enumerate(ins) if re.match(r'(global|buffer|flat)_load', t)]
    serial = 0
    for k, i in enumerate(loads):
        nxt = loads[k + 1] if k + 1 < len(loads) else len(ins)
        for j in range(i + 1, min(i + 1 + win, nxt + 1, len(ins))):
            if ins[j].startswith('s_waitcnt') and 'vmcnt(0)' in ins[j]:
                serial += 1
                break
    res[name] = (len(loads), serial)
for l in src:
    m = re.match(r'^(_Z\S+):', l)
    if m:
        flush(); name = m.group(1); ins = []
        continue
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'): continue
    ins.append(t.split(';')[0].strip())
flush()
for k, (n, s) in sorted(res.items(), key=lambda kv: -kv[1][1]):
    if n: print('%4d loads, %4d waited for alone  %s' % (n, s, k[:110]))
