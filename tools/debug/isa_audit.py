"""Audit of a hipcc -save-temps .s file of a kernel that names AGPRs literally / hand-counts its waits (fc_strip.hip, sim_strip.hip):
per kernel, what the COMPILER emitted outside the asm statements -- v_accvgpr_* (would collide with literally named AGPRs), scalar
loads and s_waitcnt (inside a hand-counted loop they break the counting), scratch accesses.

    python tools/debug/isa_audit.py file.s [label-substring]
"""
import sys


def audit(path, want=None, quiet=False):
    fn = None
    stats = {}
    inasm = False
    for l in open(path):
        l = l.rstrip('\n')
        if l.startswith('_Z') and ':' in l and not l.startswith('\t'):
            fn = l.split(':')[0]
            stats[fn] = dict(lines=0, mfma=0, acc_outside=0, sload_outside=0, waitcnt_outside=0, scratch=0, snop_outside=0, vmov_outside=0)
            continue
        if fn is None:
            continue
        if '#ASMSTART' in l:
            inasm = True
            continue
        if '#ASMEND' in l:
            inasm = False
            continue
        t = l.strip()
        if not t or t[0] in ';.' or t.endswith(':'):
            continue
        st = stats[fn]
        st['lines'] += 1
        if 'v_mfma' in t:
            st['mfma'] += 1
        if not inasm:
            if 'v_accvgpr' in t:
                st['acc_outside'] += 1
            if t.startswith('s_load') or t.startswith('s_buffer_load'):
                st['sload_outside'] += 1
            if 'scratch_' in t:
                st['scratch'] += 1
            if t.startswith('s_waitcnt'):
                st['waitcnt_outside'] += 1
            if t.startswith('s_nop'):
                st['snop_outside'] += 1
            if t.startswith('v_mov_b32'):
                st['vmov_outside'] += 1
    out = {k: v for k, v in stats.items() if want is None or want in k}
    if not quiet:
        for k, v in out.items():
            print(k[:70], v)
    return out


if __name__ == '__main__':
    audit(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
