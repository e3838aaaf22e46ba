#!/usr/bin/env python3
"""tools/dbg/vmcnt_audit.py FILE.s ...: per kernel of `hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only` output, the number of vector
stores and how many of them are followed by an `s_waitcnt vmcnt(0)` before the next store.  gfx950 retires loads and stores through one
in-order counter: such a wait makes the next store wait for the previous one's acknowledgement (EXPERIMENTS 7f).  Listing order, not
control flow: read the kernel's ISA before acting on a count.  tests/test_csrc_isa.py holds the thin-K kernel to zero."""
import re
import subprocess
import sys


def audit(path):
    """{mangled kernel name: dict(stores, serialized, scratch, packed_f32)}"""
    name, ev, out = None, None, {}
    for l in open(path):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            name = m.group(1)
            ev = out[name] = dict(events=[], scratch=0, packed_f32=0)
            continue
        if name is None:
            continue
        t = l.strip()
        if t.startswith('s_endpgm'):
            name = None
        elif re.match(r'(global|buffer|flat)_store', t):
            ev['events'].append('S')
        elif re.match(r'(global|buffer|flat)_(load|atomic)', t):
            ev['events'].append('L')
        elif t.startswith('scratch_'):
            ev['scratch'] += 1
        elif re.match(r'v_pk_(add|mul|fma)_f32', t):
            ev['packed_f32'] += 1
        elif t.startswith('s_waitcnt') and 'vmcnt' in t:
            ev['events'].append('0' if int(re.search(r'vmcnt\((\d+)\)', t).group(1)) == 0 else 'w')
    res = {}
    for n, e in out.items():
        s = ''.join(e['events'])
        if 'S' in s:
            res[n] = dict(stores=s.count('S'), serialized=len(re.findall(r'S[^S]*?0[^S]*?(?=S)', s)), scratch=e['scratch'], packed_f32=e['packed_f32'])
    return res


def demangle(n):
    try:
        return subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip()[:110]
    except Exception:
        return n


if __name__ == "__main__":
    for f in sys.argv[1:]:
        for n, r in audit(f).items():
            print(f"{demangle(n):110s} stores {r['stores']:3d}  store->vmcnt(0)->store {r['serialized']:3d}  scratch ops {r['scratch']:3d}  packed f32 {r['packed_f32']:3d}")
