#!/usr/bin/env python3
"""tools/dbg/vmcnt_audit.py FILE.s ...: per kernel of `hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only` output, the number of vector
stores and how many of them are followed by an `s_waitcnt vmcnt(0)` before the next store.  gfx950 retires loads and stores through one
in-order counter: such a wait makes the next store wait for the previous one's acknowledgement (EXPERIMENTS 7f).  Listing order, not
control flow: read the kernel's ISA before acting on a count."""
import re,sys,subprocess
def demangle(n):
    try: return subprocess.run(['c++filt',n],capture_output=True,text=True).stdout.strip()[:110]
    except Exception: return n
for f in sys.argv[1:]:
    lines=open(f).read().split('\n')
    name=None; ev=[]
    out={}
    for l in lines:
        m=re.match(r'^(_Z\w+):',l)
        if m: name=m.group(1); ev=[]; out[name]=ev; continue
        if name is None: continue
        t=l.strip()
        if t.startswith('s_endpgm'): name=None; continue
        if re.match(r'(global|buffer|flat)_store',t): ev.append('S')
        elif re.match(r'(global|buffer|flat)_load',t): ev.append('L')
        elif re.match(r'(global|buffer|flat)_atomic',t): ev.append('A')
        elif t.startswith('s_waitcnt') and 'vmcnt' in t:
            c=int(re.search(r'vmcnt\((\d+)\)',t).group(1)); ev.append('w%d'%c)
        elif t.startswith('.LBB'): ev.append('|')
    for n,ev in out.items():
        # serialization pattern: S ... w0 ... S with no label needed; count S followed (before next S) by w0
        s=''.join(e if len(e)==1 else ('0' if e=='w0' else 'w') for e in ev)
        nst=s.count('S')
        if nst==0: continue
        # store followed by a vmcnt(0) and then another store or load-use (anything) within the function
        ser=len(re.findall(r'S[^S]*?0[^S]*?(?=S)',s))
        print(f"{demangle(n):110s} stores {nst:3d}  store->vmcnt(0)->store {ser:3d}")
