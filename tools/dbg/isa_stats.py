#!/usr/bin/env python3
"""per-kernel ISA statistics of a `hipcc -S --cuda-device-only` listing: MFMAs, LDS-DMA loads, vmcnt(0) waits, registers, scratch.
usage: python tools/dbg/isa_stats.py file.s [name filter]"""
import re, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_Z\S+):[^\n]*\n(.*?)^\s*\.end_amdhsa_kernel", s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name:
        continue
    g = lambda pat: (re.search(pat, body) or [None, "?"])[1]
    print(name[:150])
    print("   mfma %d  lds-dma %d  vmcnt(0) %d  ds_write %d  ds_read %d  buffer/global loads %d stores %d  vgpr %s agpr %s scratch %s lds %s" % (
        len(re.findall(r"\bv_mfma", body)), len(re.findall(r"offen lds|\blds\b", body)), len(re.findall(r"s_waitcnt vmcnt\(0\)", body)),
        len(re.findall(r"\bds_write", body)), len(re.findall(r"\bds_read", body)),
        len(re.findall(r"\b(buffer|global)_load", body)), len(re.findall(r"\b(buffer|global)_store", body)),
        g(r"\.amdhsa_next_free_vgpr (\d+)"), g(r"\.amdhsa_accum_offset (\d+)"), g(r"\.amdhsa_private_segment_fixed_size (\d+)"),
        g(r"\.amdhsa_group_segment_fixed_size (\d+)")))
