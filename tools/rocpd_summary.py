#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average duration.
usage: python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.txt"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = db.execute("select name, start, end from kernels").fetchall()
agg = {}
for name, s, e in rows:
    short = re.sub(r"\(anonymous namespace\)::", "", name)
    short = re.sub(r"void ", "", short)
    short = short.split("(")[0][:110]
    a = agg.setdefault(short, [0, 0, 1 << 62, 0])
    d = e - s
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
print(f"# rocprofv3 --kernel-trace --stats summary: {len(rows)} dispatches, {tot/1e6:.2f} ms of kernel time")
print(f"{'kernel':110s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:110s} {a[0]:7d} {a[1]/1e6:10.3f} {a[1]/a[0]/1e3:10.2f} {a[2]/1e3:9.2f} {a[3]/1e3:9.2f} {100*a[1]/tot:6.2f}")
