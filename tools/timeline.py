#!/usr/bin/env python3
"""Timeline view of a rocprofv3 (rocpd sqlite) kernel trace of the DEFAULT (multi-stream) bench run: for one steady
step (sgd_kernel to sgd_kernel) how much wall time had no kernel in flight, how much had fewer than one workgroup per
CU in flight ("thin": latency-bound single kernels such as the distillation NMS), and which kernels own that time.
usage: python tools/timeline.py gpurun_out/prof/x_results.db [step_index]"""
import re
import sqlite3
import sys
from collections import defaultdict

NCU = 256
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, grid_x*grid_y*grid_z/(workgroup_x*workgroup_y*workgroup_z), queue_id "
                  "from kernels order by start").fetchall()


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"void ", "", n).split("(")[0][:60]


# step boundary: a kernel that runs exactly ONCE per step (loss_avg_kernel: the cross-image normalisers of the loss; the SGD update
# runs once per gradient bucket since round 4 and no longer marks a step)
marker = "loss_avg_kernel" if any("loss_avg_kernel" in r[0] for r in rows) else "sgd_kernel"
sgd = [i for i, r in enumerate(rows) if marker in r[0]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(sgd) // 2
lo, hi = sgd[k], sgd[k + 1]
step = rows[lo:hi]
t0, t1 = step[0][1], rows[hi][1]
ev = []
for i, (n, s, e, wgs, q) in enumerate(step):
    ev.append((s, 1, i)); ev.append((min(e, t1), -1, i))
ev.sort()
active, last = set(), t0
idle = thin = multi = 0
busy_sum = sum(min(e, t1) - s for _, s, e, _, _ in step)
owner = defaultdict(int)
for t, d, i in ev:
    dt = t - last
    if dt > 0:
        if len(active) > 1:
            multi += dt
        if not active:
            idle += dt
        else:
            wg = sum(step[j][3] for j in active)
            if wg < NCU:
                thin += dt
                for j in active:
                    owner[short(step[j][0])] += dt
    last = t
    if d > 0:
        active.add(i)
    else:
        active.discard(i)
print(f"step {k} ({marker} to {marker}): {len(step)} dispatches, wall {(t1 - t0) / 1e6:.2f} ms, sum of kernel durations {busy_sum / 1e6:.2f} ms, "
      f"two or more kernels in flight {multi / 1e6:.2f} ms, no kernel in flight {idle / 1e6:.2f} ms, "
      f"fewer than {NCU} workgroups in flight {thin / 1e6:.2f} ms; queues {sorted(set(r[4] for r in step))}")
by_name = defaultdict(lambda: [0, 0])
for n, s_, e_, _, _ in step:
    by_name[short(n)][0] += 1
    by_name[short(n)][1] += min(e_, t1) - s_
print("kernel time inside the step by symbol (launches, ms):")
for n, (c, v) in sorted(by_name.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"  {n:60s} {c:5d} {v / 1e6:8.3f}")
print("time with fewer than 256 workgroups in flight, by kernel:")
for n, v in sorted(owner.items(), key=lambda kv: -kv[1])[:20]:
    print(f"  {n:60s} {v / 1e3:9.1f} us")
