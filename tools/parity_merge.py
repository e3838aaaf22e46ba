#!/usr/bin/env python3
"""tools/parity_merge.py OUT.json IN1.json IN2.json ...: concatenate the per-seed rows of several tools/parity_seeds.py runs (disjoint
seed ranges, same evaluation) for the implementations ALL of them hold and recompute the summary / paired statistics."""
import json
import sys

import numpy as np

out, ins = sys.argv[1], [json.load(open(f)) for f in sys.argv[2:]]
keys = [k for k in ins[0]["rows"] if all(k in d["rows"] for d in ins)]
seeds = sum((d["seeds"] for d in ins), [])
assert len(set(seeds)) == len(seeds), "overlapping seed ranges"
rows = {k: sum((d["rows"][k] for d in ins), []) for k in keys}


def summary(a):
    a = np.array(a)
    return {"mean": [float(v) for v in a.mean(0)], "median": [float(v) for v in np.median(a, 0)],
            "sem": [float(v) for v in a.std(0, ddof=1) / np.sqrt(len(a))], "max": [float(v) for v in a.max(0)],
            "seeds_whole_gradient_above_1e-3": int((a[:, 1] > 1e-3).sum()), "seeds_whole_gradient_above_2e-3": int((a[:, 1] > 2e-3).sum())}


pairs = {}
for i, a in enumerate(keys):
    for b in keys[i + 1:]:
        A, B = np.array(rows[a]), np.array(rows[b])
        da = A - B
        pairs["%s_minus_%s" % (a, b)] = {"mean": [float(v) for v in da.mean(0)], "sem": [float(v) for v in da.std(0, ddof=1) / np.sqrt(len(da))],
                                         "ratio_of_means": [float(v) for v in A.mean(0) / B.mean(0)],
                                         "ratio_of_medians": [float(v) for v in np.median(A, 0) / np.median(B, 0)],
                                         "seeds_a_closer": int((da[:, 1] < 0).sum())}
res = {"what": ins[0]["what"], "merged_from": sys.argv[2:], "tags": [d.get("tag", "") for d in ins], "seeds": seeds, "rows": rows,
       "summary": {k: summary(v) for k, v in rows.items()}, "paired": pairs,
       "worst_loss_entry_rel_dev_from_fp64": {k: max(d["worst_loss_entry_rel_dev_from_fp64"].get(k, 0.0) for d in ins) for k in keys if k != "cpu_f32"}}
json.dump(res, open(out, "w"), indent=1)
for k, v in res["summary"].items():
    print("%-12s n=%d mean %.2e %.2e %.2e | median %.2e %.2e %.2e | sem(whole) %.1e | > 1e-3: %d" % (
        (k, len(seeds)) + tuple(v["mean"]) + tuple(v["median"]) + (v["sem"][1], v["seeds_whole_gradient_above_1e-3"])))
for k, v in pairs.items():
    print("%-28s whole-gradient mean diff %+.2e +- %.1e, ratio of means %.3f, of medians %.3f" % (k, v["mean"][1], v["sem"][1], v["ratio_of_means"][1], v["ratio_of_medians"][1]))
