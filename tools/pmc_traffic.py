#!/usr/bin/env python3
"""HBM-side traffic per kernel launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass:
MI355X_MICROARCH.md 'rocprofv3 PMC slots').

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/rNN_pmc_traffic.json

Units / corrections as the guide prescribes: both counters are in KiB-like units of 1024 B as rocprofv3 reports
them; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so wide coalesced reads are DOUBLED (`fetch_MB_corrected`).
WRITE_SIZE is uncalibrated (reported as is)."""
import collections
import csv
import glob
import json
import os
import re
import sys


def load(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    return agg


def meta():
    """which library the counters were taken on: the sha of its sources (erd_csrc_sha(); bench.py flags a summary whose sha is not the live library's)"""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from erd_amd import _lib
    return {"csrc_sha256": _lib.load().erd_csrc_sha().decode()}


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(fetch, key=lambda k: -fetch[k][1]):
        n, f = fetch[k]
        wn, w = write.get(k, [0, 0.0])
        if "conv_" not in k and f / max(n, 1) < 1024:
            continue
        f_mb = f / n * 1024 / 1e6
        w_mb = (w / wn * 1024 / 1e6) if wn else None
        out[k] = dict(launches=n, fetch_MB_raw=round(f_mb, 3), fetch_MB_corrected=round(2 * f_mb, 3),
                      write_MB=None if w_mb is None else round(w_mb, 3),
                      traffic_MB=None if w_mb is None else round(2 * f_mb + w_mb, 3))
    out["_meta"] = meta()
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
