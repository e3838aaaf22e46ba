#!/usr/bin/env python3
"""tools/kernel_resources.py FILE.hip [extra hipcc flags]: registers / spills / occupancy of every kernel in one source file
(device-only compile with -Rpass-analysis=kernel-resource-usage), one line per kernel.  Run from anywhere; no GPU needed."""
import os, re, subprocess, sys

src = os.path.abspath(sys.argv[1])
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only",
       "-Rpass-analysis=kernel-resource-usage", *sys.argv[2:], "-c", src, "-o", "/dev/null"]
err = subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.dirname(src)).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"remark:\s+(Function Name|[A-Za-z ]+(?:\[[^\]]*\])?): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["name"])
    name = re.sub(r"\(.*$", "", name)
    print("%-110s vgpr %4s agpr %4s spill %3s/%-3s occ %s scratch %s" % (
        name[:110], r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("VGPRs Spill", "?"), r.get("SGPRs Spill", "?"),
        r.get("Occupancy [waves/SIMD]", "?"), r.get("ScratchSize [bytes/lane]", "?")))
