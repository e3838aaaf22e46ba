#!/usr/bin/env python3
"""sha256 over the sources of liberd_hip.so exactly as erd_amd/csrc/Makefile forms it (every .hip / .h of csrc/, the Makefile and
include/erd_hip.h, concatenated in sorted path order): `erd_csrc_sha()` of a library built from this checkout returns the same string.
The PMC summaries under profiles/ record it (`_meta.csrc_sha256`), bench.py compares."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha256(root: str = ROOT) -> str:
    d = os.path.join(root, "erd_amd", "csrc")
    names = sorted([os.path.basename(f) for f in glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))] +
                   ["Makefile", "../../include/erd_hip.h"])
    h = hashlib.sha256()
    for n in names:
        h.update(open(os.path.join(d, n), "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    print(csrc_sha256())
