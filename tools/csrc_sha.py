#!/usr/bin/env python3
"""sha256 over the sources of liberd_hip.so exactly as erd_amd/csrc/Makefile forms it (its explicit SRCS list, read from the Makefile,
concatenated in that order): `erd_csrc_sha()` of a library built from this checkout returns the same string.
The PMC summaries under profiles/ record it (`_meta.csrc_sha256`), bench.py compares."""
import hashlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha256(root: str = ROOT) -> str:
    d = os.path.join(root, "erd_amd", "csrc")
    mk = open(os.path.join(d, "Makefile")).read()
    names = re.search(r"^SRCS := ((?:.*\\\n)*.*)$", mk, re.M).group(1).replace("\\\n", " ").split()
    h = hashlib.sha256()
    for n in names:
        h.update(open(os.path.join(d, n), "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    print(csrc_sha256())
