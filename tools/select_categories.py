#!/usr/bin/env python3
"""Slice a COCO annotation file to the categories at positions [START, END) of the id-sorted category list -- the
40+40 protocol's `instances_train2017_first_40_cats.json` / `..._last_40_cats.json` (reference:
scripts/select_categories.py, which hard-codes 40:80 and the suffix).

    python tools/select_categories.py annotations/instances_train2017.json 0 40 --suffix _first_40_cats
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("anno_file")
    ap.add_argument("start", type=int)
    ap.add_argument("end", type=int)
    ap.add_argument("--suffix", default=None)
    a = ap.parse_args(argv)
    from erd_amd.datasets import select_categories
    ds = json.load(open(a.anno_file))
    assert isinstance(ds, dict), f"annotation file format {type(ds)} not supported"
    out = select_categories(ds, a.start, a.end)
    suffix = a.suffix if a.suffix is not None else f"_cats_{a.start}_{a.end}"
    dst = os.path.splitext(a.anno_file)[0] + suffix + ".json"
    json.dump(out, open(dst, "w"))
    print(f"{dst}: {len(out['categories'])} categories, {len(out['images'])} images, {len(out['annotations'])} annotations")


if __name__ == "__main__":
    main()
