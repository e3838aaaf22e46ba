#!/usr/bin/env python3
"""images/s of the real-data input side (JPEG decode on `num_workers` threads -> pinned staging -> GPU resize / flip /
normalise / pad) on COCO-sized synthetic JPEGs, to set beside the step rate (64 img/s per GPU in fp32, 121 in bf16 mode).
usage: python tools/bench_datapipe.py [n_images]"""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image
from erd_amd.runner import CocoTrainData

n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
tmp = tempfile.mkdtemp(prefix="erd_dp_")
rng = np.random.RandomState(0)
images, anns = [], []
base = rng.randint(0, 256, (60, 80, 3), dtype=np.uint8)
for i in range(n):
    h, w = (480, 640) if i % 3 else (640, 480)
    arr = np.asarray(Image.fromarray(np.roll(base, i, 1)).resize((w, h), Image.BICUBIC))      # smooth content: realistic JPEG sizes
    arr = np.clip(arr.astype(np.int16) + rng.randint(-12, 12, arr.shape), 0, 255).astype(np.uint8)
    Image.fromarray(arr).save(os.path.join(tmp, f"{i:06d}.jpg"), quality=90)
    images.append(dict(id=i, file_name=f"{i:06d}.jpg", width=w, height=h))
    anns.append(dict(id=i + 1, image_id=i, category_id=1, bbox=[10.0, 20.0, 200.0, 150.0], area=30000.0, iscrowd=0))
json.dump(dict(images=images, annotations=anns, categories=[dict(id=1, name="a")]), open(os.path.join(tmp, "ann.json"), "w"))
kb = sum(os.path.getsize(os.path.join(tmp, f)) for f in os.listdir(tmp) if f.endswith(".jpg")) / n / 1024
print(f"{n} JPEGs, {kb:.0f} KB each, host cores {os.cpu_count()}")
dcfg = dict(data_root=tmp, ann_file="ann.json", data_prefix=dict(img=""), metainfo=dict(classes=("a",)))
for workers in (0, 2, 4, 8, 16):
    data = CocoTrainData(dcfg, batch_size=4, scale=(1333, 800), seed=0, num_workers=workers, prefetch_factor=3)
    for _ in data:            # warm-up epoch: file cache, resize tables, allocator
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cnt = 0
    for b in data:
        cnt += b["inputs"].shape[0]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"num_workers {workers:2d}: {cnt / dt:7.1f} img/s")
