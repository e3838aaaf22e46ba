"""CPU: annotation side of the data pipeline -- category slicing and aspect-ratio batching
against the reference's own code run in this container, COCO parsing/filter rules as known answers
(mmdet/datasets/coco.py:102-212)."""
import importlib.util
import json
import os
import runpy
import sys

import numpy as np
import pytest

from erd_amd.datasets import AspectRatioBatchSampler, CocoAnnotations, rescale_size, select_categories

REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")


def toy_coco(n_img=60, n_cat=12, seed=0):
    rng = np.random.RandomState(seed)
    cats = [dict(id=int(i), name=f"c{i}") for i in rng.permutation(np.arange(1, 3 * n_cat, 3))[:n_cat]]
    images = [dict(id=int(1000 + i), file_name=f"{i:06d}.jpg", width=int(rng.randint(20, 700)), height=int(rng.randint(20, 700)))
              for i in rng.permutation(n_img)]
    annos, aid = [], 1
    for im in images:
        for _ in range(rng.randint(0, 5)):
            x, y = rng.uniform(-5, im["width"]), rng.uniform(-5, im["height"])
            w, h = rng.choice([0.5, 8.0, 60.0]), rng.choice([0.7, 12.0, 45.0])
            annos.append(dict(id=aid, image_id=im["id"], category_id=cats[rng.randint(n_cat)]["id"],
                              bbox=[float(x), float(y), float(w), float(h)], area=float(w * h) if rng.rand() > 0.1 else 0.0,
                              iscrowd=int(rng.rand() < 0.15)))
            aid += 1
    return dict(images=images, annotations=annos, categories=cats)


@needs_ref
def test_select_categories_equals_reference_script(tmp_path, monkeypatch):
    ds = toy_coco(n_cat=80, n_img=200, seed=3)
    src = tmp_path / "instances_toy.json"
    json.dump(ds, open(src, "w"))
    monkeypatch.setattr(sys, "argv", ["select_categories.py", "--data_path", str(tmp_path), "--anno_file", "instances_toy"])
    runpy.run_path(os.path.join(REF, "scripts", "select_categories.py"), run_name="__main__")   # writes ..._last_40_cats.json
    want = json.load(open(tmp_path / "instances_toy_last_40_cats.json"))
    got = select_categories(ds, 40, 80)
    assert got == want
    first = select_categories(ds, 0, 40)
    assert not ({c["id"] for c in first["categories"]} & {c["id"] for c in got["categories"]})
    assert max(c["id"] for c in first["categories"]) < min(c["id"] for c in got["categories"])
    # the CLI writes the same thing
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import select_categories as cli
    cli.main([str(src), "40", "80", "--suffix", "_mine"])
    assert json.load(open(tmp_path / "instances_toy_mine.json")) == want


def test_coco_parse_and_filter_known_answers():
    ds = dict(
        categories=[dict(id=7, name="dog"), dict(id=3, name="cat"), dict(id=9, name="bird")],
        images=[dict(id=1, file_name="a.jpg", width=100, height=80), dict(id=2, file_name="b.jpg", width=30, height=200),
                dict(id=3, file_name="c.jpg", width=64, height=64), dict(id=4, file_name="d.jpg", width=90, height=90)],
        annotations=[
            dict(id=1, image_id=1, category_id=3, bbox=[10, 10, 20, 30], area=600, iscrowd=0),       # kept -> label of 'cat'
            dict(id=2, image_id=1, category_id=7, bbox=[95, 70, 20, 20], area=400, iscrowd=1),       # kept, ignore_flag
            dict(id=3, image_id=1, category_id=9, bbox=[1, 1, 5, 5], area=25, iscrowd=0),            # class not wanted
            dict(id=4, image_id=1, category_id=3, bbox=[100, 10, 20, 30], area=600, iscrowd=0),      # outside the image
            dict(id=5, image_id=1, category_id=3, bbox=[10, 10, 0.5, 30], area=15, iscrowd=0),       # w < 1
            dict(id=6, image_id=1, category_id=3, bbox=[10, 10, 20, 30], area=0, iscrowd=0),         # area <= 0
            dict(id=7, image_id=1, category_id=3, bbox=[10, 10, 20, 30], area=600, iscrowd=0, ignore=True),
            dict(id=8, image_id=2, category_id=7, bbox=[0, 0, 10, 10], area=100, iscrowd=0),         # image too small (30 < 32)
            dict(id=9, image_id=3, category_id=9, bbox=[0, 0, 10, 10], area=100, iscrowd=0),         # only an unwanted class
        ])
    a = CocoAnnotations(ds, classes=("cat", "dog"), data_prefix="train2017/")
    # cat ids follow the FILE's category order (dog=7 first), not the order of `classes`
    assert a.cat_ids == [7, 3] and a.cat2label == {7: 0, 3: 1}
    assert [d["img_id"] for d in a.data_list] == [1]          # 2: min_size, 3: no wanted class, 4: no annotation
    inst = a.data_list[0]["instances"]
    assert inst == [dict(bbox=[10, 10, 30, 40], bbox_label=1, ignore_flag=0), dict(bbox=[95, 70, 115, 90], bbox_label=0, ignore_flag=1)]
    assert a.data_list[0]["img_path"] == "train2017/a.jpg"
    t = CocoAnnotations(ds, classes=("cat", "dog"), test_mode=True)
    assert [d["img_id"] for d in t.data_list] == [1, 2, 3, 4]
    e = CocoAnnotations(ds, classes=("cat", "dog"), filter_empty_gt=False, min_size=0)
    assert [d["img_id"] for d in e.data_list] == [1, 2, 3, 4] and e.data_list[2]["instances"] == []
    s = a.data_sample(0, scale_factor=(2.0, 2.0), flip=True)
    assert s.gt_instances.bboxes.tolist() == [[200 - 60, 20, 200 - 20, 80]] and s.gt_instances.labels.tolist() == [1]
    assert s.ignored_instances.bboxes.shape == (1, 4) and s.metainfo["img_shape"] == (160, 200)
    with pytest.raises(AssertionError):
        CocoAnnotations(dict(ds, annotations=ds["annotations"] + [ds["annotations"][0]]), classes=("cat",))


def test_rescale_size_known_answers():
    # mmcv.imrescale((1333, 800), keep_ratio) on the usual COCO shapes
    assert rescale_size((640, 480), (1333, 800)) == (1067, 800)
    assert rescale_size((640, 427), (1333, 800)) == (1199, 800)
    assert rescale_size((500, 375), (1333, 800)) == (1067, 800)
    assert rescale_size((640, 240), (1333, 800)) == (1333, 500)
    assert rescale_size((480, 640), (1333, 800)) == (800, 1067)


@needs_ref
@pytest.mark.parametrize("bs,drop_last", [(2, False), (4, False), (4, True), (7, False)])
def test_aspect_ratio_batch_sampler_equals_reference_class(bs, drop_last):
    from oracle import ref_stub
    ref_stub.load_reference()
    reg = sys.modules["mmdet.registry"]
    if not hasattr(reg, "DATA_SAMPLERS"):
        class _R:
            def register_module(self, *a, **k):
                return lambda c: c
        reg.DATA_SAMPLERS = _R()
    spec = importlib.util.spec_from_file_location("_ref_batch_sampler", os.path.join(REF, "mmdet/datasets/samplers/batch_sampler.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    from torch.utils.data import Sampler
    a = CocoAnnotations(toy_coco(n_img=61, seed=5), classes=[f"c{i}" for i in range(1, 40, 3)], filter_empty_gt=False, min_size=0)

    class S(Sampler):
        def __init__(self, order, dataset):
            self.order, self.dataset = order, dataset

        def __iter__(self):
            return iter(self.order)

        def __len__(self):
            return len(self.order)

    order = [int(i) for i in np.random.RandomState(bs).permutation(len(a))]
    want = list(mod.AspectRatioBatchSampler(S(order, a), bs, drop_last))
    mine = AspectRatioBatchSampler(S(order, a), a, bs, drop_last)
    assert list(mine) == want
    assert len(mine) == len(mod.AspectRatioBatchSampler(S(order, a), bs, drop_last))
    assert list(mine) == want            # the buckets reset between epochs


def test_cv2_style_resize_restatement_known_answers():
    """oracle/image_ops.py (the checker of the GPU resize kernel): properties that follow from OpenCV's INTER_LINEAR
    definition; the product's coefficient tables equal the oracle's."""
    from erd_amd.datasets import linear_coeffs
    from oracle import image_ops as I
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(I.resize_linear_u8(img, 53, 37), img)                      # same size: identity
    flat = np.full((20, 30, 3), 177, dtype=np.uint8)
    assert (I.resize_linear_u8(flat, 77, 41) == 177).all()                           # constants are preserved
    for src, dst in [(480, 800), (640, 1067), (5, 8), (1333, 97), (7, 7)]:
        o, c = I.linear_coeffs(src, dst)
        assert (c.astype(int).sum(1) == 2048).all() and o.min() >= 0 and o.max() <= src - 1
        o2, c2 = linear_coeffs(src, dst)
        assert np.array_equal(o, o2) and np.array_equal(c, c2)
    # exact 2x upscale of a 1-D ramp: centres fall at 1/4 and 3/4 between source pixels; borders are clamped
    ramp = (np.arange(8, dtype=np.uint8) * 16)[None, :, None].repeat(2, 0).repeat(3, 2)
    up = I.resize_linear_u8(ramp, 16, 2)[0, :, 0]
    assert up.tolist() == [0, 4, 12, 20, 28, 36, 44, 52, 60, 68, 76, 84, 92, 100, 108, 112]
    out, sf = I.resize_flip(img, (133, 80), flip=True)
    assert out.shape == (80, 115, 3) and sf == (115 / 53, 80 / 37)
    assert np.array_equal(out, I.resize_linear_u8(img, 115, 80)[:, ::-1])


def test_prefetch_map_order_lookahead_and_errors():
    """the decoding thread pool: results in item order, at most `depth` items ahead, exceptions at their item"""
    import threading
    import time
    from erd_amd.datasets import prefetch_map
    started, lock = [], threading.Lock()

    def work(i):
        with lock:
            started.append(i)
        time.sleep(0.02 * ((7 - i) % 3))          # later items may finish first
        if i == 5:
            raise ValueError("bad image 5")
        return i * i

    assert list(prefetch_map(work, range(5), workers=0, depth=2)) == [0, 1, 4, 9, 16]
    started.clear()
    got = []
    gen = prefetch_map(work, range(8), workers=3, depth=2)
    for v in gen:
        got.append(v)
        with lock:
            assert max(started) <= len(got) + 2          # never more than `depth` items beyond the consumer
        if len(got) == 5:
            break
    gen.close()
    assert got == [0, 1, 4, 9, 16]
    import pytest
    with pytest.raises(ValueError, match="bad image 5"):
        list(prefetch_map(work, range(8), workers=2, depth=3))
    assert list(prefetch_map(work, [], workers=2, depth=2)) == []
