"""CPU: the product's synthetic weight / input spec (erd_amd/synthetic.py, what bench.py feeds the HIP path) is the same
function as the oracle's own generators -- so GPU-vs-oracle comparisons and the benchmark run on identical numbers --
and it derives its shapes from the model, not from a table."""
import numpy as np
import torch

from erd_amd import synthetic as S
from oracle import erd_oracle as O


def test_procedural_weights_match_the_oracle_generator_r50_and_r101():
    for depth, nc in ((50, 40), (101, 70)):
        want = O.procedural_state_dict(nc, depth=depth, seed=3)
        got = S.procedural_state_dict({k: tuple(v.shape) for k, v in want.items()}, seed=3)
        assert list(got) == list(want)
        for k in want:
            assert got[k].dtype == want[k].dtype and torch.equal(got[k], want[k]), k


def test_shapes_come_from_the_model():
    import os
    from erd_amd import Config, MODELS
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, "configs", "gfl_increment", "gfl_r50_fpn_1x_coco_first_40_cats.py"))
    shapes = S.state_shapes(MODELS.build(cfg.model))
    assert shapes == {k: tuple(v) for k, v in O.gfl_param_shapes(40, 50).items()}


def test_demo_batch_matches_the_oracle_generator():
    a = S.demo_batch(2, 37, 53, 40, seed=7)
    b = O.synthetic_batch(2, 37, 53, 40, seed=7)
    for xs, ys in zip(a, b):
        for x, y in zip(xs, ys):
            assert x.dtype == y.dtype and torch.equal(x, y)
    rng1, rng2 = np.random.RandomState(1), np.random.RandomState(1)
    assert np.array_equal(S.rand_bboxes(rng1, 5, 100, 80), O.rand_bboxes(rng2, 5, 100, 80))
