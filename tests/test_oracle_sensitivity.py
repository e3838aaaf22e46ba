"""CPU: documents the conditioning of the step's gradients.  The oracle (== the reference, see
test_oracle_golden.py) run twice with a 2e-6 relative input perturbation: losses move by ~1e-6, but
individual weight-gradient tensors may jump by >1e-3 when a single ReLU mask flips at this tiny image
size.  This is why the end-to-end GPU test uses median / norm criteria for gradients (test_gpu_e2e.py)
while each fused op's backward is held to a tight tolerance on its own (test_gpu_functions.py)."""
import numpy as np
import pytest
import torch

from e2e_util import f7_state_dicts
from oracle import erd_oracle as O


@pytest.mark.parametrize("c_old,depth", [(40, 50), (70, 101)])
def test_reference_gradients_are_not_elementwise_stable_at_1e3(c_old, depth):
    torch.manual_seed(0)
    tsd, ssd = f7_state_dicts(c_old, 80, depth)
    imgs, boxes, labels = O.synthetic_batch(2, 123, 153, 80 - c_old, seed=0)
    x, metas = O.preprocess(imgs)

    def run(xin):
        sd = {k: (v.clone().requires_grad_(True) if O.trainable(k) and v.dtype == torch.float32 else v)
              for k, v in ssd.items()}
        losses = O.erd_step_loss(tsd, sd, xin, boxes, labels, metas, c_old, 80, depth=depth)
        total = O.parse_losses(losses)
        total.backward()
        return float(total), {k: v.grad for k, v in sd.items() if isinstance(v, torch.Tensor) and v.requires_grad}

    l0, g0 = run(x)
    l1, g1 = run(x * (1 + 2e-6 * torch.randn_like(x)))
    assert abs(l0 - l1) / abs(l0) < 1e-4
    errs = [float((g0[k] - g1[k]).norm() / g0[k].norm()) for k in g0 if float(g0[k].norm()) > 1e-12]
    # the loss is stable; the gradients are only stable in a norm sense (ReLU mask flips):
    nerr = [abs(float(g0[k].double().norm()) - float(g1[k].double().norm())) / float(g0[k].double().norm())
            for k in g0 if float(g0[k].norm()) > 1e-12]
    print("R%d %d+%d: per-tensor rel L2 change under a 2e-6 input perturbation: median %.2e max %.2e; grad-norm change "
          "median %.2e max %.2e" % (depth, c_old, 80 - c_old, float(np.median(errs)), max(errs), float(np.median(nerr)),
                                    max(nerr)))
    assert max(errs) < 0.2
