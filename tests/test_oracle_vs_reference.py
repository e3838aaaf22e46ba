"""CPU, build container only: the oracle restatement against the REAL reference source executed live through
oracle/ref_stub.py (skipped where /root/reference does not exist, e.g. on the GPU box -- there the committed
fixtures of test_oracle_golden.py are the pin)."""
import pytest
import torch

from oracle import erd_oracle as O
from oracle import ref_stub

pytestmark = pytest.mark.skipif(not ref_stub.available(), reason="reference tree not present")


def test_full_step_losses_and_grads_match_reference_source():
    from e2e_util import f7_state_dicts
    ref = ref_stub.load_reference()
    teacher, student = ref_stub.build_reference_erd()
    tsd, ssd = f7_state_dicts()
    teacher.load_state_dict(tsd, strict=True)
    student.load_state_dict(ssd, strict=True)
    ref_stub.attach_teacher(student, teacher, 40)
    student.train()
    imgs, boxes, labels = O.synthetic_batch(2, 150, 187, 40, seed=11)       # not the fixture's inputs
    x, metas = O.preprocess(imgs)
    samples = []
    for i in range(2):
        ds = ref.DetDataSample(metainfo=metas[i])
        ds.gt_instances = ref.InstanceData(bboxes=boxes[i], labels=labels[i])
        samples.append(ds)
    ref_losses = student.loss(x, samples)
    sd = {k: (v.clone().requires_grad_(True) if O.trainable(k) and v.dtype == torch.float32 else v)
          for k, v in ssd.items()}
    losses = O.erd_step_loss(tsd, sd, x, boxes, labels, metas, 40, 80)
    for k in ref_losses:
        a = torch.stack([v.detach() for v in ref_losses[k]])
        b = torch.stack([v.detach() for v in losses[k]])
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-8), k
    O.parse_losses(ref_losses).backward()
    O.parse_losses(losses).backward()
    for k, p in student.named_parameters():
        if k.startswith("ori_model.") or p.grad is None:
            continue
        assert float((p.grad - sd[k].grad).abs().max()) <= 1e-4 * float(p.grad.abs().max()) + 1e-10, k
    # the trainable set is the reference's (frozen_stages=1; BN gamma/beta of layers 2-4 train)
    assert {k for k, p in student.named_parameters() if p.requires_grad and not k.startswith("ori_model.")} == \
        {k for k in sd if O.trainable(k)}
