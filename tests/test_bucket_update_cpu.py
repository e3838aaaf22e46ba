"""CPU: the per-bucket update protocol of engine.BucketedGradSync (no process group: `reduce=False`, as ERDTrainer runs it at one
rank).  A bucket is handed to `on_bucket` exactly once per step, in bucket order, never by a report from one of its own
parameters (the launches that still read the bucket's weights belong to the block that reports), the last one by wait();
buckets hold whole groups; the per-bucket SGD equals the one-launch update bit for bit (it is elementwise)."""
import torch

from erd_amd.engine import BucketedGradSync, FlatParams


def _net():
    torch.manual_seed(0)
    convs = [torch.nn.Conv2d(4, 8, 3, padding=1)] + [torch.nn.Conv2d(8, 8, 3, padding=1) for _ in range(5)] + [torch.nn.Conv2d(8, 2, 1)]
    layers = []
    for c in convs[:-1]:
        layers += [c, torch.nn.ReLU()]
    return torch.nn.Sequential(*layers, convs[-1])


def test_buckets_hold_whole_groups_and_a_small_tail():
    net = _net()
    named = list(net.named_parameters()); named.reverse()
    key = lambda n: n.split(".")[0]                   # weight + bias of one convolution stay together
    flat = FlatParams(named, "cpu", bucket_bytes=5000, tail_bytes=2000, group_key=key)
    assert len(flat.buckets) >= 3 and flat.buckets[0][0] == 0 and flat.buckets[-1][1] == flat.total
    for (s, e, mem), nxt in zip(flat.buckets, flat.buckets[1:] + [None]):
        assert nxt is None or e == nxt[0]
        keys = [key(flat.names[i]) for i in mem]
        for k in set(keys):                           # every parameter of a group is in this bucket
            assert sum(1 for n in flat.names if key(n) == k) == keys.count(k)
    assert (flat.buckets[-1][1] - flat.buckets[-1][0]) * 4 <= 2000        # the tail: the first layer alone
    assert (flat.buckets[-2][1] - flat.buckets[-2][0]) * 4 > 2000


def test_per_bucket_update_protocol_and_result():
    lr, mom, wd = 0.1, 0.9, 1e-4
    x = torch.randn(2, 4, 6, 6, generator=torch.Generator().manual_seed(5))

    def run(per_bucket: bool):
        net = _net()
        named = list(net.named_parameters()); named.reverse()
        flat = FlatParams(named, "cpu", bucket_bytes=5000, tail_bytes=2000, group_key=lambda n: n.split(".")[0])
        released = []
        reporting = [None]

        def sgd(s, e):
            g = flat.grad[s:e] + wd * flat.data[s:e]
            flat.momentum[s:e].mul_(mom).add_(g)
            flat.data[s:e].sub_(lr * flat.momentum[s:e])

        def on_bucket(b):
            released.append((b, reporting[0]))
            sgd(*flat.buckets[b][:2])

        sync = BucketedGradSync(flat, reduce=False, on_bucket=on_bucket if per_bucket else None)
        for i, p in enumerate(flat.params):           # note which bucket the reporting parameter belongs to
            inner = p._erd_sink_notify                 # (the hook BucketedGradSync registered for this parameter)

            def outer(q, i=i, inner=inner):
                reporting[0] = flat.bucket_of[i]
                inner(q)
            p._post_accumulate_grad_hooks.clear()
            p.register_post_accumulate_grad_hook(outer)
        for step in range(3):
            flat.zero_grad()
            sync.arm()
            released.clear()
            net(x).square().mean().backward()
            in_backward = list(released)
            reporting[0] = None
            sync.wait()
            if per_bucket:
                nb = len(flat.buckets)
                assert [b for b, _ in released] == list(range(nb))                    # once each, in order
                assert all(who is not None and who != b for b, who in in_backward)    # never by one of its own parameters
                assert released[-1] == (nb - 1, None) and len(in_backward) == nb - 1  # the tail by wait(), the rest inside backward
                assert sync.issued_in_backward == nb - 1 and sync.late_buckets == 0
            else:
                sgd(0, flat.total)
        return flat.data.clone(), flat.momentum.clone()

    a, b = run(True), run(False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
