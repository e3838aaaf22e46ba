"""CPU, world_size 2 over gloo: the data-parallel pieces that do not need a GPU -- bucket layout in backward
order, hook-driven per-bucket all-reduce on the flat gradient buffer, and the reduce_mean convention of the
two loss normalisers (dist_utils.py:59-65)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from erd_amd.engine import BucketedGradSync, FlatParams
        torch.manual_seed(0)                                   # same init on every rank
        net = torch.nn.Sequential(torch.nn.Conv2d(4, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 8, 3, padding=1),
                                  torch.nn.ReLU(), torch.nn.Conv2d(8, 2, 1))
        named = list(net.named_parameters()); named.reverse()
        flat = FlatParams(named, "cpu", bucket_bytes=1024)     # tiny buckets -> several all-reduces
        assert len(flat.buckets) >= 2 and flat.buckets[0][0] == 0 and flat.buckets[-1][1] == flat.total
        assert all(p.data_ptr() >= flat.data.data_ptr() for p in flat.params)           # params live in the flat buffer
        sync = BucketedGradSync(flat)
        g = torch.Generator().manual_seed(100 + rank)          # different data per rank
        x = torch.randn(2, 4, 6, 6, generator=g)
        flat.zero_grad(); sync.arm()
        net(x).square().mean().backward()
        sync.wait()
        mine = flat.grad.clone() / world
        # reference: gather every rank's local gradient and average
        flat.zero_grad()
        net(x).square().mean().backward()
        loc = [torch.zeros_like(flat.grad) for _ in range(world)]
        dist.all_gather(loc, flat.grad.clone())
        ref = sum(loc) / world
        ok = torch.allclose(mine, ref, rtol=1e-6, atol=1e-8)
        # reduce_mean of the two loss normalisers -- the product's own function (erd_amd/dist_utils.py), as the ERD
        # loss calls it on its 2-float vector [sum num_pos, sum weight_targets]
        from erd_amd.dist_utils import reduce_mean, world_size
        local = torch.tensor([3.0 + rank, 10.0 * (rank + 1)])
        avg = reduce_mean(local)
        ok = ok and world_size() == world and torch.allclose(avg, torch.tensor([3.5, 15.0]))
        ok = ok and torch.equal(local, torch.tensor([3.0 + rank, 10.0 * (rank + 1)]))       # the input is not modified
        # a parameter without a gradient on ONE rank only (rank 1 leaves the last layer -- the FIRST bucket in backward
        # order -- out of its loss): that rank completes the bucket late.  Collectives must still be enqueued in the same
        # order on both ranks (mismatched order = wrong buffers paired, or a hang), the result is the plain average, and
        # the late rank says so
        import warnings
        flat.zero_grad(); sync.arm()
        with warnings.catch_warnings(record=True) as wrn:
            warnings.simplefilter("always")
            h = net[3](net[2](net[1](net[0](x))))
            (net[4](h).square().mean() if rank == 0 else h.square().mean()).backward()
            sync.wait()
        mine = flat.grad.clone() / world
        flat.zero_grad()
        h = net[3](net[2](net[1](net[0](x))))
        (net[4](h).square().mean() if rank == 0 else h.square().mean()).backward()
        loc = [torch.zeros_like(flat.grad) for _ in range(world)]
        dist.all_gather(loc, flat.grad.clone())
        ok = ok and torch.allclose(mine, sum(loc) / world, rtol=1e-6, atol=1e-8)
        if rank == 1:
            ok = ok and sync.late_buckets >= 1 and any("issued late" in str(m.message) for m in wrn)
        else:
            ok = ok and sync.late_buckets == 0
        # the CU reserve of the whole-chip grids (ERDTrainer.tune_cu_reserve): every rank times the candidates itself, all ranks must
        # adopt the SAME one -- the candidate whose SLOWEST rank was fastest, not each rank's own favourite
        from erd_amd.dist_utils import agree_on_fastest
        cases = [([1.0, 0.9, 1.2], [1.0, 1.1, 0.8], 0),      # rank 0 alone would pick 1, rank 1 alone 2: max over ranks says 0
                 ([1.3, 1.0, 1.1], [1.2, 1.05, 1.0], 1),
                 ([1.0, 1.0, 1.0], [1.0, 1.0, 1.0], 0)]      # a tie keeps the first candidate (reserve 0)
        for t0, t1, want in cases:
            pick = agree_on_fastest(t0 if rank == 0 else t1)
            both = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(both, torch.tensor([pick]))
            ok = ok and pick == want and all(int(b) == want for b in both)
        q.put((rank, bool(ok), len(flat.buckets)))
    finally:
        dist.destroy_process_group()


def test_bucketed_grad_sync_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res


def _worker_update(rank, world, port, q):
    """per-bucket update at world size 2: all-reduce -> on_bucket per bucket, released by the first report from another bucket;
    rank 1 leaves the last layer out of its loss in the second step (its first bucket stays open until wait())"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from erd_amd.engine import BucketedGradSync, FlatParams
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Conv2d(4, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 8, 3, padding=1),
                                  torch.nn.ReLU(), torch.nn.Conv2d(8, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 2, 1))
        named = list(net.named_parameters()); named.reverse()
        flat = FlatParams(named, "cpu", bucket_bytes=2000, tail_bytes=1500, group_key=lambda n: n.split(".")[0])
        lr = 0.05
        order = []

        def on_bucket(b):                               # plain SGD on the bucket's slice with the MEAN gradient
            s, e, _ = flat.buckets[b]
            order.append(b)
            flat.data[s:e].sub_(lr * flat.grad[s:e] / world)

        sync = BucketedGradSync(flat, on_bucket=on_bucket)
        x = torch.randn(2, 4, 6, 6, generator=torch.Generator().manual_seed(100 + rank))
        ok = len(flat.buckets) >= 3
        for step in range(2):
            before = flat.data.clone()
            # reference: every rank's local gradient of the CURRENT parameters, gathered and averaged
            flat.zero_grad()
            loss = lambda: (net(x) if (step == 0 or rank == 0) else net[:6](x)).square().mean()
            loss().backward()
            loc = [torch.zeros_like(flat.grad) for _ in range(world)]
            dist.all_gather(loc, flat.grad.clone())
            want = before - lr * sum(loc) / world
            flat.zero_grad(); sync.arm(); order.clear()
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                loss().backward()
                sync.wait()
            ok = ok and order == list(range(len(flat.buckets)))
            ok = ok and torch.allclose(flat.data, want, rtol=1e-6, atol=1e-8)
            ok = ok and sync.late_buckets == (1 if (step == 1 and rank == 1) else 0)
        both = [torch.zeros_like(flat.data) for _ in range(world)]
        dist.all_gather(both, flat.data.clone())
        ok = ok and torch.equal(both[0], both[1])       # the ranks stay bit-identical
        q.put((rank, bool(ok), len(flat.buckets)))
    finally:
        dist.destroy_process_group()


def test_per_bucket_update_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_update, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
