"""GPU: the real trainer at WORLD SIZE 2 on the one GPU a test box has (VERDICT r3 "missing" item 1).
Two processes, both on cuda:0, process group `gloo` (ERD_DIST_BACKEND: RCCL refuses two ranks on one device; gloo collectives
are staged through host memory by erd_amd/dist_utils.py -- a correctness vehicle, not a performance path).  Each rank runs
`ERDTrainer` with every default on (three gradient-producing streams joined per bucket, teacher look-ahead, deferred update,
fused C2 + C3 all-reduce inside the loss, 1/world folded into the SGD kernel) on ITS OWN two images of 800x1344 for three
steps (default run: one image per rank at the full size, two steps).  Checked against the oracle's two-rank evaluation of the reference's data-parallel semantics:
  * C2 / C3: what the loss hands to `reduce_mean` and what comes back are the reference's two rank means
    (gfl_head_increment_erd.py:390-391, 406-407; dist_utils.py:59-65), the second one clamped to >= 1 after the mean;
  * the all-reduced gradient of step 0 is the mean of the two ranks' oracle gradients (D9: each rank's distillation terms are
    sums over its LOCAL images -- the gradient mean over ranks does not turn them into a mean over the global batch);
  * per-rank losses follow the oracle's two-rank SGD trajectory over three steps (tools/dist_train.sh:11-19 + DDP:
    every rank applies the same averaged gradient, lr = 0.01 x world x bs / 16);
  * both ranks hold bit-identical gradients after the all-reduce and bit-identical parameters at the end; no late bucket."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_real_trainer_world2_follows_the_two_rank_oracle(tmp_path):
    """Two cases in ONE pair of rank processes (a rank's start is 15-40 s of box time): 224 x 288 with 4 MB buckets (three steps), then
    the full size with the default 32 MB buckets (two steps; ERD_TEST_FULL=1: three -- the second step already runs on weights both
    ranks updated from the summed gradient, and the oracle's two-rank trajectory on the host is the test's time)."""
    import world2_worker as Wk
    WORLD = 2
    FULL = os.environ.get("ERD_TEST_FULL", "0") == "1"
    # (height, width, steps, bucket MB, images per rank): the full-size case with ONE image per rank in the default run (two with
    #  ERD_TEST_FULL=1) -- the oracle's two-rank trajectory on the host is 8 s per image and pass
    cases = [(224, 288, 3, 4, 2), (800, 1333, 3 if FULL else 2, 32, 2 if FULL else 1)]
    port = str(_free_port())
    env = dict(os.environ, ERD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "world2_worker.py"), str(r), str(WORLD), port, str(tmp_path)]
                              + ["%d,%d,%d,%d,%d" % c for c in cases], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(WORLD)]
    # ---- meanwhile, on the host: the oracle's two-rank trajectories ------------------------------------------------------------------
    # (the two cases side by side on the host's cores: torch's CPU operators release the GIL)
    from concurrent.futures import ThreadPoolExecutor
    threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, 32))        # (oneDNN oversubscribes on a 128-core host: bench.py's thread sweep)
    try:
        with ThreadPoolExecutor(len(cases)) as pool:
            refs = list(pool.map(lambda c: _oracle_two_rank_trajectory(Wk, WORLD, c[0], c[1], c[2], c[4]), cases))
    finally:
        torch.set_num_threads(threads)
    outs = []
    for p in procs:
        out, _ = p.communicate(timeout=1200)
        outs.append(out)
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d:\n%s" % (r, outs[r][-3000:])
    for i, ((H, W, STEPS, bucket_mb, _), ref) in enumerate(zip(cases, refs)):
        res = [torch.load(os.path.join(tmp_path, f"case{i}_rank{r}.pt"), weights_only=False) for r in range(WORLD)]
        _check_case(WORLD, H, W, STEPS, bucket_mb, res, *ref)
    # the CU reserve of the whole-chip grids (ERDTrainer.tune_cu_reserve, probed with the group live): one value on both ranks, set in
    # the library, one of the candidates
    rv = [torch.load(os.path.join(tmp_path, f"reserve_rank{r}.pt"), weights_only=False) for r in range(WORLD)]
    assert all(v["probed"] for v in rv) and rv[0]["cu_reserve"] == rv[1]["cu_reserve"] and rv[0]["cu_reserve"] in (0, 8), rv
    assert all(v["library_reserve"] == v["cu_reserve"] for v in rv), rv


def _oracle_two_rank_trajectory(Wk, WORLD, H, W, STEPS, BS):
    from e2e_util import f7_state_dicts
    from oracle import erd_oracle as O
    # ---- meanwhile, on the host: the oracle's two-rank trajectory ------------------------------------------------------------
    tsd, ssd = f7_state_dicts()
    names = [k for k, v in ssd.items() if O.trainable(k) and v.dtype == torch.float32]
    batches = [Wk.rank_batches(r, H, W, BS) for r in range(WORLD)]
    base_lr = Wk.LR * WORLD * BS / 16
    lrs = [base_lr * (Wk.WARM_START + (1 - Wk.WARM_START) * it / max(Wk.WARM - 1, 1) if it < Wk.WARM else 1.0) for it in range(STEPS)]
    if True:
        sd = {k: v.clone() for k, v in ssd.items()}
        bufs, ref_rows, ref_factors, ref_grad0 = {}, [], [], None
        for it in range(STEPS):
            local = []
            for r in range(WORLD):           # every rank's LOCAL normalisers (forward only)
                x, boxes, labels, metas = batches[r][it % 2]
                with torch.no_grad():
                    _, aux = O.erd_step_loss(tsd, sd, x, boxes, labels, metas, 40, 80, return_aux=True)
                local.append((aux["num_pos_local"], aux["weight_sum_local"]))
            # reduce_mean in the reference's arithmetic: fp32, divide by world, then sum (dist_utils.py:59-65)
            fac = sum(torch.tensor(l, dtype=torch.float32) / WORLD for l in local)
            ref_factors.append((local, fac))
            grads, rows = [], []
            for r in range(WORLD):
                x, boxes, labels, metas = batches[r][it % 2]
                leaf = {k: (sd[k].clone().requires_grad_(True) if k in names else sd[k]) for k in sd}
                losses = O.erd_step_loss(tsd, leaf, x, boxes, labels, metas, 40, 80, rank_mean_factors=(float(fac[0]), float(fac[1])))
                total = O.parse_losses(losses)
                total.backward()
                row = {k: float(sum(v.detach().mean() for v in vs)) for k, vs in losses.items()}
                row["loss"] = float(total.detach())
                rows.append(row)
                grads.append({k: leaf[k].grad for k in names})
                del leaf, losses, total
            mean = {k: sum(g[k] for g in grads) / WORLD for k in names}          # DDP: the gradient mean over ranks
            if it == 0:
                ref_grad0 = {k: v.clone() for k, v in mean.items()}
            ref_rows.append(rows)
            O.sgd_momentum_step({k: sd[k] for k in names}, mean, bufs, lrs[it], Wk.MOM, Wk.WD)
    return ssd, sd, names, lrs, ref_rows, ref_factors, ref_grad0


def _check_case(WORLD, H, W, STEPS, bucket_mb, res, ssd, sd, names, lrs, ref_rows, ref_factors, ref_grad0):
    for r, d in enumerate(res):
        assert d["backend"] == "gloo" and d["device"] == 0 and d["late_buckets"] == 0, (r, d["backend"], d["device"], d["late_buckets"], d["missing"][:12], d["repeats"][:12])
        print("rank %d: %d buckets, parameters that reported a gradient more than once in a step (deduplicated by the sync): %s"
              % (r, d["buckets"], d["repeats"][:6]))
        assert d["lrs"] == pytest.approx(lrs, rel=1e-12)
        assert d["buckets"] >= (4 if bucket_mb == 32 else 12)      # whole blocks per bucket
    # ---- C2 / C3 ---------------------------------------------------------------------------------------------------------------
    for it in range(STEPS):
        local, fac = ref_factors[it]
        sent = [res[r]["reduce_mean"][it][0] for r in range(WORLD)]
        back = [res[r]["reduce_mean"][it][1] for r in range(WORLD)]
        assert torch.equal(back[0], back[1])
        assert torch.equal(back[0], sum(s / WORLD for s in sent))                 # divide, then sum, in fp32: the reference's order
        # (the second entry is a sum of sigmoid(student logit) over a few dozen positives: at step 0 both sides evaluate the same
        #  weights; after an update the two trajectories' weights differ by their ~1e-3 gradient noise, and so do these logits)
        tol = 2e-4 if it == 0 else 3e-3
        for r in range(WORLD):
            assert float(sent[r][0]) == local[r][0]                                # sum of max(num_pos, 1) over the rank's images: an integer
            assert float(sent[r][1]) == pytest.approx(local[r][1], rel=tol), (it, r)   # sum of the positives' quality scores (student logits)
        assert float(back[0][0]) == float(fac[0]) and float(back[0][1]) == pytest.approx(float(fac[1]), rel=tol)
    # ---- both ranks hold the same gradient / parameters, bit for bit -------------------------------------------------------------
    for k in res[0]["grad0"]:
        assert torch.equal(res[0]["grad0"][k], res[1]["grad0"][k]), k
        assert torch.equal(res[0]["params"][k], res[1]["params"][k]), k
    # ---- the averaged gradient of step 0 against the mean of the two ranks' oracle gradients ---------------------------------------
    num = sum(float((res[0]["grad0"][k].double() - ref_grad0[k].double()).pow(2).sum()) for k in names)
    den = sum(float(ref_grad0[k].double().pow(2).sum()) for k in names)
    per = sorted(float((res[0]["grad0"][k] - ref_grad0[k]).norm() / ref_grad0[k].norm()) for k in names if float(ref_grad0[k].norm()) > 1e-12)
    whole = (num / den) ** 0.5
    print("world 2, %dx%d: all-reduced gradient of step 0 vs the mean of the two oracle gradients: whole %.2e, per-tensor median %.2e, worst %.2e"
          % (H, W, whole, per[len(per) // 2], per[-1]))
    # (two fp32 implementations of a ReLU network: test_gpu_parity_full.py has the fp64-anchored argument for these sizes of bound;
    # the small-image case sits on the coarser noise floor test_oracle_sensitivity.py documents)
    full = H >= 800
    assert whole < (2.5e-3 if full else 2e-2) and per[len(per) // 2] < (2.5e-3 if full else 2e-2), (whole, per[len(per) // 2])
    assert set(res[0]["grad0"]) == set(names)
    # ---- per-rank losses along the trajectory ----------------------------------------------------------------------------------------
    rel = lambda a, b: abs(a - b) / max(abs(b), 1e-7)
    for it in range(STEPS):
        for r in range(WORLD):
            got, want = res[r]["logs"][it], ref_rows[it][r]
            print("step %d rank %d: %s" % (it, r, "  ".join("%s %.1e" % (k, rel(got[k], v)) for k, v in want.items())))
            assert got["loss"] == pytest.approx(want["loss"], rel=1e-3 if full else 5e-3), (it, r, got["loss"], want["loss"])
            if it < 2:
                for k, v in want.items():
                    assert rel(got[k], v) <= (1e-3 if full else 5e-3), (it, r, k, got[k], v)
    # ---- parameter displacement after three averaged updates -------------------------------------------------------------------------
    n2 = sum(float((res[0]["params"][k].double() - ssd[k].double() - (sd[k].double() - ssd[k].double())).pow(2).sum()) for k in names)
    d2 = sum(float((sd[k].double() - ssd[k].double()).pow(2).sum()) for k in names)
    print("world 2: displacement after %d steps, rel L2 error vs the oracle's two-rank trajectory: %.2e" % (STEPS, (n2 / d2) ** 0.5))
    assert (n2 / d2) ** 0.5 < (1e-2 if full else 5e-2)
