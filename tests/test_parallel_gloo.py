"""CPU, world_size 2, gloo: the ray-sharded data-parallel path (a-nerf_amd/parallel.py).

The per-rank compute is stood in by the CPU oracle (tests may use it as a checker): rank r renders its
contiguous ray shard, takes the mean loss over the shard, and GradBucket.all_reduce_mean must reproduce the
gradient of the single-process global-batch mean loss -- the DP contract of SURVEY.md 8(e)."""
import importlib
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    parallel = importlib.import_module("a-nerf_amd.parallel")
    oracle = importlib.import_module("anerf_oracle")
    from cases import build
    c = build("eval_s32")
    n = 32                                    # even split over 2 ranks
    t = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32)
    P = oracle.params_from_numpy(c["Pc"], requires_grad=True)
    names = sorted(P)
    cfg = oracle.OracleConfig()
    rb = oracle.make_ray_batch(t(c["rays_o"][:n]), t(c["rays_d"][:n]))
    target = t(np.random.default_rng(5).random((n, 3)))

    def loss_on(sl):
        out = oracle.render_rays(cfg, P, None, rb[sl], t(c["skts"][:n][sl]), t(c["cyls"][:n][sl]), 16)
        return oracle.nerf_loss(out, target[sl], torch.ones(1, 3))[0]

    lo, hi = parallel.shard_rays(n, rank, world)
    assert (lo, hi) == (rank * 16, rank * 16 + 16)
    loss_on(slice(lo, hi)).backward()
    bucket = parallel.GradBucket([P[k] for k in names])
    flat = bucket.all_reduce_mean()
    got = {k: P[k].grad.clone() for k in names}
    assert flat.numel() == sum(P[k].numel() for k in names)
    # single-process reference on the full batch
    for k in names:
        P[k].grad = None
    loss_on(slice(0, n)).backward()
    err = max(float((got[k] - P[k].grad).abs().max() / (P[k].grad.abs().max() + 1e-12)) for k in names)
    # ragged split (n = 31 -> shards of 16 and 15): weighting each rank's mean-loss gradient by shard_weight() gives the
    # global-batch mean gradient again; the plain 1/world average does not
    n2 = 31
    for k in names:
        P[k].grad = None
    l2, h2 = parallel.shard_rays(n2, rank, world)
    loss_on(slice(l2, h2)).backward()
    w = parallel.shard_weight(n2, rank, world)
    assert abs(w - (h2 - l2) * world / n2) < 1e-12
    parallel.GradBucket([P[k] for k in names]).all_reduce_mean(weight=w)
    got2 = {k: P[k].grad.clone() for k in names}
    for k in names:
        P[k].grad = None
    loss_on(slice(0, n2)).backward()
    err = max(err, max(float((got2[k] - P[k].grad).abs().max() / (P[k].grad.abs().max() + 1e-12)) for k in names))
    # frame assembly: every rank ends up with all rows in order
    local = torch.arange(lo, hi, dtype=torch.float32)[:, None].repeat(1, 3)
    full = parallel.gather_rays(local, n)
    ok_gather = bool(torch.equal(full[:, 0], torch.arange(n, dtype=torch.float32)))
    # ragged split (n = 31): last rank is short
    lo2, hi2 = parallel.shard_rays(31, rank, world)
    full2 = parallel.gather_rays(torch.arange(lo2, hi2, dtype=torch.float32)[:, None], 31)
    ok_gather &= bool(torch.equal(full2[:, 0], torch.arange(31, dtype=torch.float32)))
    q.put((rank, err, ok_gather))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_gradient_equals_global_batch_gradient():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, ok in res:
        assert err < 1e-4, (rank, err)
        assert ok


def test_bucket_single_process_is_identity():
    parallel = importlib.import_module("a-nerf_amd.parallel")
    a, b = torch.nn.Parameter(torch.randn(3, 4)), torch.nn.Parameter(torch.randn(5))
    a.grad, b.grad = torch.randn(3, 4), torch.randn(5)
    ga, gb = a.grad.clone(), b.grad.clone()
    assert parallel.GradBucket([a, b]).all_reduce_mean() is None
    assert torch.equal(a.grad, ga) and torch.equal(b.grad, gb)
    assert parallel.shard_rays(10, 1, 4) == (3, 6) and parallel.shard_rays(10, 3, 4) == (9, 10)
