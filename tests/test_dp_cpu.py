"""Multi-process CPU tests (gloo, world_size 2) of the view-sharded data-parallel path:
replicas stay bit-identical and equal a single-process run that averages both views' gradients each step."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class TinyGaussians:
    """Duck-types the part of the reference's GaussianModel the DP hooks touch: parameter groups, Adam with
    eps=1e-15, step() = optimizer.step(); optimizer.zero_grad() (gaussian_model.py:465-497)."""

    def __init__(self, seed=0, P=257):
        g = torch.Generator().manual_seed(seed)
        self.xyz = torch.nn.Parameter(torch.randn(P, 3, generator=g))
        self.sh = torch.nn.Parameter(torch.randn(P, 16, 3, generator=g))
        self.opacity = torch.nn.Parameter(torch.randn(P, 1, generator=g))
        self.unused = torch.nn.Parameter(torch.randn(P, 2, generator=g))     # gets no gradient: must not deadlock
        self.optimizer = torch.optim.Adam([{"params": [self.xyz], "lr": 1e-2}, {"params": [self.sh], "lr": 2e-3},
                                           {"params": [self.opacity], "lr": 5e-2}, {"params": [self.unused], "lr": 1e-2}],
                                          lr=0.0, eps=1e-15)

    def parameters(self):
        return [self.xyz, self.sh, self.opacity, self.unused]

    def step(self):
        self.optimizer.step()
        self.optimizer.zero_grad()


def _loss(model, view):
    g = torch.Generator().manual_seed(1000 + view)
    w = torch.randn(257, 3, generator=g)
    return ((model.xyz * w).sum(-1).tanh() * torch.sigmoid(model.opacity[:, 0])).sum() + \
        (model.sh.mean(1) * w).square().sum() * 0.1


def _worker(rank, world, port, steps, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from relightable3dgaussian_amd import dp
    model = TinyGaussians()
    red = dp.GradAllReducer(model.parameters(), bucket_bytes=20_000)      # several buckets
    assert len(red.buckets) >= 2
    dp.patch_model_step(model, red)
    views = dp.shard_views(list(range(2 * steps)), rank, world)
    for i in range(steps):
        _loss(model, views[i]).backward()
        model.step()
    stats = [torch.full((5,), float(rank + 1)) for _ in range(4)] + [torch.tensor([rank * 3.0, 7.0 - rank])]
    dp.reduce_densification_stats(*stats)
    assert torch.equal(stats[0], torch.full((5,), 3.0)) and torch.equal(stats[4], torch.tensor([3.0, 7.0]))
    torch.save([p.detach().clone() for p in model.parameters()], os.path.join(out_dir, "rank%d.pt" % rank))
    dist.destroy_process_group()


def test_two_rank_replicas_identical_and_match_averaged_single_process(tmp_path):
    steps, world = 6, 2
    mp.spawn(_worker, args=(world, _free_port(), steps, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    for a, b in zip(r0, r1):
        assert torch.equal(a, b), "replicas diverged"
    # single process: average the two views' gradients by hand
    model = TinyGaussians()
    for i in range(steps):
        (0.5 * (_loss(model, 2 * i) + _loss(model, 2 * i + 1))).backward()
        model.step()
    for a, b in zip(r0, model.parameters()):
        assert torch.allclose(a, b.detach(), rtol=1e-5, atol=1e-6)


def test_shard_views_partition():
    from relightable3dgaussian_amd import dp
    views = list(range(10))
    parts = [dp.shard_views(views, r, 4) for r in range(4)]
    assert sorted(sum(parts, [])) == views and all(len(p) in (2, 3) for p in parts)
