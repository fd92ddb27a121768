"""View-sharded data parallelism for the training hot path; one process per GPU, `torch.distributed`
(backend "nccl" == RCCL over xGMI on MI355X, "gloo" in the CPU tests).

The reference is single-process, one camera per iteration (train.py:114-127).  Here every rank keeps a full replica of
the Gaussians, renders a DIFFERENT camera, and the per-Gaussian gradients are averaged with bucketed asynchronous
all-reduces that start while the backward is still running (the rasterizer's parameter gradients are final before the
shading backward has finished), so the collective overlaps the tail of the backward; the optimizer step waits for it.
Identical Adam steps + rank-identical densification keep the replicas bit-identical.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce is per-link bound, so buckets are large
(default 64 MiB; the whole stage-2 gradient is ~150 MB at 300k Gaussians => 3 collectives per step).
"""
import torch
import torch.distributed as dist


def shard_views(views, rank=None, world_size=None):
    """Cameras rank, rank+W, rank+2W, ... -- independent units, no data-path collective."""
    rank = dist.get_rank() if rank is None else rank
    world_size = dist.get_world_size() if world_size is None else world_size
    return list(views)[rank::world_size]


class GradAllReducer:
    """Bucketed, asynchronous gradient averaging.

    Gradients live as views into flat per-bucket buffers (no pack/unpack copies).  A post-accumulate-grad hook counts
    the parameters of a bucket that are final for this backward; when the last one lands the bucket's all-reduce is
    launched with async_op=True.  `finish()` (call it right before optimizer.step()) waits and divides by world size.
    Parameters are bucketed in REVERSE registration order, i.e. roughly the order autograd finishes them."""

    def __init__(self, params, bucket_bytes=64 << 20, process_group=None, average=True):
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.average = average
        self.buckets = []          # each: dict(flat, params, ready, handle)
        cur, cur_bytes = [], 0
        for p in reversed(self.params):
            nbytes = p.numel() * p.element_size()
            if cur and cur_bytes + nbytes > bucket_bytes:
                self._close(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self._close(cur)
        self._hooks = []
        for bi, b in enumerate(self.buckets):
            for p in b["params"]:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))

    def _close(self, plist):
        n = sum(p.numel() for p in plist)
        flat = torch.zeros(n, dtype=plist[0].dtype, device=plist[0].device)
        off = 0
        for p in plist:
            p.grad = flat[off:off + p.numel()].view_as(p)       # gradients accumulate straight into the bucket
            off += p.numel()
        self.buckets.append(dict(flat=flat, params=plist, ready=0, handle=None))

    def _make_hook(self, bi):
        def hook(_param):
            b = self.buckets[bi]
            b["ready"] += 1
            if b["ready"] == len(b["params"]) and self.world > 1:
                b["handle"] = dist.all_reduce(b["flat"], group=self.group, async_op=True)
        return hook

    def finish(self):
        for b in self.buckets:
            if self.world > 1:
                if b["handle"] is None:          # a parameter got no gradient this step: reduce now (zeros included)
                    b["handle"] = dist.all_reduce(b["flat"], group=self.group, async_op=True)
                b["handle"].wait()
                if self.average:
                    b["flat"].div_(self.world)
            b["handle"] = None
            b["ready"] = 0

    def zero_grad(self):
        for b in self.buckets:
            b["flat"].zero_()

    def remove(self):
        for h in self._hooks:
            h.remove()


def reduce_densification_stats(xyz_gradient_accum, normal_gradient_accum, denom, weights_accum, max_radii2D,
                               process_group=None):
    """One fused sum + one max all-reduce of the densification statistics (gaussian_model.py:931-937, train.py:164) so
    that densify/prune decisions are identical on every rank."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    tensors = [xyz_gradient_accum, normal_gradient_accum, denom, weights_accum]
    flat = torch.cat([t.reshape(-1).float() for t in tensors])
    dist.all_reduce(flat, group=process_group)
    off = 0
    for t in tensors:
        t.copy_(flat[off:off + t.numel()].view_as(t))
        off += t.numel()
    dist.all_reduce(max_radii2D, op=dist.ReduceOp.MAX, group=process_group)


def patch_model_step(model, reducer):
    """Wrap `model.step()` (GaussianModel.step gaussian_model.py:495-497 / DirectLightMap.step direct_light_map.py:25-27:
    optimizer.step(); optimizer.zero_grad()) so the averaged gradients are in place first -- train.py stays untouched."""
    inner = model.step

    def step(*a, **k):
        reducer.finish()
        model.optimizer.step()
        reducer.zero_grad()          # keep the bucket views; the reference's zero_grad() would detach them
        return None
    model.step = step
    model._r3dg_unpatched_step = inner
    return model
