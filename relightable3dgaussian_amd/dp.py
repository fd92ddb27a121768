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


def allreduce_optimizer_grads(optimizer, process_group=None, average=True, bucket_bytes=64 << 20):
    """Average (or sum) the gradients of every parameter an optimizer holds, in flat buckets of <= `bucket_bytes` launched
    asynchronously one after the other and waited for together.  Looks the parameters up at call time, so it keeps working
    when densification has replaced the optimizer's tensors (gaussian_model.py:718-750 cat_tensors_to_optimizer /
    _prune_optimizer).  Parameters without a gradient are skipped -- every rank runs the same graph on its own view, so the
    set is the same on all ranks (e.g. the iteration in which densify_and_prune rebuilt the parameters: train.py:166-176
    runs it BEFORE gaussians.step(), and the fresh tensors have no gradient yet)."""
    if not dist.is_initialized():
        return 0
    world = dist.get_world_size(process_group)
    if world == 1:
        return 0
    grads = [p.grad for g in optimizer.param_groups for p in g["params"] if p.grad is not None]
    buckets, cur, cur_bytes = [], [], 0
    for g in grads:
        nbytes = g.numel() * g.element_size()
        if cur and (cur_bytes + nbytes > bucket_bytes or g.dtype != cur[0].dtype):
            buckets.append(cur)
            cur, cur_bytes = [], 0
        cur.append(g)
        cur_bytes += nbytes
    if cur:
        buckets.append(cur)
    pending = []
    for b in buckets:
        flat = torch.cat([g.reshape(-1) for g in b])
        pending.append((b, flat, dist.all_reduce(flat, group=process_group, async_op=True)))
    for b, flat, handle in pending:
        handle.wait()
        if average:
            flat.div_(world)
        off = 0
        for g in b:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
    return len(buckets)


def patch_reference_classes(Scene, GaussianModel, DirectLightMap=None, rank=None, world=None, process_group=None):
    """data parallelism over camera views for the reference's UNMODIFIED train.py, applied to its classes from
    outside (tools/run_reference.py --dp N executes train.py afterwards; the file itself is not touched):
      * Scene.getTrainCameras (scene/__init__.py:103-104) returns cameras rank, rank + W, ... of the list every rank shuffled
        alike (random.seed(0), utils/general_utils.py:164; scene/__init__.py:81-83) -- train.py:114-119 then draws from ITS shard;
      * GaussianModel.step / DirectLightMap.step (gaussian_model.py:495-497, direct_light_map.py:25-27) average the gradients
        over the ranks in front of optimizer.step();
      * GaussianModel.add_densification_stats (gaussian_model.py:931-937) adds the SUM over the ranks of what this iteration's
        views contribute to xyz_gradient_accum, normal_gradient_accum, denom and weights_accum (each rank computes its increment
        from its own view's gradients, one fused all-reduce of the four increments); max_radii2D, which train.py:164-165 updates
        itself right behind that call, is max-reduced in front of its consumer densify_and_prune (:890-914) and in front of every
        step().  All five are therefore the same on every rank at every iteration boundary: clone / split / prune decisions are
        identical, torch.normal in densify_and_split (:816) draws the same numbers (same seed, same sequence of generator
        calls), and a checkpoint written by rank 0 holds the statistics of ALL views, as a single-process run's does.
    Returns the previous attributes (restore with `unpatch_reference_classes`)."""
    rank = dist.get_rank(process_group) if rank is None else rank
    world = dist.get_world_size(process_group) if world is None else world
    saved = dict(Scene=(Scene, "getTrainCameras", Scene.getTrainCameras), step=(GaussianModel, "step", GaussianModel.step),
                 densify=(GaussianModel, "densify_and_prune", GaussianModel.densify_and_prune),
                 stats=(GaussianModel, "add_densification_stats", GaussianModel.add_densification_stats))
    get_cameras, model_step, densify = Scene.getTrainCameras, GaussianModel.step, GaussianModel.densify_and_prune
    add_stats = GaussianModel.add_densification_stats
    live = dist.is_initialized() and dist.get_world_size(process_group) > 1
    SUMS = ("xyz_gradient_accum", "normal_gradient_accum", "denom", "weights_accum")

    def getTrainCameras(self, *a, **k):
        return shard_views(get_cameras(self, *a, **k), rank, world)

    def reduce_radii(self):
        if live and getattr(self, "_r3dg_radii_dirty", False) and self.max_radii2D.numel():
            dist.all_reduce(self.max_radii2D, op=dist.ReduceOp.MAX, group=process_group)
        self._r3dg_radii_dirty = False

    def add_densification_stats(self, *a, **k):
        if not live:
            return add_stats(self, *a, **k)
        before = [getattr(self, n).clone() for n in SUMS]
        out = add_stats(self, *a, **k)
        delta = torch.cat([(getattr(self, n) - b).reshape(-1) for n, b in zip(SUMS, before)])
        dist.all_reduce(delta, group=process_group)
        off = 0
        for n, b in zip(SUMS, before):
            t = getattr(self, n)
            t.copy_(b + delta[off:off + t.numel()].view_as(t))
            off += t.numel()
        self._r3dg_radii_dirty = True              # train.py:164-165 touches max_radii2D right behind this call
        return out

    def step(self):
        reduce_radii(self)
        allreduce_optimizer_grads(self.optimizer, process_group)
        return model_step(self)

    def densify_and_prune(self, *a, **k):
        reduce_radii(self)
        return densify(self, *a, **k)
    Scene.getTrainCameras, GaussianModel.step, GaussianModel.densify_and_prune = getTrainCameras, step, densify_and_prune
    GaussianModel.add_densification_stats = add_densification_stats
    if DirectLightMap is not None:
        light_step = DirectLightMap.step
        saved["light"] = (DirectLightMap, "step", light_step)

        def step_light(self):
            allreduce_optimizer_grads(self.optimizer, process_group)
            return light_step(self)
        DirectLightMap.step = step_light
    return saved


def unpatch_reference_classes(saved):
    for cls, name, fn in saved.values():
        setattr(cls, name, fn)
