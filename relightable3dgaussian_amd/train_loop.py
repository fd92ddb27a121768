"""Stage-1 training loop around the fused iteration: the densification schedule of the reference's train.py:158-175
(statistics every iteration, densify_and_prune every `densification_interval` after `densify_from_iter`, size threshold
once past the first opacity reset, reset_opacity every `opacity_reset_interval`) and its from-scratch initialisation
(`GaussianModel.create_from_pcd`, scene/gaussian_model.py:409-441: isotropic log-scale from the mean squared 3-NN
distance via distCUDA2, identity rotations, opacity 0.1, SH dc from the point colours).

Only what the hot path needs: no dataset readers, logging, checkpoints or evaluation -- cameras and ground-truth images
are whatever the caller hands in (synthetic.py in the tests and tools).  Data parallel: every rank runs this loop on its
own view shard; gradients and densification statistics are reduced inside FusedStage1Step, and the split's random
table comes from a generator seeded identically on all ranks, so the replicas stay identical.
"""
import math
import types

import torch

from .fused_step import FusedStage1Step
from .knn_ops import distCUDA2

C0 = 0.28209479177387814          # RGB2SH (utils/sh_utils.py:130-131)


def position_lr(step, lr_init, lr_final, lr_delay_mult=0.01, max_steps=30_000, lr_delay_steps=0):
    """The xyz learning rate of iteration `step` (get_expon_lr_func, utils/general_utils.py:30-63, as
    GaussianModel.training_setup instantiates it, scene/gaussian_model.py:488-491): log-linear interpolation from lr_init
    to lr_final over max_steps, optionally eased in over lr_delay_steps."""
    if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
        return 0.0
    if lr_delay_steps > 0:
        delay_rate = lr_delay_mult + (1 - lr_delay_mult) * math.sin(0.5 * math.pi * min(max(step / lr_delay_steps, 0), 1))
    else:
        delay_rate = 1.0
    t = min(max(step / max_steps, 0), 1)
    return delay_rate * math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)


class Schedule(types.SimpleNamespace):
    """The OptimizationParams fields the loop reads (arguments/__init__.py:70-106), same defaults."""

    def __init__(self, **kw):
        super().__init__(iterations=30_000, position_lr_init=0.00016, position_lr_final=0.0000016,
                         position_lr_delay_mult=0.01, position_lr_max_steps=30_000, normal_lr=0.01, sh_lr=0.0025,
                         opacity_lr=0.05,
                         scaling_lr=0.005, rotation_lr=0.001, percent_dense=0.001, densification_interval=100,
                         opacity_reset_interval=3000, densify_from_iter=500, densify_until_iter=10_000,
                         densify_grad_threshold=0.0002, densify_grad_normal_threshold=2e-9, normal_densify_from_iter=0,
                         min_opacity=0.005)
        self.__dict__.update(kw)


def create_from_points(points, colors, normals=None, device="cuda"):
    """Raw parameters of a fresh model (create_from_pcd): -> namespace with xyz, normal, scaling, rotation, opacity,
    features_dc [P,1,3], features_rest [P,15,3]."""
    xyz = points.to(device=device, dtype=torch.float32).contiguous()
    P = xyz.shape[0]
    dist2 = distCUDA2(xyz).clamp_min(1e-7)                                  # mean squared distance to the 3 NN
    scaling = torch.log(torch.sqrt(dist2))[:, None].repeat(1, 3).contiguous()
    rotation = torch.zeros(P, 4, device=device)
    rotation[:, 0] = 1
    opacity = torch.full((P, 1), math.log(0.1 / 0.9), device=device)        # inverse_sigmoid(0.1)
    dc = ((colors.to(device=device, dtype=torch.float32) - 0.5) / C0)[:, None, :].contiguous()
    rest = torch.zeros(P, 15, 3, device=device)
    if normals is None:
        normals = torch.zeros(P, 3, device=device)
        normals[:, 2] = 1
    return types.SimpleNamespace(xyz=xyz, normal=normals.to(device=device, dtype=torch.float32).contiguous(),
                                 scaling=scaling, rotation=rotation, opacity=opacity, features_dc=dc,
                                 features_rest=rest)


def train_stage1(init, cameras, images, background, extent, schedule=None, iterations=None, seed=0,
                 white_background=True, process_group=None, on_iteration=None, masks=None, loss_weights=None,
                 poll_interval=32, replay_dropped=True):
    """Runs `iterations` fused stage-1 iterations over the (camera, image) pairs in round-robin order (the reference
    draws a random permutation, train.py:115-119; the order is the caller's) with the reference's densification
    schedule.  Returns (FusedStage1Step, history) where history lists (iteration, event, rows) for every
    densify / reset.  The objective is the reference's stage-1 loss with the lambdas of script/run_nerf.sh:7-14
    (train_step.STAGE1_WEIGHTS; `loss_weights` overrides them); `masks[v]` [1,H,W] is view v's object mask
    (Camera.image_mask; None = all ones).
    The fused iteration's bounded forward drops a view on the device when it needs more tile instances than the capacity
    learned so far (fused_step._BoundedForward): the loop asks every `poll_interval` iterations and before every densify
    (`poll_overflow`: one 4-byte read-back, grows the capacity, takes the step back from Adam's count) and lists what was
    dropped as (iteration, "dropped_views", n) in the history.  The reference trains on every view (it sizes the binning state
    from the count it reads back, rasterizer_impl.cu:291): with `replay_dropped` (default; single GPU) the dropped views are
    trained on right there, through the exact two-phase forward (`FusedStage1Step.replay_dropped`), and listed as
    (iteration, "replayed_views", [the iterations they were dropped in]) -- at most `poll_interval` iterations late."""
    sch = schedule or Schedule()
    n_iter = sch.iterations if iterations is None else iterations
    step = FusedStage1Step(init, lr=sch.sh_lr, lr_rest_scale=1.0 / 20.0, process_group=process_group,
                           lrs=dict(xyz=sch.position_lr_init * extent, normal=sch.normal_lr, scaling=sch.scaling_lr,
                                    rotation=sch.rotation_lr, opacity=sch.opacity_lr, shs=sch.sh_lr),
                           loss_weights=loss_weights)
    step.enable_densification()
    gen = torch.Generator(device=step.dev).manual_seed(seed)
    history = []
    xyz_group = step.opt.groups[step._opt_order.index("xyz")]
    last_densify = None
    view_of = {}                                       # forward pass of the step object -> view index (for replays)
    for it in range(1, n_iter + 1):
        # gaussians.update_learning_rate(iteration) (train.py:101): the position rate decays log-linearly
        xyz_group["lr"] = position_lr(it, sch.position_lr_init * extent, sch.position_lr_final * extent,
                                      sch.position_lr_delay_mult, sch.position_lr_max_steps)
        v = (it - 1) % len(cameras)
        collecting = it < sch.densify_until_iter
        if not collecting and step.stats is not None:
            step.stats = None                                                # train.py:160: statistics only while densifying
        step.iteration = it                                                  # depth-variance schedule, render.py:202
        view_of[step._iter + 1] = v                                          # (the step object counts its forward passes)
        step.forward_backward(cameras[v], background, images[v], None if masks is None else masks[v])
        densify_now = collecting and it > sch.densify_from_iter and it % sch.densification_interval == 0
        replay_now = False
        # (also on the two iterations behind a densification: P has just grown, and so has the instance count the bounded
        # forward's capacity was sized for -- a dropped view is then counted back out of Adam's step count at once instead of
        # up to poll_interval iterations later)
        after_densify = last_densify is not None and it - last_densify in (1, 2)
        if densify_now or after_densify or (poll_interval and it % poll_interval == 0) or it == n_iter:
            dropped = step.poll_overflow()
            if dropped:
                history.append((it, "dropped_views", dropped))
                replay_now = replay_dropped and not step.dp
        if collecting:
            if densify_now:
                size_threshold = 20 if it > sch.opacity_reset_interval else None
                normal_thr = sch.densify_grad_normal_threshold if it > sch.normal_densify_from_iter else 99999
                # densify precedes gaussians.step() (train.py:167-177), which then finds .grad = None on the freshly
                # replaced parameters and updates nothing: the optimizer step of this iteration is skipped
                info = step.densify_and_prune(sch.densify_grad_threshold, sch.min_opacity, extent, size_threshold,
                                              normal_thr, percent_dense=sch.percent_dense, generator=gen)
                history.append((it, "densify", info["rows_out"]))
                last_densify = it
                skip_step = True
            else:
                skip_step = False
            if it % sch.opacity_reset_interval == 0 or (white_background and it == sch.densify_from_iter):
                step.reset_opacity()
                history.append((it, "reset_opacity", step.P))
        else:
            skip_step = False
        if not skip_step:
            step.optimizer_step()
        if replay_now:
            # (behind this iteration's own optimizer step / densification: a replay overwrites the gradient buffers)
            def inputs_of(i):
                w = view_of[i]
                view_of[step._iter + 1] = w
                step.iteration = it - 1       # (__call__ advances it: EVERY replayed view runs with this iteration's schedule weight)
                return cameras[w], background, images[w], None if masks is None else masks[w]
            # on_iteration(it, step) must still see iteration `it`'s outputs, not those of the last replayed view
            keep = (getattr(step, "last_outs", None), getattr(step, "viewspace_grad", None))
            redone = step.replay_dropped(inputs_of)
            step.iteration = it
            step.last_outs, step.viewspace_grad = keep
            if redone:
                history.append((it, "replayed_views", redone))
        # forward passes the last poll has looked at and not reported as dropped can never be replayed: forget their views
        pending = set(getattr(step, "dropped_iterations", ()))
        for k in [k for k in view_of if k <= getattr(step, "_drop_polled", 0) and k not in pending]:
            del view_of[k]
        if on_iteration is not None:
            on_iteration(it, step)
    return step, history
