"""The reference's training checkpoint format (`chkpnt<iteration>.pth`), both directions, for the fused iterations.

train.py:190-203 writes `torch.save((gaussians.capture(), iteration), path)`; `GaussianModel.capture()`
(scene/gaussian_model.py:113-141) is the list
    [active_sh_degree, _xyz, _normal, _shs_dc, _shs_rest, _scaling, _rotation, _opacity, max_radii2D, weights_accum,
     xyz_gradient_accum, normal_gradient_accum, denom, optimizer.state_dict(), spatial_lr_scale]
    (+ [_base_color, _roughness, _incidents_dc, _incidents_rest, _visibility_dc, _visibility_rest] for render_type neilf)
and `restore()` / `create_from_ckpt()` (:143-182, :358-408) read it back -- stage 2 of every run script starts from the
stage-1 file (`-c .../3dgs/chkpnt30000.pth`).  `capture` produces exactly that object from a FusedStage1Step /
FusedStage2Step (the optimizer entry is the state_dict of a real torch.optim.Adam built with the reference's groups, so
`optimizer.load_state_dict` accepts it); `restore` turns such an object (ours or the reference's) back into raw
parameters, Adam moments and densification statistics.  Plain PyTorch on whatever device the tensors live on; not on
the hot path.
"""
import types

import torch
from torch import nn

STAGE1_GROUPS = ("xyz", "normal", "rotation", "scaling", "opacity", "f_dc", "f_rest")          # training_setup order
PBR_GROUPS = ("base_color", "roughness", "incidents_dc", "incidents_rest", "visibility_dc", "visibility_rest")
STAT_NAMES = ("weights_accum", "xyz_gradient_accum", "normal_gradient_accum", "denom")


def reference_learning_rates(spatial_lr_scale=1.0):
    """The param_group rates GaussianModel.training_setup installs (scene/gaussian_model.py:465-486) from the defaults of
    arguments/__init__.py:70-97 (position_lr_init 1.6e-4 * spatial_lr_scale, normal 0.01, rotation 0.001, scaling 0.005,
    opacity 0.05, sh 0.0025 and /20, base_color / roughness 0.01, light 0.001 / 0.0001, visibility 0.0025)."""
    return dict(xyz=0.00016 * spatial_lr_scale, normal=0.01, rotation=0.001, scaling=0.005, opacity=0.05, f_dc=0.0025,
                f_rest=0.0025 / 20.0, base_color=0.01, roughness=0.01, incidents_dc=0.001, incidents_rest=0.0001,
                visibility_dc=0.0025, visibility_rest=0.0025)


def _named_tensors(step):
    """Reference group name -> (parameter, exp_avg, exp_avg_sq) views of a fused step object (its single [P,16,3] SH /
    incident tensors are split into the dc / rest groups the reference keeps)."""
    if getattr(step, "_chain_deferred", None) is not None or getattr(step, "_early_pending", False):
        step.flush()             # (a single-GPU iteration left the incident-light group's update pending or running)
    order = step._opt_order
    mom = {k: (step.opt.groups[i]["exp_avg"], step.opt.groups[i]["exp_avg_sq"]) for i, k in enumerate(order)}
    out = {}
    for k in ("xyz", "normal", "rotation", "scaling", "opacity", "base_color", "roughness"):
        if k in mom:
            out[k] = (getattr(step, k), mom[k][0], mom[k][1])
    out["f_dc"] = (step.shs[:, :1], mom["shs"][0][:, :1], mom["shs"][1][:, :1])
    out["f_rest"] = (step.shs[:, 1:], mom["shs"][0][:, 1:], mom["shs"][1][:, 1:])
    if "incidents" in mom:
        out["incidents_dc"] = (step.incidents[:, :1], mom["incidents"][0][:, :1], mom["incidents"][1][:, :1])
        out["incidents_rest"] = (step.incidents[:, 1:], mom["incidents"][0][:, 1:], mom["incidents"][1][:, 1:])
    return out


def step_learning_rates(step):
    """{reference group name: lr} of the groups a fused step owns, read from its optimizer (FusedAdam.groups): the joined
    [P,16,3] SH tensors split back into the dc / rest rates (`lr` / `lr_tail`)."""
    out = {}
    opt = getattr(step, "opt", None)
    if opt is None:
        return out
    for k, g in zip(step._opt_order, opt.groups):
        if g.get("lr") is None:
            continue
        tail = g.get("lr_tail")
        if k == "shs":
            out["f_dc"], out["f_rest"] = float(g["lr"]), float(g["lr"] if tail is None else tail)
        elif k == "incidents":
            out["incidents_dc"], out["incidents_rest"] = float(g["lr"]), float(g["lr"] if tail is None else tail)
        elif k != "env":
            out[k] = float(g["lr"])
    return out


def capture(step, iteration, spatial_lr_scale=1.0, active_sh_degree=3, learning_rates=None):
    """-> the object train.py saves: `(GaussianModel.capture() list, iteration)`.  `step`: FusedStage1Step or
    FusedStage2Step (stage 2 adds the six PBR entries; the baked-visibility SH groups this repo does not train are
    written as zeros of the reference's shapes).  `learning_rates`: {group name: lr} for the param_groups -- torch's
    Optimizer.load_state_dict ADOPTS the saved groups' hyper-parameters, so a reference GaussianModel.restore(is_training=
    True) on this file trains with exactly these rates; the default is what the step itself trains with
    (step_learning_rates: the scheduled xyz rate, the stage-2 rates of the run script), and the reference's own training_setup
    values (reference_learning_rates) for the groups the step does not own.
    Frozen groups (learning rate 0: every geometry group under script/run_syn4.sh / run_dtu.sh): the fused iteration launches
    no Adam for them, so their moments are written as they were when the group froze (zeros for a group that never trained),
    whereas the reference's torch.optim.Adam keeps accumulating exp_avg / exp_avg_sq (and `step`) for lr = 0 groups.  The
    PARAMETERS are identical either way; a reference run that restores this file and then RAISES those rates starts from
    different optimizer state than it would from its own chkpnt."""
    named = _named_tensors(step)
    P = step.xyz.shape[0]
    dev = step.xyz.device
    pbr = "base_color" in named
    if pbr:
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        named["visibility_dc"] = (z(P, 1, 1), z(P, 1, 1), z(P, 1, 1))
        named["visibility_rest"] = (z(P, 15, 1), z(P, 15, 1), z(P, 15, 1))
    names = STAGE1_GROUPS + (PBR_GROUPS if pbr else ())
    params = {n: nn.Parameter(named[n][0].detach().clone().contiguous().requires_grad_(True)) for n in names}
    lrs = reference_learning_rates(spatial_lr_scale)
    lrs.update(step_learning_rates(step))          # what the step actually trains with (scheduled xyz rate, stage-2 rates)
    lrs.update(learning_rates or {})
    optimizer = torch.optim.Adam([{"params": [params[n]], "lr": float(lrs[n]), "name": n} for n in names],
                                 lr=0.0, eps=1e-15)
    steps = float(step.opt.step_count)
    for n in names:
        if steps > 0:
            optimizer.state[params[n]] = {"step": torch.tensor(steps), "exp_avg": named[n][1].detach().clone().contiguous(),
                                          "exp_avg_sq": named[n][2].detach().clone().contiguous()}
    stats = getattr(step, "stats", None)
    col = lambda name: (getattr(stats, name).detach().clone().view(-1, 1) if stats is not None
                        else torch.zeros(P, 1, dtype=torch.float32, device=dev))
    max_radii = stats.max_radii2D.detach().clone() if stats is not None else torch.zeros(P, dtype=torch.float32, device=dev)
    captured = [int(active_sh_degree), params["xyz"], params["normal"], params["f_dc"], params["f_rest"],
                params["scaling"], params["rotation"], params["opacity"], max_radii, col("weights_accum"),
                col("xyz_gradient_accum"), col("normal_gradient_accum"), col("denom"), optimizer.state_dict(),
                float(spatial_lr_scale)]
    if pbr:
        captured.extend(params[n] for n in PBR_GROUPS)
    return captured, int(iteration)


def restore(checkpoint, device=None, pbr=False, allow_unsafe=False):
    """`checkpoint`: the `(captured list, iteration)` object (or a path to one).  -> namespace with the raw parameters
    under this repo's names (xyz, normal, scaling, rotation, opacity, features_dc, features_rest [, base_color, roughness,
    incidents_dc, incidents_rest, visibility_dc, visibility_rest]), `moments` {reference group name: (exp_avg, exp_avg_sq)}
    (empty without optimizer state), `adam_steps`, `stats` {name: [P] tensor} + `max_radii2D`, `active_sh_degree`,
    `spatial_lr_scale`, `iteration`.  A stage-1 file (15 entries) can be handed to FusedStage1Step as its `params`.
    `pbr=True` is the stage-1 -> stage-2 hand-off of every run script (`train.py -t neilf -c .../3dgs/chkpnt30000.pth`):
    as GaussianModel.create_from_ckpt does for a 15-entry file (scene/gaussian_model.py:381-403), base_color, roughness,
    incidents and the visibility SH groups are ZERO-initialised, so the result (+ an `env` of the caller's,
    DirectLightMap) can be handed to FusedStage2Step.  The reference-faithful hand-off does NOT carry the Adam state over:
    train.py calls create_from_ckpt(restore_optimizer=True) before training_setup, when `optimizer` is still None, and
    swallows the exception -- stage 2 starts with fresh moments and step count.  `load_moments` is for resuming the SAME
    stage from this repo's own checkpoints.
    `allow_unsafe`: retry a file that torch's weights-only loader refuses with full unpickling (trusted files only)."""
    if isinstance(checkpoint, (str, bytes)) or hasattr(checkpoint, "__fspath__"):
        # tensors, Parameters, dicts, lists -- everything GaussianModel.capture() writes -- load without arbitrary
        # unpickling.  A file the safe loader refuses is NOT retried with the unsafe one (a malicious file would only have
        # to fail the first attempt) unless the caller says so for a file it trusts; a corrupt file raises either way.
        import pickle
        try:
            # train.py stores spatial_lr_scale as the numpy float64 the dataset reader computed (scene/__init__.py:86,
            # train.py:190-203): numpy's scalar reconstructor and dtype are allow-listed for this load, nothing else is
            import numpy as np
            extra = [np.dtype, type(np.dtype(np.float64)), type(np.dtype(np.float32))]
            core = getattr(np, "_core", None) or getattr(np, "core")
            extra.append(core.multiarray.scalar)
            with torch.serialization.safe_globals(extra):
                checkpoint = torch.load(checkpoint, map_location="cpu", weights_only=True)
        except pickle.UnpicklingError:
            if not allow_unsafe:
                raise
            checkpoint = torch.load(checkpoint, map_location="cpu", weights_only=False)
    captured, iteration = checkpoint
    if len(captured) not in (15, 21):
        raise RuntimeError("not a GaussianModel checkpoint: %d entries" % len(captured))
    to = (lambda t: t.detach().to(device)) if device is not None else (lambda t: t.detach())
    out = types.SimpleNamespace(active_sh_degree=int(captured[0]), iteration=int(iteration),
                                spatial_lr_scale=float(captured[14]))
    for name, idx in (("xyz", 1), ("normal", 2), ("features_dc", 3), ("features_rest", 4), ("scaling", 5), ("rotation", 6),
                      ("opacity", 7)):
        setattr(out, name, to(captured[idx]).clone().contiguous())
    out.max_radii2D = to(captured[8]).clone().reshape(-1)
    out.stats = {n: to(captured[9 + i]).clone().reshape(-1) for i, n in enumerate(STAT_NAMES)}
    names = STAGE1_GROUPS
    if len(captured) == 21:
        for j, n in enumerate(PBR_GROUPS):
            setattr(out, n, to(captured[15 + j]).clone().contiguous())
        names = STAGE1_GROUPS + PBR_GROUPS
    elif pbr:
        P = out.xyz.shape[0]
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=out.xyz.device)
        out.base_color, out.roughness = z(P, 3), z(P, 1)
        out.incidents_dc, out.incidents_rest = z(P, 1, 3), z(P, 15, 3)
        out.visibility_dc, out.visibility_rest = z(P, 1, 1), z(P, 15, 1)
    opt = captured[13]
    out.moments, out.adam_steps = {}, 0
    group_names = [g.get("name") for g in opt.get("param_groups", [])]
    for g in opt.get("param_groups", []):
        st = opt["state"].get(g["params"][0])
        if st is None:
            continue
        out.moments[g["name"]] = (to(st["exp_avg"]).clone(), to(st["exp_avg_sq"]).clone())
        out.adam_steps = max(out.adam_steps, int(float(st["step"])))
    out.group_names = group_names or list(names)
    return out


def load_moments(step, restored):
    """Adam state of a restored checkpoint into a fused step built from it (`FusedStageNStep(restored, ...)`): exp_avg /
    exp_avg_sq of every group the optimizer knows, dc / rest halves re-joined, and the shared step count.  For resuming
    the same stage; the reference's stage-1 -> stage-2 hand-off starts from fresh moments (see restore)."""
    order = step._opt_order
    slot = {k: step.opt.groups[i] for i, k in enumerate(order)}
    m = restored.moments
    for k in ("xyz", "normal", "rotation", "scaling", "opacity", "base_color", "roughness"):
        if k in slot and k in m:
            slot[k]["exp_avg"].copy_(m[k][0])
            slot[k]["exp_avg_sq"].copy_(m[k][1])
    for joined, dc, rest in (("shs", "f_dc", "f_rest"), ("incidents", "incidents_dc", "incidents_rest")):
        if joined in slot and dc in m and rest in m:
            slot[joined]["exp_avg"].copy_(torch.cat([m[dc][0], m[rest][0]], 1))
            slot[joined]["exp_avg_sq"].copy_(torch.cat([m[dc][1], m[rest][1]], 1))
    step.opt.step_count = int(restored.adam_steps)
