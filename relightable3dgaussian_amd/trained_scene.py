"""Workloads whose splat statistics are NOT the i.i.d. log-normal ones of synthetic.make_scene: every bench
number and most parity cases run on splats of sigma ~ 2.8 px and ~6 tiles each, while a trained, densified scene has heavy-tailed
radii and tile lists.

* `train_scene`: a scene produced the way the reference produces one -- stage 1 from a few thousand random points
  (GaussianModel.create_from_pcd, scene/gaussian_model.py:409-441) with the densification schedule of train.py:158-175
  (compressed in time) against views of a hidden teacher, through this repo's own train_loop.train_stage1.  A few seconds on the
  MI355X.  The result depends on the order of float atomics in the backward, so it is not bit-reproducible: parity tests train
  one and then compare both rasterizers on THAT scene.
* `heavy_tail_scene`: the synthetic scene with a fraction of the splats scaled up (default 1 % x 20): rectangles of thousands of
  tiles beside the six-tile ones.
* `binning_stats`: what the front end sees of a scene in one view -- num_rendered, the tile-list length distribution, the
  rectangle-size tail.
Scenes come back in synthetic.make_scene's format (activated values on the CPU), so every consumer of that works on them."""
import math

import torch

from . import synthetic as syn


def add_stage2_extras(scene, seed=0):
    """The PBR attributes make_scene(stage2=True) draws, for a scene that has none (same distributions, gaussian_model.py:51-52,
    direct_light_map.py:14)."""
    g = torch.Generator().manual_seed(seed + 17)
    P = scene["xyz"].shape[0]
    scene = dict(scene)
    scene["base_color"] = 0.03 + 0.77 * torch.sigmoid(torch.randn(P, 3, generator=g))
    scene["roughness"] = 0.09 + 0.9 * torch.sigmoid(torch.randn(P, 1, generator=g))
    scene["incidents"] = 0.02 * torch.randn(P, 16, 3, generator=g)
    scene["env"] = 0.5 * torch.rand(1, 16, 32, 3, generator=g)
    return scene


def heavy_tail_scene(P=300_000, frac=0.01, factor=20.0, seed=0, stage2=True, scale_log_mean=-4.6):
    """make_scene with a seeded `frac` of the splats `factor` times larger (all three axes)."""
    scene = syn.make_scene(P=P, seed=seed, stage2=stage2, scale_log_mean=scale_log_mean)
    g = torch.Generator().manual_seed(seed + 991)
    big = torch.rand(P, generator=g) < frac
    scene["scales"] = torch.where(big[:, None], scene["scales"] * factor, scene["scales"]).contiguous()
    scene["heavy_tail"] = dict(frac=frac, factor=factor, n_big=int(big.sum()))
    return scene


def scene_of_step(step):
    """A FusedStage1Step's raw parameters as a make_scene-format dict (activations of gaussian_model.py:183-232)."""
    with torch.no_grad():
        return dict(xyz=step.xyz.detach().cpu().clone(),
                    normal=torch.nn.functional.normalize(step.normal.detach(), dim=-1, eps=1e-3).cpu(),
                    scales=torch.exp(step.scaling.detach()).cpu(),
                    rotations=torch.nn.functional.normalize(step.rotation.detach()).cpu(),
                    opacity=torch.sigmoid(step.opacity.detach()).cpu(), shs=step.shs.detach().cpu().clone().contiguous(),
                    sh_degree=3, M=16)


def train_scene(dev, res=800, views=24, teacher_points=60_000, teacher_scale=-3.6, init_points=4000, iterations=3000,
                densify_until=2600, densification_interval=100, densify_from=200, opacity_reset_interval=1000,
                grad_threshold=0.00005, grad_normal_threshold=999.0, seed=0, stage2=True, history_out=None):
    """-> make_scene-format scene trained from `init_points` random points against `views` renders of a hidden teacher
    (teacher_points splats of log-scale mean teacher_scale), `iterations` stage-1 iterations with the reference's densification
    schedule compressed into them (train.py:158-175: statistics every iteration, densify_and_prune every
    `densification_interval` from `densify_from` until `densify_until`, the size threshold once past the first opacity reset).
    `grad_normal_threshold`: 999 = script/run_dtu.sh:16 (clone / split on the position gradient only); the reference's default 2e-9
    (arguments/__init__.py:105) selects every Gaussian whose normal has a gradient and grows this scene to 1.6 M rows in 2 400
    iterations."""
    from . import train_loop
    from .bench_core import GaussianParams, render_stage1
    torch.manual_seed(seed)
    cams = [c.to(dev) for c in syn.orbit_cameras(views, width=res, height=res)]
    bg = torch.ones(3, device=dev)
    with torch.no_grad():
        teacher = GaussianParams(syn.make_scene(P=teacher_points, seed=seed + 3, stage2=False, scale_log_mean=teacher_scale), dev, False)
        gts = [render_stage1(teacher, c, bg)[2].clone() for c in cams]
        del teacher
    g = torch.Generator().manual_seed(seed + 5)
    pts = torch.rand(init_points, 3, generator=g) * 2.6 - 1.3                    # dataset_readers.py:290-297
    cols = torch.rand(init_points, 3, generator=g)
    init = train_loop.create_from_points(pts, cols, device=dev)
    extent, _ = syn.cameras_extent(cams)
    sch = train_loop.Schedule(densify_from_iter=densify_from, densification_interval=densification_interval,
                              densify_until_iter=densify_until, opacity_reset_interval=opacity_reset_interval,
                              densify_grad_threshold=grad_threshold, densify_grad_normal_threshold=grad_normal_threshold,
                              percent_dense=0.01)
    step, history = train_loop.train_stage1(init, cams, gts, bg, extent=extent, schedule=sch, iterations=iterations, seed=seed)
    torch.cuda.synchronize()
    if history_out is not None:
        history_out.extend(history)
    scene = scene_of_step(step)
    del step
    torch.cuda.empty_cache()
    if stage2:
        scene = add_stage2_extras(scene, seed)
    return scene


@torch.no_grad()
def binning_stats(scene, cam, dev):
    """One forward of `scene` in view `cam`: num_rendered, tile-list lengths (mean over non-empty tiles, 99.9th percentile,
    max), rectangle sizes (tiles per visible Gaussian: mean, 99.9th percentile, max; the share of Gaussians above 32 tiles --
    the wave-cooperative expansion of the binning kernels -- and of the instances they carry), radii."""
    from . import rasterizer_ops as ro
    P = scene["xyz"].shape[0]
    H, W = cam.image_height, cam.image_width
    empty = torch.Tensor([])
    t = lambda k: scene[k].to(dev).contiguous()
    out = ro.rasterize_gaussians(torch.ones(3, device=dev), t("xyz"), torch.zeros(P, 0, device=dev), empty, t("opacity"), t("scales"),
                                 t("rotations"), 1.0, empty, cam.world_view_transform.contiguous(),
                                 cam.full_proj_transform.contiguous(), cam.tanfovx, cam.tanfovy, cam.cx, cam.cy, H, W,
                                 t("shs"), 3, cam.camera_center.contiguous(), False, True, False)
    R = int(out[0])
    st = ro.decode_state(out[-3], out[-2], out[-1], P, R, H, W)
    touched = st["tiles_touched"].long()
    live = touched > 0
    tl = (st["ranges"][:, 1].long() - st["ranges"][:, 0].long())
    tl_live = tl[tl > 0].float()
    rect = touched[live].float()
    radii = out[9][live].float()
    q = lambda x, p: float(torch.quantile(x, p)) if x.numel() else 0.0
    big = touched > 32
    return dict(points=P, visible=int(live.sum()), num_rendered=R,
                tile_list=dict(tiles=int(tl.numel()), non_empty=int(tl_live.numel()), mean=round(float(tl_live.mean()), 1) if tl_live.numel() else 0.0,
                               p999=round(q(tl_live, 0.999)), max=int(tl.max()) if tl.numel() else 0),
                rect_tiles=dict(mean=round(float(rect.mean()), 2) if rect.numel() else 0.0, p999=round(q(rect, 0.999)),
                                max=int(touched.max()) if P else 0,
                                above_32_tiles_frac=round(float(big.sum()) / max(1, int(live.sum())), 5),
                                above_32_tiles_instance_share=round(float(touched[big].sum()) / max(1.0, float(touched.sum())), 4)),
                radius_px=dict(mean=round(float(radii.mean()), 2) if radii.numel() else 0.0, p999=round(q(radii, 0.999)),
                               max=int(radii.max()) if radii.numel() else 0))
