"""The reference's stage-2 training iteration assembled from PINNED parts -- TEST INFRASTRUCTURE ONLY (tests/ and nothing
else imports it).  It is the "reference pipeline" side of the north-star quality clause (PSNR within 0.1 dB):

    rasterize fwd / bwd   the REAL reference kernels (forward.cu / backward.cu compiled unmodified into oracle/_ref by
                          oracle/build_ref.py), behind an autograd.Function that routes the nine gradients exactly as
                          gaussian_renderer/r3dg_rasterization.py:132-183 does
    visibility            the REAL reference trace kernel (trace.cu:196-286) over the exact LBVH (the reference's own
                          internal boxes are racy on gfx950, tests/test_reference_gpu.py), rays = oracle/shading.py's
                          restatement of fibonacci_sphere_sampling (pinned by tests/golden/fibonacci_reference.npz)
    shading               oracle/shading.py under autograd (pinned by tests/golden/shading_reference.npz to neilf.py:339-407)
    render_view glue      gaussian_renderer/neilf.py:74-209 (is_training=True: S=16 feature row :120-122, pbr composite :179)
    loss                  neilf.py:212-318 with script/run_nerf.sh:20-39's weights (lambda_pbr 1, lambda_light 0.01,
                          lambda_env_smooth 0.01, everything else 0) through the pinned ssim / tv_loss / rgb_to_srgb
    optimizer             torch.optim.Adam(eps=1e-15) (gaussian_model.py:486)
Nothing of this repo's HIP library runs in it except the LBVH *build* (bit-identical to the reference's node table)."""
import torch
import torch.nn.functional as F

from . import reference_gpu as rg
from . import shading as osh


class _RefRasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, opacities, scales, rotations, features, cam, bg):
        a = [t.detach().contiguous() for t in (means3D, sh, opacities, scales, rotations, features)]
        fw = rg.rasterize_forward(bg, a[0], a[5], None, a[2], a[3], a[4], 1.0, None, cam.world_view_transform,
                                  cam.full_proj_transform, cam.tanfovx, cam.tanfovy, cam.cx, cam.cy, cam.image_height,
                                  cam.image_width, a[1], 3, cam.camera_center)
        # only what the backward reads -- NOT the output tensors: an output held by its own grad_fn's ctx is a reference cycle
        # that nothing but the garbage collector breaks (277 MB per iteration at 300k Gaussians / 800x800 stayed allocated: the
        # 1000-iteration PSNR run ended at 285 GiB)
        ctx.fw = {k: fw[k] for k in ("buffers", "num_rendered", "radii")}
        ctx.shapes = {k: (fw[k].shape, fw[k].dtype) for k in ("color", "opacity", "depth", "feature")}
        ctx.cam, ctx.bg, ctx.a = cam, bg, a
        n_contrib = fw["n_contrib"]
        ctx.mark_non_differentiable(n_contrib, fw["normal"], fw["xyz"])
        return fw["color"], fw["opacity"], fw["depth"], fw["feature"], fw["normal"], fw["xyz"], n_contrib

    @staticmethod
    def backward(ctx, gC, gO, gD, gF, _gn, _gx, _gc):
        cam, a = ctx.cam, ctx.a
        dev = a[0].device
        z = lambda g, k: torch.zeros(ctx.shapes[k][0], dtype=ctx.shapes[k][1], device=dev) if g is None else g.contiguous()
        fw = ctx.fw
        g = rg.rasterize_backward(fw, ctx.bg, a[0], a[5], None, a[3], a[4], 1.0, None, cam.world_view_transform,
                                  cam.full_proj_transform, cam.tanfovx, cam.tanfovy, z(gC, "color"), z(gO, "opacity"),
                                  z(gD, "depth"), z(gF, "feature"), a[1], 3, cam.camera_center, True)
        ctx.fw = ctx.a = None                              # (the scratch buffers go with the iteration)
        return g["mean3D"], g["mean2D"], g["sh"], g["opacity"], g["scale"], g["rot"], g["feature"], None, None


class ReferenceStage2:
    """Raw parameters (a bench_core.GaussianParams) trained by the reference pipeline."""

    def __init__(self, params, sample_num, lr, tree_builder):
        from relightable3dgaussian_amd.train_step import inverse_covariance
        self.p, self.K = params, sample_num
        with torch.no_grad():
            xyz, scales, rot = params.xyz.detach(), params.get_scaling().detach(), params.get_rotation().detach()
            normal = params.get_normal().detach()
            nodes, aabbs = tree_builder(xyz, scales, rot)                      # exact LBVH (see module docstring)
            dirs, areas = osh.fibonacci_sphere_sampling(normal, sample_num)
            dirs = dirs.contiguous()
            rays_o = (xyz[:, None, :] + 0.05 * dirs).contiguous()              # bvh/__init__.py:63
            cinv = inverse_covariance(scales, rot)
            _, vis = rg.bvh_trace_opacity(nodes, aabbs, rays_o, dirs, xyz.contiguous(), cinv,
                                          params.get_opacity().detach()[:, 0].contiguous(), normal.contiguous())
            self.visibility, self.incident_dirs, self.incident_areas = vis[..., None], dirs, areas
        self.opt = torch.optim.Adam(params.parameters(), lr=lr, eps=1e-15)

    def render(self, cam, bg):
        p = self.p
        means3D = p.xyz
        means2D = torch.zeros_like(means3D, requires_grad=True)
        base_color = 0.03 + 0.77 * torch.sigmoid(p.base_color)
        roughness = 0.09 + 0.9 * torch.sigmoid(p.roughness)
        normal = p.get_normal()
        incidents = torch.cat([p.incidents_dc, p.incidents_rest], 1)
        viewdirs = F.normalize(cam.camera_center - means3D, dim=-1)
        env = F.softplus(p.env)[0]
        r = osh.rendering_equation(base_color, roughness, normal.detach(), viewdirs, incidents, env, self.visibility,
                                   self.incident_dirs, self.incident_areas)
        xyz_h = torch.cat([means3D, torch.ones_like(means3D[:, :1])], -1)
        depths = (xyz_h @ cam.world_view_transform)[:, 2:3]
        features = torch.cat([depths, depths.square(), r["pbr"], normal, base_color, roughness, r["diffuse_light"],
                              r["incident_visibility"]], -1)
        outs = _RefRasterize.apply(means3D, means2D, p.get_shs(), p.get_opacity(), p.get_scaling(), p.get_rotation(),
                                   features, cam, bg)
        return outs, r["diffuse_light"], env

    def loss(self, cam, bg, gt):
        from relightable3dgaussian_amd.train_step import image_loss, rgb_to_srgb, tv_loss
        (image, opacity, depth, feature, pseudo_normal, xyz, n_contrib), diffuse_light, env = self.render(cam, bg)
        feat = feature / opacity.clamp_min(1e-5) * (n_contrib > 0)
        pbr_img = rgb_to_srgb(feat[2:5] * opacity + (1 - opacity) * bg[:, None, None])
        loss = image_loss(image, gt) + 1.0 * image_loss(pbr_img, gt)
        mean_light = diffuse_light.mean(-1, keepdim=True).expand_as(diffuse_light)
        loss = loss + 0.01 * F.l1_loss(diffuse_light, mean_light)
        loss = loss + 0.01 * tv_loss(env.permute(2, 0, 1))
        return loss, image

    def step(self, cam, bg, gt):
        loss, image = self.loss(cam, bg, gt)
        loss.backward()
        self.opt.step()
        self.opt.zero_grad(set_to_none=False)
        return float(loss.detach()), image.detach()
