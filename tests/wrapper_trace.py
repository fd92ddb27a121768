"""Call-trace harness for the autograd face of the rasterizer: drives a `GaussianRasterizer`-shaped
wrapper with a recording fake backend and returns what the backend saw -- positional arguments of `rasterize_gaussians` /
`rasterize_gaussians_backward` (which named input or setting arrived where, shape, dtype, device) -- and which input
receives which of the backend's nine gradients.  Used twice: tests/golden/make_densify_golden.py runs it on the REFERENCE's
wrapper (gaussian_renderer/r3dg_rasterization.py) to write tests/golden/wrapper_trace_reference.json, and
tests/test_host_mirrors_cpu.py runs it on this repo's relightable3dgaussian_amd/rasterizer.py.  Pure CPU."""
import torch

P, S, H, W = 6, 4, 5, 7
TAGS = dict(means3D=11.0, means2D=12.0, opacities=13.0, shs=14.0, colors_precomp=15.0, scales=16.0, rotations=17.0,
            cov3D_precomp=18.0, features=19.0, bg=21.0, viewmatrix=22.0, projmatrix=23.0, campos=24.0)
SHAPES = dict(means3D=(P, 3), means2D=(P, 3), opacities=(P, 1), shs=(P, 16, 3), colors_precomp=(P, 3), scales=(P, 3),
              rotations=(P, 4), cov3D_precomp=(P, 6), features=(P, S), bg=(3,), viewmatrix=(4, 4), projmatrix=(4, 4),
              campos=(3,))
SCALARS = dict(image_height=H, image_width=W, tanfovx=0.31, tanfovy=0.37, cx=3.25, cy=2.75, scale_modifier=1.5,
               sh_degree=2, prefiltered=False, backward_geometry=True, computer_pseudo_normal=True, debug=False)
GRAD_TAGS = (101.0, 102.0, 103.0, 104.0, 105.0, 106.0, 107.0, 108.0, 109.0)      # order of the backend's 9-tuple
BUFFER_TAGS = dict(geomBuffer=31, binningBuffer=32, imgBuffer=33)


def _describe(a, by_tag):
    if isinstance(a, torch.Tensor):
        tag = None
        if a.numel() > 0:
            v = float(a.detach().flatten()[0])
            tag = by_tag.get(v)
        return dict(kind="tensor", shape=list(a.shape), dtype=str(a.dtype), device=a.device.type, input=tag)
    return dict(kind=type(a).__name__, value=a)


def run(settings_cls, rasterizer_cls, install_backend, variant):
    """variant: "sh_scale" (shs + scales/rotations + features), "color_cov" (colors_precomp + cov3D_precomp, no features).
    install_backend(forward_fn, backward_fn) plugs the fakes in place of the compiled extension."""
    by_tag = {v: k for k, v in TAGS.items()}
    by_tag.update({float(v): k for k, v in BUFFER_TAGS.items()})
    by_tag.update({201.0: "grad_out_color", 202.0: "grad_out_opacity", 203.0: "grad_out_depth", 204.0: "grad_out_feature",
                   41.0: "radii"})
    t = {k: torch.full(SHAPES[k], TAGS[k]) for k in TAGS}
    leaves = ("means3D", "means2D", "opacities", "shs", "colors_precomp", "scales", "rotations", "cov3D_precomp", "features")
    for k in leaves:
        t[k].requires_grad_(True)
    trace = {}
    nS = S if variant == "sh_scale" else 0

    def fake_forward(*args):
        trace["forward_args"] = [_describe(a, by_tag) for a in args]
        z = lambda *s: torch.zeros(*s)
        outs = (7, torch.zeros(H, W, dtype=torch.int32), z(3, H, W), z(1, H, W), z(1, H, W), z(nS, H, W), z(3, H, W),
                z(3, H, W), z(P, 1), torch.full((P,), 41, dtype=torch.int32),
                torch.full((8,), BUFFER_TAGS["geomBuffer"], dtype=torch.uint8),
                torch.full((8,), BUFFER_TAGS["binningBuffer"], dtype=torch.uint8),
                torch.full((8,), BUFFER_TAGS["imgBuffer"], dtype=torch.uint8))
        return outs

    def fake_backward(*args):
        trace["backward_args"] = [_describe(a, by_tag) for a in args]
        shapes = ((P, 3), (P, 3), (P, 1), (P, 3), (P, nS), (P, 6), (P, 16, 3), (P, 3), (P, 4))
        return tuple(torch.full(s, g) for s, g in zip(shapes, GRAD_TAGS))

    install_backend(fake_forward, fake_backward)
    rs = settings_cls(bg=t["bg"], viewmatrix=t["viewmatrix"], projmatrix=t["projmatrix"], campos=t["campos"], **SCALARS)
    kw = dict(shs=t["shs"], scales=t["scales"], rotations=t["rotations"], features=t["features"]) if variant == "sh_scale" \
        else dict(colors_precomp=t["colors_precomp"], cov3D_precomp=t["cov3D_precomp"])
    outs = rasterizer_cls(rs)(t["means3D"], t["means2D"], t["opacities"], **kw)
    trace["n_outputs"] = len(outs)
    trace["num_rendered"] = int(outs[0])
    color, opacity, depth, feature = outs[2], outs[3], outs[4], outs[5]
    loss = (color * 201.0).sum() + (opacity * 202.0).sum() + (depth * 203.0).sum()
    if feature.numel():
        loss = loss + (feature * 204.0).sum()
    loss.backward()
    routing = {}
    for k in leaves:
        g = t[k].grad
        routing[k] = None if g is None else float(g.flatten()[0]) if g.numel() else "empty"
    trace["grad_routing"] = routing
    return trace


def run_raytracer(raytracer_cls, install_backend):
    """The same for `RayTracer` (bvh/__init__.py:28-71): what `create_bvh` and `trace_bvh_opacity` receive (the ray origins
    already offset by 0.05 d) and how the result dict is shaped."""
    g = torch.Generator().manual_seed(9)
    n, R, K = 12, 5, 3
    means = torch.randn(n, 3, generator=g)
    scales = torch.rand(n, 3, generator=g) * 0.1 + 0.02
    rot = torch.nn.functional.normalize(torch.randn(n, 4, generator=g), dim=-1)
    trace = {}

    def fake_create(means3D, scales_, rotations, nodes, aabbs):
        trace["create_args"] = [dict(shape=list(a.shape), dtype=str(a.dtype)) for a in (means3D, scales_, rotations, nodes, aabbs)]
        return torch.full((2 * n - 1, 5), 7, dtype=torch.int32), torch.full((2 * n - 1, 6), 8.0), torch.full((n,), 9)

    def fake_trace(*args):
        trace["trace_args"] = [dict(shape=list(a.shape), dtype=str(a.dtype), first=float(a.flatten()[0])) for a in args]
        trace["rays_o_seen"] = args[2].clone()
        lead = args[2].shape[:-1]
        return torch.full(lead, 2, dtype=torch.int32), torch.full(lead, 0.5)

    install_backend(fake_create, fake_trace)
    tracer = raytracer_cls(means, scales, rot)
    rays_o = torch.randn(R, K, 3, generator=g)
    rays_d = torch.nn.functional.normalize(torch.randn(R, K, 3, generator=g), dim=-1)
    symm = torch.rand(n, 6, generator=g)
    opac = torch.rand(n, generator=g)
    normals = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    res = tracer.trace_visibility(rays_o, rays_d, means, symm, opac, normals)
    trace["offset_ok"] = bool(torch.allclose(trace.pop("rays_o_seen"), rays_o + 0.05 * rays_d, rtol=0, atol=1e-7))
    trace["result"] = {k: dict(shape=list(v.shape), dtype=str(v.dtype), first=float(v.flatten()[0])) for k, v in sorted(res.items())}
    return trace
