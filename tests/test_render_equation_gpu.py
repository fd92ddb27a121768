"""GPU parity of the render_equation.{cu,h} contract model (HIP) vs the as-written CPU restatement
(oracle/shading_oracle.c).  Tolerance: forward |err| <= 1e-6 + 1e-4*max|ref|, backward |err| <= 1e-6 + 5e-4*max|ref|: the
spherical-Gaussian lobe exp((2/r^2)(h.n-1)) has a sharpness of up to ~800 at roughness 0.05, so one fp32 ulp of h.n
(dot-product order, FMA) moves D by ~1e-4 relative, and the wave reduction sums in a different order than the
reference's serial loop.  The backward reproduces the
reference's quirks Q1-Q4; Q5 (racy dL_ddirect_shs) is the well-defined sum on both sides."""
import numpy as np
import pytest
import torch

from tests.helpers import report

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ok(name, got, ref, rtol, atol):
    from tests.helpers import to_np
    ok, msg = report(name, got, to_np(ref).reshape(tuple(got.shape)), rtol, atol)
    print(msg)
    assert ok, msg


def _inputs(P, Si, Sd, Sv, seed):
    g = torch.Generator().manual_seed(seed)
    nrm = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1)
    nrm[0] = torch.tensor([0.0, 0.0, -1.0])
    return dict(base_color=torch.rand(P, 3, generator=g), roughness=0.05 + 0.9 * torch.rand(P, 1, generator=g),
                metallic=torch.rand(P, 1, generator=g), normals=nrm,
                viewdirs=torch.nn.functional.normalize(nrm + 0.8 * torch.randn(P, 3, generator=g), dim=-1),
                inc=0.3 * torch.randn(P, Si, 3, generator=g), direct=0.3 * torch.randn(1, Sd, 3, generator=g),
                vis=0.5 * torch.randn(P, Sv, 1, generator=g))


ORDER = ("base_color", "roughness", "metallic", "normals", "viewdirs", "inc", "direct", "vis")


@pytest.mark.parametrize("P,K,Si,Sd,Sv,train", [(2000, 24, 16, 16, 16, False), (500, 64, 16, 16, 16, True),
                                                 (300, 100, 16, 9, 4, False), (700, 64, 9, 16, 16, False)])
def test_contract_model_forward_backward(P, K, Si, Sd, Sv, train):
    from oracle import render_equation as ore
    from relightable3dgaussian_amd import shading_ops as so
    inp = _inputs(P, Si, Sd, Sv, seed=P + K)
    d = [inp[k].to(DEV) for k in ORDER]
    pbr, dirs, dl = so.render_equation_forward(*d, K, train)
    rnd = so.render_equation_forward.last_rand.cpu() if train else None
    r_pbr, r_dirs, r_dl = ore.forward(*[inp[k] for k in ORDER], K, rnd)
    # with the random angle, theta = rnd*2*pi + k*delta reaches ~1e2 rad where one fp32 ulp is 8e-6: whether the
    # multiply-add is fused (GPU, and nvcc by default) or not (oracle) moves the direction by ~1e-5
    _ok("incident_dirs", dirs, r_dirs, 0, 5e-5 if train else 5e-6)
    if train:   # evaluate the rest on identical directions
        dirs = torch.from_numpy(r_dirs).to(DEV)
    # (in the random-angle case the 1e-5 direction noise is amplified by the lobe sharpness: only a sanity bound)
    _ok("pbr", pbr, r_pbr, 3e-2 if train else 1e-4, 1e-6)
    _ok("diffuse_light", dl, r_dl, 1e-3 if train else 1e-4, 1e-6)
    if not train:
        outs = so.render_equation_forward_complex(*d, K)
        refs = ore.forward_complex(*[inp[k] for k in ORDER], K)
        for name, a, b in zip(("pbr", "incident_dirs", "incident_lights", "local_lights", "global_lights", "visibility",
                               "diffuse_light", "local_diffuse_light", "accum", "rgb_d", "rgb_s"), outs, refs):
            _ok("complex/" + name, a, b, 1e-4, 5e-6)
    g = torch.Generator().manual_seed(99)
    g_pbr, g_dl = torch.randn(P, 3, generator=g), torch.randn(P, 3, generator=g)
    # backward consumes the forward's incident_dirs (both sides get the ORACLE's so the comparison is apples to apples)
    grads = so.render_equation_backward(*d, K, torch.from_numpy(r_dirs).to(DEV), g_pbr.to(DEV), g_dl.to(DEV))
    refs = ore.backward(*[inp[k] for k in ORDER], K, r_dirs, g_pbr, g_dl)
    for name, a, b in zip(("dL_dbase_color", "dL_droughness", "dL_dmetallic", "dL_dnormals", "dL_dviewdirs",
                           "dL_dincidents_shs", "dL_ddirect_shs", "dL_dvisibility_shs"), grads, refs):
        _ok(name, a, b, 5e-4, 1e-6)


# ---------------------------------------------------------------------------------------------------------------------
# Pin to the REAL reference: r3dg-rasterization/render_equation.cu compiled unmodified for gfx950 by oracle/build_ref.py
# (oracle/_ref/libr3dg_reference_shading.so).  Compared: (i) this repo's HIP kernels, (ii) the CPU restatement
# oracle/shading_oracle.c -- which is thereby pinned too.  Same tolerances as above.  Quirks as they fall out:
#   Q1-Q4 are properties of the reference's arithmetic and are reproduced -> every gradient but dL_ddirect_shs is compared
#         on the full batch;
#   Q5   dL_ddirect_shs is a non-atomic `+=` from all P threads in the reference (:447), i.e. its value at P>1 is whatever
#         survives the race; the well-defined quantity is the sum of the per-Gaussian contributions, so the reference is
#         run on single Gaussians (P=1: no race) and their sum is compared with the HIP / oracle result on that subset;
#   Q2   with S_direct > S_incident the reference's loop runs past the incidents row (out-of-bounds) -> only
#         S_direct <= S_incident is compared with it.
# ---------------------------------------------------------------------------------------------------------------------
def _need_ref_shading():
    from oracle import reference_gpu as rg
    if not rg.shading_available():
        pytest.skip("oracle/_ref/libr3dg_reference_shading.so not built (python -m oracle.build_ref needs /root/reference)")
    return rg


@pytest.mark.parametrize("P,K,Si,Sd,Sv,train", [(2000, 24, 16, 16, 16, False), (500, 64, 16, 16, 16, True),
                                                 (300, 100, 16, 9, 4, False), (4096, 64, 16, 16, 16, False),
                                                 (1000, 384, 16, 16, 16, False)])
def test_contract_model_matches_real_reference(P, K, Si, Sd, Sv, train):
    rg = _need_ref_shading()
    from oracle import render_equation as ore
    from relightable3dgaussian_amd import shading_ops as so
    inp = _inputs(P, Si, Sd, Sv, seed=7 * P + K)
    cpu = [inp[k] for k in ORDER]
    d = [x.to(DEV) for x in cpu]
    pbr, dirs, dl = so.render_equation_forward(*d, K, train)
    rnd = so.render_equation_forward.last_rand if train else None
    r_pbr, r_dirs, r_dl = rg.render_equation_forward(*d, K, train, rnd)
    o_pbr, o_dirs, o_dl = ore.forward(*cpu, K, rnd.cpu() if train else None)
    torch.cuda.synchronize()
    for tag, (a_pbr, a_dirs, a_dl) in (("hip", (pbr, dirs, dl)), ("oracle", (o_pbr, o_dirs, o_dl))):
        _ok(tag + "/incident_dirs vs reference", torch.as_tensor(a_dirs), r_dirs, 0, 5e-5 if train else 5e-6)
        # (random-angle variant: the 1e-5 direction noise is amplified by the lobe sharpness -- sanity bound only)
        _ok(tag + "/pbr vs reference", torch.as_tensor(a_pbr), r_pbr, 3e-2 if train else 1e-4, 1e-6)
        _ok(tag + "/diffuse_light vs reference", torch.as_tensor(a_dl), r_dl, 1e-3 if train else 1e-4, 1e-6)
    names = ("pbr", "incident_dirs", "incident_lights", "local_lights", "global_lights", "visibility", "diffuse_light",
             "local_diffuse_light", "accum", "rgb_d", "rgb_s")
    if not train:
        outs = so.render_equation_forward_complex(*d, K)
        refs = rg.render_equation_forward_complex(*d, K)
        orcs = ore.forward_complex(*cpu, K)
        for name, a, o, b in zip(names, outs, orcs, refs):
            _ok("hip/complex/%s vs reference" % name, a, b, 1e-4, 5e-6)
            _ok("oracle/complex/%s vs reference" % name, torch.as_tensor(o).reshape(b.shape), b, 1e-4, 5e-6)
    g = torch.Generator().manual_seed(99)
    g_pbr, g_dl = torch.randn(P, 3, generator=g), torch.randn(P, 3, generator=g)
    # all three backward passes consume the REFERENCE forward's directions
    grads = so.render_equation_backward(*d, K, r_dirs, g_pbr.to(DEV), g_dl.to(DEV))
    refs = rg.render_equation_backward(*d, K, r_dirs, g_pbr.to(DEV), g_dl.to(DEV))
    orcs = ore.backward(*cpu, K, r_dirs.cpu(), g_pbr, g_dl)
    gnames = ("dL_dbase_color", "dL_droughness", "dL_dmetallic", "dL_dnormals", "dL_dviewdirs", "dL_dincidents_shs",
              "dL_ddirect_shs", "dL_dvisibility_shs")
    for name, a, o, b in zip(gnames, grads, orcs, refs):
        if name == "dL_ddirect_shs":
            continue                                                      # Q5, below
        _ok("hip/%s vs reference" % name, a, b, 5e-4, 1e-6)
        _ok("oracle/%s vs reference" % name, torch.as_tensor(np.asarray(o, np.float32)).reshape(b.shape), b, 5e-4, 1e-6)
    # Q5: race-free reference = one Gaussian per launch, summed
    sub = list(range(0, P, max(P // 12, 1)))[:12]
    acc = torch.zeros(1, Sd, 3, dtype=torch.float64, device=DEV)
    for i in sub:
        di = [x[i:i + 1] if x.shape[0] == P else x for x in d]
        acc += rg.render_equation_backward(*di, K, r_dirs[i:i + 1], g_pbr[i:i + 1].to(DEV), g_dl[i:i + 1].to(DEV))[6].double()
    idx = torch.tensor(sub)
    ds = [x[idx.to(DEV)] if x.shape[0] == P else x for x in d]
    ours = so.render_equation_backward(*ds, K, r_dirs[idx.to(DEV)], g_pbr[idx].to(DEV), g_dl[idx].to(DEV))[6]
    cs = [x[idx] if x.shape[0] == P else x for x in cpu]
    orc = ore.backward(*cs, K, r_dirs[idx.to(DEV)].cpu(), g_pbr[idx], g_dl[idx])[6]
    _ok("hip/dL_ddirect_shs vs race-free reference sum", ours, acc, 5e-4, 1e-6)
    _ok("oracle/dL_ddirect_shs vs race-free reference sum", torch.as_tensor(np.asarray(orc)).reshape(acc.shape), acc, 5e-4, 1e-6)
