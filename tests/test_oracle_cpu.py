"""CPU tests (no GPU): the oracle is checked against (a) golden vectors produced by running the reference's own
Python code (tests/golden/make_golden.py) and (b) an independent pure-PyTorch restatement whose autograd supplies
the backward; plus the C-ABI library must load and export every symbol include/r3dg_hip.h declares."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from tests.helpers import fwd_args, make_case, report

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _gold(name):
    return dict(np.load(os.path.join(GOLD, name)))


def _ok(name, got, ref, rtol, atol):
    ok, msg = report(name, got, ref, rtol, atol)
    print(msg)
    assert ok, msg


# ---------------------------------------------------------------- rasterizer oracle: C vs PyTorch restatement
def test_c_oracle_matches_torch_oracle_forward_and_backward():
    from oracle import rasterizer as orc, torch_rasterizer as trz
    case = make_case(P=2500, W=112, H=96, S=5, seed=1)
    a = fwd_args(case)
    out = orc.rasterize_gaussians(*a[:-3], want_margin=True)
    st = out[-1]
    leaf = {k: case[k].clone().requires_grad_(True) for k in ("means3D", "opacity", "scales", "rotations", "shs")}
    feat = case["features"].clone().requires_grad_(True)
    res = trz.rasterize(a[0], leaf["means3D"], feat, None, leaf["opacity"], leaf["scales"], leaf["rotations"], 1.0, None,
                        a[9], a[10], a[11], a[12], a[13], a[14], case["H"], case["W"], leaf["shs"], 3, a[19])
    assert out[0] == res["num_rendered"]
    assert np.array_equal(st["radii"], res["radii"].numpy())
    assert np.array_equal(st["tiles_touched"], res["pre"]["tiles_touched"].numpy().astype(np.uint32))
    assert np.array_equal(st["point_list"].astype(np.int64), res["binning"]["point_list"].numpy())
    assert np.array_equal(st["ranges"].astype(np.int64), res["binning"]["ranges"].numpy())
    mism = res["n_contrib"].numpy() != out[1]
    assert not (mism & (st["margin"] > 1e-4)).any()
    for name, idx, key in (("color", 2, "color"), ("opacity", 3, "opacity"), ("depth", 4, "depth"),
                           ("feature", 5, "feature"), ("surface_xyz", 7, "surface_xyz"), ("weights", 8, "weights")):
        _ok(name, res[key], out[idx], 2e-5, 1e-5)
    H, W, S = case["H"], case["W"], case["S"]
    g = torch.Generator().manual_seed(7)
    gC, gO, gD, gF = (torch.randn(c, H, W, generator=g) for c in (3, 1, 1, S))
    loss = (res["color"] * gC).sum() + (res["opacity"] * gO).sum() + (res["depth"] * gD).sum() + (res["feature"] * gF).sum()
    loss.backward()
    gr = orc.rasterize_gaussians_backward(a[0], a[1], a[2], out[9], None, a[5], a[6], 1.0, None, a[9], a[10], a[11],
                                          a[12], gC, gO, gD, gF, a[17], 3, a[19], st, True)
    d_mean2D, d_colors, d_opacity, d_means3D, d_feature, d_cov3D, d_sh, d_scales, d_rot, _ = gr
    _ok("dL_dmeans3D", d_means3D, leaf["means3D"].grad, 2e-4, 1e-6)
    _ok("dL_dopacity", d_opacity, leaf["opacity"].grad, 2e-4, 1e-6)
    _ok("dL_dscales", d_scales, leaf["scales"].grad, 2e-4, 1e-6)
    _ok("dL_drot", d_rot, leaf["rotations"].grad, 2e-4, 1e-6)
    _ok("dL_dsh", d_sh, leaf["shs"].grad, 2e-4, 1e-6)
    _ok("dL_dfeature", d_feature, feat.grad, 2e-4, 1e-6)
    _ok("dL_dmean2D.xy", d_mean2D[:, :2], res["pre"]["p_proj_xy"].grad, 2e-4, 1e-6)


def test_oracle_sort_is_stable_and_ranges_cover_list():
    from oracle import rasterizer as orc
    case = make_case(P=1500, W=100, H=70, S=0, seed=9, scale_log_mean=-2.5)
    st = orc.rasterize_gaussians(*fwd_args(case)[:-3])[-1]
    keys, vals = st["keys"], st["point_list"]
    assert (np.diff(keys.astype(np.int64)) >= 0).all()
    same = np.diff(keys.astype(np.int64)) == 0
    assert (np.diff(vals.astype(np.int64))[same] > 0).all(), "ties must keep ascending Gaussian index"
    rg = st["ranges"].astype(np.int64)
    assert (rg[:, 1] - rg[:, 0]).sum() == st["num_rendered"]


# ---------------------------------------------------------------- golden vectors from the reference's Python
def test_oracle_sh_matches_reference_eval_sh():
    from oracle import rasterizer as orc
    gd = _gold("eval_sh_reference.npz")
    sh = np.ascontiguousarray(gd["sh"].transpose(0, 2, 1))     # reference layout [P,3,16] -> op layout [P,16,3]
    P = sh.shape[0]
    for deg in range(4):
        rgb = np.zeros((P, 3), np.float32)
        clamped = np.zeros((P, 3), np.uint8)
        campos = np.zeros(3, np.float32)
        orc.lib().r3dgo_sh_to_rgb(P, deg, 16, gd["dirs"].ctypes.data_as(C.c_void_p), campos.ctypes.data_as(C.c_void_p),
                                  sh.ctypes.data_as(C.c_void_p), clamped.ctypes.data_as(C.c_void_p),
                                  rgb.ctypes.data_as(C.c_void_p))
        want = np.maximum(gd["deg%d" % deg] + 0.5, 0)
        _ok("sh deg %d" % deg, rgb, want, 0, 2e-6)
        assert np.array_equal(clamped.astype(bool), (gd["deg%d" % deg] + 0.5) < 0) or \
            np.abs((gd["deg%d" % deg] + 0.5))[clamped.astype(bool) != ((gd["deg%d" % deg] + 0.5) < 0)].max() < 1e-6


def test_oracle_cov3d_matches_reference_covariance():
    from oracle import rasterizer as orc, torch_rasterizer as trz
    gd = _gold("covariance_reference.npz")
    P = gd["scales"].shape[0]
    cov = np.zeros((P, 6), np.float32)
    orc.lib().r3dgo_cov3d(P, gd["scales"].ctypes.data_as(C.c_void_p), C.c_float(float(gd["modifier"])),
                          gd["rotations"].ctypes.data_as(C.c_void_p), cov.ctypes.data_as(C.c_void_p))
    _ok("cov3D (C oracle)", cov, gd["cov3D"], 1e-5, 1e-9)
    cov_t = trz.cov3d_from_scale_rot(torch.from_numpy(gd["scales"]), float(gd["modifier"]),
                                     torch.from_numpy(gd["rotations"]))
    _ok("cov3D (torch oracle)", cov_t, gd["cov3D"], 1e-5, 1e-9)


def test_oracle_shading_matches_reference_rendering_equation():
    from oracle import shading
    gd = _gold("shading_reference.npz")
    t = {k: torch.from_numpy(v) for k, v in gd.items()}
    leaves = {k: t[k].clone().requires_grad_(True) for k in ("base_color", "roughness", "viewdirs", "incidents", "env_raw")}
    env = torch.nn.functional.softplus(leaves["env_raw"])[0]
    out = shading.rendering_equation(leaves["base_color"], leaves["roughness"], t["normals"], leaves["viewdirs"],
                                     leaves["incidents"], env, t["visibility"], t["incident_dirs"], t["incident_areas"])
    for k_out, k_ref in (("pbr", "pbr"), ("diffuse_light", "diffuse_light"), ("specular", "specular"),
                         ("incident_lights", "incident_lights_mean"),
                         ("local_incident_lights", "local_incident_lights_mean"),
                         ("global_incident_lights", "global_incident_lights_mean"),
                         ("incident_visibility", "incident_visibility_mean")):
        _ok(k_out, out[k_out], t[k_ref], 2e-5, 1e-6)
    loss = (out["pbr"] * t["g_pbr"]).sum() + (out["diffuse_light"] * t["g_diffuse_light"]).sum()
    loss.backward()
    for k in ("base_color", "roughness", "viewdirs", "incidents", "env_raw"):
        _ok("d_" + k, leaves[k].grad, t["d_" + k], 1e-4, 1e-6)


def test_oracle_shading_matches_reference_on_the_fixed_ray_set_fixture():
    """tests/golden/shading_reference_frs.npz (make_frs_golden.py): the reference's rendering_equation on an UNPERTURBED Fibonacci
    ray set, shading normal != ray normal, ray normals next to -z, smooth Gaussians -- the fixture the fixed-ray-set kernels are
    pinned to on the GPU; here: the oracle reproduces it, and its ray set is the product's (sampling.py) ray set."""
    from oracle import shading
    from relightable3dgaussian_amd import sampling
    gd = _gold("shading_reference_frs.npz")
    t = {k: torch.from_numpy(v) for k, v in gd.items()}
    dirs, areas = sampling.fibonacci_sphere_sampling(t["ray_normals"], t["incident_dirs"].shape[1])
    assert float((dirs - t["incident_dirs"]).abs().max()) < 2e-6 and torch.equal(areas, t["incident_areas"])
    leaves = {k: t[k].clone().requires_grad_(True) for k in ("base_color", "roughness", "viewdirs", "incidents", "env_raw")}
    env = torch.nn.functional.softplus(leaves["env_raw"])[0]
    out = shading.rendering_equation(leaves["base_color"], leaves["roughness"], t["normals"], leaves["viewdirs"],
                                     leaves["incidents"], env, t["visibility"], t["incident_dirs"], t["incident_areas"])
    # (fp32 vs fp32 with smooth Gaussians: the GGX denominator is the ill-conditioned term, see tests/test_shading_gpu.py)
    _ok("pbr", out["pbr"], t["pbr"], 2e-4, 1e-6)
    _ok("diffuse_light", out["diffuse_light"], t["diffuse_light"], 2e-5, 1e-6)
    ((out["pbr"] * t["g_pbr"]).sum() + (out["diffuse_light"] * t["g_diffuse_light"]).sum()).backward()
    for k in ("base_color", "roughness", "viewdirs", "incidents", "env_raw"):
        _ok("d_" + k, leaves[k].grad, t["d_" + k], 1e-3, 1e-6)


def test_oracle_env_lookup_matches_reference_envlight():
    from oracle import shading
    gd = _gold("envlight_reference.npz")
    got = shading.env_lookup(torch.from_numpy(gd["envmap"]), torch.from_numpy(gd["dirs"]),
                             torch.from_numpy(gd["transform"]))
    _ok("EnvLight.direct_light", got, gd["light"], 2e-5, 1e-6)


def test_oracle_and_host_glue_match_the_reference_eval_frame():
    """tests/golden/pipeline_reference_relight.npz (make_relight_golden.py: the reference's render_view(is_training=False) with
    its EnvLight and a per-frame light.transform): the per-Gaussian 28-channel rows the reference's Python hands to the
    rasterizer equal oracle/shading.rendering_equation (float64, with the transform) on the fixture's caches -- the oracle the
    relight kernels are tested against is pinned on the eval branch too -- and the host-side composites
    (relight.env_directions + the sRGB curve: env_only / render_env / pbr_env, neilf.py:198-203) equal the reference's maps when
    fed the reference's own rasterizer outputs."""
    import types
    from oracle import shading
    from relightable3dgaussian_amd import relight
    from relightable3dgaussian_amd.synthetic import SynthCamera
    z = dict(np.load(os.path.join(GOLD, "pipeline_reference_relight.npz")))
    t = lambda k: torch.from_numpy(z[k])
    fovx, fovy, tanx, tany, cx, cy = [float(v) for v in z["cam_scalars"]]
    cam = SynthCamera(int(z["H"]), int(z["W"]), fovx, fovy, tanx, tany, cx, cy, t("wvt"), t("fpt"), t("campos"))
    assert np.abs(z["wvt"] - z["ref_wvt"]).max() < 2e-6 and np.abs(z["fpt"] - z["ref_fpt"]).max() < 2e-6
    d = lambda k: t(k).double()
    base = 0.03 + 0.77 * torch.sigmoid(d("raw_base_color"))
    rough = 0.09 + 0.9 * torch.sigmoid(d("raw_roughness"))
    normal = torch.nn.functional.normalize(t("raw_normal"), dim=-1, eps=1e-3).double()
    view = torch.nn.functional.normalize(d("campos") - d("raw_xyz"), dim=-1)
    inc = torch.cat([d("raw_incidents_dc"), d("raw_incidents_rest")], 1)
    for tag, tr in (("a", d("T_a")), ("b", d("T_b")), ("n", None)):
        ref = shading.rendering_equation(base, rough, normal, view, inc, d("envmap"), d("visibility"), d("incident_dirs"),
                                         d("incident_areas"), tr)
        f = z[tag + "_features"]
        depth = (torch.cat([d("raw_xyz"), torch.ones(f.shape[0], 1, dtype=torch.float64)], -1) @ d("wvt"))[:, 2:3]
        want = torch.cat([depth, depth.square(), ref["pbr"], normal, base, rough, ref["diffuse_light"], ref["specular"],
                          ref["incident_lights"], ref["local_incident_lights"], ref["global_incident_lights"],
                          ref["incident_visibility"]], -1)
        assert want.shape == f.shape == (f.shape[0], 28)
        for c0, c1, tol in ((0, 2, 1e-5), (2, 5, 5e-4), (5, 15, 1e-4), (15, 18, 5e-4), (18, 28, 1e-4)):     # (fp32 reference vs float64)
            ok, msg = report("%s rows[%d:%d]" % (tag, c0, c1), f[:, c0:c1], want[:, c0:c1], tol, 1e-6)
            assert ok, msg
        trf = None if tr is None else tr.float()
        env_rgb = relight.env_directions(cam, t("envmap"), trf)
        ok, msg = report(tag + " env_only", relight.rgb_to_srgb(env_rgb), z[tag + "_map_env_only"], 0.0, 2e-4)      # (fp32: c2w as an inverse there, a transpose here; sRGB slope 12.92 near black)
        assert ok, msg
        if tag == "a":
            op, img = t("a_map_opacity"), t("a_map_render")
            feat = t("a_feature_image") / op.clamp_min(1e-5) * (t("a_num_contrib") > 0)
            ok, msg = report("a render_env", img + (1 - op) * relight.rgb_to_srgb(env_rgb), z["a_map_render_env"], 0.0, 2e-4)
            assert ok, msg
            ok, msg = report("a pbr_env", relight.rgb_to_srgb(feat[2:5] * op + (1 - op) * env_rgb), z["a_map_pbr_env"], 0.0, 2e-4)
            assert ok, msg


def test_oracle_ray_set_matches_reference_fibonacci():
    from oracle import shading
    gd = _gold("fibonacci_reference.npz")
    dirs, areas = shading.fibonacci_sphere_sampling(torch.from_numpy(gd["normals"]), gd["dirs"].shape[1])
    _ok("fibonacci dirs", dirs, gd["dirs"], 0, 2e-6)
    _ok("fibonacci areas", areas, gd["areas"], 0, 1e-6)


# ---------------------------------------------------------------- the C ABI
def test_hip_library_exports_every_declared_symbol():
    from relightable3dgaussian_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = set()
    for fn in os.listdir(os.path.join(root, "include")):
        if fn.endswith(".h"):
            src = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", fn)).read(), flags=re.S)
            names |= set(re.findall(r"\b(r3dg_\w+)\s*\(", src))
    names.discard("r3dg_alloc_fn")
    assert len(names) >= 15
    L = C.CDLL(_lib.LIB_PATH)
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, "declared in include/*.h but not exported: %s" % missing
    lib = _lib.lib()                                        # argtypes resolve for every bound symbol
    assert lib.r3dg_version() >= 100
    assert lib.r3dg_geometry_state_bytes(1000) > 0 and lib.r3dg_max_features_forward() >= 33
    # enum r3dg_option <-> _lib.OPTIONS (host calls only: no device involved), range checks, round trip
    hdr = open(os.path.join(root, "include", "r3dg_hip.h")).read()
    enum = re.search(r"enum r3dg_option \{(.*?)\};", hdr, re.S).group(1)
    declared = [n[len("R3DG_OPT_"):] for n in re.findall(r"\b(R3DG_OPT_\w+)\b", re.sub(r"/\*.*?\*/", "", enum, flags=re.S))]
    assert declared[-1] == "COUNT" and tuple(declared[:-1]) == _lib.OPTIONS
    assert _lib.get_option("TILE_BINNING") == 2 and _lib.get_option("CULL") == 1 and _lib.get_option("RESERVE_CUS") == 0
    _lib.set_option("RESERVE_CUS", 8)
    assert _lib.get_option("RESERVE_CUS") == 8
    _lib.set_option("RESERVE_CUS", 0)
    for bad in ((len(_lib.OPTIONS), 0), (-1, 0), (_lib.OPTIONS.index("TILE_BINNING"), 3), (_lib.OPTIONS.index("CULL"), -1)):
        assert lib.r3dg_set_option(*bad) != 0
    assert _lib.get_option("TILE_BINNING") == 2 and lib.r3dg_bounded_forward_supported(800, 800) == 1
    assert lib.r3dg_bounded_forward_supported(2560, 1664) == 0 and lib.r3dg_shade_frs_supported(64, 16, 16, 32) == 1
    assert lib.r3dg_shade_frs_supported(30, 16, 16, 32) == 0 and lib.r3dg_shade_frs_tables_bytes(64) == 4 * 512 * 4


def test_option_context_restore_stack_is_per_thread():
    """the library's current context is thread-local, so the Python object's restore stack has to be too.  Two threads
    enter the SAME context object in an interleaved order (A in, B in, A out, B out -- with one shared stack A would pop B's saved
    handle); afterwards each thread is back at the context it started from, and the context sees its own values inside.  No
    kernel is launched: option contexts are host state."""
    import ctypes as C
    import threading
    from relightable3dgaussian_amd import _lib
    outer, ctx = _lib.OptionContext(CULL=0), _lib.OptionContext(CULL=1, RESERVE_CUS=8)
    step = [threading.Event() for _ in range(4)]
    seen = {}

    def current():
        prev = C.c_void_p()
        _lib.check(_lib.lib().r3dg_context_make_current(None, C.byref(prev)), "probe")      # -> previous; install it again
        _lib.check(_lib.lib().r3dg_context_make_current(prev, None), "probe")
        return prev.value

    def thread_a():
        with outer:                                    # A starts inside another context, B at the process defaults
            base = current()
            with ctx:
                step[0].set(); step[1].wait(5)         # B enters while A is inside
                seen["a_in"] = current()
            step[2].set(); step[3].wait(5)             # A has left; B still inside, then leaves
            seen["a_out"] = (current(), base)

    def thread_b():
        step[0].wait(5)
        base = current()
        with ctx:
            step[1].set(); step[2].wait(5)
            seen["b_in"] = current()
        seen["b_out"] = (current(), base)
        step[3].set()

    ta, tb = threading.Thread(target=thread_a), threading.Thread(target=thread_b)
    ta.start(); tb.start(); ta.join(10); tb.join(10)
    assert not ta.is_alive() and not tb.is_alive()
    assert seen["a_in"] == seen["b_in"] == ctx._h
    assert seen["a_out"][0] == seen["a_out"][1] == outer._h and seen["b_out"][0] == seen["b_out"][1] and not seen["b_out"][0]
    assert not _lib._live_entries                       # nobody is inside any context any more


def test_host_mirror_rejects_bad_inputs_without_gpu():
    from relightable3dgaussian_amd import rasterizer_ops
    import pytest
    with pytest.raises(RuntimeError):
        rasterizer_ops.rasterize_gaussians(torch.zeros(3), torch.zeros(10, 2), torch.zeros(10, 0), torch.Tensor([]),
                                           torch.zeros(10, 1), torch.zeros(10, 3), torch.zeros(10, 4), 1.0,
                                           torch.Tensor([]), torch.eye(4), torch.eye(4), 1.0, 1.0, 8.0, 8.0, 16, 16,
                                           torch.zeros(10, 16, 3), 3, torch.zeros(3), False, True, False)
    with pytest.raises(RuntimeError):      # CPU tensors are refused: there is no CPU fallback
        rasterizer_ops.rasterize_gaussians(torch.zeros(3), torch.zeros(10, 3), torch.zeros(10, 0), torch.Tensor([]),
                                           torch.zeros(10, 1), torch.zeros(10, 3), torch.zeros(10, 4), 1.0,
                                           torch.Tensor([]), torch.eye(4), torch.eye(4), 1.0, 1.0, 8.0, 8.0, 16, 16,
                                           torch.zeros(10, 16, 3), 3, torch.zeros(3), False, True, False)


# ---------------------------------------------------------------- BVH
def test_leaf_boxes_match_reference_raytracer_init():
    from oracle import bvh as ob
    from relightable3dgaussian_amd import bvh as host_bvh
    gd = _gold("bvh_leaf_reference.npz")
    nodes, aabbs = ob.leaf_boxes(gd["means3D"], gd["scales"], gd["rotations"])
    assert np.array_equal(nodes, gd["nodes_init"]) and np.array_equal(aabbs, gd["aabbs_init"])
    n2, a2 = host_bvh.leaf_boxes(*(torch.from_numpy(gd[k]) for k in ("means3D", "scales", "rotations")))
    assert np.array_equal(n2.numpy(), gd["nodes_init"]) and np.array_equal(a2.numpy(), gd["aabbs_init"])


def _bvh_case(P, seed, K=8, dup=False):
    from oracle import bvh as ob, shading
    from oracle.torch_rasterizer import cov3d_from_scale_rot
    from relightable3dgaussian_amd import synthetic as syn
    sc = syn.make_scene(P=max(P, 4), seed=seed, scale_log_mean=-3.2)
    sc = {k: (v[:P] if torch.is_tensor(v) and v.shape[0] >= P and k != "env" else v) for k, v in sc.items()}
    if dup and P > 10:                       # coincident Gaussians -> identical Morton codes (tie-break on index)
        sc["xyz"][P // 2:P // 2 + 5] = sc["xyz"][0]
        sc["scales"][P // 2:P // 2 + 5] = sc["scales"][0]
        sc["rotations"][P // 2:P // 2 + 5] = sc["rotations"][0]
    dirs, _ = shading.fibonacci_sphere_sampling(sc["normal"], K)
    cinv = cov3d_from_scale_rot(1 / sc["scales"], 1.0, sc["rotations"]).contiguous()
    rays_o = (sc["xyz"][:, None, :] + 0.05 * dirs).contiguous()
    return sc, dirs.contiguous(), cinv, rays_o


def test_bvh_oracle_tree_is_wellformed_and_trace_equals_bruteforce():
    from oracle import bvh as ob
    for P, seed, dup in ((1, 0, False), (2, 1, False), (3, 2, False), (1500, 3, True)):
        sc, dirs, cinv, rays_o = _bvh_case(P, seed, dup=dup)
        nodes, aabbs = ob.leaf_boxes(sc["xyz"].numpy(), sc["scales"].numpy(), sc["rotations"].numpy())
        n2, a2, m = ob.create_bvh(nodes, aabbs)
        assert n2[0, 0] == -1 and n2[0, 4] == P and (np.diff(m.astype(np.int64)) > 0).all()
        if P > 1:
            for i in range(P - 1):          # parent links, box containment, leaf counts
                l, r = n2[i, 1], n2[i, 2]
                assert n2[l, 0] == i and n2[r, 0] == i and n2[i, 4] == n2[l, 4] + n2[r, 4]
                assert (a2[i, :3] <= np.minimum(a2[l, :3], a2[r, :3])).all() and (a2[i, 3:] >= np.maximum(a2[l, 3:], a2[r, 3:])).all()
        assert sorted(n2[P - 1:, 3].tolist()) == list(range(P))
        args = (n2, a2, rays_o.numpy(), dirs.numpy(), sc["xyz"].numpy(), cinv.numpy(), sc["opacity"][:, 0].numpy(),
                sc["normal"].numpy())
        cnt, vis = ob.trace_bvh_opacity(*args)
        cb, prod = ob.trace_bruteforce(*args)
        want = np.where(prod < 0.9, 0.0, prod)
        near = np.abs(prod - 0.9) < 1e-5
        assert np.abs(vis - want)[~near].max() < 1e-5
        assert np.array_equal(cnt[vis > 0], cb[vis > 0])


def _transport_case(P, K, seed=3):
    from relightable3dgaussian_amd import sampling
    g = torch.Generator().manual_seed(seed)
    normals = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1, eps=1e-3)
    normals[0] = torch.tensor([0.0, 0.0, -1.0])                      # the n_z + 1 <= 0 branch of rotation_between_z
    normals[1] = torch.tensor([0.0, 0.0, 1.0])
    base = 0.03 + 0.77 * torch.rand(P, 3, generator=g)
    rough = 0.09 + 0.9 * torch.rand(P, 1, generator=g)
    view = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1)
    inc = 0.4 * torch.randn(P, 16, 3, generator=g)
    env = 2.0 * torch.rand(16, 32, 3, generator=g)
    vis = (torch.rand(P, K, 1, generator=g) > 0.3).float() * torch.rand(P, K, 1, generator=g)
    dirs, areas = sampling.fibonacci_sphere_sampling(normals, K)
    zs = sampling.fibonacci_z_samples(K, "cpu")[0].t().contiguous()               # [K,3], row k = (x_k, y_k, z_k)
    return dict(normals=normals, base=base, rough=rough, view=view, inc=inc, env=env, vis=vis, dirs=dirs, areas=areas, zs=zs)


def _transport_in_torch(c):
    """shade_build_transport_kernel + shade_forward_transport_kernel (csrc/shading_transport.hpp), statement for statement."""
    import math
    from oracle import shading as osh
    n, dirs, vis, areas, inc, zs = c["normals"], c["dirs"], c["vis"], c["areas"], c["inc"], c["zs"]
    P = n.shape[0]
    # --- builder: radiance -> transport in place, 13 constants per Gaussian
    radiance = osh.env_lookup(c["env"], dirs)                                      # what r3dg_shade_build_taps caches
    Y = osh.sh_basis(3, dirs)
    loc = torch.einsum("pkm,pmc->pkc", Y, inc).clamp_min(0)
    glob = radiance * vis
    lin = loc + glob
    area_ndi = areas * (n[:, None] * dirs).sum(-1, keepdim=True).clamp_min(0)
    transport = lin * area_ndi
    consts = torch.cat([transport.mean(1), lin.mean(1), loc.mean(1), glob.mean(1), vis.mean(1)], -1)     # [P,13]
    # --- frame kernel: per-Gaussian setup (gauss_setup), the rotation, the loop over the table
    v1, v2, cp = -n[:, 1], n[:, 0], (n[:, 2] + 1).clamp_min(1e-7)
    regular = n[:, 2] + 1 > 0
    one, zero = torch.ones(P), torch.zeros(P)
    R = [torch.where(regular, 1 + (-v2 * v2) / cp, -one), torch.where(regular, v1 * v2 / cp, zero), torch.where(regular, v2, zero),
         None, torch.where(regular, 1 + (-v1 * v1) / cp, -one), torch.where(regular, -v1, zero),
         torch.where(regular, -v2, zero), torch.where(regular, v1, zero),
         torch.where(regular, 1 + (-v2 * v2 - v1 * v1) / cp, -one)]
    R[3] = R[1]
    rx = R[0][:, None] * zs[None, :, 0] + R[1][:, None] * zs[None, :, 1] + R[2][:, None] * zs[None, :, 2]
    ry = R[3][:, None] * zs[None, :, 0] + R[4][:, None] * zs[None, :, 1] + R[5][:, None] * zs[None, :, 2]
    rz = R[6][:, None] * zs[None, :, 0] + R[7][:, None] * zs[None, :, 1] + R[8][:, None] * zs[None, :, 2]
    raw = torch.stack([rx, ry, rz], -1)
    L = raw * torch.rsqrt((raw * raw).sum(-1, keepdim=True).clamp_min(1e-24))
    view, rough, base = c["view"], c["rough"], c["base"]
    V = view / view.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    N0 = n / n.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    N = N0 * torch.sign((V * N0).sum(-1, keepdim=True))
    NoV = (N * V).sum(-1).clamp(1e-6, 1.0)
    a = rough[:, 0] ** 2
    a2, kk = a * a, (a + 2 * rough[:, 0] + 1.0) / 8.0
    nom1 = NoV * (1 - kk) + kk
    u = (L + V[:, None]) / 2.0
    Hh = u * torch.rsqrt((u * u).sum(-1, keepdim=True).clamp_min(1e-24))
    NoL = (N[:, None] * L).sum(-1).clamp(1e-6, 1.0)
    NoH = (N[:, None] * Hh).sum(-1).clamp(1e-6, 1.0)
    VoH = (V[:, None] * Hh).sum(-1).clamp(1e-6, 1.0)
    p2 = torch.exp2((-5.55473 * VoH - 6.98316) * VoH)
    frac = (0.04 + 0.96 * p2) * a2[:, None]
    nom0 = NoH * NoH * (a2[:, None] - 1) + 1
    nom2 = NoL * (1 - kk[:, None]) + kk[:, None]
    nom = (4 * math.pi * nom0 * nom0 * nom1[:, None] * nom2).clamp(1e-6, 4 * math.pi)
    spec = frac / nom
    S = (spec[..., None] * transport).mean(1)
    out = torch.cat([base / math.pi * consts[:, 0:3] + S, consts[:, 0:3], S, consts[:, 3:6], consts[:, 6:9], consts[:, 9:12],
                     consts[:, 12:13]], -1)
    return radiance, transport, consts, L, out


def test_transport_cache_formulation_equals_the_rendering_equation():
    """The opt-in relight cache (csrc/shading_transport.hpp) regroups rendering_equation (neilf.py:339-371) into a
    view-independent part and a per-frame GGX sum and regenerates each direction from the normal and the K-entry Fibonacci
    table.  That arithmetic, transcribed in torch, must reproduce the oracle's 19 shading outputs, and the regenerated
    directions must be the cached ones -- including the table's [K,3] layout and the rotation's orientation."""
    from oracle import shading as osh
    c = _transport_case(300, 37)
    want = osh.rendering_equation(c["base"], c["rough"], c["normals"], c["view"], c["inc"], c["env"], c["vis"], c["dirs"],
                                  c["areas"])
    _rad, _tr, _consts, L, out = _transport_in_torch(c)
    assert float((L - c["dirs"]).abs().max()) < 2e-6, "regenerated directions differ from the cached ones"
    ref = torch.cat([want["pbr"], want["diffuse_light"], want["specular"], want["incident_lights"],
                     want["local_incident_lights"], want["global_incident_lights"], want["incident_visibility"]], -1)
    err = (out - ref).abs().max(0).values / ref.abs().max(0).values.clamp_min(1e-6)
    # the view-independent columns are the same sums; the GGX columns see the regenerated directions (1e-7 off the cached
    # ones) through an ill-conditioned lobe: 1e-4, the bound the shading parity tests use for that term
    assert out.shape == (300, 19) and float(err[[3, 4, 5] + list(range(9, 19))].max()) < 2e-6, err
    assert float(err[[0, 1, 2, 6, 7, 8]].max()) < 1e-4, err


@pytest.mark.parametrize("P,K,uniform,regen", [(130, 37, True, True), (9, 100, False, False), (64, 64, True, False)])
def test_transport_kernels_source_runs_in_the_cpu_wave_emulation(tmp_path, P, K, uniform, regen):
    """The two kernels of csrc/shading_transport.hpp -- the files hipcc compiles for gfx950 -- compiled for the HOST against
    tests/emu/hip_emu.hpp (every lane a thread, __shfl_xor / __shared__ emulated in lock step) and run on the CPU: indexing,
    the in-place radiance -> transport rewrite, partially filled last waves and workgroups, the wave reductions and the 19
    output columns, against the torch transcription above.  (Written without GPU access; the hardware run of the same source
    is tests/test_relight_gpu.py.)"""
    import ctypes as C
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("needs g++ (C++20)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_path = os.path.join(tmp_path, "libtransport_emu.so")
    r = subprocess.run(["g++", "-std=c++20", "-O1", "-pthread", "-shared", "-fPIC", "-w",
                        "-I", os.path.join(root, "relightable3dgaussian_amd", "csrc"),
                        os.path.join(root, "tests", "emu", "transport_emu.cpp"), "-o", lib_path], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lib = C.CDLL(lib_path)
    c = _transport_case(P, K, seed=5)
    if not uniform:
        c["areas"] = c["areas"] * (0.5 + torch.rand(P, K, 1, generator=torch.Generator().manual_seed(1)))
    radiance, transport, consts, _L, out = _transport_in_torch(c)
    f = lambda t: np.ascontiguousarray(t.detach().numpy(), np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    buf = f(radiance).copy()                                        # [P,K,3]: radiance in, transport out
    got_consts = np.full((P, 16), np.nan, np.float32)
    normals, inc, vis, dirs, areas = f(c["normals"]), f(c["inc"]), f(c["vis"]), f(c["dirs"]), f(c["areas"])
    lib.emu_shade_build_transport(C.c_int(P), C.c_int(K), C.c_int(16), p(normals), p(inc), p(vis), p(dirs),
                                  None if uniform else p(areas), C.c_float(float(c["areas"][0, 0, 0])), p(buf), p(got_consts))
    scale = float(transport.abs().max())
    assert np.abs(buf - f(transport)).max() <= 2e-6 * scale, "transport"
    assert np.abs(got_consts[:, :13] - f(consts)).max() <= 2e-6 * float(consts.abs().max()) and (got_consts[:, 13:] == 0).all()
    base, rough, view, zs = f(c["base"]), f(c["rough"]), f(c["view"]), f(c["zs"])
    got = np.full((P, 19), np.nan, np.float32)
    lib.emu_shade_forward_transport(C.c_int(P), C.c_int(K), p(base), p(rough), p(normals), p(view), p(buf), p(got_consts),
                                    p(zs), None if regen else p(dirs), p(got))
    want = f(out)
    err = np.abs(got - want).max(0) / np.maximum(np.abs(want).max(0), 1e-6)
    assert np.isfinite(got).all() and err[[3, 4, 5] + list(range(9, 19))].max() < 2e-6, err        # copied constants
    assert err[[0, 1, 2, 6, 7, 8]].max() < 1e-4, err                # the GGX columns (ill-conditioned lobe, other rsqrt / exp2)
