"""GPU parity: HIP rasterizer (through the C ABI / `r3dg_rasterization._C`) vs the CPU oracle.

Tolerances (stated per buffer):
  * integer work -- radii, tiles_touched, point_offsets, sorted keys, point_list, ranges, num_rendered: BIT-EXACT
  * n_contrib: exact except pixels whose oracle threshold margin is < 1e-4 (a decision that depends on the
    last ulps of exp(); the oracle reports the margin per pixel)
  * forward float buffers: |err| <= 1e-5 + 2e-5*max|ref|;  weights (float atomics): 1e-4 relative
  * gradients (float atomics vs double-accumulated oracle): |err| <= 1e-6 + 2e-3*max|ref|
"""
import numpy as np
import pytest
import torch

from tests.helpers import fwd_args, make_case, report, to_np

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _opt(**kw):
    """r3dg_set_option by name (include/r3dg_hip.h enum r3dg_option)."""
    from relightable3dgaussian_amd import _lib
    for k, v in kw.items():
        _lib.set_option(k, v)


def _run_forward(case, debug=False):
    from r3dg_rasterization import _C
    return _C.rasterize_gaussians(*fwd_args(case, DEV, debug))


def _oracle_forward(case):
    from oracle import rasterizer as orc
    return orc.rasterize_gaussians(*fwd_args(case)[:-3], want_margin=True)


def _check_forward(case, label=""):
    from relightable3dgaussian_amd.rasterizer_ops import decode_state
    out = _run_forward(case)
    torch.cuda.synchronize()
    ref = _oracle_forward(case)
    st_ref = ref[-1]
    P, H, W = case["P"], case["H"], case["W"]
    R = out[0]
    msgs, ok_all = [], True
    assert R == ref[0], "%s num_rendered %d vs oracle %d" % (label, R, ref[0])
    st = decode_state(out[10], out[11], out[12], P, R, H, W)

    def exact(name, got, want):
        nonlocal ok_all
        got, want = to_np(got), np.asarray(want)
        same = np.array_equal(got.astype(np.int64), want.astype(np.int64))
        msgs.append("%-14s exact=%s%s" % (name, same, "" if same else "  mismatches=%d first=%s" % (
            (got.astype(np.int64) != want.astype(np.int64)).sum(),
            np.argwhere(got.astype(np.int64) != want.astype(np.int64))[:5].tolist())))
        ok_all &= same

    exact("radii", out[9], st_ref["radii"])
    exact("tiles_touched", st["tiles_touched"], st_ref["tiles_touched"])
    exact("point_offsets", st["point_offsets"], st_ref["offsets"] if "offsets" in st_ref else np.zeros(P))
    vis = st_ref["radii"] > 0
    # depth bits decide the sort order: must be bit-identical on visible Gaussians
    exact("depth bits", to_np(st["depths"]).view(np.int32)[vis], st_ref["depths"].view(np.int32)[vis])
    if R > 0:
        exact("keys", st["keys"], st_ref["keys"].astype(np.int64))
        exact("point_list", st["point_list"], st_ref["point_list"])
    exact("ranges", st["ranges"], st_ref["ranges"])

    def close(name, got, want, rtol, atol):
        nonlocal ok_all
        ok, m = report(name, got, want, rtol, atol)
        msgs.append(m)
        ok_all &= ok

    close("means2D", to_np(st["means2D"])[vis], st_ref["means2D"][vis], 0, 0)     # same op order, no FMA: exact
    close("conic_opacity", to_np(st["conic_opacity"])[vis], st_ref["conic_opacity"][vis], 0, 0)
    if case["colors"] is None:
        close("rgb", to_np(st["rgb"])[vis], st_ref["rgb"][vis], 0, 0)
        exact("clamped", to_np(st["clamped"])[vis], st_ref["clamped"][vis])
    if case["cov3D"] is None:
        close("cov3D", to_np(st["cov3D"])[vis], st_ref["cov3D"][vis], 0, 0)

    nc, nc_ref = to_np(out[1]), ref[1]
    mism = nc != nc_ref
    border = st_ref["margin"] < 1e-4
    hard = mism & ~border
    msgs.append("n_contrib      mismatches %d (borderline %d, hard %d)" % (mism.sum(), (mism & border).sum(), hard.sum()))
    ok_all &= not hard.any()
    # float buffers are compared where the discrete outcome provably agrees: a borderline alpha-vs-1/255 decision
    # adds or drops a whole alpha*T*c term (up to 4e-3) without necessarily changing n_contrib
    good = ~mism & ~border
    for name, idx in (("color", 2), ("opacity", 3), ("depth", 4), ("feature", 5)):
        g_, r_ = to_np(out[idx]), ref[idx]
        if g_.size:
            close(name, g_[:, good], r_[:, good], 2e-5, 1e-5)
    close("final_T", to_np(st["final_T"])[good], st_ref["final_T"][good], 2e-5, 1e-6)
    # pseudo normal / surface xyz depend on neighbours: compare where the 3x3 neighbourhood agrees
    nb = np.ones_like(good)
    pad = np.pad(good, 1, mode="edge")
    for dy in range(3):
        for dx in range(3):
            nb &= pad[dy:dy + H, dx:dx + W]
    close("surface_xyz", to_np(out[7])[:, good], ref[7][:, good], 2e-5, 1e-5)
    # The pseudo normal is normalize(ga x gb) of finite differences of surface_xyz: its conditioning is
    # (|ga|+|gb|)/|ga x gb|.  With delta = the fp32 noise of the differences (3e-6 * max|xyz|) the admissible
    # error per pixel is 2e-3 + 10*delta*(|ga|+|gb|)/|ga x gb|; flat/degenerate pixels are thereby exempt.
    xyz = ref[7].astype(np.float64)
    pz = np.pad(xyz, ((0, 0), (1, 1), (1, 1)), mode="edge")

    def sh(dy, dx):
        return pz[:, 1 + dy:1 + dy + H, 1 + dx:1 + dx + W]
    ga = -0.125 * sh(-1, -1) + 0.125 * sh(-1, 1) - 0.25 * sh(0, -1) + 0.25 * sh(0, 1) - 0.125 * sh(1, -1) + 0.125 * sh(1, 1)
    gb = -0.125 * sh(-1, -1) - 0.25 * sh(-1, 0) - 0.125 * sh(-1, 1) + 0.125 * sh(1, -1) + 0.25 * sh(1, 0) + 0.125 * sh(1, 1)
    n0 = np.linalg.norm(np.cross(ga, gb, axis=0), axis=0)
    delta = 3e-6 * max(np.abs(xyz).max(), 1e-30)
    nbound = 2e-3 + 10 * delta * (np.linalg.norm(ga, axis=0) + np.linalg.norm(gb, axis=0)) / np.maximum(n0, 1e-300)
    nerr = np.abs(to_np(out[6]).astype(np.float64) - ref[6]).max(0)
    nbad = nb & (nerr > nbound) & (n0 > 0)
    msgs.append("normal         bad %d/%d (max err on well-conditioned pixels %.3e)" % (
        nbad.sum(), nb.sum(), nerr[nb & (nbound < 1e-2)].max() if (nb & (nbound < 1e-2)).any() else 0.0))
    ok_all &= not nbad.any()
    close("weights", out[8], ref[8], 1e-4, 1e-6)
    text = "\n".join(["[%s] P=%d %dx%d S=%d R=%d" % (label, P, W, H, case["S"], R)] + msgs)
    print(text)
    assert ok_all, text
    return out, ref


CASES = {
    "sh_scale_rot_S5": dict(S=5),
    "S0": dict(S=0),
    "S16": dict(S=16, seed=2),
    "S28": dict(S=28, seed=3),
    "S33": dict(S=33, seed=4, P=1500),
    "colors_precomp": dict(S=3, use_colors=True),
    "cov_precomp": dict(S=4, use_cov=True),
    "ragged_image": dict(S=5, W=200, H=120, seed=5),
    "big_splats": dict(S=5, scale_log_mean=-1.5, P=800, seed=6),
    "camera_inside": dict(S=5, eye=(0.2, 0.1, 0.0), seed=7),
    "black_bg_deg1": dict(S=2, bg=(0.0, 0.0, 0.0), sh_degree=1),
    # fewer Gaussians than tiles: the bounded forward's projection cannot zero every tile counter itself (memset instead)
    "few_gaussians_many_tiles": dict(S=5, P=150, W=640, H=480, seed=8),
}


@pytest.mark.parametrize("name", list(CASES))
def test_forward_parity(name):
    _check_forward(make_case(**CASES[name]), name)


def test_forward_empty_and_culled():
    from r3dg_rasterization import _C
    case = make_case(P=64, S=5)
    # P == 0: kernels skipped, zero outputs (rasterize_points.cu:92)
    empty = dict(case)
    for k in ("means3D", "opacity", "scales", "rotations"):
        empty[k] = case[k][:0]
    empty["features"] = case["features"][:0]
    empty["shs"] = case["shs"][:0]
    empty["P"] = 0
    out = _C.rasterize_gaussians(*fwd_args(empty, DEV))
    assert out[0] == 0 and float(out[2].abs().sum()) == 0.0 and out[1].shape == (case["H"], case["W"])
    # everything behind the camera: nothing rendered, colour == background
    behind = dict(case)
    behind["means3D"] = case["means3D"] + torch.tensor([100.0, 0, 0])
    out = _C.rasterize_gaussians(*fwd_args(behind, DEV))
    assert out[0] == 0
    assert torch.allclose(out[2].cpu(), case["bg"][:, None, None].expand(3, case["H"], case["W"]))
    assert int(out[9].abs().sum()) == 0


def test_forward_bad_shape_raises():
    from r3dg_rasterization import _C
    case = make_case(P=64, S=5)
    args = list(fwd_args(case, DEV))
    args[1] = args[1][:, :2].contiguous()
    with pytest.raises(RuntimeError):
        _C.rasterize_gaussians(*args)
    case = make_case(P=64, S=40)
    with pytest.raises(RuntimeError):
        _C.rasterize_gaussians(*fwd_args(case, DEV))


def _check_backward(case, label, backward_geometry=True):
    from oracle import rasterizer as orc
    from r3dg_rasterization import _C
    out, ref = _check_forward(case, label + "/fwd")
    H, W, S = case["H"], case["W"], case["S"]
    g = torch.Generator().manual_seed(123)
    gC, gO, gD = torch.randn(3, H, W, generator=g), torch.randn(1, H, W, generator=g), torch.randn(1, H, W, generator=g)
    gF = torch.randn(S, H, W, generator=g)
    a = fwd_args(case, DEV)
    grads = _C.rasterize_gaussians_backward(a[0], a[1], a[2], out[9], a[3], a[5], a[6], 1.0, a[8], a[9], a[10], a[11],
                                            a[12], gC.to(DEV), gO.to(DEV), gD.to(DEV), gF.to(DEV), a[17], a[18], a[19],
                                            out[10], out[0], out[11], out[12], backward_geometry, False)
    torch.cuda.synchronize()
    c = fwd_args(case)
    # the oracle backward walks the ORACLE's forward state; borderline pixels (threshold margin < 1e-4) are zeroed in both
    # upstream gradients so both sides differentiate the same discrete structure
    nc_same = torch.from_numpy((to_np(out[1]) == ref[1]) & ~(ref[-1]["margin"] < 1e-4))
    if not bool(nc_same.all()):
        mask = nc_same[None].float()
        gC, gO, gD, gF = gC * mask, gO * mask, gD * mask, gF * mask
        grads = _C.rasterize_gaussians_backward(a[0], a[1], a[2], out[9], a[3], a[5], a[6], 1.0, a[8], a[9], a[10],
                                                a[11], a[12], gC.to(DEV), gO.to(DEV), gD.to(DEV), gF.to(DEV), a[17],
                                                a[18], a[19], out[10], out[0], out[11], out[12], backward_geometry,
                                                False)
        torch.cuda.synchronize()
    oref = orc.rasterize_gaussians_backward(c[0], c[1], c[2], ref[9], c[3], c[5], c[6], 1.0, c[8], c[9], c[10], c[11],
                                            c[12], gC, gO, gD, gF, c[17], c[18], c[19], ref[-1], backward_geometry)
    names = ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dfeatures", "dL_dcov3D", "dL_dsh",
             "dL_dscales", "dL_drotations"]
    msgs, ok_all = [], True
    for name, got, want in zip(names, grads, oref[:9]):
        ok, m = report(name, got, np.asarray(want).reshape(tuple(got.shape)), 2e-3, 1e-6)
        msgs.append(m)
        ok_all &= ok
    text = "\n".join(["[%s] backward" % label] + msgs)
    print(text)
    assert ok_all, text


BWD_CASES = {
    "S5": dict(S=5),
    "S0": dict(S=0, seed=21),
    "S16": dict(S=16, seed=22),
    "S24": dict(S=24, seed=23, P=1500),
    "S33": dict(S=33, seed=24, P=1000),
    "colors_precomp": dict(S=3, use_colors=True, seed=25),
    "cov_precomp": dict(S=4, use_cov=True, seed=26),
    "ragged_image": dict(S=5, W=200, H=120, seed=27),
    "big_splats": dict(S=5, scale_log_mean=-1.5, P=800, seed=28),
}


@pytest.mark.parametrize("name", list(BWD_CASES))
def test_backward_parity(name):
    _check_backward(make_case(**BWD_CASES[name]), name)


def test_backward_no_geometry_flag():
    _check_backward(make_case(S=5, seed=31), "bg_geom_off", backward_geometry=False)


@pytest.mark.parametrize("order", [0, 1])
def test_parity_tile_order(order, hip_lib):
    """The block -> tile map (longest-tile-first or natural) is a scheduling knob: it must not change results."""
    _opt(TILE_ORDER=order)
    try:
        _check_backward(make_case(S=16, seed=61, P=4000), "order%d" % order)
        _check_forward(make_case(S=5, seed=72, scale_log_mean=-2.0, P=1500), "order%d_big" % order)
    finally:
        _opt(TILE_ORDER=1)


@pytest.mark.parametrize("name", ["S16", "big_splats", "ragged_image", "camera_inside", "S0", "S28", "S33"])
def test_forward_without_the_block_cull(name, hip_lib):
    """The conservative per-block cull of the tile forward (render_forward_wave_kernel stages only entries that can reach
    alpha >= 1/255 somewhere in the wave's 8x8 block) must not change a single output: CULL=0 evaluates every entry."""
    case = make_case(**CASES[name])
    ref = _run_forward(case)
    _opt(CULL=0)
    try:
        out, _ = _check_forward(case, "%s_cull0" % name)
    finally:
        _opt(CULL=1)
    torch.cuda.synchronize()
    assert torch.equal(out[1], ref[1]), "n_contrib depends on the cull"
    for i in (2, 3, 4, 5):
        if ref[i].numel():
            assert torch.equal(out[i], ref[i]), (i, float((out[i] - ref[i]).abs().max()))
    assert torch.allclose(out[8], ref[8], rtol=1e-5, atol=1e-6)        # (float atomics: order)


@pytest.mark.parametrize("name", list(BWD_CASES))
def test_backward_without_the_block_cull(name, hip_lib):
    """The same for the tile backward (render_backward_wave_kernel): the oracle's gradients with CULL=0."""
    _opt(CULL=0)
    try:
        _check_backward(make_case(**BWD_CASES[name]), "%s_cull0" % name)
        if name == "S5":
            _check_backward(make_case(S=5, seed=31), "bg_geom_off_cull0", backward_geometry=False)
    finally:
        _opt(CULL=1)


@pytest.mark.parametrize("name", ["S16", "big_splats", "ragged_image", "camera_inside", "S0"])
@pytest.mark.parametrize("binning", [0, 1, 2])
def test_tile_binned_order_equals_global_sort(name, binning, hip_lib):
    """All three orderings (0: global radix sort of (tile|depth) keys; 1: radix partition by tile + per-tile LDS sort; 2, the
    default: instances emitted straight into their tile's segment + per-tile LDS sort) must give the oracle's sorted keys,
    point list, ranges and point offsets bit for bit (checked inside _check_forward)."""
    _opt(TILE_BINNING=binning)
    try:
        _check_forward(make_case(**CASES[name]), "%s_binning%d" % (name, binning))
    finally:
        _opt(TILE_BINNING=2)


def test_tile_binned_order_long_tiles(hip_lib):
    """Tiles longer than the small (4096) and the big (16384) in-LDS capacities: few pixels, many large splats."""
    case = make_case(P=40000, W=64, H=48, S=2, scale_log_mean=-1.2, seed=91)
    a = _run_forward(case)
    try:
        _opt(TILE_BINNING=1)
        a1 = _run_forward(case)
        _opt(TILE_BINNING=0)
        b = _run_forward(case)
    finally:
        _opt(TILE_BINNING=2)
    torch.cuda.synchronize()
    for i in (1, 2, 3, 4, 5):
        assert torch.equal(a1[i], b[i]), i
    from relightable3dgaussian_amd.rasterizer_ops import decode_state
    P, H, W = case["P"], case["H"], case["W"]
    assert a[0] == b[0]
    sa, sb = decode_state(a[10], a[11], a[12], P, a[0], H, W), decode_state(b[10], b[11], b[12], P, b[0], H, W)
    lens = (sa["ranges"][:, 1] - sa["ranges"][:, 0])
    assert int(lens.max()) > 16384, "case no longer exercises the global-memory tile sort (max %d)" % int(lens.max())
    for k in ("keys", "point_list", "ranges", "point_offsets"):
        assert torch.equal(torch.as_tensor(sa[k]), torch.as_tensor(sb[k])), k
    for i in (1, 2, 3, 4, 5):
        assert torch.equal(a[i], b[i]), i


@pytest.mark.parametrize("ties", ["pairs", "long_run", "all_equal", "none"])
def test_long_tile_sort_orders_equal_depths_by_index(ties, hip_lib):
    """The long-tile sort (csrc/radix_sort.hip: segmented radix over the depth bits that differ inside the tile) against the
    global radix sort of the reference formulation, with Gaussians at EXACTLY equal depth in
    tiles longer than 4096 instances: pairs (the index-rank fix-up), a run of 300 (the fall-back to the network on the unique
    (depth, index) key), and a frame whose every Gaussian has the same depth (no depth bit differs at all)."""
    case = make_case(P=24000, W=64, H=48, S=2, scale_log_mean=-1.2, seed=191)
    xyz = case["means3D"].clone()
    if ties == "pairs":
        xyz[12000:] = xyz[:12000]
    elif ties == "long_run":
        xyz[500:800] = xyz[500]
        xyz[12000:13000] = xyz[:1000]
    elif ties == "all_equal":
        # every centre on one plane orthogonal to the viewing direction: view-space z is one value up to rounding, so force it
        cam = case["cam"]
        wv = cam.world_view_transform                                        # row-vector convention: p_view = [p 1] @ wv
        zcol = wv[:3, 2]
        depth = xyz @ zcol + wv[3, 2]
        xyz = xyz - (depth - depth.mean())[:, None] * zcol[None] / (zcol @ zcol)
    case["means3D"] = xyz.contiguous()
    try:
        a = _run_forward(case)
        _opt(TILE_BINNING=1)
        a1 = _run_forward(case)
        _opt(TILE_BINNING=0)
        b = _run_forward(case)
    finally:
        _opt(TILE_BINNING=2)
    torch.cuda.synchronize()
    from relightable3dgaussian_amd.rasterizer_ops import decode_state
    P, H, W = case["P"], case["H"], case["W"]
    assert a[0] == b[0] == a1[0]
    sa, sa1, sb = (decode_state(o[10], o[11], o[12], P, o[0], H, W) for o in (a, a1, b))
    lens = sa["ranges"][:, 1] - sa["ranges"][:, 0]
    assert int(lens.max()) > 4096
    keys = torch.as_tensor(sb["keys"]).to(torch.int64)
    if ties != "none":
        assert int((keys[1:] == keys[:-1]).sum()) > (100 if ties != "all_equal" else 1000), "case holds no equal (tile, depth) keys"
    for k in ("keys", "point_list", "ranges", "point_offsets"):
        assert torch.equal(torch.as_tensor(sa[k]), torch.as_tensor(sb[k])), (k, "direct binning")
        assert torch.equal(torch.as_tensor(sa1[k]), torch.as_tensor(sb[k])), (k, "radix partition")
    for i in (1, 2, 3, 4, 5):
        assert torch.equal(a[i], b[i]), i


def test_tile_binned_order_many_tiles(hip_lib):
    """1600x1200 (DTU size, BASELINE config 3): 7500 tiles = 13 tile-id bits, i.e. the two-pass (stable) partition."""
    case = make_case(P=20000, W=1600, H=1200, S=3, scale_log_mean=-2.6, seed=93)
    a = _run_forward(case)
    _opt(TILE_BINNING=0)
    try:
        b = _run_forward(case)
    finally:
        _opt(TILE_BINNING=2)
    torch.cuda.synchronize()
    from relightable3dgaussian_amd.rasterizer_ops import decode_state
    P, H, W = case["P"], case["H"], case["W"]
    assert a[0] == b[0] and a[0] > 0
    sa, sb = decode_state(a[10], a[11], a[12], P, a[0], H, W), decode_state(b[10], b[11], b[12], P, b[0], H, W)
    for k in ("keys", "point_list", "ranges", "point_offsets"):
        assert torch.equal(torch.as_tensor(sa[k]), torch.as_tensor(sb[k])), k
    for i in (1, 2, 3, 4, 5):
        assert torch.equal(a[i], b[i]), i


def test_tile_binned_order_falls_back_above_the_lds_histogram(hip_lib):
    """2064x2048 = 16512 tiles: more than the direct binning's LDS histogram holds, so the radix-partition path runs
    under the default setting -- same lists as the global sort."""
    case = make_case(P=6000, W=2064, H=2048, S=0, scale_log_mean=-2.2, seed=97)
    a = _run_forward(case)
    _opt(TILE_BINNING=0)
    try:
        b = _run_forward(case)
    finally:
        _opt(TILE_BINNING=2)
    torch.cuda.synchronize()
    from relightable3dgaussian_amd.rasterizer_ops import decode_state
    P, H, W = case["P"], case["H"], case["W"]
    assert a[0] == b[0] and a[0] > 0
    sa, sb = decode_state(a[10], a[11], a[12], P, a[0], H, W), decode_state(b[10], b[11], b[12], P, b[0], H, W)
    for k in ("keys", "point_list", "ranges", "point_offsets"):
        assert torch.equal(torch.as_tensor(sa[k]), torch.as_tensor(sb[k])), k


def _same_lists_as_the_global_sort(case):
    a = _run_forward(case)
    _opt(TILE_BINNING=0)
    try:
        b = _run_forward(case)
    finally:
        _opt(TILE_BINNING=2)
    torch.cuda.synchronize()
    from relightable3dgaussian_amd.rasterizer_ops import decode_state
    P, H, W = case["P"], case["H"], case["W"]
    assert a[0] == b[0] and a[0] > 0
    sa, sb = decode_state(a[10], a[11], a[12], P, a[0], H, W), decode_state(b[10], b[11], b[12], P, b[0], H, W)
    for k in ("keys", "point_list", "ranges", "point_offsets"):
        assert torch.equal(torch.as_tensor(sa[k]), torch.as_tensor(sb[k])), k
    return a, sa


def test_tile_binned_order_large_rectangles_clustered_at_low_indices(hip_lib):
    """What a densified scene looks like to the binning kernels (round 6): the oldest Gaussians sit at the lowest indices and are
    the largest.  The first 3 000 of 30 000 splats are 25 x larger (hundreds to thousands of tiles each, some the whole grid):
    the wave chunks dealt round robin over the workgroups and the workgroup-shared list of rectangles above 32 tiles must give
    the lists of the global sort -- with P not a multiple of 64 or of a workgroup's share."""
    case = make_case(P=30011, W=800, H=800, S=0, scale_log_mean=-3.6, seed=131)
    case["scales"] = case["scales"].clone()
    case["scales"][:3000] *= 25.0
    a, st = _same_lists_as_the_global_sort(case)
    touched = torch.as_tensor(st["tiles_touched"]).long()
    assert int((touched[:3000] > 32).sum()) > 2000 and int(touched.max()) >= 2000         # the case is what it claims to be
    # ... and with EVERY rectangle of a workgroup large (2048 list entries per workgroup: the list's capacity)
    case["scales"][:] = case["scales"].clamp_min(0.5)
    small = dict(case, P=4100)
    for k in ("means3D", "features", "opacity", "scales", "rotations", "shs"):
        small[k] = case[k][:4100].contiguous()
    _same_lists_as_the_global_sort(small)


def test_tile_binned_order_at_the_lds_limit(hip_lib):
    """2048 x 2048 = 16 384 tiles: the largest grid the direct binning takes -- 64 KB of LDS counters + the list of large
    rectangles behind them, i.e. more dynamic LDS than a launch may ask for without the per-device function attribute."""
    case = make_case(P=9001, W=2048, H=2048, S=0, scale_log_mean=-2.4, seed=137)
    case["scales"] = case["scales"].clone()
    case["scales"][:200] *= 12.0
    _same_lists_as_the_global_sort(case)


def test_release_scratch_and_count_store(hip_lib):
    """r3dg_release_scratch frees the library's per-(device, stream) buffers and the next call allocates them again;
    r3dg_store_u64_to_host writes 8 bytes into pinned host memory behind the stream's work."""
    import ctypes as C
    from relightable3dgaussian_amd import _lib
    case = make_case(P=3000, W=128, H=128, S=5, seed=11)
    before = _run_forward(case)
    torch.cuda.synchronize()
    assert hip_lib.r3dg_release_scratch() == 0
    after = _run_forward(case)
    torch.cuda.synchronize()
    assert before[0] == after[0] and torch.equal(before[2], after[2])
    src = torch.tensor([0x1122334455667788], dtype=torch.int64, device=DEV)
    ring = torch.zeros(4, dtype=torch.int64).pin_memory()
    _lib.check(hip_lib.r3dg_store_u64_to_host(_lib.current_stream(), src.data_ptr(), ring[2:].data_ptr()), "store")
    torch.cuda.synchronize()
    assert ring.tolist() == [0, 0, 0x1122334455667788, 0]
    assert hip_lib.r3dg_store_u64_to_host(_lib.current_stream(), None, ring.data_ptr()) != 0      # NULL source: R3DG_EINVAL


def test_full_size_properties(hip_lib):
    """BASELINE size (300k Gaussians, 800x800, S=16): size-independent properties of the forward state and outputs, and
    linearity of the backward in its upstream gradient."""
    from relightable3dgaussian_amd import synthetic as syn
    from relightable3dgaussian_amd.rasterizer_ops import decode_state
    from r3dg_rasterization import _C
    P, RES, S = 300_000, 800, 16
    sc = syn.make_scene(P=P, seed=0, stage2=False)
    cam = syn.orbit_cameras(100, width=RES, height=RES)[3].to(DEV)
    d = {k: v.to(DEV) for k, v in sc.items() if torch.is_tensor(v)}
    g = torch.Generator().manual_seed(1)
    feat = torch.rand(P, S, generator=g).to(DEV)
    empty, bg = torch.Tensor([]), torch.ones(3, device=DEV)
    out = _C.rasterize_gaussians(bg, d["xyz"], feat, empty, d["opacity"], d["scales"], d["rotations"], 1.0, empty,
                                 cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, cam.cx,
                                 cam.cy, RES, RES, d["shs"], 3, cam.camera_center, False, True, False)
    R, n_contrib, color, opacity, depth, feature, normal, sxyz, weights, radii, geom, binning, img = out
    st = decode_state(geom, binning, img, P, R, RES, RES)
    keys = st["keys"]
    assert R == int(st["point_offsets"][-1]) == int(st["tiles_touched"].long().sum())
    assert bool((keys[1:] >= keys[:-1]).all()), "sorted keys not ascending"
    tiles = (keys >> 32)
    rng = st["ranges"].long()
    lens = rng[:, 1] - rng[:, 0]
    assert int(lens.sum()) == R
    counts = torch.bincount(tiles, minlength=rng.shape[0])
    assert torch.equal(counts, lens), "tile ranges do not match the key histogram"
    # every instance belongs to a visible Gaussian, each Gaussian appears tiles_touched times
    inst = torch.bincount(st["point_list"].long(), minlength=P)
    assert torch.equal(inst, st["tiles_touched"].long())
    # per pixel: contributors never exceed the tile list; transmittance and opacity are consistent
    tx = (torch.arange(RES, device=DEV) // 16)
    tile_of_pix = (tx[:, None] * ((RES + 15) // 16) + tx[None, :])
    assert bool((n_contrib.long() <= lens[tile_of_pix]).all())
    T_final = st["final_T"]
    assert bool(((T_final >= 0) & (T_final <= 1)).all())
    assert torch.allclose(opacity[0], 1 - T_final, atol=2e-5)
    assert all(bool(torch.isfinite(t).all()) for t in (color, opacity, depth, feature, normal))
    # checksum of checksums: sum of per-Gaussian blending weights == sum of the opacity image
    assert abs(float(weights.double().sum()) - float(opacity.double().sum())) <= 1e-4 * float(opacity.double().sum())
    # backward is linear in the upstream gradients
    gC, gO, gD, gF = [torch.randn(c, RES, RES, generator=g).to(DEV) for c in (3, 1, 1, S)]

    def bwd(scale):
        return _C.rasterize_gaussians_backward(bg, d["xyz"], feat, radii, empty, d["scales"], d["rotations"], 1.0, empty,
                                               cam.world_view_transform, cam.full_proj_transform, cam.tanfovx,
                                               cam.tanfovy, scale * gC, scale * gO, scale * gD, scale * gF, d["shs"], 3,
                                               cam.camera_center, geom, R, binning, img, True, False)
    g1, g3 = bwd(1.0), bwd(3.0)
    for a, b in zip(g1, g3):
        scale = float(b.abs().max())
        assert float((3.0 * a - b).abs().max()) <= 2e-4 * scale + 1e-12
        assert bool(torch.isfinite(a).all())


@pytest.mark.parametrize("S,active", [(16, (2, 3, 4, 5, 6, 7)), (16, (15, 0, 9)), (5, (0, 1, 2)), (28, tuple(range(3, 20))),
                                      (16, ()), (7, (6,))])
def test_backward_active_feature_subset(S, active):
    """`active_features`: with the upstream gradient of every other feature channel zero, the subset kernel must give the
    same nine gradients as the full kernel (the skipped channels' dL_dfeatures columns stay zero)."""
    from r3dg_rasterization import _C
    from relightable3dgaussian_amd import rasterizer_ops
    case = make_case(S=S, seed=81 + S, P=4000)
    a = fwd_args(case, DEV)
    out = _C.rasterize_gaussians(*a)
    H, W = case["H"], case["W"]
    g = torch.Generator().manual_seed(5)
    gC, gO, gD = [torch.randn(c, H, W, generator=g).to(DEV) for c in (3, 1, 1)]
    gF = torch.zeros(S, H, W, device=DEV)
    for ch in active:
        gF[ch] = torch.randn(H, W, generator=g).to(DEV)

    def run(act):
        return rasterizer_ops.rasterize_gaussians_backward(
            a[0], a[1], a[2], out[9], a[3], a[5], a[6], 1.0, a[8], a[9], a[10], a[11], a[12], gC, gO, gD, gF, a[17],
            a[18], a[19], out[10], out[0], out[11], out[12], True, False, active_features=act)
    full, sub = run(None), run(active)
    torch.cuda.synchronize()
    names = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dfeatures", "dL_dcov3D", "dL_dsh",
             "dL_dscales", "dL_drotations")
    for nm, x, y in zip(names, full, sub):
        ok, msg = report(nm, y, x, 1e-4, 1e-9)        # float-atomic summation order differs between the two kernels
        assert ok, msg
    inactive = [c for c in range(S) if c not in active]
    assert float(sub[4][:, inactive].abs().max()) == 0.0 if inactive else True
    with pytest.raises(RuntimeError):
        run((S,))


@pytest.mark.parametrize("S,active", [(16, (2, 3, 4)), (16, (2, 3, 4, 5, 6, 7)), (16, (1, 2, 3, 4, 5, 6, 7)), (5, (0, 1, 2)),
                                      (5, (0, 1, 2, 3, 4)), (7, (6,)), (16, (0, 3, 5, 7, 9, 11, 13))])
def test_backward_without_a_depth_gradient_takes_the_lean_instances(S, active):
    """Round 5: a caller that passes NO depth gradient (an empty tensor = NULL, what the fused iterations do) gets the tile
    backward's lean instances -- channel vector without the depth slot, one reduction channel fewer (1..7 live feature channels
    with at least one padding slot; R3DG_OPT_BWD_LEAN = 0 switches them off).  Same nine gradients as (i) the same call with the
    lean instances off, (ii) the call with an all-zero depth-gradient IMAGE (the general instances), and (iii) the float64 CPU
    oracle; dL_dmeans2D is the FINAL gradient again after the per-Gaussian kernel converted the moments the tile kernel
    accumulates (csrc/rasterizer_preprocess_bwd.hip moments_to_gradients)."""
    import numpy as np
    from oracle import rasterizer as orc
    from r3dg_rasterization import _C
    from relightable3dgaussian_amd import _lib, rasterizer_ops
    case = make_case(S=S, seed=91 + S, P=3000, W=112, H=96)
    a = fwd_args(case, DEV)
    out = _C.rasterize_gaussians(*a)
    H, W = case["H"], case["W"]
    g = torch.Generator().manual_seed(7)
    gC, gO = [torch.randn(c, H, W, generator=g).to(DEV) for c in (3, 1)]
    gF = torch.zeros(S, H, W, device=DEV)
    for ch in active:
        gF[ch] = torch.randn(H, W, generator=g).to(DEV)
    empty = torch.Tensor([])

    def run(gD, act=active):
        return rasterizer_ops.rasterize_gaussians_backward(
            a[0], a[1], a[2], out[9], a[3], a[5], a[6], 1.0, a[8], a[9], a[10], a[11], a[12], gC, gO, gD, gF, a[17],
            a[18], a[19], out[10], out[0], out[11], out[12], True, False, active_features=act)
    lean = run(empty)
    try:
        _lib.set_option("BWD_LEAN", 0)
        plain = run(empty)
    finally:
        _lib.set_option("BWD_LEAN", 1)
    zeros = run(torch.zeros(1, H, W, device=DEV))
    torch.cuda.synchronize()
    names = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dfeatures", "dL_dcov3D", "dL_dsh",
             "dL_dscales", "dL_drotations")
    for nm, x, y, z in zip(names, lean, plain, zeros):
        for tag, other in (("lean off", y), ("zero depth image", z)):
            ok, msg = report(nm + " vs " + tag, x, other, 1e-4, 1e-9)     # float-atomic summation order differs
            assert ok, msg
    assert float(lean[0][:, 2].abs().max()) == 0.0                       # no depth gradient: the z side channel stays zero
    c = fwd_args(case)
    ref = orc.rasterize_gaussians(*c[:-3])
    oref = orc.rasterize_gaussians_backward(c[0], c[1], c[2], ref[9], c[3], c[5], c[6], 1.0, c[8], c[9], c[10], c[11], c[12],
                                            gC.cpu(), gO.cpu(), torch.zeros(1, H, W), gF.cpu(), c[17], c[18], c[19], ref[-1], True)
    for nm, x, y in zip(names, lean, oref[:9]):
        ok, msg = report(nm + " vs oracle", x, np.asarray(y, np.float64).reshape(tuple(x.shape)), 2e-3, 1e-7)
        assert ok, msg


@pytest.mark.parametrize("S,active,size", [(16, None, (128, 128)), (16, (2, 3, 4, 8, 9, 10, 11, 12, 13, 14), (150, 97)),
                                           (5, (0,), (128, 128)), (28, tuple(range(3, 20)), (64, 200)), (36, None, (96, 96)),
                                           (16, (), (128, 128))])
def test_backward_features_only_equals_the_full_backward(S, active, size):
    """r3dg_rasterize_backward_features (frozen geometry, script/run_syn4.sh:27-33): dL_dfeatures alone, without the
    alpha-gradient recursion -- the same values as the full backward's dL_dfeatures (backward.cu:566 does not depend on
    backward_geometry, colours, opacity or depth gradients), on ragged images and for any active-channel subset."""
    from r3dg_rasterization import _C
    from relightable3dgaussian_amd import rasterizer_ops
    W, H = size
    case = make_case(S=S, seed=31 + S, P=5000, W=W, H=H)
    a = fwd_args(case, DEV)
    out = _C.rasterize_gaussians(*a)
    g = torch.Generator().manual_seed(6)
    gC, gO, gD = [torch.randn(c, H, W, generator=g).to(DEV) for c in (3, 1, 1)]
    gF = torch.zeros(S, H, W, device=DEV)
    for ch in (range(S) if active is None else active):
        gF[ch] = torch.randn(H, W, generator=g).to(DEV)
    full = rasterizer_ops.rasterize_gaussians_backward(
        a[0], a[1], a[2], out[9], a[3], a[5], a[6], 1.0, a[8], a[9], a[10], a[11], a[12], gC, gO, gD, gF, a[17], a[18], a[19],
        out[10], out[0], out[11], out[12], True, False, active_features=active)[4]
    only = rasterizer_ops.rasterize_gaussians_backward_features(case["P"], S, H, W, gF, out[10], out[0], out[11], out[12],
                                                                active_features=active)
    torch.cuda.synchronize()
    ok, msg = report("dL_dfeatures", only, full, 2e-5, 1e-9)          # float-atomic summation order differs
    assert ok, msg
    if active is None or len(active):
        assert float(only.abs().max()) > 0
    else:
        assert float(only.abs().max()) == 0.0
    with pytest.raises(RuntimeError):
        rasterizer_ops.rasterize_gaussians_backward_features(case["P"], S, H, W, gF, out[10], out[0], out[11], out[12],
                                                             active_features=(S,))


@pytest.mark.parametrize("name", ["S16", "big_splats", "ragged_image"])
def test_cull_is_exact(name, hip_lib):
    """The sub-tile cull only drops (wave, Gaussian) pairs whose every pixel fails alpha >= 1/255, so the forward
    images must be bit-identical with and without it (weights: float atomics, order-dependent -> tolerance)."""
    case = make_case(**CASES[name])
    try:
        _opt(CULL=0)
        a = _run_forward(case)
        _opt(CULL=1)
        b = _run_forward(case)
    finally:
        _opt(CULL=1)
    torch.cuda.synchronize()
    assert a[0] == b[0]
    for i, nm in ((1, "n_contrib"), (2, "color"), (3, "opacity"), (4, "depth"), (5, "feature"), (6, "normal"),
                  (7, "surface_xyz")):
        assert torch.equal(a[i], b[i]), nm
    assert torch.allclose(a[8], b[8], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("N", [12, 16, 32, 64])
@pytest.mark.parametrize("dpp", [0, 1])
def test_transpose_reduce_selftest(N, dpp, hip_lib):
    from relightable3dgaussian_amd import _lib
    g = torch.Generator().manual_seed(N + dpp)
    x = torch.randn(64, N, generator=g)
    xin = x.to(DEV)
    out = torch.zeros(64, device=DEV)
    chan = torch.zeros(64, dtype=torch.int32, device=DEV)
    owner = torch.zeros(64, dtype=torch.int32, device=DEV)
    st = hip_lib.r3dg_selftest_transpose_reduce(_lib.current_stream(), N, dpp, xin.data_ptr(), out.data_ptr(),
                                                chan.data_ptr(), owner.data_ptr())
    _lib.check(st, "selftest")
    torch.cuda.synchronize()
    want = x.double().sum(0)
    chan, owner, out = chan.cpu(), owner.cpu(), out.cpu()
    assert sorted(chan[owner.bool()].tolist()) == list(range(N)), "owners must cover every channel exactly once"
    err = (out.double() - want[chan.long()]).abs().max()
    print("transpose_reduce N=%d dpp=%d max err %.3e" % (N, dpp, err))
    assert err < 1e-4


def test_radix_sort_stable(hip_lib):
    from relightable3dgaussian_amd import _lib
    for n, end_bit, seed in ((1, 44, 0), (63, 44, 1), (4097, 33, 2), (100_000, 44, 3), (1_000_003, 45, 4)):
        g = torch.Generator().manual_seed(seed)
        # few distinct keys -> many ties: stability is what is being tested
        hi = torch.randint(0, 1 << (end_bit - 32), (n,), generator=g, dtype=torch.int64)
        lo = torch.randint(0, 50, (n,), generator=g, dtype=torch.int64) * 0x01010101
        keys = (hi << 32) | lo
        vals = torch.arange(n, dtype=torch.int32)
        order = torch.sort(keys, stable=True).indices
        k_in, v_in = keys.to(DEV), vals.to(DEV)
        k_out, v_out = torch.empty_like(k_in), torch.empty_like(v_in)
        temp = torch.empty(int(hip_lib.r3dg_sort_temp_bytes(n)), dtype=torch.uint8, device=DEV)
        st = hip_lib.r3dg_sort_pairs(_lib.current_stream(), n, k_in.data_ptr(), v_in.data_ptr(), k_out.data_ptr(),
                                     v_out.data_ptr(), end_bit, temp.data_ptr())
        _lib.check(st, "sort_pairs")
        torch.cuda.synchronize()
        assert torch.equal(k_out.cpu(), keys[order]), "keys not sorted (n=%d)" % n
        assert torch.equal(v_out.cpu().long(), order), "sort not stable (n=%d)" % n


def test_mark_visible():
    from oracle import rasterizer as orc
    from r3dg_rasterization import _C
    case = make_case(P=5000, eye=(0.2, 0.1, 0.0))
    cam = case["cam"]
    got = _C.mark_visible(case["means3D"].to(DEV), cam.world_view_transform.to(DEV), cam.full_proj_transform.to(DEV))
    want = orc.mark_visible(case["means3D"], cam.world_view_transform)
    assert np.array_equal(got.cpu().numpy(), want)


def test_autograd_wrapper_matches_ops():
    """The nn.Module face (reference wrapper API) returns the op's outputs and routes gradients to every input."""
    from r3dg_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    case = make_case(P=2000, S=5, seed=51)
    cam = case["cam"].to(DEV)
    rs = GaussianRasterizationSettings(case["H"], case["W"], cam.tanfovx, cam.tanfovy, cam.cx, cam.cy,
                                       case["bg"].to(DEV), 1.0, cam.world_view_transform, cam.full_proj_transform,
                                       3, cam.camera_center, False, True, True, False)
    leaves = {k: case[k].to(DEV).requires_grad_(True) for k in ("means3D", "opacity", "scales", "rotations", "shs",
                                                               "features")}
    means2D = torch.zeros_like(leaves["means3D"], requires_grad=True)
    outs = GaussianRasterizer(rs)(leaves["means3D"], means2D, leaves["opacity"], shs=leaves["shs"],
                                  scales=leaves["scales"], rotations=leaves["rotations"], features=leaves["features"])
    assert len(outs) == 10
    loss = outs[2].sum() + outs[3].mean() + outs[4].mean() + outs[5].mean()
    loss.backward()
    for k, v in leaves.items():
        assert v.grad is not None and torch.isfinite(v.grad).all(), k
    assert means2D.grad is not None and means2D.grad.shape == (case["P"], 3)


@pytest.mark.parametrize("name", ["S16", "big_splats", "ragged_image", "few_gaussians_many_tiles"])
@pytest.mark.parametrize("side_stream", [False, True])
def test_bounded_forward_equals_the_two_phase_forward(name, side_stream, hip_lib):
    """r3dg_rasterize_forward_begin_bounded / _finish_bounded (no host read-back of num_rendered; binning state laid out
    for a capacity) give the same images, n_contrib, tile ranges and per-tile lists, bit for bit, as the reference-shaped
    forward -- with the ordering on the caller's stream or on a second one -- and the count stays readable on the device."""
    from relightable3dgaussian_amd import rasterizer_ops as ro
    case = make_case(**CASES[name])
    a = _run_forward(case)
    R = a[0]
    cap = R + 1000
    flag = torch.full((4,), 7.0, device=DEV)
    count = torch.zeros(1, dtype=torch.int32, device=DEV)
    order = torch.cuda.Stream() if side_stream else None
    pending = ro.rasterize_gaussians_begin(*fwd_args(case, DEV, False), capacity=cap, overflow_flag=flag,
                                           overflow_count=count, ordering_stream=order)
    b = pending.finish()
    torch.cuda.synchronize()
    assert b[0] == cap
    P, H, W = case["P"], case["H"], case["W"]
    assert int(ro.num_rendered_of(b[10], P)) == R
    assert float(flag[0]) == 0.0 and int(count) == 0
    for i in (1, 2, 3, 4, 5, 6, 7, 9):
        assert torch.equal(a[i], b[i]), "output %d differs" % i
    assert float((a[8] - b[8]).abs().max()) <= 1e-5 * float(a[8].abs().max())      # weights: float atomics, order only
    sa, sb = ro.decode_state(a[10], a[11], a[12], P, R, H, W), ro.decode_state(b[10], b[11], b[12], P, cap, H, W)
    assert torch.equal(torch.as_tensor(sa["ranges"]), torch.as_tensor(sb["ranges"]))
    for k in ("keys", "point_list"):
        assert torch.equal(torch.as_tensor(sa[k])[:R], torch.as_tensor(sb[k])[:R]), k
    assert torch.equal(torch.as_tensor(sa["point_offsets"]), torch.as_tensor(sb["point_offsets"]))
    # (the bounded front end is folded -- block sums scanned by the tile scan, tile order from an extra block of the emit kernel:
    # a tile missing from that order would show as a background-only tile in the images compared above)
    # the backward takes the capacity where the reference passes num_rendered (it selects the state layout)
    from r3dg_rasterization import _C
    gC, gO, gD, gF = [torch.randn(c, H, W, device=DEV) for c in (3, 1, 1, a[5].shape[0])]
    args = fwd_args(case, DEV, False)

    def bwd(o, Rb):
        return _C.rasterize_gaussians_backward(args[0], args[1], args[2], o[9], args[3], args[5], args[6], 1.0, args[8],
                                               args[9], args[10], args[11], args[12], gC, gO, gD, gF, args[17], args[18],
                                               args[19], o[10], Rb, o[11], o[12], True, False)
    ga, gb = bwd(a, R), bwd(b, cap)
    torch.cuda.synchronize()
    for i, (x, y) in enumerate(zip(ga, gb)):
        scale = float(x.abs().max()) + 1e-30
        assert float((x - y).abs().max()) <= 2e-4 * scale, "gradient %d differs (atomics order only)" % i


def test_bounded_forward_drops_a_frame_that_does_not_fit(hip_lib):
    """capacity < num_rendered: nothing is written out of bounds, the frame comes out empty (background, n_contrib 0), the
    flag is raised and the running count incremented; the next frame with enough room is complete again."""
    from relightable3dgaussian_amd import rasterizer_ops as ro
    case = make_case(**CASES["S16"])
    a = _run_forward(case)
    R = a[0]
    flag = torch.zeros(4, device=DEV)
    count = torch.zeros(1, dtype=torch.int32, device=DEV)
    b = ro.rasterize_gaussians_begin(*fwd_args(case, DEV, False), capacity=R - 1, overflow_flag=flag,
                                     overflow_count=count).finish()
    torch.cuda.synchronize()
    P, H, W = case["P"], case["H"], case["W"]
    assert float(flag[0]) == 1.0 and int(count) == 1
    assert int(ro.num_rendered_of(b[10], P)) == R
    assert int(b[1].abs().max()) == 0                                   # n_contrib
    bg = fwd_args(case, DEV, False)[0]
    assert torch.equal(b[2], bg[:, None, None].expand_as(b[2]).contiguous())
    assert float(b[3].abs().max()) == 0.0                               # opacity
    st = ro.decode_state(b[10], b[11], b[12], P, R - 1, H, W)
    assert int(torch.as_tensor(st["ranges"]).abs().max()) == 0
    c = ro.rasterize_gaussians_begin(*fwd_args(case, DEV, False), capacity=R, overflow_flag=flag,
                                     overflow_count=count).finish()
    torch.cuda.synchronize()
    assert float(flag[0]) == 0.0 and int(count) == 1
    for i in (1, 2, 3, 4, 5):
        assert torch.equal(a[i], c[i]), i
