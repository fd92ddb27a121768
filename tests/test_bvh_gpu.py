"""GPU parity of the LBVH build and the visibility trace vs the CPU oracle (oracle/bvh_oracle.c).
  * build: node table (parent, left, right, object id, leaf count), every box and the 64-bit Morton codes BIT-EXACT
    (integer topology from fp32 Morton codes computed in the reference's operation order, no FMA);
  * trace: visibility within 1e-5 and the {0, >=0.9} class identical except rays whose product lies within 1e-5 of
    0.9; hit counts equal on rays without early-out."""
import numpy as np
import pytest
import torch

from tests.test_oracle_cpu import _bvh_case

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("P,seed,dup", [(1, 0, False), (2, 1, False), (3, 2, False), (1000, 3, True),
                                        (100_000, 4, True)])
def test_bvh_build_and_trace_parity(P, seed, dup):
    from oracle import bvh as ob
    from bvh_tracing import RayTracer
    K = 8 if P > 10000 else 16
    sc, dirs, cinv, rays_o = _bvh_case(P, seed, K=K, dup=dup)
    nodes, aabbs = ob.leaf_boxes(sc["xyz"].numpy(), sc["scales"].numpy(), sc["rotations"].numpy())
    n_ref, a_ref, m_ref = ob.create_bvh(nodes, aabbs)
    rt = RayTracer(sc["xyz"].to(DEV), sc["scales"].to(DEV), sc["rotations"].to(DEV))
    torch.cuda.synchronize()
    assert np.array_equal(rt.morton.cpu().numpy().astype(np.uint64), m_ref), "Morton codes differ"
    assert np.array_equal(rt.tree.cpu().numpy(), n_ref), "node table differs"
    a_got = rt.aabb.cpu().numpy()
    leaf_bad = (a_got[P - 1:] != a_ref[P - 1:]).any(1).sum()
    int_bad = (a_got[:P - 1] != a_ref[:P - 1]).any(1).sum()
    print("P=%d box rows differing: leaves %d, internal %d (max |diff| %.3e)" % (P, leaf_bad, int_bad,
                                                                               np.abs(a_got - a_ref).max()))
    assert leaf_bad == 0 and int_bad == 0, "boxes differ"
    # trace through the reference-shaped entry point (RayTracer.trace_visibility adds the 0.05*d offset itself)
    o_unoffset = sc["xyz"][:, None, :].expand_as(dirs).contiguous()
    res = rt.trace_visibility(o_unoffset.to(DEV), dirs.to(DEV), sc["xyz"].to(DEV), cinv.to(DEV),
                              sc["opacity"][:, 0].contiguous().to(DEV), sc["normal"].to(DEV))
    torch.cuda.synchronize()
    from relightable3dgaussian_amd import bvh_ops
    assert int(bvh_ops.trace_bvh_opacity.last_overflow.item()) == 0
    vis = res["visibility"][..., 0].cpu().numpy()
    cnt = res["contribute"][..., 0].cpu().numpy()
    # oracle uses the same offset origin the host computed (fp32 add in torch)
    ro = (o_unoffset + dirs * 0.05).numpy()
    cnt_ref, vis_ref = ob.trace_bvh_opacity(n_ref, a_ref, ro, dirs.numpy(), sc["xyz"].numpy(), cinv.numpy(),
                                            sc["opacity"][:, 0].numpy(), sc["normal"].numpy())
    _, prod = ob.trace_bruteforce(n_ref, a_ref, ro, dirs.numpy(), sc["xyz"].numpy(), cinv.numpy(),
                                  sc["opacity"][:, 0].numpy(), sc["normal"].numpy()) if P <= 1000 else (None, None)
    near = np.abs(vis_ref - 0.9) < 1e-5 if prod is None else np.abs(prod - 0.9) < 1e-5
    cls_diff = ((vis == 0) != (vis_ref == 0)) & ~near
    # an early-out that fires at a different visit leaves the same 0; otherwise values agree to fp32 rounding
    print("P=%d rays=%d  vis==0: %.3f  max|err| %.3e  class mismatches %d" % (
        P, vis.size, (vis == 0).mean(), np.abs(vis - vis_ref)[~cls_diff].max(), cls_diff.sum()))
    assert vis.shape == dirs.shape[:-1]
    assert cls_diff.sum() == 0
    same_cls = (vis == 0) == (vis_ref == 0)
    assert np.abs(vis - vis_ref)[same_cls].max() < 1e-5
    both = (vis > 0) & (vis_ref > 0)
    assert np.array_equal(cnt[both], cnt_ref[both])


def test_bvh_ops_shapes_and_errors():
    from bvh_tracing import _C
    sc, dirs, cinv, rays_o = _bvh_case(50, 9)
    with pytest.raises(RuntimeError):
        _C.create_bvh(sc["xyz"].to(DEV), sc["scales"].to(DEV), sc["rotations"].to(DEV),
                      torch.zeros(10, 5, dtype=torch.int32, device=DEV), torch.zeros(10, 6, device=DEV))



@pytest.mark.parametrize("P,seed,K", [(1, 0, 4), (3, 1, 8), (40, 2, 16), (3000, 3, 8)])
def test_trace_bvh_hit_lists_are_consistent(P, seed, K):
    """trace_bvh (bvh/src/trace.cu:8-192): per-ray hit lists.  Properties that hold by construction: the list length is the
    sum of the counts; every ray's segment is sorted by t with the rejected entries (id -1, t = 1e6) last; accepted
    entries satisfy t = (mean - o).d >= 0.01 and position = o + t d; the candidate set of a ray contains every Gaussian
    whose own leaf box the ray hits with tmax > 0 (brute force)."""
    from bvh_tracing import _C
    from relightable3dgaussian_amd import bvh as hb
    sc, dirs, cinv, rays_o = _bvh_case(P, seed, K=K, dup=P > 100)
    d = {k: v.to(DEV) for k, v in sc.items() if torch.is_tensor(v)}
    nodes, aabbs = hb.leaf_boxes(d["xyz"], d["scales"], d["rotations"])
    nodes, aabbs, _ = _C.create_bvh(d["xyz"], d["scales"], d["rotations"], nodes, aabbs)
    ro, rd = rays_o.reshape(-1, 3).to(DEV), dirs.reshape(-1, 3).to(DEV)
    cnt, pts, pos, rid = _C.trace_bvh(nodes, aabbs, ro, rd, d["xyz"], cinv.to(DEV), d["opacity"][:, 0].contiguous())
    torch.cuda.synchronize()
    N = ro.shape[0]
    assert cnt.shape == (N, 1) and cnt.dtype == torch.int32
    n = int(cnt.sum())
    if n == 0:
        assert pts.shape == (0, 1) and pos.shape == (0, 3) and rid.shape == (0, 3)
        return
    assert pts.shape == (n, 1) and pos.shape == (n, 3) and rid.shape == (n, 1)
    assert torch.equal(rid[:, 0].long(), torch.repeat_interleave(torch.arange(N, device=DEV), cnt[:, 0].long()))
    r = rid[:, 0].long()
    t = ((pos - ro[r]) * rd[r]).sum(-1) / (rd[r] * rd[r]).sum(-1)           # position = o + t d
    ok = pts[:, 0] >= 0
    g = pts[ok, 0].long()
    t_mean = ((d["xyz"][g] - ro[r[ok]]) * rd[r[ok]]).sum(-1)
    if g.numel():
        assert (t_mean >= 0.01 - 1e-6).all() and (t[ok] - t_mean).abs().max().item() < 1e-4 * max(1.0, t_mean.abs().max().item())
    assert (t[~ok] > 1e5).all()
    same = r[1:] == r[:-1]
    assert (t[1:][same] >= t[:-1][same] - 1e-4 * t[:-1][same].abs().clamp_min(1.0)).all(), "a ray's entries are not sorted by t"
    # brute force: a Gaussian whose own leaf box is hit (tmax > 0) and whose projected distance passes must be listed
    if P <= 100:
        leaf = aabbs[P - 1:]
        obj = nodes[P - 1:, 3].long()
        for ray in range(0, N, max(1, N // 24)):
            o, dd = ro[ray].cpu().double(), rd[ray].cpu().double()
            listed = set(pts[r == ray, 0].tolist())
            for j in range(P):
                lo, hi = leaf[j, :3].cpu().double(), leaf[j, 3:].cpu().double()
                t0, t1 = (lo - o) / dd, (hi - o) / dd
                tmin, tmax = torch.minimum(t0, t1).max().item(), torch.maximum(t0, t1).min().item()
                tm = float(((d["xyz"][obj[j]].cpu().double() - o) * dd).sum())
                if tmax > 1e-4 and tmin < tmax - 1e-6 and tm > 0.011 and tmin + 1e-5 < tm < tmax - 1e-5:
                    assert int(obj[j]) in listed, (ray, j)


@pytest.mark.parametrize("P,seed,K", [(7, 3, 16), (5000, 5, 64), (60_000, 6, 16)])
def test_trace_formulations_give_identical_results(P, seed, K):
    """R3DG_OPT_TRACE_FORMULATION: 0 thread-per-ray over the reference's tables, 2 packed 64-byte records, 3 persistent waves with
    per-XCD queues, 4 (default) phase-separated bodies with the current node in a register -- same per-ray visit order and
    arithmetic, so transmittance and hit counts are bit-identical (incl. a ray count that is not a multiple of 64)."""
    from bvh_tracing import RayTracer
    from relightable3dgaussian_amd import _lib
    sc, dirs, cinv, rays_o = _bvh_case(P, seed, K=K, dup=True)
    rt = RayTracer(sc["xyz"].to(DEV), sc["scales"].to(DEV), sc["rotations"].to(DEV))
    o = sc["xyz"][:, None, :].expand_as(dirs).contiguous().to(DEV)[:, :K - 1].contiguous()
    d = dirs.to(DEV)[:, :K - 1].contiguous()
    L = _lib.lib()
    got = {}
    try:
        for mode in (0, 2, 3, 4):
            _lib.set_option("TRACE_FORMULATION", mode)
            res = rt.trace_visibility(o, d, sc["xyz"].to(DEV), cinv.to(DEV), sc["opacity"][:, 0].contiguous().to(DEV),
                                      sc["normal"].to(DEV))
            torch.cuda.synchronize()
            got[mode] = (res["visibility"].clone(), res["contribute"].clone())
    finally:
        _lib.set_option("TRACE_FORMULATION", 4)
    for mode in (2, 3, 4):
        assert torch.equal(got[mode][0], got[0][0]), "visibility differs in mode %d" % mode
        assert torch.equal(got[mode][1], got[0][1]), "hit counts differ in mode %d" % mode
    # the COUNTING instantiation of the default kernel (R3DG_OPT_TRACE_COUNT_VISITS, bench.py's node-visits/s): same results; every
    # ray takes at least the root's node step (P > 1), a leaf step per accepted Gaussian at least, and the sums are the
    # traversal's, i.e. the same on a second run
    from relightable3dgaussian_amd import bvh_ops
    counts = []
    try:
        _lib.set_option("TRACE_COUNT_VISITS", 1)
        for _ in range(2):
            bvh_ops.VISITS[:] = [0, 0, 0]
            res = rt.trace_visibility(o, d, sc["xyz"].to(DEV), cinv.to(DEV), sc["opacity"][:, 0].contiguous().to(DEV),
                                      sc["normal"].to(DEV))
            counts.append(tuple(bvh_ops.VISITS))
            assert torch.equal(res["visibility"], got[4][0]) and torch.equal(res["contribute"], got[4][1])
    finally:
        _lib.set_option("TRACE_COUNT_VISITS", 0)
        bvh_ops.VISITS[:] = [0, 0, 0]
    nodes, leaves, rays = counts[0]
    assert counts[0] == counts[1] and rays == P * (K - 1)
    assert nodes >= rays and leaves >= int(got[4][1].sum()) and nodes < 2000 * rays


@pytest.mark.parametrize("P", [2, 700, 70_000])
def test_bvh_build_with_caller_initialised_leaf_counters(P):
    """The reference ADDS the children's leaf counters into column 4 of whatever node table it is handed
    (construct.cu:246-262); its own RayTracer hands zeros, which the build answers from the Karras ranges without any
    inter-thread synchronisation.  Any other content takes the reference-shaped walk: same table as the oracle either way,
    same boxes in both."""
    from oracle import bvh as ob
    from relightable3dgaussian_amd import bvh as B, bvh_ops
    sc, dirs, cinv, rays_o = _bvh_case(P, 17, K=4, dup=True)
    nodes0, aabbs0 = ob.leaf_boxes(sc["xyz"].numpy(), sc["scales"].numpy(), sc["rotations"].numpy())
    got = {}
    for init in (0, 3):
        n_in = nodes0.copy()
        n_in[:P - 1, 4] = init
        n_ref, a_ref, m_ref = ob.create_bvh(n_in.copy(), aabbs0.copy())
        nodes, aabbs = B.leaf_boxes(sc["xyz"].to(DEV), sc["scales"].to(DEV), sc["rotations"].to(DEV))
        nodes[:P - 1, 4] = init
        tree, box, morton = bvh_ops.create_bvh(sc["xyz"].to(DEV), sc["scales"].to(DEV), sc["rotations"].to(DEV), nodes, aabbs)
        torch.cuda.synchronize()
        assert np.array_equal(tree.cpu().numpy(), n_ref), "node table differs (initial counter %d)" % init
        assert np.array_equal(box.cpu().numpy(), a_ref), "boxes differ (initial counter %d)" % init
        got[init] = box.clone()
    assert torch.equal(got[0], got[3])
