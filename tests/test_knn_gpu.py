"""GPU parity of distCUDA2 (simple_knn drop-in) vs the brute-force CPU oracle: bit-exact (exact 3-NN, fp32 distances in
the reference's operation order, no FMA on either side)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _oracle(pts):
    from oracle import _build
    lib = C.CDLL(_build.build())
    pts = np.ascontiguousarray(pts, np.float32)
    out = np.zeros(pts.shape[0], np.float32)
    lib.knno_dist2(pts.shape[0], pts.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out


@pytest.mark.parametrize("P,seed", [(4, 0), (5, 1), (1000, 2), (3000, 3), (20000, 4)])
def test_dist2_matches_bruteforce(P, seed):
    from simple_knn._C import distCUDA2
    g = torch.Generator().manual_seed(seed)
    pts = torch.rand(P, 3, generator=g) * 2.6 - 1.3
    if P > 100:
        pts[10:14] = pts[5]          # coincident points -> zero distances
    got = distCUDA2(pts.cuda()).cpu().numpy()
    want = _oracle(pts.numpy())
    assert np.array_equal(got, want), "max |diff| %g" % np.abs(got - want).max()


def test_dist2_rejects_bad_input():
    from simple_knn._C import distCUDA2
    with pytest.raises(RuntimeError):
        distCUDA2(torch.zeros(10, 2).cuda())
