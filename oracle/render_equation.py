"""CPU oracle of the render_equation.{cu,h} contract model -- TEST INFRASTRUCTURE ONLY (Python face of
oracle/shading_oracle.c; restates the .cu as written, quirks included; parity unpinned by the reference)."""
import ctypes as C

import numpy as np

from . import _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_build.build())
    return _lib


def _f(a):
    return np.ascontiguousarray(a.detach().cpu().numpy() if hasattr(a, "detach") else a, np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def forward(base_color, roughness, metallic, normals, viewdirs, inc, direct, vis, K, rand_float=None):
    a = [_f(x) for x in (base_color, roughness, metallic, normals, viewdirs, inc, direct, vis)]
    P, Si, Sd, Sv = a[0].shape[0], a[5].shape[1], a[6].shape[1], a[7].shape[1]
    dirs, pbr, dl = np.zeros((P, K, 3), np.float32), np.zeros((P, 3), np.float32), np.zeros((P, 3), np.float32)
    rf = _f(rand_float) if rand_float is not None else None
    lib().reo_forward(P, Si, Sd, Sv, *[_p(x) for x in a], K, _p(rf), _p(dirs), _p(pbr), _p(dl))
    return pbr, dirs, dl


def forward_complex(base_color, roughness, metallic, normals, viewdirs, inc, direct, vis, K):
    a = [_f(x) for x in (base_color, roughness, metallic, normals, viewdirs, inc, direct, vis)]
    P, Si, Sd, Sv = a[0].shape[0], a[5].shape[1], a[6].shape[1], a[7].shape[1]
    z = lambda *s: np.zeros(s, np.float32)
    outs = [z(P, K, 3), z(P, 3), z(P, K, 3), z(P, K, 3), z(P, K, 3), z(P, K, 1), z(P, 3), z(P, 3), z(P, 1), z(P, 3), z(P, 3)]
    lib().reo_forward_complex(P, Si, Sd, Sv, *[_p(x) for x in a], K, *[_p(o) for o in outs])
    dirs, pbr = outs[0], outs[1]
    return (pbr, dirs) + tuple(outs[2:])


def backward(base_color, roughness, metallic, normals, viewdirs, inc, direct, vis, K, incident_dirs, dL_dpbr, dL_ddl):
    a = [_f(x) for x in (base_color, roughness, metallic, normals, viewdirs, inc, direct, vis)]
    P, Si, Sd, Sv = a[0].shape[0], a[5].shape[1], a[6].shape[1], a[7].shape[1]
    d, gp, gd = _f(incident_dirs), _f(dL_dpbr), _f(dL_ddl)
    z = lambda *s: np.zeros(s, np.float32)
    outs = [z(P, 3), z(P, 1), z(P, 1), z(P, 3), z(P, 3), z(P, Si, 3)]
    ddirect = np.zeros((1, Sd, 3), np.float64)
    dvis = z(P, Sv, 1)
    lib().reo_backward(P, Si, Sd, Sv, *[_p(x) for x in a], K, _p(d), _p(gp), _p(gd), *[_p(o) for o in outs],
                       _p(ddirect), _p(dvis))
    return tuple(outs) + (ddirect, dvis)
