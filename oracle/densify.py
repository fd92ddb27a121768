"""TEST INFRASTRUCTURE ONLY -- CPU (numpy, fp32) restatement of the reference's densification bookkeeping.

Restates, as written and in the reference's order of operations:
  * `GaussianModel.add_densification_stats` (scene/gaussian_model.py:931-937) + the max-radii update of train.py:164-165;
  * `GaussianModel.densify_and_prune` (:893-915) = `densify_and_clone` (:846-891) -> `densify_and_split` (:790-844)
    -> opacity / weight / size prune, with the optimiser-state surgery of `cat_tensors_to_optimizer` (:721-744),
    `_prune_optimizer` (:681-698), `densification_postfix` (:746-788), `prune_points` (:700-719);
  * `GaussianModel.prune` (:917-929) and `reset_opacity` (:563-566, `replace_tensor_to_optimizer` :667-679).

Pinned by tests/golden/densify_reference_*.npz = inputs/outputs of the reference's own class executed on CPU
(tests/golden/make_densify_golden.py).  Quirks kept on purpose:
  Q1  `densification_postfix` zeroes `max_radii2D` before `densify_and_prune` evaluates `big_points_vs`, so the
      screen-size test of a densify call never fires (it does in `prune`);
  Q2  the rows appended by clone/split enter the final prune with `weights_accum = 1`;
  Q3  the split uses `padded_grad >= threshold` on the signed mean (no norm), the clone `norm(...) >= threshold`;
  Q4  NaN statistics (0/0) count as 0, x/0 = inf stays inf.
Nothing outside tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
import numpy as np

F32 = np.float32


def _f(x):
    return np.ascontiguousarray(x, dtype=F32)


def accumulate(stats, viewspace_grad, normal_grad, radii, weights):
    """stats: dict of xyz_gradient_accum[P,1], normal_gradient_accum[P,1], denom[P,1], weights_accum[P,1],
    max_radii2D[P]; updated in place."""
    vis = np.asarray(radii) > 0
    stats["weights_accum"] += _f(weights).reshape(-1, 1)
    g = _f(viewspace_grad)[:, :2]
    stats["xyz_gradient_accum"][vis] += np.sqrt((g[vis] * g[vis]).sum(-1, dtype=F32, keepdims=True), dtype=F32)
    n = _f(normal_grad)
    stats["normal_gradient_accum"][vis] += np.sqrt((n[vis] * n[vis]).sum(-1, dtype=F32, keepdims=True), dtype=F32)
    stats["denom"][vis] += F32(1)
    stats["max_radii2D"][vis] = np.maximum(stats["max_radii2D"][vis], np.asarray(radii)[vis].astype(F32))


def _rotmat(q):
    q = _f(q)
    q = q / np.sqrt((q * q).sum(-1, dtype=F32, keepdims=True), dtype=F32)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.zeros((q.shape[0], 3, 3), F32)
    one, two = F32(1), F32(2)
    R[:, 0, 0] = one - two * (y * y + z * z)
    R[:, 0, 1] = two * (x * y - r * z)
    R[:, 0, 2] = two * (x * z + r * y)
    R[:, 1, 0] = two * (x * y + r * z)
    R[:, 1, 1] = one - two * (x * x + z * z)
    R[:, 1, 2] = two * (y * z - r * x)
    R[:, 2, 0] = two * (x * z - r * y)
    R[:, 2, 1] = two * (y * z + r * x)
    R[:, 2, 2] = one - two * (x * x + y * y)
    return R


def _sigmoid(x):
    return (F32(1) / (F32(1) + np.exp(-_f(x), dtype=F32))).astype(F32)


class Model:
    """params / exp_avg / exp_avg_sq: dicts name -> [P, ...] fp32 arrays; must hold 'xyz', 'scaling', 'rotation',
    'opacity' (raw, pre-activation) plus any number of pass-through groups."""

    def __init__(self, params, exp_avg, exp_avg_sq, stats):
        self.p = {k: _f(v).copy() for k, v in params.items()}
        self.m = {k: _f(v).copy() for k, v in exp_avg.items()}
        self.v = {k: _f(v).copy() for k, v in exp_avg_sq.items()}
        self.s = {k: _f(v).copy() for k, v in stats.items()}
        self.min_margin = np.inf        # smallest relative distance of any compared quantity from its threshold

    P = property(lambda self: self.p["xyz"].shape[0])

    def _margin(self, value, thr):
        value = np.asarray(value, dtype=np.float64)
        fin = np.isfinite(value)
        if fin.any() and np.isfinite(thr):
            d = np.abs(value[fin] - thr) / max(abs(thr), 1e-30)
            if d.size:
                self.min_margin = min(self.min_margin, float(d.min()))

    def _scales(self):
        return np.exp(self.p["scaling"], dtype=F32)

    def _append(self, new, count):
        for k in self.p:
            self.p[k] = np.concatenate([self.p[k], new[k]], 0)
            z = np.zeros_like(new[k])
            self.m[k] = np.concatenate([self.m[k], z], 0)
            self.v[k] = np.concatenate([self.v[k], z], 0)
        P = self.P
        self.s["weights_accum"] = np.concatenate([self.s["weights_accum"], np.ones((count, 1), F32)], 0)
        self.s["xyz_gradient_accum"] = np.zeros((P, 1), F32)
        self.s["normal_gradient_accum"] = np.zeros((P, 1), F32)
        self.s["denom"] = np.zeros((P, 1), F32)
        self.s["max_radii2D"] = np.zeros((P,), F32)                       # Q1

    def _prune(self, mask):
        keep = ~mask
        for d in (self.p, self.m, self.v):
            for k in d:
                d[k] = d[k][keep]
        for k in self.s:
            self.s[k] = self.s[k][keep]

    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size, max_grad_normal, percent_dense,
                          normal_table, weights_threshold=1e-4, n_split=2):
        with np.errstate(divide="ignore", invalid="ignore"):
            grads = (self.s["xyz_gradient_accum"] / self.s["denom"]).astype(F32)
            grads_n = (self.s["normal_gradient_accum"] / self.s["denom"]).astype(F32)
        grads[np.isnan(grads)] = 0                                         # Q4
        grads_n[np.isnan(grads_n)] = 0
        self._margin(grads, max_grad)
        self._margin(grads_n, max_grad_normal)
        size_thr = F32(percent_dense * extent)
        # ---- clone (:846-891)
        smax = self._scales().max(1)
        self._margin(smax, float(size_thr))
        sel = (np.abs(grads[:, 0]) >= F32(max_grad)) | (np.abs(grads_n[:, 0]) >= F32(max_grad_normal))
        sel &= smax <= size_thr
        self.clone_mask = sel.copy()
        self._append({k: self.p[k][sel] for k in self.p}, int(sel.sum()))
        # ---- split (:790-844)
        P1 = self.P
        pg = np.zeros(P1, F32)
        pg[:grads.shape[0]] = grads[:, 0]
        pgn = np.zeros(P1, F32)
        pgn[:grads_n.shape[0]] = grads_n[:, 0]
        sel = (pg >= F32(max_grad)) | (pgn >= F32(max_grad_normal))       # Q3
        scales = self._scales()
        sel &= scales.max(1) > size_thr
        self.split_mask = sel.copy()
        ns = int(sel.sum())
        stds = np.tile(scales[sel], (n_split, 1))
        samples = (F32(0) + stds * _f(normal_table)[:n_split * ns]).astype(F32)
        R = np.tile(_rotmat(self.p["rotation"][sel]), (n_split, 1, 1))
        new = {k: np.tile(self.p[k][sel], (n_split,) + (1,) * (self.p[k].ndim - 1)) for k in self.p}
        new["xyz"] = (np.einsum("nij,nj->ni", R, samples).astype(F32) + np.tile(self.p["xyz"][sel], (n_split, 1))).astype(F32)
        new["scaling"] = np.log(stds / F32(0.8 * n_split), dtype=F32)
        self._append(new, n_split * ns)
        self._prune(np.concatenate([sel, np.zeros(n_split * ns, bool)]))
        # ---- final prune (:904-915)
        self._final_prune(min_opacity, extent, max_screen_size, weights_threshold)

    def _final_prune(self, min_opacity, extent, max_screen_size, weights_threshold):
        op = _sigmoid(self.p["opacity"])[:, 0]
        self._margin(op, min_opacity)
        self._margin(self.s["weights_accum"][:, 0], weights_threshold)
        mask = (op < F32(min_opacity)) | (self.s["weights_accum"][:, 0] < F32(weights_threshold))
        if max_screen_size:
            smax = self._scales().max(1)
            self._margin(smax, 0.1 * extent)
            mask |= (self.s["max_radii2D"] > F32(max_screen_size)) | (smax > F32(0.1 * extent))
        self.prune_mask = mask.copy()
        self._prune(mask)
        self.s["weights_accum"][:] = 0

    def prune(self, min_opacity, extent, max_screen_size, weights_threshold=1e-4):
        self._final_prune(min_opacity, extent, max_screen_size, weights_threshold)

    def reset_opacity(self):
        op = np.minimum(_sigmoid(self.p["opacity"]), F32(0.01))
        self.p["opacity"] = np.log(op / (F32(1) - op), dtype=F32)
        self.m["opacity"] = np.zeros_like(self.p["opacity"])
        self.v["opacity"] = np.zeros_like(self.p["opacity"])
