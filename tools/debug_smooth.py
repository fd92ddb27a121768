"""Where do the streamed and the three-pass smoothness kernels differ?  (debug aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from relightable3dgaussian_amd import _lib
DEV = "cuda"
L = _lib.lib()
for (H, W) in [(64, 64), (37, 50), (5, 7), (120, 161)]:
    for weights, acc_normal, masked in [((1.0, 0.5, 1.0), 0, True), ((1.0, 0.0, 0.0), 0, False), ((0.0, 0.5, 1.0), 1, True), ((0.0, 0.0, 1.0), 1, False)]:
        g = torch.Generator().manual_seed(H * 1000 + W)
        N = H * W
        opacity = torch.rand(1, H, W, generator=g).to(DEV)
        opacity[0, : H // 4] *= 1e-6
        feature = (torch.rand(16, H, W, generator=g) * 1.5 - 0.2).to(DEV) * opacity
        n_contrib = (torch.rand(H, W, generator=g) > 0.15).to(torch.int32).to(DEV)
        gt = torch.rand(3, H, W, generator=g).to(DEV)
        mask = torch.rand(1, H, W, generator=g).to(DEV) if masked else None
        wb, wr, wl = (w / (3.0 * N) for w in weights)
        s = _lib.current_stream()
        outs = []
        for fused in (False, True):
            d_op = torch.full((1, H, W), 0.25, device=DEV)
            d_f = torch.full((16, H, W), -3.0, device=DEV)
            sums = torch.zeros(3, 32, device=DEV)
            args = (W, H, opacity.data_ptr(), feature.data_ptr(), n_contrib.data_ptr())
            if fused:
                _lib.check(L.r3dg_stage2_smooth_fused(s, *args, gt.data_ptr(), _lib.ptr(mask), wb, wr, wl, acc_normal, d_op.data_ptr(),
                                                      d_f.data_ptr(), sums.data_ptr()), "smooth_fused")
            else:
                scratch = torch.empty(30 * N, device=DEV)
                _lib.check(L.r3dg_stage2_smooth_forward(s, *args, gt.data_ptr(), _lib.ptr(mask), wb, wr, wl, scratch.data_ptr(),
                                                        sums.data_ptr()), "smooth_forward")
                _lib.check(L.r3dg_stage2_smooth_backward(s, *args, _lib.ptr(mask), scratch.data_ptr(), wb, wr, wl, acc_normal,
                                                         d_op.data_ptr(), d_f.data_ptr()), "smooth_backward")
            torch.cuda.synchronize()
            outs.append((d_op.cpu(), d_f.cpu(), sums.sum(1).cpu()))
        (o0, f0, s0), (o1, f1, s1) = outs
        bad = (f0 != f1).nonzero()
        rel = ((f0 - f1).abs() / f0.abs().clamp_min(1e-30))
        print(H, W, weights, acc_normal, masked, "differing:", len(bad), "of", f0.numel(), "max rel %.3g" % float(rel.max()),
              "sums", s0.tolist(), s1.tolist())
        if len(bad):
            ch = sorted(set(bad[:, 0].tolist()))
            ys = bad[:, 1]; xs = bad[:, 2]
            print("   channels", ch, "rows", sorted(set(ys.tolist()))[:12], "cols", sorted(set(xs.tolist()))[:12])
            for b in bad[:5].tolist():
                print("   ", b, float(f0[tuple(b)]), float(f1[tuple(b)]))
        bo = (o0 != o1).nonzero()
        if len(bo):
            print("   opacity grad differs at", len(bo), [(b, float(o0[tuple(b)]), float(o1[tuple(b)])) for b in bo[:3].tolist()])
