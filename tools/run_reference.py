#!/usr/bin/env python
"""Run a script of an UNMODIFIED NJU-3DV/Relightable3DGaussian checkout against this repo's drop-in extension packages
(SURVEY.md 8(f) n4):

    python tools/run_reference.py --reference /path/to/Relightable3DGaussian -- train.py -s <data> -m <out> --eval ...

What it does, and nothing else: (1) puts this repo in front of sys.path, so the reference's `from r3dg_rasterization import _C`
(gaussian_renderer/r3dg_rasterization.py:8), `from bvh_tracing import _C` (bvh/__init__.py:8) and
`from simple_knn._C import distCUDA2` (scene/gaussian_model.py:13) resolve to the HIP-backed packages at the repo root
instead of the CUDA extensions the reference would JIT-compile; (2) installs stand-ins for third-party packages that are
missing from the image (tools/reference_shims.py; real packages are used when they import); (3) executes the script as
`__main__` with the reference directory as sys.path[0], exactly like `python train.py ...` started there.

`--cpu-oracle` is TEST INFRASTRUCTURE for boxes without a GPU (tests/test_reference_scripts_cpu.py): the three extension
modules are then backed by the CPU oracle (oracle/*.c) and the reference's hard-coded device="cuda" is redirected to the CPU
(tests/reference_cpu_backend.py).  It exists to show that the reference's own training loop runs unchanged across this
repo's extension boundary; it is not a product path and never a fallback -- without the flag a missing GPU or library
fails loudly inside the first op."""
import argparse
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--reference", default=os.environ.get("R3DG_REFERENCE", "/root/reference"),
                    help="checkout of NJU-3DV/Relightable3DGaussian (default: $R3DG_REFERENCE or /root/reference)")
    ap.add_argument("--cpu-oracle", action="store_true", help="(tests only) CPU oracle behind the extension modules")
    ap.add_argument("--quiet-shims", action="store_true")
    ap.add_argument("--patch-rendering-equation", action="store_true",
                    help="apply INTEGRATION.md's optional one-line patch from outside the checkout: "
                         "gaussian_renderer.neilf.rendering_equation = this repo's fused op (same signature)")
    ap.add_argument("script", help="script inside the checkout, e.g. train.py")
    ap.add_argument("args", nargs=argparse.REMAINDER)
    ns = ap.parse_args(argv)
    ref = os.path.abspath(ns.reference)
    script = ns.script if os.path.isabs(ns.script) else os.path.join(ref, ns.script)
    if not os.path.isfile(script):
        sys.exit("run_reference: %s does not exist" % script)
    for p in (ROOT, os.path.join(ROOT, "tools")):
        if p not in sys.path:
            sys.path.insert(0, p)
    # the reference's `torch.load(checkpoint_path)` calls (scene/gaussian_model.py:359, direct_light_map.py) predate
    # torch 2.6's weights_only=True default, and its checkpoints hold a numpy scalar (spatial_lr_scale)
    os.environ.setdefault("TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD", "1")
    import reference_shims
    reference_shims.install(verbose=not ns.quiet_shims)
    if ns.cpu_oracle:
        from tests import reference_cpu_backend
        reference_cpu_backend.install()
    else:
        import bvh_tracing  # noqa: F401  (the drop-in packages: import errors surface here, before the script starts)
        import r3dg_rasterization  # noqa: F401
        import simple_knn  # noqa: F401
    # `python train.py` puts the script's directory first; the reference's own packages (scene, utils, arguments, bvh, ...)
    # must win over same-named directories elsewhere, the three extension packages are not shadowed by anything in it
    sys.path.insert(0, os.path.dirname(script))
    if ns.patch_rendering_equation:
        import gaussian_renderer.neilf as neilf                      # (the reference's module, found through sys.path[0])
        from relightable3dgaussian_amd import shading_ops
        neilf.rendering_equation = shading_ops.rendering_equation    # same signature as neilf.py:339-371
        print("[run_reference] gaussian_renderer.neilf.rendering_equation -> relightable3dgaussian_amd.shading_ops.rendering_equation")
    sys.argv = [script] + [a for a in ns.args if a != "--"]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
