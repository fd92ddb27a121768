#!/usr/bin/env python
"""Run a script of an UNMODIFIED NJU-3DV/Relightable3DGaussian checkout against this repo's drop-in extension packages
(SURVEY.md 8(f) n4):

    python tools/run_reference.py --reference /path/to/Relightable3DGaussian -- train.py -s <data> -m <out> --eval ...
    python tools/run_reference.py --dp 8 --patch-rendering-equation -- train.py -s <data> -m <out> -t neilf -c <ckpt> ...

What it does, and nothing else: (1) puts this repo in front of sys.path, so the reference's `from r3dg_rasterization import _C`
(gaussian_renderer/r3dg_rasterization.py:8), `from bvh_tracing import _C` (bvh/__init__.py:8) and
`from simple_knn._C import distCUDA2` (scene/gaussian_model.py:13) resolve to the HIP-backed packages at the repo root
instead of the CUDA extensions the reference would JIT-compile; (2) installs stand-ins for third-party packages that are
missing from the image (tools/reference_shims.py; real packages are used when they import); (3) executes the script as
`__main__` with the reference directory as sys.path[0], exactly like `python train.py ...` started there.

`--dp N`: data parallelism over camera views for the unmodified train.py -- one process per GPU
(`HIP_VISIBLE_DEVICES=<rank>`, so the `cuda:0` that utils/general_utils.py:167 pins is a different device in every process),
torch.distributed over RCCL, and, applied to the reference's classes from outside before the script starts
(relightable3dgaussian_amd.dp.patch_reference_classes): `Scene.getTrainCameras` -> cameras rank::N of the identically shuffled
list; gradient averaging in front of `GaussianModel.step` / `DirectLightMap.step`; the densification statistics summed over the
ranks inside `add_densification_stats` (`max_radii2D`: max, in front of `densify_and_prune` / `step`); and every file output
(scene.save, torch.save, cfg_args / cameras.json / input.ply, training visualisations, TensorBoard: train.py:186-203,
utils/system_utils.py:52-61, scene/__init__.py:66-79) left to rank 0.  Replicas stay bit-identical
(same seed, same decisions); N views go into every optimizer step, learning rates as given (DESIGN.md section 5 on what that does
to the optimisation path).

There is no CPU path here: without a GPU or the built library the first op fails loudly.  (The tests drive this launcher on a
box without a GPU through tests/run_reference_cpu.py, which passes its own `install_backend`; nothing of that lives here.)"""
import argparse
import os
import runpy
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _hip_backend():
    import bvh_tracing  # noqa: F401  (the drop-in packages: import errors surface here, before the script starts)
    import r3dg_rasterization  # noqa: F401
    import simple_knn  # noqa: F401


def spawn_ranks(ns, argv, entry):
    """The parent of a `--dp N` run: N copies of this command, one per GPU, rank 0 on this terminal, the others' output in
    <replica dir or /tmp>/rank<r>.log (printed if they fail).  Exit code: the first non-zero one."""
    import tempfile
    port = _free_port()
    logdir = ns.dp_replica_dir or tempfile.mkdtemp(prefix="r3dg_dp_")
    os.makedirs(logdir, exist_ok=True)
    procs = []
    for r in range(ns.dp):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(ns.dp), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), R3DG_DP_RANK=str(r),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        if not ns.dp_share_device:
            env["HIP_VISIBLE_DEVICES"] = str(r)          # this rank's GPU is its cuda:0 (utils/general_utils.py:167)
        log = None if r == 0 else open(os.path.join(logdir, "rank%d.log" % r), "w")
        procs.append((subprocess.Popen([sys.executable, entry] + list(argv), env=env, stdin=subprocess.DEVNULL,
                                       stdout=log, stderr=subprocess.STDOUT if log else None), log))
    # poll ALL ranks: a rank that dies leaves the others waiting in a collective -- possibly for ever -- so the first non-zero
    # exit ends the run, whichever rank it was
    import time
    code, failed, live = 0, None, set(range(ns.dp))
    while live:
        for r in sorted(live):
            rc = procs[r][0].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0 and code == 0:
                code, failed = rc, r
                for q, _ in procs:
                    if q.poll() is None:
                        q.terminate()
        if live:
            time.sleep(0.2)
    for p, log in procs:
        if log:
            log.close()
    if failed is not None and failed > 0:
        sys.stderr.write("[run_reference] rank %d failed (%d):\n%s\n" % (
            failed, code, open(os.path.join(logdir, "rank%d.log" % failed)).read()[-4000:]))
    return code


def install_data_parallel(ns):
    """Child of a `--dp N` run, before the script starts: process group, class patches, rank-0-only outputs.
    -> init_globals for runpy (rank > 0: an `open` that sends the script's own file writes to /dev/null)."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if ns.dp_backend == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group(ns.dp_backend)
    import scene as ref_scene                                    # the reference's packages (sys.path[0] is the checkout)
    import scene.direct_light_map as ref_light
    import utils.system_utils as ref_system
    from relightable3dgaussian_amd import dp
    dp.patch_reference_classes(ref_scene.Scene, ref_scene.GaussianModel, ref_light.DirectLightMap, rank, world)
    print("[run_reference] rank %d of %d: cameras %d::%d, gradients averaged in front of step(), densification statistics "
          "reduced inside add_densification_stats, file outputs on rank 0" % (rank, world, rank, world), flush=True)
    if rank == 0:
        return {}
    # ---- rank > 0 writes nothing into the model directory -------------------------------------------------------------------
    import builtins
    import torchvision.utils as tvu

    def quiet_open(file, mode="r", *a, **k):
        if any(c in mode for c in "wax+"):
            return builtins.open(os.devnull, mode, *a, **k)
        return builtins.open(file, mode, *a, **k)
    ref_scene.open = quiet_open                  # cameras.json, input.ply (scene/__init__.py:66-79)
    ref_system.open = quiet_open                 # cfg_args (utils/system_utils.py:55)
    ref_system.TENSORBOARD_FOUND = False         # (utils/system_utils.py:60)
    ref_scene.Scene.save = lambda self, iteration: None              # point_cloud.ply (train.py:188)
    tvu.save_image = lambda *a, **k: None                            # training visualisations, eval images (train.py:317)
    real_save = torch.save
    if ns.dp_replica_dir:
        # (debugging / tests) the checkpoints of the other ranks, to compare the replicas: <dir>/rank<r>/<file name>
        mine = os.path.join(ns.dp_replica_dir, "rank%d" % rank)
        os.makedirs(mine, exist_ok=True)
        torch.save = lambda obj, f, *a, **k: real_save(obj, os.path.join(mine, os.path.basename(f)) if isinstance(f, str) else f,
                                                       *a, **k)
    else:
        torch.save = lambda *a, **k: None                            # chkpnt*.pth, env_light_chkpnt*.pth (train.py:192-201)
    return {"open": quiet_open}                  # train.py's own open() calls (eval: <name>_loss.txt)


def main(argv=None, install_backend=None, entry=None):
    """`install_backend`: what stands behind the three extension packages (default: this repo's HIP packages; the CPU tests pass
    their oracle backend); `entry`: the script `--dp` re-executes for every rank (default: this file)."""
    argv = sys.argv[1:] if argv is None else list(argv)
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--reference", default=os.environ.get("R3DG_REFERENCE", "/root/reference"),
                    help="checkout of NJU-3DV/Relightable3DGaussian (default: $R3DG_REFERENCE or /root/reference)")
    ap.add_argument("--quiet-shims", action="store_true")
    ap.add_argument("--patch-rendering-equation", action="store_true",
                    help="apply INTEGRATION.md's optional one-line patch from outside the checkout: "
                         "gaussian_renderer.neilf.rendering_equation = this repo's fused op (same signature)")
    ap.add_argument("--dp", type=int, default=1, help="data parallel over camera views: N processes, one per GPU")
    ap.add_argument("--dp-backend", default="nccl", choices=["nccl", "gloo"], help="nccl = RCCL (default); gloo: tests")
    ap.add_argument("--dp-share-device", action="store_true",
                    help="do not give every rank its own HIP_VISIBLE_DEVICES (tests on a box with fewer devices than ranks)")
    ap.add_argument("--dp-replica-dir", default=None,
                    help="(debugging) ranks > 0 write their checkpoints to <dir>/rank<r>/ instead of dropping them")
    ap.add_argument("script", help="script inside the checkout, e.g. train.py")
    ap.add_argument("args", nargs=argparse.REMAINDER)
    ns = ap.parse_args(argv)
    ref = os.path.abspath(ns.reference)
    script = ns.script if os.path.isabs(ns.script) else os.path.join(ref, ns.script)
    if not os.path.isfile(script):
        sys.exit("run_reference: %s does not exist" % script)
    if ns.dp > 1 and "R3DG_DP_RANK" not in os.environ:
        sys.exit(spawn_ranks(ns, argv, entry or os.path.abspath(__file__)))
    for p in (ROOT, os.path.join(ROOT, "tools")):
        if p not in sys.path:
            sys.path.insert(0, p)
    # the reference's `torch.load(checkpoint_path)` calls (scene/gaussian_model.py:359, direct_light_map.py) predate
    # torch 2.6's weights_only=True default, and its checkpoints hold a numpy scalar (spatial_lr_scale)
    os.environ.setdefault("TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD", "1")
    import reference_shims
    reference_shims.install(verbose=not ns.quiet_shims and os.environ.get("R3DG_DP_RANK", "0") == "0")
    (install_backend or _hip_backend)()
    # `python train.py` puts the script's directory first; the reference's own packages (scene, utils, arguments, bvh, ...)
    # must win over same-named directories elsewhere, the three extension packages are not shadowed by anything in it
    sys.path.insert(0, os.path.dirname(script))
    if ns.patch_rendering_equation:
        import gaussian_renderer.neilf as neilf                      # (the reference's module, found through sys.path[0])
        from relightable3dgaussian_amd import shading_ops
        neilf.rendering_equation = shading_ops.rendering_equation    # same signature as neilf.py:339-371
        print("[run_reference] gaussian_renderer.neilf.rendering_equation -> relightable3dgaussian_amd.shading_ops.rendering_equation")
    init_globals = install_data_parallel(ns) if ns.dp > 1 else {}
    sys.argv = [script] + [a for a in ns.args if a != "--"]
    try:
        runpy.run_path(script, init_globals=init_globals, run_name="__main__")
    finally:
        if ns.dp > 1:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()


if __name__ == "__main__":
    main()
