"""the reference's OWN `train.py`, byte for byte, run end to end across this repo's extension
boundary -- in this container, where /root/reference exists and no GPU does, so the three extension modules are backed by
the CPU oracle (tests/run_reference_cpu.py = tools/run_reference.py + tests/reference_cpu_backend.py).  What this proves is the seam, not
the kernels (those are pinned on the GPU by tests/test_reference_gpu.py and test_reference_pipeline_gpu.py): the module
names, call signatures, tuple layouts, opaque state buffers, dtype / shape conventions and the data formats either side
(Blender-format dataset written by synthetic.write_blender_dataset, `points3d.ply`, `chkpnt*.pth`) are what an unmodified
reference checkout needs -- through create_from_pcd (distCUDA2), the training loop with densify_and_prune / reset_opacity /
Adam, checkpoint capture, then stage 2 (`-t neilf -c chkpnt`): create_from_ckpt, update_visibility (create_bvh +
trace_bvh_opacity), the shading integral, env light, and this repo's checkpoint reader on the files train.py wrote.
Skipped where the reference tree is absent (the GPU box)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "train.py")), reason="needs the reference checkout")


def _write_dataset(root, n_views=6, res=40, P=900):
    """A Blender-format scene the reference's loader accepts: views of a small teacher scene rendered by the CPU oracle,
    plus points3d.ply (so readNerfSyntheticInfo does not draw its 100 000 random points, dataset_readers.py:288-299)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import reference_shims as shims
    from oracle import rasterizer as orc
    from relightable3dgaussian_amd import synthetic as syn
    from tests.helpers import fwd_args
    sc = syn.make_scene(P=P, seed=5, stage2=False, scale_log_mean=-2.4)
    cams = syn.orbit_cameras(n_views, width=res, height=res)
    images = []
    for cam in cams:
        case = dict(P=P, W=res, H=res, S=0, bg=torch.ones(3), means3D=sc["xyz"], features=torch.zeros(P, 0),
                    opacity=sc["opacity"], scales=sc["scales"], rotations=sc["rotations"], shs=sc["shs"], degree=3, cam=cam,
                    colors=None, cov3D=None)
        out = orc.rasterize_gaussians(*fwd_args(case)[:-3])
        images.append(torch.from_numpy(np.asarray(out[2], np.float32)).clamp(0, 1))
    syn.write_blender_dataset(root, cams, images, split="train")
    syn.write_blender_dataset(root, cams[:2], images[:2], split="test")
    g = np.random.default_rng(3)
    keep = g.permutation(P)[:600]
    xyz = sc["xyz"].numpy()[keep] + 0.02 * g.standard_normal((600, 3)).astype(np.float32)
    data = np.empty(600, dtype=[("x", "f4"), ("y", "f4"), ("z", "f4"), ("nx", "f4"), ("ny", "f4"), ("nz", "f4"),
                                ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    nrm = sc["normal"].numpy()[keep]
    for i, n in enumerate(("x", "y", "z")):
        data[n], data["n" + n] = xyz[:, i], nrm[:, i]
    rgb = (255 * g.random((600, 3))).astype(np.uint8)
    data["red"], data["green"], data["blue"] = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    shims.PlyData([shims.PlyElement.describe(data, "vertex")]).write(os.path.join(root, "points3d.ply"))
    return len(cams)


def _run(args, timeout=900, launcher=()):
    cmd = [sys.executable, os.path.join(ROOT, "tests", "run_reference_cpu.py"), "--reference", REF] + list(launcher) + ["--"] + args
    env = dict(os.environ, OMP_NUM_THREADS="4" if not launcher else "2", PYTHONPATH=ROOT)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "R3DG_DP_RANK"):
        env.pop(k, None)
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT, stdin=subprocess.DEVNULL)


def test_ply_stand_in_round_trip(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import reference_shims as shims
    data = np.zeros(5, dtype=[("x", "f4"), ("f_dc_0", "f4"), ("red", "u1")])
    data["x"], data["f_dc_0"], data["red"] = np.arange(5), np.linspace(-1, 1, 5), [0, 1, 2, 254, 255]
    for text in (False, True):
        path = os.path.join(tmp_path, "t%d.ply" % text)
        shims.PlyData([shims.PlyElement.describe(data, "vertex")], text=text).write(path)
        back = shims.PlyData.read(path)
        v = back["vertex"]
        assert [p.name for p in back.elements[0].properties] == ["x", "f_dc_0", "red"]
        assert np.array_equal(v["x"], data["x"]) and np.allclose(v["f_dc_0"], data["f_dc_0"]) and v["red"].dtype == np.uint8
        assert np.array_equal(v["red"], data["red"])


def test_reference_train_py_and_relighting_py_run_unchanged(tmp_path):
    data, out1, out2 = (os.path.join(tmp_path, d) for d in ("data", "stage1", "stage2"))
    os.makedirs(data)
    _write_dataset(data)
    # stage 1 (script/run_nerf.sh:7-14 flags; a short schedule that still reaches densify_and_prune and reset_opacity)
    r = _run(["train.py", "-s", data, "-m", out1, "--data_device", "cpu", "--lambda_normal_render_depth", "0.01",
              "--lambda_normal_smooth", "0.01", "--lambda_mask_entropy", "0.1", "--lambda_depth_var", "1e-2",
              "--iterations", "14", "--densify_from_iter", "3", "--densification_interval", "4",
              "--opacity_reset_interval", "9", "--test_interval", "7", "--checkpoint_interval", "14", "--save_interval", "14",
              "--save_training_vis", "--save_training_vis_iteration", "7"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "Training complete." in r.stdout
    ck1 = os.path.join(out1, "chkpnt14.pth")
    assert os.path.isfile(ck1) and os.path.isfile(os.path.join(out1, "point_cloud", "iteration_14", "point_cloud.ply"))
    assert os.path.isfile(os.path.join(out1, "visualize", "000007.png"))
    from relightable3dgaussian_amd import checkpoint
    st1 = checkpoint.restore(ck1)
    assert st1.iteration == 14 and st1.xyz.shape[0] != 600, "densify_and_prune never changed the row count"
    assert all(torch.isfinite(getattr(st1, k)).all() for k in ("xyz", "normal", "features_dc", "scaling", "opacity"))
    assert st1.moments and st1.adam_steps > 0                        # the optimizer state train.py captured
    # stage 2 (script/run_nerf.sh:20-39): from the stage-1 checkpoint, K = 16 rays per Gaussian
    r = _run(["train.py", "-s", data, "-m", out2, "-c", ck1, "--data_device", "cpu", "-t", "neilf", "--sample_num", "16",
              "--position_lr_init", "0.000016", "--position_lr_final", "0.00000016", "--normal_lr", "0.001", "--sh_lr",
              "0.00025", "--opacity_lr", "0.005", "--scaling_lr", "0.0005", "--rotation_lr", "0.0001", "--iterations", "18",
              "--lambda_base_color_smooth", "0", "--lambda_roughness_smooth", "0", "--lambda_light_smooth", "0",
              "--lambda_light", "0.01", "--lambda_env_smooth", "0.01", "--test_interval", "1000", "--checkpoint_interval",
              "18", "--save_interval", "18", "--save_training_vis", "--save_training_vis_iteration", "2",
              # in the real schedule stage 2 starts at iteration 30001, past densify_until_iter (10 000): the neilf render
              # package has no 'weights' entry for add_densification_stats (train.py:161-162)
              "--densify_until_iter", "10"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "Training complete." in r.stdout
    ck2 = os.path.join(out2, "chkpnt18.pth")
    assert os.path.isfile(ck2) and os.path.isfile(os.path.join(out2, "env_light_chkpnt18.pth"))
    st2 = checkpoint.restore(ck2)
    assert st2.iteration == 18 and st2.xyz.shape[0] == st1.xyz.shape[0]          # (no densification in stage 2)
    assert st2.base_color.shape == (st2.xyz.shape[0], 3) and st2.incidents_rest.shape[1:] == (15, 3)
    assert float(st2.base_color.abs().max()) > 0 and float(st2.incidents_dc.abs().max()) > 0, "PBR groups never trained"
    # relighting.py (composition + relight under an environment map, :102-170): two copies of the trained object from the
    # point_cloud.ply train.py wrote (GaussianModel.save_ply -> load_ply), each under its own similarity transform, a camera
    # trajectory, a light that turns with the frames -- update_visibility at K rays, render_neilf(is_training=False), EnvLight
    import json
    ply = os.path.join(out2, "point_cloud", "iteration_18", "point_cloud.ply")
    assert os.path.isfile(ply)
    cfg, cap = os.path.join(tmp_path, "relight_cfg"), os.path.join(tmp_path, "capture")
    os.makedirs(cfg)
    eye = [1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0]
    moved = [0.6, 0, 0, 1.1, 0, 0.6, 0, 0.2, 0, 0, 0.6, 0, 0, 0, 0, 1.0]
    json.dump({"a": {"path": ply, "transform": eye}, "b": {"path": ply, "transform": moved}},
              open(os.path.join(cfg, "transform.json"), "w"))
    from relightable3dgaussian_amd import synthetic as syn
    traj, lights = {}, {}
    for i, cam in enumerate(syn.orbit_cameras(3, width=40, height=32)):
        traj[str(i)] = cam.world_view_transform.t().reshape(-1).tolist()                  # W2C, row major
        a = 0.4 * i
        lights[str(i)] = [float(np.cos(a)), float(-np.sin(a)), 0.0, float(np.sin(a)), float(np.cos(a)), 0.0, 0.0, 0.0, 1.0]
    json.dump({"camera": {"width": 40, "height": 32, "fov": 40}, "trajectory": traj}, open(os.path.join(cfg, "trajectory.json"), "w"))
    json.dump({"transform": lights}, open(os.path.join(cfg, "light_transform.json"), "w"))
    r = _run(["relighting.py", "-co", cfg, "-e", os.path.join(REF, "env_map", "envmap3.png"), "--output", cap, "--sample_num",
              "16", "--capture_list", "pbr_env,render_env,base_color,normal", "-bg", "0"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "Totally %d points loaded." % (2 * st2.xyz.shape[0]) in r.stdout
    from PIL import Image
    for kind in ("pbr_env", "render_env", "base_color", "normal"):
        for i in range(3):
            img = np.asarray(Image.open(os.path.join(cap, kind, "frame_%d.png" % i)))
            assert img.shape[:2] == (32, 40) and img.std() > 0, (kind, i)


def _flat_tensors(obj, prefix=""):
    """(name, tensor) of everything tensor-like inside a captured checkpoint (tuples, lists, dicts, Parameters)."""
    if isinstance(obj, torch.Tensor):
        yield prefix, obj.detach()
    elif isinstance(obj, dict):
        for k in sorted(obj, key=str):
            yield from _flat_tensors(obj[k], "%s.%s" % (prefix, k))
    elif isinstance(obj, (list, tuple)):
        for i, v in enumerate(obj):
            yield from _flat_tensors(v, "%s[%d]" % (prefix, i))


def test_reference_train_py_data_parallel_keeps_replicas_identical(tmp_path):
    """`run_reference.py --dp 2` around the reference's unmodified train.py -- two processes (gloo here, RCCL on
    a GPU node), cameras rank::2 of the identically shuffled list, gradients averaged in front of GaussianModel.step /
    DirectLightMap.step, the densification statistics summed over the ranks inside add_densification_stats, file outputs on rank 0 only.
    Stage 1 across three densifications and an opacity reset, then stage 2 from its checkpoint: the replicas' checkpoints
    (parameters, Adam moments, statistics) are bit-identical and the model directory holds ONE set of files."""
    data, out1, out2, rep1, rep2 = (os.path.join(tmp_path, d) for d in ("data", "stage1", "stage2", "replicas1", "replicas2"))
    os.makedirs(data)
    _write_dataset(data)
    dp = ["--dp", "2", "--dp-backend", "gloo", "--dp-share-device", "--quiet-shims"]
    r = _run(["train.py", "-s", data, "-m", out1, "--data_device", "cpu", "--lambda_normal_render_depth", "0.01",
              "--lambda_normal_smooth", "0.01", "--lambda_mask_entropy", "0.1", "--lambda_depth_var", "1e-2",
              "--iterations", "14", "--densify_from_iter", "3", "--densification_interval", "4",
              "--opacity_reset_interval", "9", "--test_interval", "7", "--checkpoint_interval", "14", "--save_interval", "14",
              "--save_training_vis", "--save_training_vis_iteration", "7"], launcher=dp + ["--dp-replica-dir", rep1])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "rank 0 of 2: cameras 0::2" in r.stdout and "Training complete." in r.stdout
    ck1 = os.path.join(out1, "chkpnt14.pth")
    other1 = os.path.join(rep1, "rank1", "chkpnt14.pth")
    assert os.path.isfile(ck1) and os.path.isfile(other1)
    a, b = torch.load(ck1, weights_only=False), torch.load(other1, weights_only=False)
    assert a[1] == b[1] == 14
    ta, tb = dict(_flat_tensors(a[0])), dict(_flat_tensors(b[0]))
    assert ta.keys() == tb.keys() and len(ta) > 20
    for k in ta:
        assert ta[k].shape == tb[k].shape and torch.equal(ta[k], tb[k]), "replicas diverged in stage 1: " + k
    rows = [t.shape[0] for k, t in ta.items() if t.dim() == 2 and t.shape[1] == 3][0]
    assert rows != 600, "densify_and_prune never changed the row count"
    # the other rank left nothing in the model directory: one checkpoint, one point cloud, one visualisation per saved iteration
    listing = sorted(os.path.relpath(os.path.join(d, f), out1) for d, _, fs in os.walk(out1) for f in fs)
    assert listing.count("chkpnt14.pth") == 1 and not any("rank" in f for f in listing), listing
    assert os.path.isfile(os.path.join(out1, "point_cloud", "iteration_14", "point_cloud.ply"))
    assert "Training complete." in open(os.path.join(rep1, "rank1.log")).read()
    # a data-parallel run is a different optimisation path than one process (two views per step), not a reordering of it:
    # the two ranks really saw different cameras
    assert "cameras 1::2" in open(os.path.join(rep1, "rank1.log")).read()
    # ---- stage 2 from the stage-1 checkpoint: environment light (DirectLightMap.step) included -----------------------------
    r = _run(["train.py", "-s", data, "-m", out2, "-c", ck1, "--data_device", "cpu", "-t", "neilf", "--sample_num", "16",
              "--position_lr_init", "0.000016", "--position_lr_final", "0.00000016", "--normal_lr", "0.001", "--sh_lr",
              "0.00025", "--opacity_lr", "0.005", "--scaling_lr", "0.0005", "--rotation_lr", "0.0001", "--iterations", "18",
              "--lambda_base_color_smooth", "0", "--lambda_roughness_smooth", "0", "--lambda_light_smooth", "0",
              "--lambda_light", "0.01", "--lambda_env_smooth", "0.01", "--test_interval", "1000", "--checkpoint_interval",
              "18", "--save_interval", "18", "--densify_until_iter", "10"], launcher=dp + ["--dp-replica-dir", rep2])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    for name in ("chkpnt18.pth", "env_light_chkpnt18.pth"):
        a = torch.load(os.path.join(out2, name), weights_only=False)
        b = torch.load(os.path.join(rep2, "rank1", name), weights_only=False)
        ta, tb = dict(_flat_tensors(a[0])), dict(_flat_tensors(b[0]))
        assert ta.keys() == tb.keys() and ta
        for k in ta:
            assert torch.equal(ta[k], tb[k]), "replicas diverged in stage 2 (%s): %s" % (name, k)
    env = torch.load(os.path.join(out2, "env_light_chkpnt18.pth"), weights_only=False)[0][0]
    assert float(env.detach().std()) > 0


def test_data_parallel_reference_patches_on_stand_in_classes():
    """relightable3dgaussian_amd.dp.patch_reference_classes on stand-ins with the reference's attribute names, one process
    (no group: the collectives are skipped): the camera shard, and that step() / densify_and_prune still reach the originals."""
    from relightable3dgaussian_amd import dp
    calls = []

    class Scene:
        def getTrainCameras(self, scale=1.0):
            return list(range(10))

    class Model:
        def __init__(self):
            self.optimizer = torch.optim.Adam([torch.nn.Parameter(torch.zeros(3))], lr=0.1)
            self.xyz_gradient_accum = self.normal_gradient_accum = self.denom = self.weights_accum = torch.zeros(4, 1)
            self.max_radii2D = torch.zeros(4)

        def step(self):
            calls.append("step")

        def densify_and_prune(self, a, b=2):
            calls.append(("densify", a, b))

        def add_densification_stats(self, v, f, w):
            calls.append("stats")

    class Light(Model):
        def step(self):
            calls.append("light")
    saved = dp.patch_reference_classes(Scene, Model, Light, rank=1, world=4)
    assert Scene().getTrainCameras() == [1, 5, 9] and Scene().getTrainCameras(2.0) == [1, 5, 9]
    m, l = Model(), Light()
    m.step(), l.step(), m.densify_and_prune(7, b=3), m.add_densification_stats(None, None, None)
    assert calls == ["step", "light", ("densify", 7, 3), "stats"]
    dp.unpatch_reference_classes(saved)
    assert Scene().getTrainCameras() == list(range(10))
