"""Implementation of bench.py (kept in the package so tests can import pieces of it)."""
import json
import math
import os
import time

import torch
import torch.distributed as dist

from . import _lib, synthetic as syn
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


class GaussianParams:
    """Raw (pre-activation) parameters of the synthetic scene, as the reference's GaussianModel holds them
    (scene/gaussian_model.py:183-232): log-scales, unnormalised quaternions, opacity logits, SH dc/rest."""

    def __init__(self, scene, device, stage2):
        def p(t):
            return torch.nn.Parameter(t.to(device).contiguous())
        self.xyz = p(scene["xyz"])
        self.normal = p(scene["normal"])
        self.scaling = p(torch.log(scene["scales"]))
        self.rotation = p(scene["rotations"])
        self.opacity = p(torch.logit(scene["opacity"].clamp(1e-4, 1 - 1e-4)))
        self.features_dc = p(scene["shs"][:, :1].clone())
        self.features_rest = p(scene["shs"][:, 1:].clone())
        self.stage2 = stage2
        if stage2:
            self.base_color = p(torch.logit(((scene["base_color"] - 0.03) / 0.77).clamp(1e-4, 1 - 1e-4)))
            self.roughness = p(torch.logit(((scene["roughness"] - 0.09) / 0.9).clamp(1e-4, 1 - 1e-4)))
            self.incidents_dc = p(scene["incidents"][:, :1].clone())
            self.incidents_rest = p(scene["incidents"][:, 1:].clone())
            self.env = p(scene["env"].clone())

    def parameters(self):
        names = ["xyz", "normal", "scaling", "rotation", "opacity", "features_dc", "features_rest"]
        if self.stage2:
            names += ["base_color", "roughness", "incidents_dc", "incidents_rest", "env"]
        return [getattr(self, n) for n in names]

    # activations exactly as the reference applies them before calling the op (gaussian_model.py:183-232)
    def get_scaling(self):
        return torch.exp(self.scaling)

    def get_rotation(self):
        return torch.nn.functional.normalize(self.rotation)

    def get_opacity(self):
        return torch.sigmoid(self.opacity)

    def get_shs(self):
        return torch.cat([self.features_dc, self.features_rest], 1)

    def get_normal(self):
        return torch.nn.functional.normalize(self.normal, dim=-1, eps=1e-3)


def raster_settings(cam, bg, sh_degree=3):
    return GaussianRasterizationSettings(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy, cam.cx, cam.cy,
                                         bg, 1.0, cam.world_view_transform, cam.full_proj_transform, sh_degree,
                                         cam.camera_center, False, True, True, False)


def render_stage1(params, cam, bg):
    """The op-level content of gaussian_renderer/render.py:15-130 (features = [normal, depth, depth^2], S=5)."""
    means3D = params.xyz
    means2D = torch.zeros_like(means3D, requires_grad=True)
    xyz_h = torch.cat([means3D, torch.ones_like(means3D[:, :1])], -1)
    depths = (xyz_h @ cam.world_view_transform)[:, 2:3]
    features = torch.cat([params.get_normal(), depths, depths.square()], -1)
    outs = GaussianRasterizer(raster_settings(cam, bg))(
        means3D, means2D, params.get_opacity(), shs=params.get_shs(), scales=params.get_scaling(),
        rotations=params.get_rotation(), features=features)
    return outs


def loss_stage1(outs, gt, image_mask=None, weights=None, iteration=0):
    """The reference's stage-1 objective (gaussian_renderer/render.py:137-223, flags of script/run_nerf.sh:7-14)."""
    from .train_step import stage1_loss
    return stage1_loss(outs, gt, image_mask, weights, iteration)


def _algorithmic_bytes(stage, P, R, N, S, K=64, S_bwd=None, chain_kernel=False):
    """SURVEY.md 8(d) per-unit figures x the units one launch processes (fp32).  `S_bwd`: the feature channels the TIMED
    backward launch carries (the iteration passes `active_features`: channels whose upstream gradient is zero are neither read
    nor written) -- round 3 priced the launch with all S channels, twice what it moves.  The instance-ordering stages are priced with
    the bytes of the formulation that RUNS (direct tile binning), not with the reference formulation's
    (its 152 R for the global radix sort is kept in `reference_formulation_MB` of the bench line):
      duplicate_with_keys (profile stage of tile_count + tile_scan + tile_emit): the projection outputs of every Gaussian are
        read twice (count pass, emit pass: means2D 8 + depth 4 + radius 4 + tiles_touched 4 = 20 B each time) and every
        instance is written ONCE as its 8-byte (depth | index) entry;
      sort_pairs (tile_order + tile_sort): each entry is read and written once by the in-LDS sort (2 x 8 R) and leaves as the
        4-byte point_list entry + the 8-byte sorted key the state layout promises (12 R)."""
    return {
        "preprocess": 311.0 * P + 8.0 * P,
        "duplicate_with_keys": 40.0 * P + 8.0 * R,
        "sort_pairs": 28.0 * R,
        "identify_tile_ranges": 8.0 * R,
        "render_forward": (44.0 + 4 * S + 8.0) * R + (28.0 + 4 * S) * N,
        "pseudo_normal": 44.0 * N,
        "render_backward": (124.0 + 12 * (S if S_bwd is None else S_bwd)) * R + (28.0 + 4 * (S if S_bwd is None else S_bwd)) * N,
        "preprocess_backward": (679.0 + 4 * S) * P,
        # live shading model over the fixed ray set (round 4: no direction stream): per sample visibility 4 + lookup record 8;
        # per Gaussian: material 28 + normal 12 + view 12 + ray normal 12 + rotated coefficients 192 + validity 1 + 7 outputs 28
        # (285), backward + upstream gradients 24 + gradients written 28 + coefficient gradient 192 (529).  (SURVEY 8d priced the
        # reference's cache layout, direction 12 + visibility 4 per sample: (260+16K) / (476+16K) -- what the GENERAL kernels move.)
        # (+ 28: the seven results a second time, straight into the rasterizer's feature rows -- no pack kernel)
        "shade_forward": (313.0 + 12 * K) * P,
        "shade_backward": (529.0 + 12 * K) * P,
        "shade_forward_general": (260.0 + 16 * K) * P,
        "shade_backward_general": (476.0 + 16 * K) * P,
        # fixed ray set: the coefficient rotation, once each way (48 floats + the normal read, 48 floats written)
        # (chain_kernel: the incident-light chain as one kernel -- rotated gradient 192 + p, m, v 576 in; p, m, v 576 + gradient
        # 192 + rotated coefficients 192 out; normal 12, validity 1 -- whose Adam traffic then leaves `adam_step`)
        "shade_frs_aux": 1741.0 * P if chain_kernel else 2 * (48.0 + 3 + 48) * 4 * P,
        # relight under a fixed light: 12 B of cached transport per sample; per Gaussian albedo, roughness, normal, view
        # direction (40 B) + 16 cached constants (64 B) read, 19 outputs (76 B) written
        "shade_forward_transport": (180.0 + 12 * K) * P,
        # Adam: 28 B per parameter float (p, g, m, v read; p, m, v written); 127 floats per Gaussian in stage 2
        "adam_step": 28.0 * (127 - (48 if chain_kernel else 0)) * P,
        # glue: activations 68 B read + 72 B written + the nine feature-row columns they own (36 B); feature row (general
        # shading kernels only) 40+76 read, 64 written; loss 27 maps read, 20 written
        "stage2_activate": 176.0 * P,
        "stage2_pack_features": 180.0 * P,
        "stage2_unpack_gradients": 164.0 * P,
        "stage2_activate_backward": (68.0 + 64 + 28 + 44 + 72) * P,
        "stage2_loss": (27.0 + 20.0) * 4 * N,
        # relight frame glue: S=28 feature row (19 + 13 floats read, 28 written); composite (19 maps read ... 76 B / pixel)
        "relight_pack_features": 188.0 * P,
        "relight_compose": 76.0 * N,
    }[stage]


def cpu_baseline(scene, cams, S, budget_s, points=300_000, width=800, height=800):
    """The reported CPU baseline (north_star): a pure-PyTorch rasterize forward of the SAME view of the SAME scene
    (oracle/torch_rasterizer.py, the autograd restatement of forward.cu) timed on the host cores in a bounded child
    process; core count stated.  `extra.c_port` keeps the round-1 figure: the oracle C port (1 thread) doing rasterize
    forward+backward view after view for ~budget_s seconds."""
    same_view = pytorch_cpu_rasterize(points, width, height, S, timeout_s=240)
    extra = {"c_port": c_port_baseline(scene, cams, S, budget_s),
             "pytorch_cpu_config0": pytorch_cpu_rasterize(2000, 400, 400, 5, 60, config0=True)}
    sec = same_view.get("seconds_per_view")
    return dict(value=None if sec is None else round(1.0 / sec, 4), unit="views/s (rasterize forward)",
                cores=same_view.get("threads"), kind="port", seconds_per_view=sec if sec is not None else ">240",
                sample=same_view.get("what", same_view.get("failed")), logical_cpus=os.cpu_count(), extra=extra)


def c_port_baseline(scene, cams, S, budget_s):
    """Oracle (C port of the reference algorithm, 1 thread) rasterize forward+backward of the SAME scene, one view
    after another until ~budget_s seconds of CPU work are spent (a bounded sample of the GPU workload)."""
    import numpy as np
    from oracle import rasterizer as orc
    P = scene["xyz"].shape[0]
    feat = torch.rand(P, S)
    t_f = t_b = 0.0
    views = 0
    R = 0
    while t_f + t_b < budget_s and views < len(cams):
        cam = cams[views]
        args = (torch.ones(3), scene["xyz"], feat, None, scene["opacity"], scene["scales"], scene["rotations"], 1.0,
                None, cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, cam.cx, cam.cy,
                cam.image_height, cam.image_width, scene["shs"], 3, cam.camera_center)
        t0 = time.time()
        out = orc.rasterize_gaussians(*args)
        t_f += time.time() - t0
        H, W = cam.image_height, cam.image_width
        g = [np.full((c, H, W), 1.0 / (H * W), np.float32) for c in (3, 1, 1, S)]
        t0 = time.time()
        orc.rasterize_gaussians_backward(args[0], args[1], feat, out[9], None, args[5], args[6], 1.0, None, args[9],
                                         args[10], args[11], args[12], g[0], g[1], g[2], g[3], args[17], 3, args[19],
                                         out[-1], True)
        t_b += time.time() - t0
        views += 1
        R = out[0]
    return dict(value=round(views / (t_f + t_b), 4), unit="iters/s", cores=1,
                sample="%d views %dx%d, %d Gaussians, R~%d, rasterize fwd+bwd S=%d only (oracle C port, fp32, 1 thread; "
                       "fwd %.1fs + bwd %.1fs of CPU time); shading/Adam not included" % (
                           views, W, H, P, R, S, t_f, t_b))


_CPU_RASTERIZE_SCRIPT = """
import json, os, sys, time
sys.path.insert(0, %(root)r)
import torch
from oracle import torch_rasterizer as tr
from relightable3dgaussian_amd import synthetic as syn
torch.set_num_threads(%(threads)d)
P, WID, HEI, S = %(P)d, %(width)d, %(height)d, %(S)d
if %(config0)r:
    sc0 = syn.make_scene(P=P, seed=0, stage2=False, scale_log_mean=-3.0)
    cam0 = syn.orbit_cameras(4, width=WID, height=HEI)[0]
else:                                      # the bench's own scene and its view 0
    sc0 = syn.make_scene(P=P, seed=0, stage2=False)
    cam0 = syn.orbit_cameras(100, width=WID, height=HEI)[0]
f0 = torch.rand(P, S)
t0 = time.time()
with torch.no_grad():
    o0 = tr.rasterize(torch.ones(3), sc0["xyz"], f0, None, sc0["opacity"], sc0["scales"], sc0["rotations"], 1.0, None,
                      cam0.world_view_transform, cam0.full_proj_transform, cam0.tanfovx, cam0.tanfovy, cam0.cx, cam0.cy,
                      HEI, WID, sc0["shs"], 3, cam0.camera_center)
print(json.dumps(dict(seconds_per_view=round(time.time() - t0, 3), num_rendered=int(o0["num_rendered"]))))
"""


def pytorch_cpu_rasterize(points, width, height, S, timeout_s=60, config0=False):
    """Pure-PyTorch CPU rasterize forward (oracle/torch_rasterizer.py) of one view, timed on the host cores.  Runs in a
    child process with a hard time limit and a thread count taken from the CPU affinity mask (capped at 8): on a box
    whose container sees more logical CPUs than it may use, os.cpu_count() OpenMP threads make the thousands of tiny
    tensor ops of the tile loop crawl for minutes -- a reported baseline must never be able to stall the bench.
    config0=True is BASELINE configs[0] (2k random Gaussians, 400x400), else the bench's own scene, view 0."""
    import subprocess
    import sys
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    threads = max(1, min(8, usable))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="")
    what = "pure-PyTorch CPU rasterize forward (oracle/torch_rasterizer.py), %d Gaussians, one %dx%d view, S=%d, %d threads" % (
        points, width, height, S, threads)
    script = _CPU_RASTERIZE_SCRIPT % dict(root=root, threads=threads, P=points, width=width, height=height, S=S,
                                          config0=bool(config0))
    try:
        r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True,
                           timeout=timeout_s, env=env, stdin=subprocess.DEVNULL)
        line = [x for x in r.stdout.splitlines() if x.startswith("{")]
        if r.returncode != 0 or not line:
            return {"failed": (r.stderr or r.stdout)[-300:], "threads": threads}
        doc = json.loads(line[-1])
        return dict(seconds_per_view=doc["seconds_per_view"], threads=threads, logical_cpus=os.cpu_count(),
                    what=what + ", num_rendered=%d" % doc["num_rendered"])
    except subprocess.TimeoutExpired:
        return {"failed": what + ": no result within %d s" % timeout_s, "threads": threads, "logical_cpus": os.cpu_count()}
    except Exception as e:
        return {"failed": repr(e)}


def _kernel_names(stage):
    """Kernel(s) a profile stage times, dominant first (names as tools/pmc_*.py shorten them)."""
    return {"sort_pairs": ["tile_sort_small_kernel", "partition_scatter_kernel"],
            "duplicate_with_keys": ["tile_emit_kernel", "duplicate_with_keys_kernel"],      # (stage name kept from K5)
            # (the training iteration runs the fixed-ray-set kernels, csrc/shading_frs.hpp; the row kernel is what a caller with
            # other caches gets)
            "shade_forward": ["shade_forward_frs_kernel", "shade_forward_row_kernel"],
            "shade_backward": ["shade_backward_frs_kernel", "shade_backward_kernel"],
            "shade_forward_general": ["shade_forward_row_kernel"], "shade_backward_general": ["shade_backward_kernel"],
            "shade_frs_aux": ["frs_incident_chain_kernel", "frs_rotate_kernel"],
            "render_forward": ["render_forward_wave_kernel", "render_forward_kernel"],
            "render_backward": ["render_backward_wave_kernel", "render_backward_kernel"],
            "shade_forward_transport": ["shade_forward_transport_kernel"],
            "adam_step": ["adam_kernel"], "bvh_trace": ["trace_opacity_phased_kernel", "trace_opacity_persistent_kernel"],
            }.get(stage, [stage + "_kernel", stage])


def _pmc_file_rows(kind, stage, workload=None):
    """(kernel name, its row, file name, stale?) from the newest committed counter file profiles/rNN_pmc_<kind>[_<workload>].json
    that holds one of `stage`'s kernels (`workload`: None = the training kernels' pass, "relight" = the relight frame's pass --
    the same kernel runs with other template instances and sizes there; falls back to the training pass).  stale: the file carries the sha256 of the kernel's source at collection time (tools/pmc_*.py,
    kernel_sources.stamp) and it differs from the tree this process runs from -- None when the file predates the stamps."""
    import glob
    from . import kernel_sources
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    paths = []
    for w in ([workload] if workload else []) + [None]:
        paths += sorted(glob.glob(os.path.join(root, "profiles", "r[0-9][0-9]_pmc_%s%s.json" % (kind, "_" + w if w else ""))),
                        reverse=True)
    for path in paths:
        try:
            doc = json.load(open(path))
        except Exception:
            continue
        for name in _kernel_names(stage):
            v = doc.get("kernels", {}).get(name)
            if v is None:
                continue
            stamp = (doc.get("sources") or {}).get(name)
            stale = None if not stamp else bool(stamp.get("sha256") != kernel_sources.source_of(name)[1])
            return name, v, os.path.basename(path), stale
    return None


def pmc_traffic(stage, workload=None):
    """HBM bytes per launch of `stage`'s kernel from the committed rocprofv3 PMC passes (profiles/*_pmc_traffic*.json;
    FETCH_SIZE / WRITE_SIZE collected in separate passes on the same workload), or None."""
    hit = _pmc_file_rows("traffic", stage, workload)
    if hit is None:
        return None
    name, v, source, stale = hit
    return dict(bytes_corrected=v["hbm_bytes_corrected"], bytes_raw=v["hbm_bytes_raw"], kernel=name, source=source, stale=stale,
                note="corrected = 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md); profiles/r02_pmc_calibration.json: "
                     "the factor 2 holds for streaming reads, random sub-line gathers are counted at 64 B per "
                     "request (factor 1) and a 4-byte atomic as 32 written bytes, so for the gather / atomic "
                     "heavy tile kernels the truth lies between bytes_raw and bytes_corrected")


def pmc_valu(stage, workload=None):
    """VALU-issue evidence for `stage`'s kernel from the committed rocprofv3 SQ-counter passes (profiles/*_pmc_valu*.json,
    tools/pmc_valu.py): fraction of the kernel's duration the SIMDs spend issuing VALU instructions, occupancy, LDS bank
    conflicts -- shown beside every HBM fraction because the tile and shading kernels are VALU-bound -- or None."""
    hit = _pmc_file_rows("valu", stage, workload)
    if hit is None:
        return None
    name, v, source, stale = hit
    out = {k: v[k] for k in ("valu_issue_frac", "valu_busy_frac", "waves_per_simd", "lds_bank_conflict_frac",
                             "lds_issue_frac", "trans_frac", "mfma_busy_frac", "salu_issue_frac", "wait_frac",
                             "issue_stall_frac", "wait_lds_frac", "vgprs", "lds_bytes") if k in v}
    out.update(kernel=name, source=source, stale=stale)
    return out


def valu_bound_of(stage, measured_ms, workload=None):
    """The VALU-issue bound of `stage`'s kernel from the committed SQ-counter pass (profiles/*_pmc_valu*.json, collected on the
    launch configuration the iteration runs): a wave64 VALU instruction occupies its SIMD's issue port for 4 cycles (16 lanes
    per cycle; v_pk_* and transcendentals longer, so this is a LOWER bound on the time), the device has 1024 SIMDs:
        bound_ms = SQ_INSTS_VALU x 4 / (1024 x clock);   frac = bound_ms / the launch's HIP-event time in the iteration."""
    hit = _pmc_file_rows("valu", stage, workload)
    if hit is None:
        return None
    name, v, source, stale = hit
    if "SQ_INSTS_VALU" not in v.get("counters_mean_per_dispatch", {}) or not v.get("clock_ghz"):
        return None
    wi = float(v["counters_mean_per_dispatch"]["SQ_INSTS_VALU"])
    clock = float(v["clock_ghz"])
    bound_ms = wi * 4.0 / (1024.0 * clock * 1e9) * 1e3
    return dict(wave_instr=round(wi), cycles_per_instr=4, simds=1024, clock_ghz=clock, bound_ms=round(bound_ms, 4),
                frac=None if not measured_ms else round(bound_ms / measured_ms, 4), valu_busy_frac=v.get("valu_busy_frac"),
                duration_ms_under_pmc=None if v.get("duration_us_under_pmc") is None else round(v["duration_us_under_pmc"] / 1e3, 4),
                kernel=name, source=source, stale=stale)


def kernel_table(prof, n_sampled, P, R, N, S, K, rename=None, S_bwd=None, workload=None, alone=None, chain_kernel=False):
    """{stage: avg_ms, launches, ms per iteration/frame, algorithmic MB, achieved GB/s, fraction of the 8 TB/s HBM peak,
    VALU-issue fraction (committed PMC evidence)} from the in-library HIP-event timing.
    `alone` = (prof, n_sampled) of a few more iterations / frames run with every launch on ONE stream: `alone_ms_per_iteration`
    is then the stage's time with nothing beside it -- an event bracket on a side stream also counts the time its kernel spent
    sharing the CUs with another stream's kernel (round 4: the 0.03 ms emit kernel read 0.415 ms beside the relight shading
    kernel and was named the dominant kernel of the frame)."""
    kernels = {}
    alone_prof, alone_n = alone if alone is not None else ({}, 1)
    for name, (ms, cnt) in prof.items():
        if cnt == 0:
            continue
        name = (rename or {}).get(name, name)        # (one profile stage, several kernels: the caller knows which one ran)
        # a stage may take several launches per iteration (the Adam groups go out in two or three launches, SSIM in
        # four): `algorithmic_MB` is the stage's bytes per ITERATION, so it is divided by the stage's time per iteration
        per_iter = max(1, round(cnt / n_sampled))
        avg_ms = ms / cnt
        step_ms = avg_ms * per_iter
        try:
            by = _algorithmic_bytes(name, P, R, N, S, K, S_bwd, chain_kernel)
        except KeyError:
            by = None
        row = dict(avg_ms=round(avg_ms, 4), launches=cnt, launches_per_iteration=per_iter,
                   ms_per_iteration=round(step_ms, 4), algorithmic_MB=None if by is None else round(by / 1e6, 1),
                   achieved_GBs=None if by is None else round(by / (step_ms * 1e-3) / 1e9, 1),
                   hbm_frac=None if by is None else round(by / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
        if name == "sort_pairs":
            # what the reference's formulation (one global radix sort of 44-bit keys, K6) would move for the same instances
            row["reference_formulation_MB"] = round(152.0 * R / 1e6, 1)
        v = pmc_valu(name, workload)
        if v is not None:
            row["valu"] = v
        kernels[name] = row
    for name, (ms, cnt) in alone_prof.items():
        name = (rename or {}).get(name, name)
        if cnt and name in kernels:
            row = kernels[name]
            row["alone_ms_per_iteration"] = round(ms / cnt * max(1, round(cnt / max(1, alone_n))), 4)
            if row.get("algorithmic_MB"):
                row["alone_hbm_frac"] = round(row["algorithmic_MB"] * 1e6 / (row["alone_ms_per_iteration"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    if "adam_step" in kernels:
        #  inside the iteration the gradients this launch reads were written microseconds earlier and sit in
        # the 256 MiB last-level cache: its in-iteration rate can exceed what HBM alone delivers
        kernels["adam_step"]["note"] = ("hbm_frac inside the iteration is LLC-ASSISTED (the gradients were just written and sit in the "
                                        "256 MiB Infinity Cache) -- not an HBM figure; the kernel's HBM fraction is the COLD one: "
                                        "tools/kbench_adam.py, 0.154 ms per 958 MB = 6.21 TB/s = 0.78 of the peak at 300k Gaussians "
                                        "(DESIGN.md section 6), and alone_hbm_frac (one stream, warm) beside it")
    return kernels


def roofline_of(kernels, note, workload=None):
    """The dominant kernel of the timed region.  `achieved` / `peak` / `frac` are the HBM figures the contract asks for
    (algorithmic bytes of THIS launch configuration / HIP-event time in the timed region / 8 TB/s); `bound` names what the
    counters say limits the kernel: "valu" when its VALU-issue bound (`valu_bound`) explains more of the launch's time than its
    HBM fraction does -- "unknown" when the committed counters describe an OLDER source of the kernel than the one just timed
    (`stale`).  WHICH kernel is dominant is decided on durations that stream concurrency does not inflate: the smaller of the
    time inside the iteration and the time on one stream (`alone_ms_per_iteration`, see kernel_table) when the latter exists."""
    def own_time(k):
        row = kernels[k]
        t = row.get("ms_per_iteration", 0.0)
        return min(t, row["alone_ms_per_iteration"]) if row.get("alone_ms_per_iteration") else t
    dom = max(kernels, key=own_time)
    row = kernels[dom]
    ach = row["achieved_GBs"]
    tr = pmc_traffic(dom, workload)
    vb = valu_bound_of(dom, row["ms_per_iteration"], workload)
    frac = None if ach is None else round(ach / HBM_PEAK_GBS, 4)
    flags = [x.get("stale") for x in (tr, vb) if x]
    # True: a counter file describes an older source of this kernel; None: no file says which source it describes; False: current
    stale = True if any(f is True for f in flags) else (None if (not flags or any(f is None for f in flags)) else False)
    bound = "hbm"
    if vb is not None and vb.get("frac") is not None and (frac is None or vb["frac"] > frac):
        bound = "valu"
    if stale:
        bound = "unknown"
    return dict(bound=bound, kernel=dom, achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=frac,
                traffic=None if tr is None else tr["bytes_corrected"], traffic_detail=tr, valu_bound=vb,
                valu=row.get("valu"), avg_kernel_ms=row["ms_per_iteration"], alone_kernel_ms=row.get("alone_ms_per_iteration"),
                algorithmic_MB=row.get("algorithmic_MB"), stale=stale,
                counters_source=", ".join(sorted({x["source"] for x in (tr, vb) if x})) or None,
                selected_by="largest own time (min of in-iteration and one-stream HIP-event time)" if any(
                    "alone_ms_per_iteration" in v for v in kernels.values()) else "largest HIP-event time in the timed region",
                note=note)


@torch.no_grad()
def trace_visit_stats(renderer, seconds):
    """SURVEY.md 8(d) K18 "report rays/s and node-visits/s": the renderer's visibility update once more with the COUNTING
    instantiation of the trace kernel (R3DG_OPT_TRACE_COUNT_VISITS; same traversal, results discarded) -- node steps (a slab test
    of both children) and leaf steps (one Gaussian evaluated) per ray, and per second of the UNCOUNTED update that took
    `seconds` (the time `visibility_Mrays_per_s` is quoted on: tree build + ray generation + trace)."""
    from . import bvh_ops
    from .train_step import update_visibility
    r = renderer
    try:
        _lib.set_option("TRACE_COUNT_VISITS", 1)
        bvh_ops.VISITS[:] = [0, 0, 0]
        update_visibility(r.xyz, r.a_scales, r.a_rot, r.a_opacity, r.a_normal, r.K)
        nodes, leaves, rays = bvh_ops.VISITS
    finally:
        _lib.set_option("TRACE_COUNT_VISITS", 0)
        bvh_ops.VISITS[:] = [0, 0, 0]
    if rays == 0:
        return None
    return dict(rays=rays, node_steps=nodes, leaf_steps=leaves, node_steps_per_ray=round(nodes / rays, 2),
                leaf_steps_per_ray=round(leaves / rays, 2), node_visits_per_s=round((nodes + leaves) / seconds),
                note="node_visits_per_s = (node steps + leaf steps) / visibility_seconds; a node step tests both children")


@torch.no_grad()
def relight_bench(params, cams, dev, frames, K):
    """Relight / eval rendering (relighting.py:114-170, neilf.py:98-209 eval branch): per frame the shading integral at
    K samples under a fixed HDR environment map + rasterize forward with the S=28 eval feature row + the environment
    composite, through relight.RelightRenderer (all glue in HIP); the same frames through the drop-in ops + PyTorch glue
    (relight.frame_reference, the shape the reference's render_view has) are timed beside it."""
    from . import relight
    g = torch.Generator().manual_seed(7)
    envmap = (3.0 * torch.rand(256, 512, 3, generator=g) ** 2).to(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # (R3DG_RELIGHT_CACHE=radiance: A/B against the full integral per frame)
    cache = os.environ.get("R3DG_RELIGHT_CACHE", "transport")
    renderer = relight.RelightRenderer(params, envmap, K,             # builds the BVH and traces P x K visibility rays
                                       cache=cache, regenerate_dirs=os.environ.get("R3DG_RELIGHT_DIRS", "regen") != "load")
    torch.cuda.synchronize()
    t_vis = time.perf_counter() - t0
    try:
        visits = trace_visit_stats(renderer, t_vis)
    except Exception as e:                        # (a side measurement)
        visits = {"failed": repr(e)}
    bg = torch.zeros(3, device=dev)

    def timed(fn, n):
        for i in range(3):
            fn(cams[i])
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(n):
            fn(cams[(3 + i) % len(cams)])
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n

    L = _lib.lib()
    for i in range(3):
        renderer.frame(cams[i], bg)
    torch.cuda.synchronize()
    L.r3dg_profile_enable(1)
    torch.cuda.synchronize()
    R_seen = []
    t = time.perf_counter()
    for i in range(frames):
        L.r3dg_profile_pause(0 if i % 4 == 0 else 1)          # per-kernel HIP events on every 4th frame
        R_seen.append(renderer.frame(cams[(3 + i) % len(cams)], bg)["num_rendered"])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / frames
    prof = _lib.profile_read()
    L.r3dg_profile_enable(0)
    # the same frames with every launch on ONE stream (kernel_table `alone`): what each kernel takes with nothing beside it
    alone = None
    try:
        keep = renderer._order_stream
        renderer._order_stream = None
        renderer.frame(cams[0], bg)
        torch.cuda.synchronize()
        L.r3dg_profile_enable(1)
        n_alone = 6
        for i in range(n_alone):
            renderer.frame(cams[(3 + i) % len(cams)], bg)
        torch.cuda.synchronize()
        alone = (_lib.profile_read(), n_alone)
    except Exception:
        alone = None
    finally:
        renderer._order_stream = keep
        L.r3dg_profile_enable(0)
    dt_ref = timed(lambda cam: relight.frame_reference(renderer, cam, bg), max(3, frames // 4))
    # the same frames with only the sampled radiance cached (the full integral per frame: what a light that changes now and
    # then, or parameters that are still being trained, would pay) -- same process, same visibility caches
    radiance_fps = None
    if cache == "transport":
        try:
            r2 = relight.RelightRenderer.__new__(relight.RelightRenderer)
            r2.__dict__.update(renderer.__dict__)
            r2.cache = "radiance"
            r2._taps = r2._taps_key = r2._taps_ref = r2._light_key = r2._light_ref = None
            r2._light_changes, r2._area_key = 0, None
            r2.shade_out, r2.features = torch.empty_like(renderer.shade_out), torch.empty_like(renderer.features)
            radiance_fps = round(1.0 / timed(lambda cam: r2.frame(cam, bg), max(3, frames // 2)), 2)
            del r2
        except Exception as e:
            radiance_fps = {"failed": repr(e)}
    # a light that turns with every frame (configs/nerf_syn_light, configs/tnt: light_transform.json holds one rotation per
    # frame, relighting.py:162-163): the cached lookups of the static-light frames above are then rebuilt per frame
    rotating = None
    try:
        n_rot = max(3, frames // 2)
        rots = []
        for i in range(n_rot + 3):
            a = 2.0 * math.pi * i / 300.0
            rots.append(torch.tensor([[math.cos(a), -math.sin(a), 0.0], [math.sin(a), math.cos(a), 0.0], [0.0, 0.0, 1.0]],
                                     device=dev))
        it = iter(rots)
        dt_rot = timed(lambda cam: renderer.frame(cam, bg, env_transform=next(it)), n_rot)
        rotating = dict(fps=round(1.0 / dt_rot, 2), ms_per_frame=round(1e3 * dt_rot, 3), frames=n_rot,
                        what="one new light rotation per frame (relighting.py:160-161): the light-independent half of the "
                             "transport is cached once (r3dg_shade_build_split), the lat-long lookup of every sample happens "
                             "inside the per-frame kernel (r3dg_shade_forward_split)")
    except Exception as e:                     # a side measurement: never fail the bench on it
        rotating = {"failed": repr(e)}
    P = params.xyz.shape[0]
    H, W = cams[0].image_height, cams[0].image_width
    R_mean = float(sum(R_seen)) / max(1, len(R_seen))
    kernels = kernel_table(prof, max(1, sum(1 for i in range(frames) if i % 4 == 0)), P, R_mean, H * W, 28, K,
                           rename={"shade_forward": "shade_forward_transport"} if cache == "transport" else None,
                           workload="relight", alone=alone)
    roof = roofline_of(kernels, "relight frame (%s at K=%d + rasterize forward S=28 + composite): achieved = "
                       "algorithmic bytes per launch / HIP-event kernel time" % (
                           "GGX lobe against the cached transport, 12 B per sample," if cache == "transport"
                           else "shading forward", K), workload="relight")
    return dict(relight_fps=round(1.0 / dt, 2), relight_ms_per_frame=round(1e3 * dt, 3), relight_K=K, relight_cache=cache,
                relight_fps_radiance_cache=radiance_fps,
                relight_features=28, relight_fps_pytorch_glue=round(1.0 / dt_ref, 2), visibility_rays=P * K,
                visibility_seconds=round(t_vis, 3), visibility_Mrays_per_s=round(P * K / t_vis / 1e6, 1),
                visibility_node_visits_per_s=(visits or {}).get("node_visits_per_s"), visibility_trace_visits=visits,
                num_rendered=R_mean, roofline_relight=roof, kernels=kernels, relight_rotating_light=rotating,
                relight_note="relight_fps: moving camera under a FIXED light (configs/teaser, configs/nerf_syn): the "
                             "view-independent part of the integral is cached per sample (RelightRenderer's default, "
                             "cache='transport'); relight_fps_radiance_cache: only the looked-up radiance is reused; "
                             "relight_rotating_light: the light turns with every frame -- the light-independent half of the "
                             "transport is reused, the lookup happens in the kernel")


@torch.no_grad()
def relight_bench_sharded(model, cams, dev, frames, K, rank, world, group=None):
    """Relight / eval rendering over `world` ranks (BASELINE.json configs[4] "view-sharded render"): the frames of
    the trajectory are independent units -- the reference walks them one after another (relighting.py:114-185) -- so rank r renders
    frames r, r + W, r + 2W, ... of a trajectory of `frames` x W frames on its own replica of the scene, with NO data-path
    collective.  The one exchange is in the set-up: `update_visibility` traces ceil(P/W) ray bundles per rank against the
    replicated BVH and ONE all-gather assembles the [P,K,1] visibility (train_step.update_visibility).  Timing as the contract
    asks: barrier + synchronize on both sides, MAX over ranks; relight_fps = all frames of all ranks / that time (weak scaling:
    `frames` per rank)."""
    from . import relight
    g = torch.Generator().manual_seed(7)
    envmap = (3.0 * torch.rand(256, 512, 3, generator=g) ** 2).to(dev)
    torch.cuda.synchronize()
    dist.barrier(group=group)
    t0 = time.perf_counter()
    renderer = relight.RelightRenderer(model, envmap, K, process_group=group)       # sharded trace + all-gather
    torch.cuda.synchronize()
    t_vis = time.perf_counter() - t0
    bg = torch.zeros(3, device=dev)
    mine = list(range(rank, frames * world, world))           # this rank's frames of the trajectory
    for i in range(3):
        renderer.frame(cams[(rank + i * world) % len(cams)], bg)
    torch.cuda.synchronize()
    dist.barrier(group=group)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nr = 0
    for f in mine:
        nr += renderer.frame(cams[f % len(cams)], bg)["num_rendered"]
    torch.cuda.synchronize()
    own = time.perf_counter() - t0
    dist.barrier(group=group)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    cpu = dist.get_backend(group) == "gloo"
    t = torch.tensor([elapsed, t_vis], dtype=torch.float64, device="cpu" if cpu else dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    per_rank = [torch.zeros(2, dtype=torch.float64, device=t.device) for _ in range(world)]
    dist.all_gather(per_rank, torch.tensor([own, float(nr)], dtype=torch.float64, device=t.device), group=group)
    elapsed, t_vis = float(t[0]), float(t[1])
    fps_rank = [len(mine) / float(x[0]) for x in per_rank]
    P = renderer.P
    return dict(relight_fps=round(world * len(mine) / elapsed, 2), relight_ms_per_frame=round(1e3 * elapsed / len(mine), 3),
                relight_K=K, relight_cache=renderer.cache, relight_features=28, frames_per_rank=len(mine),
                per_rank_fps_min=round(min(fps_rank), 2), per_rank_fps_max=round(max(fps_rank), 2),
                num_rendered=float(sum(float(x[1]) for x in per_rank)) / max(1, world * len(mine)),
                visibility_rays=P * K, visibility_seconds=round(t_vis, 3),
                visibility_Mrays_per_s=round(P * K / t_vis / 1e6, 1),
                sharding="frames rank::world of a %d-frame trajectory, replicas only (no data-path collective); visibility: "
                         "ceil(P/W) ray bundles per rank + one all-gather" % (frames * world))


# (see bench.py: an OpenMP pool sized for the host runs into the container's CPU quota; tools that import this module directly
# get the cap here)
torch.set_num_threads(min(torch.get_num_threads(), 8))

# the per-kernel HIP events are live on every EVENT_EVERY-th step of the timed region (a step with ~40 event pairs between its
# kernels runs ~13 % slower: at every 8th step the timed block measured 624 it/s against 636 without events)
EVENT_EVERY = 10


def host_cpu_quota():
    """The container's CPU quota and how often it was enforced so far (cgroup v2 cpu.max / cpu.stat; {} when the files are
    absent): a timed region that ran into the quota was stalled by the host, not by the GPU, and the line says so."""
    out = {}
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        out["quota_cores"] = None if quota == "max" else round(float(quota) / float(period), 2)
        for ln in open("/sys/fs/cgroup/cpu.stat"):
            k, v = ln.split()
            if k in ("nr_throttled", "throttled_usec"):
                out[k] = int(v)
    except (OSError, ValueError):
        pass
    return out


def _finite_json(x):
    """A child's JSON document with every non-finite float replaced by a string: the ONE line the parent prints must stay
    strict JSON whatever a side measurement produced (Python's json would print a bare NaN)."""
    if isinstance(x, float) and not math.isfinite(x):
        return repr(x)
    if isinstance(x, dict):
        return {k: _finite_json(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_finite_json(v) for v in x]
    return x


def dp_path_one_rank(args, timeout_s=180, fake_comm_gbs=None, steps=None):
    """The data-parallel iteration (bucketed async all-reduces on RCCL's stream, reduced skip flag, deferred incident-light
    update) over a ONE-rank RCCL group -- what the path's own structure costs before any byte crosses xGMI -- measured by a
    child `bench.py` with R3DG_DP_SINGLE_RANK=1 (fused_step._world_of) on the same workload."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, R3DG_DP_SINGLE_RANK="1", R3DG_DIST_BACKEND="nccl")
    if fake_comm_gbs is not None:
        # priced rehearsal: every bucket's identity all-reduce is followed by a spin of its 8-rank ring time at this bus
        # bandwidth (fused_step._allreduce_async)
        env["R3DG_DP_FAKE_COMM_GBS"] = str(fake_comm_gbs)
        env["R3DG_DP_FAKE_COMM_WORLD"] = "8"
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", str(steps or args.steps), "--warmup",
           str(args.warmup), "--points", str(args.points), "--res", str(args.res), "--width", str(getattr(args, "width", 0)), "--height",
           str(getattr(args, "height", 0)), "--objective", getattr(args, "objective", "nerf"), "--sample-num", str(args.sample_num),
           "--no-cpu-baseline", "--no-other-configs", "--relight-frames", "0", "--repeats", "0"]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env, stdin=subprocess.DEVNULL)
        line = [x for x in r.stdout.splitlines() if x.startswith("{")]
        if r.returncode != 0 or not line:
            return {"failed": (r.stderr or r.stdout)[-400:]}
        doc = _finite_json(json.loads(line[-1]))
        if fake_comm_gbs is not None:
            return dict(assumed_bus_GBs=fake_comm_gbs, iters_per_s_one_rank=doc["value"], ms_per_step=doc["ms_per_step"],
                        exposed_comm_ms=doc.get("exposed_comm_ms"), exposed_comm_ms_by_bucket=doc.get("exposed_comm_ms_by_bucket"),
                        comm_buckets=doc.get("comm_buckets"), predicted_8gpu_iters_per_s=round(8.0 * doc["value"], 1))
        return dict(iters_per_s=doc["value"], ms_per_step=doc["ms_per_step"], exposed_comm_ms=doc.get("exposed_comm_ms"),
                    exposed_comm_ms_by_bucket=doc.get("exposed_comm_ms_by_bucket"), comm_buckets=doc.get("comm_buckets"),
                    reserved_cus_for_comm=doc.get("reserved_cus_for_comm"),
                    what="the same iteration through the data-parallel path over a ONE-rank RCCL group (identity collectives; "
                         "the persistent kernels leave `reserved_cus_for_comm` CUs to RCCL)")
    except subprocess.TimeoutExpired:
        return {"failed": "no result within %d s" % timeout_s}
    except Exception as e:
        return {"failed": repr(e)}


# learning rates of the stage-2 schedules (the bench's headline uses one small rate for every group: throughput does not depend
# on the values, only on WHICH groups train).  run_syn4.sh:27-33 / run_dtu.sh:29-35 freeze the geometry groups.
SYN4_LRS = dict(xyz=0.0, normal=0.0, scaling=0.0, rotation=0.0, opacity=0.0, shs=0.0, shs_rest=0.0, base_color=0.01,
                roughness=0.01, incidents=0.001, incidents_rest=0.0001, env=0.1)


def config_rate(dev, points, width, height, stage=2, sample_num=64, objective="nerf", steps=16, warmup=4, relight_samples=0,
                relight_frames=0, stage_ms=False, scene=None):
    """One of the other BASELINE configurations on the synthetic scene (single GPU, short run): iters/s of the fused training
    iteration with that configuration's objective and schedule (`objective`: "nerf" = script/run_nerf.sh, "syn4" =
    script/run_syn4.sh / run_dtu.sh: edge-aware smoothness terms + frozen geometry), the measured num_rendered, and optionally
    relight FPS at `relight_samples` rays per Gaussian."""
    from . import fused_step, relight, train_step
    given = scene is not None            # (`scene`: a make_scene-format dict instead of the i.i.d. synthetic one, e.g. trained_scene.py)
    if not given:
        scene = syn.make_scene(P=points, seed=0, stage2=stage == 2)
    points = scene["xyz"].shape[0]
    cams = [c.to(dev) for c in syn.orbit_cameras(100, width=width, height=height)[:4]]
    bg = torch.ones(3, device=dev)
    params = GaussianParams(scene, dev, stage == 2)
    with torch.no_grad():
        teacher = GaussianParams(scene if given else syn.make_scene(P=points, seed=0, stage2=False), dev, False)
        teacher.features_dc.add_(0.05 * torch.randn_like(teacher.features_dc))
        gts = [render_stage1(teacher, c, bg)[2].clone() for c in cams]
        del teacher
    out = dict(points=points, image="%dx%d" % (width, height))
    if stage == 2:
        syn4 = objective == "syn4"
        step_fn = fused_step.FusedStage2Step(params, sample_num, lr=1e-4, lrs=SYN4_LRS if syn4 else None,
                                             loss_weights=train_step.STAGE2_WEIGHTS_SYN4 if syn4 else None)
        out.update(sample_num=sample_num, objective="script/run_syn4.sh + run_dtu.sh: smoothness terms, geometry frozen"
                   if syn4 else "script/run_nerf.sh", frozen_geometry=step_fn.frozen_geometry)
    else:
        step_fn = fused_step.FusedStage1Step(params, lr=1e-4)
        out.update(objective="script/run_nerf.sh stage 1 (the reference's real regularisers)")
    for i in range(warmup):
        step_fn(cams[i % 4], bg, gts[i % 4])
    # two timed blocks, the faster one reported (both listed): these short side rows run once inside a long process, and a single
    # host hiccup inside ten timed steps -- seen once in round 6: 360 it/s for the DTU row between runs of 607-619 on the same
    # tree -- must not stand as the configuration's rate
    blocks = []
    for b in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step_fn(cams[(warmup + i) % 4], bg, gts[(warmup + i) % 4])
        step_fn.flush()
        torch.cuda.synchronize()
        blocks.append((time.perf_counter() - t0) / steps)
    dt = min(blocks)
    counts = step_fn.rendered_counts(steps)
    out.update(iters_per_s=round(1.0 / dt, 2), ms_per_step=round(1e3 * dt, 3), num_rendered=round(sum(counts) / max(1, len(counts))),
               iters_per_s_blocks=[round(1.0 / x, 2) for x in blocks], dropped_steps=step_fn.poll_overflow())
    if stage_ms:
        # per-stage HIP-event times of 3 more iterations (outside the timed region: the event pairs cost host time)
        L = _lib.lib()
        L.r3dg_profile_enable(1)
        for i in range(3):
            step_fn(cams[i % 4], bg, gts[i % 4])
        step_fn.flush()
        torch.cuda.synchronize()
        prof = _lib.profile_read()
        L.r3dg_profile_enable(0)
        out["stage_ms"] = {k: round(ms / 3, 4) for k, (ms, n) in prof.items() if n}
    if stage == 2 and relight_samples and relight_frames:
        step_fn.visibility = step_fn.incident_dirs = step_fn.incident_areas = None
        step_fn._taps = step_fn._taps_src = step_fn._frs = None
        torch.cuda.empty_cache()
        envmap = (3.0 * torch.rand(256, 512, 3, generator=torch.Generator().manual_seed(7)) ** 2).to(dev)
        t0 = time.perf_counter()
        r = relight.RelightRenderer(step_fn, envmap, relight_samples)
        torch.cuda.synchronize()
        t_vis = time.perf_counter() - t0
        try:
            visits = trace_visit_stats(r, t_vis)
        except Exception as e:
            visits = {"failed": repr(e)}
        zbg = torch.zeros(3, device=dev)
        for i in range(2):
            r.frame(cams[i], zbg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nr = 0
        for i in range(relight_frames):
            nr += r.frame(cams[i % 4], zbg)["num_rendered"]
        torch.cuda.synchronize()
        dtf = (time.perf_counter() - t0) / relight_frames
        out.update(relight_fps=round(1.0 / dtf, 2), relight_ms_per_frame=round(1e3 * dtf, 3), relight_samples=relight_samples,
                   relight_num_rendered=round(nr / relight_frames), visibility_rays=points * relight_samples,
                   visibility_seconds=round(t_vis, 3), visibility_Mrays_per_s=round(points * relight_samples / t_vis / 1e6, 1),
                   visibility_trace_visits=visits)
    return out


def distribution_rows(dev, width, height, sample_num=64, relight_samples=384, iid=None):
    """Workloads whose splat statistics are not the i.i.d. log-normal ones every other number is quoted on (VERDICT r5 weak 8;
    trained_scene.py): a scene TRAINED here from 4 000 random points with the reference's densification schedule, and the
    synthetic scene with 1 % of its splats 20 x larger.  For each: what the front end sees in view 0 (`binning`: num_rendered,
    tile-list lengths, rectangle sizes), the stage-2 training rate with its per-stage times, relight FPS, views dropped by the
    bounded forward -- and, against the i.i.d. scene's row `iid` (config_rate(..., stage_ms=True) of the headline scene), every
    stage's time PER MILLION INSTANCES relative to the i.i.d. scene's (`per_instance_vs_iid`; > 2 = a cliff to explain)."""
    from . import trained_scene as ts
    rows = {}
    cam0 = syn.orbit_cameras(100, width=width, height=height)[0].to(dev)
    if iid is None:
        iid = config_rate(dev, 300_000, width, height, sample_num=sample_num, stage_ms=True)
    iid_R = float(iid["num_rendered"])

    def relative(row):
        out = {}
        for k, ms in (row.get("stage_ms") or {}).items():
            base = (iid.get("stage_ms") or {}).get(k)
            if base and ms and k in ("duplicate_with_keys", "sort_pairs", "render_forward", "render_backward", "identify_tile_ranges"):
                out[k] = round((ms / row["num_rendered"]) / (base / iid_R), 2)
        if row.get("ms_per_step") and iid.get("ms_per_step"):
            out["whole_step"] = round((row["ms_per_step"] / row["num_rendered"]) / (iid["ms_per_step"] / iid_R), 2)
        return out
    for name, make in (("trained_scene", lambda: ts.train_scene(dev, res=max(width, height) if width == height else 800)),
                       ("heavy_tail_1pct_x20", lambda: ts.heavy_tail_scene())):
        try:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sc = make()
            torch.cuda.synchronize()
            t_make = time.perf_counter() - t0
            row = config_rate(dev, 0, width, height, sample_num=sample_num, steps=12, warmup=4, relight_samples=relight_samples,
                              relight_frames=6, stage_ms=True, scene=sc)
            row["binning"] = ts.binning_stats(sc, cam0, dev)
            row["scene_seconds"] = round(t_make, 2)
            row["per_instance_vs_iid"] = relative(row)
            row.pop("visibility_trace_visits", None)
            rows[name] = row
            del sc
            torch.cuda.empty_cache()
        except Exception as e:                   # a side measurement
            rows[name] = {"failed": repr(e)}
    rows["iid_reference_row"] = {k: iid.get(k) for k in ("points", "iters_per_s", "ms_per_step", "num_rendered", "stage_ms")}
    rows["what"] = ("trained_scene: stage 1 from 4 000 random points, 3 000 iterations, densify_and_prune every 100 from 200 to 2 600 "
                    "(train.py:158-175 compressed), position-gradient threshold 5e-5, against 24 views of a hidden 60 000-splat teacher "
                    "(trained_scene.train_scene); heavy_tail: make_scene with a seeded 1 % of the splats x 20")
    return rows


def unfused_rate(dev, points, width, height, sample_num=64, shading="hip", steps=10, warmup=3):
    """The REFERENCE'S loop shape over the drop-in ops (train.py:114-127 with the flags of script/run_nerf.sh:20-39): per
    iteration the Python glue of gaussian_renderer/neilf.py (activations, feature row, losses as PyTorch ops), the three
    extension calls through autograd, loss.backward(), one torch.optim.Adam step over the parameter groups, zero_grad -- what a
    user gets who only swaps the extension packages (north_star: "train.py drops in unchanged").  `shading`: "hip" = the
    shading integral is this repo's op (INTEGRATION.md's one-line rendering_equation patch); "pytorch" = the reference's own
    pure-PyTorch rendering_equation (no patch at all).  iters/s of `steps` iterations, host-synchronous like train.py
    (loss.item() every iteration for the progress bar: train.py:133)."""
    from . import train_step
    scene = syn.make_scene(P=points, seed=0, stage2=True)
    cams = [c.to(dev) for c in syn.orbit_cameras(100, width=width, height=height)[:4]]
    bg = torch.ones(3, device=dev)
    params = GaussianParams(scene, dev, True)
    with torch.no_grad():
        teacher = GaussianParams(syn.make_scene(P=points, seed=0, stage2=False), dev, False)
        teacher.features_dc.add_(0.05 * torch.randn_like(teacher.features_dc))
        gts = [render_stage1(teacher, c, bg)[2].clone() for c in cams]
        del teacher
    step_fn = train_step.Stage2Step(params, scene, dev, sample_num, shading=shading)
    opt = torch.optim.Adam([{"params": [q], "lr": 1e-4} for q in params.parameters()], eps=1e-15)   # (one group per tensor,
    #                                                                     foreach implementation: gaussian_model.py:465-497)

    def one(i):
        loss, outs = step_fn(cams[i % 4], bg, gts[i % 4])
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        return float(loss)                                   # (the reference reads the loss every iteration: train.py:133)
    for i in range(warmup):
        one(i)
    torch.cuda.synchronize()
    peak0 = torch.cuda.max_memory_allocated()
    t0 = time.perf_counter()
    for i in range(steps):
        one(warmup + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out = dict(points=points, image="%dx%d" % (width, height), sample_num=sample_num, iters_per_s=round(1.0 / dt, 2),
               ms_per_step=round(1e3 * dt, 3), target_iters_per_s=40, meets_target=bool(1.0 / dt >= 40.0),
               peak_memory_GB=round(max(peak0, torch.cuda.max_memory_allocated()) / 2 ** 30, 2),
               shading="this repo's op (shading_ops.rendering_equation)" if shading == "hip"
               else "the reference's pure-PyTorch rendering_equation (neilf.py:339-407), unpatched")
    del step_fn, opt, params
    torch.cuda.empty_cache()
    return out


def densify_bench(points, res, dev, views=8):
    """One densify_and_prune of the stage-1 model after `views` iterations of statistics: wall time
    of the whole call (plan + count read-back + normal table + one-launch gather + buffer re-allocation) and the gather's
    algorithmic bytes (every surviving parameter / Adam-moment float read once and written once)."""
    from . import fused_step
    scene = syn.make_scene(P=points, seed=0, stage2=False)
    cams = [c.to(dev) for c in syn.orbit_cameras(100, width=res, height=res)[:views]]
    bg = torch.ones(3, device=dev)
    params = GaussianParams(scene, dev, False)
    with torch.no_grad():
        teacher = GaussianParams(syn.make_scene(P=points, seed=0, stage2=False), dev, False)
        teacher.features_dc.add_(0.05 * torch.randn_like(teacher.features_dc))
        gts = [render_stage1(teacher, c, bg)[2].clone() for c in cams]
        del teacher
    step = fused_step.FusedStage1Step(params, lr=1e-4)
    step.enable_densification()
    for i in range(views):
        step(cams[i], bg, gts[i])
    st = step.stats
    mean_grad = st.xyz_gradient_accum / st.denom.clamp_min(1)
    thr = float(mean_grad[mean_grad > 0].median())          # half of the visible Gaussians clone or split
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    info = step.densify_and_prune(thr, 0.005, 2.6, 20, 99999, percent_dense=0.01,
                                  generator=torch.Generator(device=dev).manual_seed(0))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    floats = 3 + 3 + 3 + 4 + 1 + 48
    moved = info["rows_out"] * floats * 4 * 3 * 2
    step(cams[0], bg, gts[0])                                 # the loop goes on at the new size
    torch.cuda.synchronize()
    return dict(rows_in=points, rows_out=info["rows_out"], cloned=info["cloned"], split=info["split"],
                call_ms=round(1e3 * dt, 3), gather_algorithmic_MB=round(moved / 1e6, 1))


COMPACT_LIMIT = 4096      # bytes: the LAST stdout line must stay a document the driver's tail can hold

_OTHER_SHORT = (("configs[1] stage-1", "stage1_800"), ("sample_num 384 as BASELINE", "syn4_K384"),
                ("the same at the script's own sample_num 64", "syn4_K64"), ("stage-2 run_nerf.sh objective at sample_num 384", "nerf_K384"),
                ("configs[3] DTU", "dtu_1600x1200"), ("configs[4] composition", "compose_2M"),
                ("stage1_densify_and_prune", "densify_call_ms"), ("splat distributions", "distributions"),
                ("rendering_equation = this repo's op", "dropin_loop_patched"),
                ("rendering_equation = the reference's pure-PyTorch", "dropin_loop_unpatched"),
                ("data_parallel_path_one_rank_rccl", "dp_one_rank_rccl"))


def _roofline_compact(r):
    if not r:
        return None
    out = {k: r.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_kernel_ms", "algorithmic_MB")}
    if isinstance(out["traffic"], float):
        out["traffic"] = round(out["traffic"])
    vb = r.get("valu_bound") or {}
    out["valu_bound"] = {k: vb.get(k) for k in ("frac", "bound_ms", "wave_instr")} if vb else None
    for k in ("stale", "counters_source", "alone_kernel_ms"):
        if r.get(k) is not None:
            out[k] = r[k]
    return out


def compact(result):
    """The LAST stdout line of bench.py: the contract's keys + roofline + cpu_baseline + one number per side measurement,
    < COMPACT_LIMIT bytes so that a driver that keeps only the tail of stdout still holds a parseable headline (round 4's
    23 KB document left `BENCH_r04.parsed` null).  The full document goes to gpurun_out/bench_full.json and to stderr."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "exposed_comm_ms", "exposed_comm_ms_by_bucket", "dp_buckets", "reserved_cus_for_comm", "plumbing_only")
    c = {k: result[k] for k in keep if k in result}
    cfg = result.get("config") or {}
    c["config"] = {"workload": str(cfg.get("workload", ""))[:260], "parallelism": str(cfg.get("parallelism", "")).split(" ")[0]}
    c["roofline"] = _roofline_compact(result.get("roofline"))
    if result.get("roofline_relight"):
        c["roofline_relight"] = _roofline_compact(result["roofline_relight"])
    cbk = result.get("comm_buckets")
    if cbk:
        c["comm_buckets"] = {k: {a: v.get(a) for a in ("MB", "ready_us", "released_us", "collective_ms", "bus_GBs")} for k, v in cbk.items()}
    rl = result.get("relight") or {}
    if rl and "frames_per_rank" in rl:
        c["relight_fps"] = rl.get("relight_fps")
        c["relight"] = {k: rl.get(k) for k in ("relight_K", "frames_per_rank", "per_rank_fps_min", "per_rank_fps_max",
                                               "visibility_Mrays_per_s", "visibility_seconds")}
    elif rl:
        c["relight_fps"] = rl.get("relight_fps")
        c["relight"] = {"K": rl.get("relight_K"), "fps_radiance_cache": rl.get("relight_fps_radiance_cache"),
                        "fps_turning_light": (rl.get("relight_rotating_light") or {}).get("fps"),
                        "fps_pytorch_glue": rl.get("relight_fps_pytorch_glue"),
                        "visibility_Mrays_per_s": rl.get("visibility_Mrays_per_s"),
                        "visibility_node_visits_per_s": rl.get("visibility_node_visits_per_s")}
    sp = result.get("spread_iters_per_s") or {}
    if sp:
        c["spread_iters_per_s"] = {k: sp.get(k) for k in ("min", "median", "max", "blocks")}
    cb = result.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                             "seconds_per_view": cb.get("seconds_per_view"), "sample": str(cb.get("sample"))[:170]}
    oc = result.get("other_configs")
    if isinstance(oc, dict):
        short = {}
        for key, v in oc.items():
            name = next((b for a, b in _OTHER_SHORT if a in key), key[:24])
            if not isinstance(v, dict):
                short[name] = v
            elif "failed" in v or "skipped" in v:
                short[name] = "failed" if "failed" in v else "skipped"
            elif name == "distributions":
                for sub in ("trained_scene", "heavy_tail_1pct_x20"):
                    r = v.get(sub) or {}
                    short[sub] = {"iters_per_s": r.get("iters_per_s"), "points": r.get("points"), "num_rendered": r.get("num_rendered"),
                                  "relight_fps": r.get("relight_fps"), "dropped": r.get("dropped_steps"),
                                  "step_per_instance_vs_iid": (r.get("per_instance_vs_iid") or {}).get("whole_step")} \
                        if "failed" not in r else "failed"
            elif name == "densify_call_ms":
                short[name] = v.get("call_ms")
            else:
                short[name] = v.get("iters_per_s")
                if v.get("relight_fps") is not None:
                    short[name + "_relight_fps"] = v["relight_fps"]
                if v.get("visibility_Mrays_per_s") is not None:
                    short[name + "_visibility_Mrays_per_s"] = v["visibility_Mrays_per_s"]
                pr = v.get("priced_all_reduce_8_ranks")
                if isinstance(pr, dict):
                    short["priced_8gpu_iters_per_s"] = {a.split()[0]: (b.get("predicted_8gpu_iters_per_s") if isinstance(b, dict) else None)
                                                        for a, b in pr.items()}
        c["other_configs_iters_per_s"] = short
    hc = result.get("host_cpu") or {}
    if hc:
        c["host_throttled_ms"] = hc.get("throttled_ms_in_timed_region")
    dc = result.get("device_clock") or {}
    if dc.get("shader_clock_ghz_under_valu_load") is not None:
        c["shader_clock_ghz"] = dc["shader_clock_ghz_under_valu_load"]
    c["full"] = "gpurun_out/bench_full.json (also on stderr)"
    c = _finite_json(c)
    line = json.dumps(c, allow_nan=False, separators=(",", ":"))
    # never exceed the limit: drop the least important groups first
    for k in ("spread_iters_per_s", "relight", "other_configs_iters_per_s", "roofline_relight"):
        if len(line) < COMPACT_LIMIT:
            break
        c.pop(k, None)
        line = json.dumps(c, allow_nan=False, separators=(",", ":"))
    return line


def flush_c_stdio():
    """Push out what C libraries hold in their stdio buffers.  RCCL prints its version banner to stdout through C stdio when the
    communicator is created; with stdout a pipe that text sits in a buffer until the process exits -- i.e. it would land BEHIND the
    JSON line, and a driver that parses the last stdout line of `bench.py --gpus N` would find "Librccl path : ..." there
    (measured on the one-rank RCCL group, round 6)."""
    import ctypes
    import sys
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


def close_stdout():
    """Nothing this process writes to fd 1 from here on reaches the terminal / pipe (ranks > 0 after their part is done; rank 0
    behind its JSON line): library teardown messages cannot follow the ONE line the contract promises."""
    import sys
    try:
        flush_c_stdio()
        fd = os.open(os.devnull, os.O_WRONLY)
        os.dup2(fd, 1)
        os.close(fd)
    except OSError:
        pass


def emit(result):
    """rank 0: the full document to gpurun_out/bench_full.json and stderr, then the compact line -- the ONE JSON line on stdout."""
    full = json.dumps(_finite_json(result), allow_nan=False)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "bench_full.json"), "w") as fh:
            fh.write(full + "\n")
    except OSError:
        pass
    import sys
    sys.stdout.flush()
    sys.stderr.write("bench_full: " + full + "\n")          # (stderr: stdout carries exactly ONE JSON line, the contract's)
    sys.stderr.flush()
    flush_c_stdio()                                          # RCCL's buffered banner first, so that the JSON line is the LAST one
    print(compact(result), flush=True)
    close_stdout()


def _plumbing_only(args, world, rank, backend):
    """The launcher / rendezvous / max-over-ranks reduction / rank-0 print path of run() with no kernels: what a CPU box
    can check of `bench.py --gpus N` (tests/test_dp_cpu.py).  value is null: nothing was measured."""
    if world > 1:
        dist.init_process_group("gloo" if not torch.cuda.is_available() else backend)
        dist.barrier()
    t0 = time.perf_counter()
    t = torch.tensor([time.perf_counter() - t0 + 1e-3 * rank], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        seen = torch.ones(1, dtype=torch.float64)
        dist.all_reduce(seen)
        assert int(seen.item()) == world
    result = None
    if rank == 0:
        result = {"metric": "plumbing only (no kernels run)", "value": None, "unit": "iters/s", "n_gpus": world,
                  "steps": args.steps, "warmup": args.warmup, "plumbing_only": True}
        emit(result)
    if world > 1:
        dist.destroy_process_group()
    return result


def run(args):
    # the host driver only supports dmabuf IPC: must be in the environment before the first HIP call initialises the runtime
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    t_start = time.perf_counter()

    def side_budget(want_s):
        """Seconds a side measurement in a child process may still take: the default run is meant to finish within minutes,
        so the side measurements share what is left of ~5 minutes since the start (0 = skip it); R3DG_BENCH_NO_CHILDREN=1 skips all."""
        if os.environ.get("R3DG_BENCH_NO_CHILDREN") == "1":
            return 0
        left = 300.0 - (time.perf_counter() - t_start)
        return int(min(want_s, left)) if left >= 45.0 else 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # one process per GPU; R3DG_DIST_BACKEND=gloo lets the multi-rank path be exercised on a box with fewer GPUs than
    # ranks (ranks then share devices) -- test use only, the measured configuration is nccl (= RCCL)
    backend = os.environ.get("R3DG_DIST_BACKEND", "nccl")
    if getattr(args, "plumbing_only", False):
        return _plumbing_only(args, world, rank, backend)
    dev_index = local_rank % max(1, torch.cuda.device_count()) if world > 1 else 0
    # R3DG_DP_SINGLE_RANK=1: a ONE-rank process group whose iteration still takes the data-parallel path
    # (fused_step._world_of) -- the RCCL calls of `--gpus N` exercised on a box with one GPU; the collectives are identities
    dp = world > 1 or os.environ.get("R3DG_DP_SINGLE_RANK") == "1"
    if dp:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1 and "MASTER_PORT" not in os.environ:           # not under a launcher: rendezvous with ourselves
            import socket
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                port = sock.getsockname()[1]
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
        torch.cuda.set_device(dev_index)
        # the process group then brackets every collective with timed events on RCCL's own stream (Work._get_duration): what
        # FusedStage2Step.comm_table reads the per-bucket collective time from
        os.environ.setdefault("TORCH_NCCL_ENABLE_TIMING", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    L = _lib.lib()
    for name in _lib.OPTIONS:                            # tuning experiments only: R3DG_OPT_<NAME>=<value>
        if os.environ.get("R3DG_OPT_" + name):
            _lib.set_option(name, int(os.environ["R3DG_OPT_" + name]))

    stage2 = args.stage == 2
    scene = syn.make_scene(P=args.points, seed=0, stage2=stage2)
    W_img = getattr(args, "width", 0) or args.res
    H_img = getattr(args, "height", 0) or args.res
    cams_cpu = syn.orbit_cameras(100, width=W_img, height=H_img)
    cams = [c.to(dev) for c in cams_cpu]
    bg = torch.ones(3, device=dev)
    params = GaussianParams(scene, dev, stage2)
    fused = not getattr(args, "unfused", False)
    opt = None if fused else torch.optim.Adam(params.parameters(), lr=1e-4, eps=1e-15, fused=True)
    if fused and not stage2:
        from . import fused_step
        step_fn = fused_step.FusedStage1Step(params, lr=1e-4)
        S = 5
    elif fused:
        # the whole iteration through the fused glue kernels + one-launch Adam (fused_step.py); gradients are averaged
        # over ranks inside (three buckets of one flat slab; see fused_step.py and DESIGN.md section 5)
        from . import fused_step
        syn4 = getattr(args, "objective", "nerf") == "syn4"
        from . import train_step as _ts
        step_fn = fused_step.FusedStage2Step(params, args.sample_num, lr=1e-4, lrs=SYN4_LRS if syn4 else None,
                                             loss_weights=_ts.STAGE2_WEIGHTS_SYN4 if syn4 else None,
                                             bounded=os.environ.get("R3DG_BOUNDED", "1") != "0")      # (A/B experiments)
        S = 16
    elif stage2:
        from . import train_step
        step_fn = train_step.Stage2Step(params, scene, dev, args.sample_num)
        S = 16
    else:
        step_fn = None
        S = 5

    # ground-truth images: renders of a perturbed "teacher" copy, resident in HBM before timing
    with torch.no_grad():
        teacher = GaussianParams(syn.make_scene(P=args.points, seed=0, stage2=False), dev, False)
        teacher.features_dc.add_(0.05 * torch.randn_like(teacher.features_dc))
        my_views = list(range(rank, 100, world))
        gts = {}
        for v in my_views[: max(4, (args.steps + args.warmup))]:
            gts[v] = render_stage1(teacher, cams[v], bg)[2].clone()
        del teacher
    gt_views = list(gts.keys())
    reducer = None
    if world > 1 and not fused:
        from . import dp as dp_autograd
        reducer = dp_autograd.GradAllReducer(params.parameters())     # bucketed async all-reduce, overlaps the backward tail

    R_seen = []

    def one_step(i):
        v = gt_views[i % len(gt_views)]
        cam = cams[v]
        if fused:
            outs = step_fn(cam, bg, gts[v])                  # forward + loss + backward + all-reduce + Adam
            if not hasattr(step_fn, "rendered_counts"):      # (the bounded forward keeps the count off the host)
                R_seen.append(outs[0])
            return None
        if stage2:
            loss, outs = step_fn(cam, bg, gts[v])
        else:
            outs = render_stage1(params, cam, bg)
            loss = loss_stage1(outs, gts[v])
        R_seen.append(outs[0])
        loss.backward()
        if reducer is not None:
            reducer.finish()
        opt.step()
        if reducer is not None:
            reducer.zero_grad()
        else:
            opt.zero_grad(set_to_none=False)
        return loss

    for i in range(args.warmup):
        one_step(i)
    torch.cuda.synchronize()
    if dp:
        dist.barrier()
    L.r3dg_profile_enable(0 if os.environ.get("R3DG_BENCH_NOPROFILE") else 1)
    if dp and fused and hasattr(step_fn, "measure_comm"):
        step_fn.measure_comm = True          # (two event records per waited bucket: what the compute stream stalls on)
    torch.cuda.synchronize()
    quota0 = host_cpu_quota()
    t0 = time.perf_counter()
    # per-kernel HIP-event timing is live inside the timed region but sampled (every EVENT_EVERY-th step): each event pair costs
    # ~2 us of host time, ~40 pairs per step, and the events themselves sit between the kernels on the stream
    for i in range(args.steps):
        L.r3dg_profile_pause(0 if (i % EVENT_EVERY == 0 and not os.environ.get("R3DG_BENCH_NOPROFILE")) else 1)
        one_step(args.warmup + i)
    if fused:
        step_fn.flush()                  # (world > 1) the last iteration's deferred incident-light update
    torch.cuda.synchronize()
    if dp:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    quota1 = host_cpu_quota()
    host_cpu = dict(quota_cores=quota1.get("quota_cores"),
                    throttled_periods_in_timed_region=quota1.get("nr_throttled", 0) - quota0.get("nr_throttled", 0),
                    throttled_ms_in_timed_region=round(1e-3 * (quota1.get("throttled_usec", 0) - quota0.get("throttled_usec", 0)), 1),
                    throttled_periods_before=quota0.get("nr_throttled"),
                    what="cgroup cpu.max / cpu.stat around the timed region: periods in which the container's CPU quota "
                         "stalled its threads (a host effect; 0 = the timed steps were not held up by it)") if quota1 else None
    prof = _lib.profile_read()
    L.r3dg_profile_enable(0)
    # the shader clock this box sustains under VALU load, measured on the device right behind the timed region (boxes of the pool
    # differed by 8 % on identical code): what `roofline.valu_bound` and a comparison between two bench lines should be read with
    try:
        ghz, nwaves = _lib.shader_clock_ghz(dev)
        device_clock = dict(shader_clock_ghz_under_valu_load=round(ghz, 3), waves=nwaves,
                            what="r3dg_clock_probe behind the timed region: shader-clock cycles per wall-clock tick over a "
                                 "device-filling grid of FMA-only waves")
    except Exception as e:                         # (never the reason a bench line is missing)
        device_clock = dict(error=str(e)[:120])
    exposed_comm = None
    exposed_by_bucket = {}
    if dp and fused and hasattr(step_fn, "exposed_comm_ms"):
        exposed_comm, exposed_by_bucket = step_fn.exposed_comm_ms(split=True)
        step_fn.measure_comm = False
    if fused and hasattr(step_fn, "rendered_counts"):
        R_seen = step_fn.rendered_counts(args.steps)
        dropped = step_fn.poll_overflow()
        if dropped:
            raise RuntimeError("bench: %d timed iterations were dropped by the bounded forward (capacity too small)" % dropped)
    if dp:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # spread of the headline number: the same block of --steps iterations repeated (N=1; event timing off)
    blocks = [world * args.steps / elapsed]
    if world == 1 and getattr(args, "repeats", 0) > 0:
        L.r3dg_profile_pause(1)
        for r in range(args.repeats):
            torch.cuda.synchronize()
            tb = time.perf_counter()
            for i in range(args.steps):
                one_step(args.warmup + (r + 1) * args.steps + i)
            if fused:
                step_fn.flush()
            torch.cuda.synchronize()
            blocks.append(args.steps / (time.perf_counter() - tb))
    blocks_sorted = sorted(blocks)
    spread = dict(blocks=len(blocks), steps_per_block=args.steps, min=round(blocks_sorted[0], 2),
                  median=round(blocks_sorted[len(blocks) // 2], 2), max=round(blocks_sorted[-1], 2),
                  note="iters/s of the timed block (`value`, event timing on every %dth step) and of %d more blocks "
                       "(event timing off)" % (EVENT_EVERY, len(blocks) - 1))

    # a few more iterations with every launch on ONE stream (FusedStage2Step.serial_streams): each stage's time with nothing
    # beside it, for the choice of the dominant kernel (kernel_table `alone`, roofline_of)
    alone = None
    if fused and stage2 and world == 1 and not dp and hasattr(step_fn, "serial_streams") and not os.environ.get("R3DG_BENCH_NOPROFILE") \
            and not os.environ.get("R3DG_BENCH_NO_ALONE"):          # (rocprofv3 runs: keep the trace to the pipelined iterations)
        try:
            step_fn.serial_streams = True
            one_step(args.warmup)
            torch.cuda.synchronize()
            L.r3dg_profile_pause(0)
            L.r3dg_profile_enable(1)
            n_alone = 4
            for i in range(n_alone):
                one_step(args.warmup + 1 + i)
            torch.cuda.synchronize()
            alone = (_lib.profile_read(), n_alone)
        except Exception:
            alone = None
        finally:
            step_fn.serial_streams = False
            L.r3dg_profile_enable(0)

    relight = None
    comm_buckets = exposed_split = None
    if dp and fused and hasattr(step_fn, "comm_table"):
        comm_buckets = step_fn.comm_table()
    if stage2 and args.relight_frames > 0:
        step_fn.visibility = step_fn.incident_dirs = step_fn.incident_areas = None   # free the K=train caches
        if hasattr(step_fn, "_taps"):
            step_fn._taps = step_fn._taps_src = None
        torch.cuda.empty_cache()
        if world == 1:
            relight = relight_bench(step_fn if fused else params, cams, dev, args.relight_frames, args.relight_samples)
        else:
            # the other half of BASELINE.json's metric at N > 1: frames sharded over the ranks (replicas only)
            relight = relight_bench_sharded(step_fn if fused else params, cams, dev, args.relight_frames, args.relight_samples,
                                            rank, world)
    result = None
    if rank == 0:
        P, N = args.points, W_img * H_img
        R_mean = float(sum(R_seen[-args.steps:])) / max(1, args.steps)
        n_sampled = max(1, sum(1 for i in range(args.steps) if i % EVENT_EVERY == 0))     # steps whose launches were timed
        # feature channels the timed backward launch carried (FusedStage2Step: 3 of 16 under the run_nerf.sh objective)
        S_bwd = getattr(step_fn, "last_active_features", None)
        S_bwd = None if S_bwd is None else len(S_bwd)
        general = fused and stage2 and getattr(step_fn, "_frs", None) is None
        kernels = kernel_table(prof, n_sampled, P, R_mean, N, S, args.sample_num, S_bwd=S_bwd,
                               rename={"shade_forward": "shade_forward_general", "shade_backward": "shade_backward_general"}
                               if general else None, alone=alone,
                               # (whole single-GPU iterations run the incident-light chain as one kernel: fused_step.py)
                               chain_kernel=bool(fused and stage2 and getattr(step_fn, "_chain_kernel", False) and
                                                 getattr(step_fn, "_pre_rotated", None) is not None and not dp))
        if not kernels:                   # experiments with the in-library event timing switched off
            kernels = {"none": dict(avg_ms=0.0, launches=0, ms_per_iteration=0.0, algorithmic_MB=None, achieved_GBs=None)}
        roofline = roofline_of(kernels, "achieved = algorithmic bytes of the TIMED launch (SURVEY.md 8(d) per-unit figures with "
                               "the %s feature channels this launch carries) / HIP-event kernel time inside the iteration; "
                               "`bound` = what the SQ counters of the same launch configuration say (`valu_bound`: wave "
                               "instructions x 4 cycles / 1024 SIMDs); the HBM fraction is reported as the contract requires" % (
                                   "S" if S_bwd is None else str(S_bwd)))
        roofline["feature_channels_in_backward"] = S_bwd
        iters_s = world * args.steps / elapsed
        result = {
            "metric": "train iters/s, synthetic lego-like %dx%d, %d Gaussians (stage-%d hot path)" % (
                W_img, H_img, P, args.stage),
            "value": round(iters_s, 3), "unit": "iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("stage-%d train iteration%s: %srasterize fwd (S=%d) + loss + rasterize bwd + Adam; "
                                    "1 view/rank/step, %d Gaussians, %dx%d, num_rendered~%.0f" + (
                                        "; objective + schedule of script/run_syn4.sh (smoothness terms, geometry frozen)"
                                        if getattr(args, "objective", "nerf") == "syn4" else "")) % (
                                       args.stage, " (fused glue kernels + one-launch Adam)" if fused else "",
                                       "shading fwd/bwd (K=%d) + " % args.sample_num if stage2 else "", S,
                                       P, W_img, H_img, R_mean),
                       "parallelism": "dp%d (views sharded over ranks; bucketed async RCCL all-reduce of per-Gaussian grads)" % world},
            "roofline": roofline, "kernels": kernels, "spread_iters_per_s": spread, "host_cpu": host_cpu,
            "device_clock": device_clock,
        }
        if dp:
            # rank 0's compute stream: mean time per iteration it stood waiting for gradient all-reduce buckets (C, then the
            # deferred B in front of the next shading forward; A is waited for on a side stream, under the shading backward)
            result["exposed_comm_ms"] = None if exposed_comm is None else round(exposed_comm, 4)
            # ... split by the bucket the stream stood waiting for, and every bucket's collective bracketed by events on a probe
            # stream (FusedStage2Step.comm_table): bytes, ready / done times inside the iteration, the collective's own time and
            # the ring bus bandwidth it amounts to -- what tells a lost 8-GPU factor apart into bandwidth (bus_GBs low), message
            # size (R3DG_DP_BUCKETS=1 against the default 3) and schedule (exposed_comm_ms_by_bucket high at a good bus_GBs)
            result["exposed_comm_ms_by_bucket"] = {k: round(v, 4) for k, v in sorted(exposed_by_bucket.items())}
            result["comm_buckets"] = comm_buckets
            result["dp_buckets"] = 1 if getattr(step_fn, "_single_bucket", False) else 3
            with step_fn._ctx:                   # (the step's own option context: the process default stays 0)
                result["reserved_cus_for_comm"] = _lib.get_option("RESERVE_CUS")
        if relight is not None and "roofline_relight" in relight:
            result["roofline_relight"] = relight.pop("roofline_relight")
        if relight is not None:
            result["relight"] = relight
        if stage2 and world == 1 and not getattr(args, "no_other_configs", False):
            # the other single-GPU BASELINE configurations on the same scene (short runs; the headline stays `value`)
            try:
                del step_fn, params
                torch.cuda.empty_cache()
                oc = result["other_configs"] = {}
                oc["configs[1] stage-1 3DGS train, 800x800"] = config_rate(dev, args.points, W_img, H_img, stage=1)
                oc["configs[2] Synthetic4Relight stage-2 (run_syn4.sh objective + schedule), sample_num 384 as BASELINE.json states"] = \
                    config_rate(dev, args.points, W_img, H_img, sample_num=384, objective="syn4", steps=10, warmup=3)
                oc["configs[2] the same at the script's own sample_num 64 (run_syn4.sh:39)"] = \
                    config_rate(dev, args.points, W_img, H_img, sample_num=64, objective="syn4")
                oc["stage-2 run_nerf.sh objective at sample_num 384"] = config_rate(
                    dev, args.points, W_img, H_img, sample_num=384, steps=10, warmup=3)
                if side_budget(60):
                    oc["configs[3] DTU stage-2 (run_dtu.sh: 1600x1200, sample_num 32, env 16, geometry frozen), one view on one GPU"] = \
                        config_rate(dev, args.points, 1600, 1200, sample_num=32, objective="syn4", steps=10, warmup=3,
                                    stage_ms=True)
                if side_budget(60):
                    oc["configs[4] composition scale: 2M Gaussians, 1800x700 (configs/teaser), train sample_num 64 + relight sample_num 384"] = \
                        config_rate(dev, 2_000_000, 1800, 700, sample_num=64, steps=8, warmup=3, relight_samples=384,
                                    relight_frames=6, stage_ms=True)
                if side_budget(50):
                    # (the i.i.d. row they are compared with is the headline itself: the timed block's step time and stage times)
                    iid = dict(points=P, iters_per_s=result["value"], ms_per_step=result["ms_per_step"], num_rendered=R_mean,
                               stage_ms={k: v.get("ms_per_iteration") for k, v in kernels.items()})
                    oc["splat distributions other than the i.i.d. one (trained scene, heavy tail)"] = distribution_rows(
                        dev, W_img, H_img, sample_num=args.sample_num, iid=iid)
                oc["stage1_densify_and_prune (one call at the bench size)"] = densify_bench(args.points, args.res, dev)
                if args.stage == 2 and side_budget(40):
                    # north_star's literal mode: the reference's loop over the drop-in ops, at the headline size
                    oc["reference loop shape (drop-in ops + autograd + torch.optim.Adam), rendering_equation = this repo's op"] = \
                        unfused_rate(dev, args.points, W_img, H_img, args.sample_num, "hip")
                    oc["reference loop shape, rendering_equation = the reference's pure-PyTorch one (unpatched train.py)"] = \
                        unfused_rate(dev, args.points, W_img, H_img, args.sample_num, "pytorch", steps=6, warmup=2)
                if args.stage == 2 and not getattr(args, "unfused", False):
                    skipped = {"skipped": "time budget of the default run used up (or R3DG_BENCH_NO_CHILDREN=1)"}
                    b = side_budget(150)
                    oc["data_parallel_path_one_rank_rccl"] = dp_path_one_rank(args, timeout_s=b) if b else skipped
                    # ... and PRICED: what the three buckets (58 / 22 / 58 MB at 300k Gaussians) would cost over 8 ranks at an
                    # assumed all-reduce bus bandwidth (xGMI: 7 links x ~153 GB/s per GPU; ring collectives are per-link bound),
                    # each bucket's identity collective followed by a spin of 2 (7/8) bytes / B on the communication stream
                    priced = {}
                    for gbs in (75, 150, 300, 500):
                        b = side_budget(60)
                        priced["%d GB/s" % gbs] = dp_path_one_rank(args, timeout_s=b, fake_comm_gbs=gbs, steps=12) if b else skipped
                    if isinstance(oc["data_parallel_path_one_rank_rccl"], dict):
                        oc["data_parallel_path_one_rank_rccl"]["priced_all_reduce_8_ranks"] = priced
                        oc["data_parallel_path_one_rank_rccl"]["priced_note"] = (
                            "rehearsal, not a measurement of xGMI: predicted_8gpu_iters_per_s = 8 views per step / the one-rank "
                            "step time with every bucket's all-reduce replaced by a spin of its ring time at the assumed bus "
                            "bandwidth; exposed_comm_ms = what the compute stream waits for buckets C and B")
            except Exception as e:
                result["other_configs"] = {"failed": repr(e)}
        if not args.no_cpu_baseline and world == 1:
            try:
                result["cpu_baseline"] = cpu_baseline(scene, cams_cpu, S, args.cpu_baseline_seconds, args.points, W_img, H_img)
            except Exception as e:  # the oracle is only a reported baseline; never fail the bench on it
                result["cpu_baseline"] = {"value": None, "unit": "views/s (rasterize forward)", "cores": None, "kind": "port",
                                          "sample": "failed: %r" % (e,)}
    if dp:
        # every rank's buffered library output (RCCL's banner) is out before rank 0 prints the line; the other ranks then close
        # their stdout for good
        flush_c_stdio()
        dist.barrier()
        if rank != 0:
            close_stdout()
    if rank == 0:
        emit(result)
    if dp:
        dist.destroy_process_group()
    return result
