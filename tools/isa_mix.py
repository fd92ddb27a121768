#!/usr/bin/env python
"""Static instruction mix of the hot kernels (no GPU needed): compiles a csrc/*.hip with the flags of
relightable3dgaussian_amd/build.py to gfx950 assembly and counts, per kernel and for its loops (spans closed by a backward branch,
largest first), the VALU / transcendental / SALU / LDS / vector-memory instructions.

    python tools/isa_mix.py profiles/rNN_isa_mix.json

Evidence for the VALU-bound reading of the tile and shading kernels: what a loop iteration issues,
next to what PMC says the SIMDs were doing (profiles/rNN_pmc_valu.json)."""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from relightable3dgaussian_amd import build as B   # noqa: E402

# (file, substring of the demangled kernel name): the template instances the default bench launches
KERNELS = [
    ("shading.hip", "shade_backward_kernel<true, true, true, false>"),
    ("shading.hip", "shade_backward_frs_kernel"),
    ("shading.hip", "shade_forward_frs_kernel"),
    ("shading.hip", "shade_forward_row_kernel<7, true, 1, true, false>"),
    ("shading.hip", "shade_forward_row_kernel<19, false, 2, true, false>"),
    ("shading.hip", "shade_forward_transport_kernel"),
    ("rasterizer_render_fwd.hip", "render_forward_wave_kernel<16, 4>"),
    ("rasterizer_render_bwd.hip", "render_backward_wave_kernel<4, true>"),
]
TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def mix(ops):
    c = collections.Counter(classify(o) for o in ops)
    c["valu_transcendental"] = sum(1 for o in ops if o.startswith(TRANS))
    c["valu_packed"] = sum(1 for o in ops if o.startswith("v_pk_"))
    c["valu_f64"] = sum(1 for o in ops if o.startswith("v_") and "_f64" in o)
    c["lds_atomic"] = sum(1 for o in ops if o.startswith("ds_add") or o.startswith("ds_max") or o.startswith("ds_min"))
    c["total"] = len(ops)
    return dict(c)


def kernels_of(src):
    flags = B.COMMON + B.EXTRA.get(src, [])
    asm = subprocess.run([B.HIPCC] + flags + ["--cuda-device-only", "-S", os.path.join(B.CSRC, src), "-o", "-"],
                         capture_output=True, text=True)
    if asm.returncode != 0:
        raise RuntimeError(asm.stderr[-2000:])
    out, name, body = {}, None, []
    for line in asm.stdout.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
            out[name] = body
            continue
        if line.startswith(".Lfunc_end"):
            name = None
        if name is not None:
            body.append(line)
    return out


def loops_of(body):
    """Loops = spans [label .. backward branch to that label], largest first (they may nest: `contains_smaller_loops`)."""
    labels, instr = {}, []
    for line in body:
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            labels[m.group(1)] = len(instr)
            continue
        t = line.strip()
        if not t or t.startswith((";", ".")):
            continue
        instr.append(t)
    spans = []
    for i, t in enumerate(instr):
        m = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", t)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            spans.append((labels[m.group(1)], i))
    ops = [t.split()[0] for t in instr]
    spans = sorted(set(spans), key=lambda s: s[0] - s[1])                  # largest first; loops may nest
    return ops, [dict(mix(ops[a:b + 1]), contains_smaller_loops=sum(1 for o in spans if o != (a, b) and a <= o[0] and o[1] <= b))
                 for a, b in spans[:5]]


def main():
    res, cache = {}, {}
    for src, want in KERNELS:
        if src not in cache:
            cache[src] = kernels_of(src)
        for mangled, body in cache[src].items():
            dem = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip()
            if want in dem:
                ops, loops = loops_of(body)
                res[re.sub(r"^void r3dg::", "", dem).split("(")[0]] = dict(file=src, whole_kernel=mix(ops),
                                                                            loops_largest_first=loops)
                break
        else:
            res[want] = {"missing": True}
    doc = {"note": "static counts from `hipcc -S` (gfx950, flags of relightable3dgaussian_amd/build.py): instructions in the kernel "
                   "text and in its loops (spans closed by a backward branch, largest first; they may nest), NOT executed counts",
           "kernels": res}
    out = sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout"
    json.dump(doc, open(out, "w"), indent=1)
    for k, v in res.items():
        if "whole_kernel" in v:
            w = v["whole_kernel"]
            sys.stderr.write("%-50s total %5d  valu %5d  trans %3d  lds %4d  vmem %3d  loops %s\n" % (
                k[:50], w["total"], w.get("valu", 0), w["valu_transcendental"], w.get("lds", 0), w.get("vmem", 0),
                [(l["total"], l.get("valu", 0)) for l in v["loops_largest_first"]]))
        else:
            sys.stderr.write("%-50s MISSING\n" % k)


if __name__ == "__main__":
    main()
